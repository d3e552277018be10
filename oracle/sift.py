"""TEST INFRASTRUCTURE ONLY: NumPy restatement of SIFT detection + description as the reference uses it
(`cv2.SIFT_create().detectAndCompute(gray, None)`, src/extract_sift.py:104-114 and src/utils.py:78-80), all in float32.

OpenCV is a third-party dependency of the reference (opencv-python, not vendored, not importable here), so this file
restates the published algorithm of its features2d module (sift.dispatch.cpp / sift.simd.hpp of OpenCV 4.5+: Lowe's
SIFT with a base image doubled by INTER_LINEAR resize (the default, enable_precise_upscale = False), 3 layers per octave,
sigma 1.6, contrast threshold 0.04, edge threshold 10, float pyramid, descriptors scaled by 512 and saturated to [0, 255];
taps are summed in ascending order and the first blur's sigma is computed in double, where OpenCV pairs symmetric taps and
uses float32, so the pyramid may differ from OpenCV's in the last place).  PARITY UNPINNED: the reference holds no keypoint or
descriptor fixtures and OpenCV cannot be run here, so nothing ties this restatement to cv2's output bit for bit; what is
pinned is that the HIP implementation equals this restatement (tests/test_gpu_sift.py) and that both behave like SIFT
(blobs found at their scale, matches across a rotation and a scale change; tests/test_sift_host.py).

Only tests/, __graft_entry__.smoke() and bench.py's CPU baseline may import this module.
"""
import math

import numpy as np

F = np.float32
N_LAYERS = 3  # nOctaveLayers
SIGMA = 1.6
CONTRAST_THR = 0.04
EDGE_THR = 10.0
INIT_SIGMA = 0.5
IMG_BORDER = 5
MAX_INTERP_STEPS = 5
ORI_HIST_BINS = 36
ORI_SIG_FCTR = 1.5
ORI_RADIUS = 4.5
ORI_PEAK_RATIO = 0.8
DESCR_WIDTH = 4
DESCR_HIST_BINS = 8
DESCR_SCL_FCTR = 3.0
DESCR_MAG_THR = 0.2
INT_DESCR_FCTR = 512.0
FIRST_OCTAVE = -1
FLT_EPSILON = float(np.finfo(np.float32).eps)


def cv_round(v):
    """cvRound: round half to even."""
    return int(np.rint(v))


def gaussian_taps(sigma):
    """getGaussianKernel(cvRound(sigma * 8 + 1) | 1, sigma, CV_32F): float32 taps that sum to 1."""
    n = cv_round(sigma * 8 + 1) | 1
    x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
    t = np.exp(-(x * x) / (2.0 * sigma * sigma)).astype(F)
    s = 1.0 / float(t.astype(np.float64).sum())
    return (t.astype(np.float64) * s).astype(F)


def blur(img, taps):
    """Separable convolution, rows then columns, BORDER_REFLECT_101, float32 multiply then add in ascending tap order."""
    h, w = img.shape
    r = (len(taps) - 1) // 2
    xs = np.arange(w)
    acc = None
    for t, k in enumerate(taps):
        term = img[:, _reflect_many(xs + t - r, w)] * F(k)
        acc = term if acc is None else (acc + term).astype(F)
    rows = acc.astype(F)
    ys = np.arange(h)
    acc = None
    for t, k in enumerate(taps):
        term = rows[_reflect_many(ys + t - r, h), :] * F(k)
        acc = term if acc is None else (acc + term).astype(F)
    return acc.astype(F)


def _reflect_many(i, n):
    """BORDER_REFLECT_101 for offsets that may exceed one reflection (tiny images of the last octaves)."""
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def upsample2x(gray_u8):
    """cv::resize(src, 2 x size, INTER_LINEAR), the branch createInitialImage takes for SIFT_create()'s default
    enable_precise_upscale = False: pixel centres aligned, dst(X) samples src at X / 2 - 0.25 with the border replicated,
    i.e. even X = 2m -> 0.25 src[m-1] + 0.75 src[m], odd X = 2m+1 -> 0.75 src[m] + 0.25 src[m+1]; rows first, then
    columns (every product and sum is exact in float32)."""
    src = gray_u8.astype(F)
    h, w = src.shape

    def axis(a, n, ax):
        X = np.arange(2 * n)
        m, odd = X // 2, (X % 2).astype(bool)
        i0 = np.where(odd, m, np.maximum(m - 1, 0))
        i1 = np.where(odd, np.minimum(m + 1, n - 1), m)
        w0 = np.where(odd, F(0.75), F(0.25)).astype(F)
        w1 = (F(1.0) - w0).astype(F)
        shape = [1, 1]
        shape[ax] = 2 * n
        return (np.take(a, i0, axis=ax) * w0.reshape(shape) + np.take(a, i1, axis=ax) * w1.reshape(shape)).astype(F)

    return axis(axis(src, w, 1), h, 0)


def downsample2(img):
    """resize(src, Size(cols / 2, rows / 2), INTER_NEAREST)."""
    H, W = img.shape
    h, w = max(H // 2, 1), max(W // 2, 1)
    ys = np.minimum(np.floor(np.arange(h) * (H / h)).astype(int), H - 1)
    xs = np.minimum(np.floor(np.arange(w) * (W / w)).astype(int), W - 1)
    return img[np.ix_(ys, xs)].copy()


def layer_sigmas():
    sig = [SIGMA]
    k = 2.0 ** (1.0 / N_LAYERS)
    for i in range(1, N_LAYERS + 3):
        prev = k ** (i - 1) * SIGMA
        total = prev * k
        sig.append(math.sqrt(total * total - prev * prev))
    return sig


def num_octaves(base_w, base_h):
    return cv_round(math.log(min(base_w, base_h)) / math.log(2.0) - 2) - FIRST_OCTAVE


def build_pyramids(gray_u8, taps_fn=gaussian_taps):
    """(gauss[o][layer], dog[o][layer]) of the doubled image."""
    base = upsample2x(gray_u8)
    sig_diff = math.sqrt(max(SIGMA * SIGMA - INIT_SIGMA * INIT_SIGMA * 4, 0.01))
    base = blur(base, taps_fn(sig_diff))
    sig = layer_sigmas()
    gauss, dog = [], []
    for o in range(num_octaves(base.shape[1], base.shape[0])):
        g = []
        for i in range(N_LAYERS + 3):
            if o == 0 and i == 0:
                g.append(base)
            elif i == 0:
                g.append(downsample2(gauss[o - 1][N_LAYERS]))
            else:
                g.append(blur(g[i - 1], taps_fn(sig[i])))
        gauss.append(g)
        dog.append([(g[i + 1] - g[i]).astype(F) for i in range(N_LAYERS + 2)])
    return gauss, dog


def find_candidates(dog_o):
    """(layer, r, c) of the scale-space extrema of one octave, in (layer, r, c) order."""
    out = []
    thr = math.floor(0.5 * CONTRAST_THR / N_LAYERS * 255)
    h, w = dog_o[0].shape
    if h <= 2 * IMG_BORDER or w <= 2 * IMG_BORDER:
        return out
    for i in range(1, N_LAYERS + 1):
        cur = dog_o[i]
        core = cur[IMG_BORDER : h - IMG_BORDER, IMG_BORDER : w - IMG_BORDER]
        is_max = core > thr
        is_min = core < -thr
        for img in (dog_o[i - 1], cur, dog_o[i + 1]):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    nb = img[IMG_BORDER + dy : h - IMG_BORDER + dy, IMG_BORDER + dx : w - IMG_BORDER + dx]
                    is_max &= core >= nb
                    is_min &= core <= nb
        rs, cs = np.nonzero(is_max | is_min)
        out.extend((i, int(r) + IMG_BORDER, int(c) + IMG_BORDER) for r, c in zip(rs, cs))
    return out


def _solve3(H, b):
    """Matx33f::solve(b, DECOMP_LU): Cramer's rule in float32, zero vector for a singular matrix."""
    a = H
    m00 = F(a[1][1] * a[2][2]) - F(a[1][2] * a[2][1])
    m01 = F(a[1][0] * a[2][2]) - F(a[1][2] * a[2][0])
    m02 = F(a[1][0] * a[2][1]) - F(a[1][1] * a[2][0])
    d = F(F(a[0][0] * m00) - F(a[0][1] * m01)) + F(a[0][2] * m02)
    if d == 0:
        return [F(0), F(0), F(0)]
    d = F(1) / d
    t0 = F(b[1] * a[2][2]) - F(a[1][2] * b[2])
    t1 = F(b[1] * a[2][1]) - F(a[1][1] * b[2])
    t2 = F(a[1][0] * b[2]) - F(b[1] * a[2][0])
    x0 = d * F(F(F(b[0] * m00) - F(a[0][1] * t0)) + F(a[0][2] * t1))
    x1 = d * F(F(F(a[0][0] * t0) - F(b[0] * m01)) + F(a[0][2] * t2))
    x2 = d * F(F(F(a[0][0] * F(-t1)) - F(a[0][1] * t2)) + F(b[0] * m02))
    return [F(x0), F(x1), F(x2)]


def _derivs(dog_o, layer, r, c):
    img, prev, nxt = dog_o[layer], dog_o[layer - 1], dog_o[layer + 1]
    img_scale = F(1.0 / 255)
    ds, sds, cds = F(img_scale * F(0.5)), img_scale, F(img_scale * F(0.25))
    dD = [F(img[r, c + 1] - img[r, c - 1]) * ds, F(img[r + 1, c] - img[r - 1, c]) * ds, F(nxt[r, c] - prev[r, c]) * ds]
    v2 = F(img[r, c] * F(2))
    dxx = F(F(img[r, c + 1] + img[r, c - 1]) - v2) * sds
    dyy = F(F(img[r + 1, c] + img[r - 1, c]) - v2) * sds
    dss = F(F(nxt[r, c] + prev[r, c]) - v2) * sds
    dxy = F(F(F(img[r + 1, c + 1] - img[r + 1, c - 1]) - img[r - 1, c + 1]) + img[r - 1, c - 1]) * cds
    dxs = F(F(F(nxt[r, c + 1] - nxt[r, c - 1]) - prev[r, c + 1]) + prev[r, c - 1]) * cds
    dys = F(F(F(nxt[r + 1, c] - nxt[r - 1, c]) - prev[r + 1, c]) + prev[r - 1, c]) * cds
    return [F(v) for v in dD], F(dxx), F(dyy), F(dss), F(dxy), F(dxs), F(dys)


def adjust_local_extremum(dog_o, octv, layer, r, c):
    """adjustLocalExtrema: returns None or (x, y, octave_packed, size, response, layer, r, c) in base-image pixels."""
    h, w = dog_o[0].shape
    xi = xr = xc = F(0)
    for it in range(MAX_INTERP_STEPS + 1):
        if it == MAX_INTERP_STEPS:
            return None
        dD, dxx, dyy, dss, dxy, dxs, dys = _derivs(dog_o, layer, r, c)
        X = _solve3([[dxx, dxy, dxs], [dxy, dyy, dys], [dxs, dys, dss]], dD)
        xi, xr, xc = F(-X[2]), F(-X[1]), F(-X[0])
        if abs(xi) < 0.5 and abs(xr) < 0.5 and abs(xc) < 0.5:
            break
        if abs(xi) > 2147483647 / 3 or abs(xr) > 2147483647 / 3 or abs(xc) > 2147483647 / 3:
            return None
        c += cv_round(xc)
        r += cv_round(xr)
        layer += cv_round(xi)
        if layer < 1 or layer > N_LAYERS or c < IMG_BORDER or c >= w - IMG_BORDER or r < IMG_BORDER or r >= h - IMG_BORDER:
            return None
    dD, dxx, dyy, dss, dxy, dxs, dys = _derivs(dog_o, layer, r, c)
    t = F(F(F(dD[0] * xc) + F(dD[1] * xr)) + F(dD[2] * xi))
    contr = F(F(dog_o[layer][r, c] * F(1.0 / 255)) + F(t * F(0.5)))
    if F(abs(contr) * F(N_LAYERS)) < F(CONTRAST_THR):
        return None
    tr = F(dxx + dyy)
    det = F(F(dxx * dyy) - F(dxy * dxy))
    if det <= 0 or F(F(tr * tr) * F(EDGE_THR)) >= F(F((EDGE_THR + 1) * (EDGE_THR + 1)) * det):
        return None
    x = F(F(c + xc) * F(1 << octv))
    y = F(F(r + xr) * F(1 << octv))
    packed = octv + (layer << 8) + (cv_round(F(F(xi + F(0.5)) * F(255))) << 16)
    size = F(F(F(SIGMA) * F(np.power(F(2), F(F(layer + xi) / F(N_LAYERS))))) * F(1 << octv)) * F(2)
    return float(x), float(y), packed, float(size), float(abs(contr)), layer, r, c


_P1 = F(0.9997878412794807 * (180 / math.pi))
_P3 = F(-0.3258083974640975 * (180 / math.pi))
_P5 = F(0.1555786518463281 * (180 / math.pi))
_P7 = F(-0.04432655554792128 * (180 / math.pi))


def fast_atan2(y, x):
    """OpenCV's fastAtan2 (degrees in [0, 360), odd polynomial of degree 7), elementwise in float32."""
    y, x = np.asarray(y, F), np.asarray(x, F)
    ax, ay = np.abs(x), np.abs(y)
    big, small = np.maximum(ax, ay), np.minimum(ax, ay)
    c = (small / (big + F(2.220446049250313e-16))).astype(F)  # the float32 sum equals `big` unless it is tiny
    c2 = (c * c).astype(F)
    a = (((((_P7 * c2).astype(F) + _P5).astype(F) * c2).astype(F) + _P3).astype(F) * c2).astype(F)
    a = ((a + _P1).astype(F) * c).astype(F)
    a = np.where(ax >= ay, a, (F(90) - a).astype(F))
    a = np.where(x < 0, (F(180) - a).astype(F), a)
    a = np.where(y < 0, (F(360) - a).astype(F), a)
    return a.astype(F)


def orientation_hist(img, px, py, radius, sigma):
    """calcOrientationHist: (smoothed 36-bin histogram, its maximum); samples in row-major window order."""
    n = ORI_HIST_BINS
    h, w = img.shape
    expf_scale = F(-1.0) / F(F(2.0) * F(sigma) * F(sigma))
    ii, jj = np.meshgrid(np.arange(-radius, radius + 1), np.arange(-radius, radius + 1), indexing="ij")
    y, x = py + ii, px + jj
    ok = (y > 0) & (y < h - 1) & (x > 0) & (x < w - 1)
    y, x, ii, jj = y[ok], x[ok], ii[ok], jj[ok]
    dx = (img[y, x + 1] - img[y, x - 1]).astype(F)
    dy = (img[y - 1, x] - img[y + 1, x]).astype(F)
    wgt = np.exp(((ii * ii + jj * jj).astype(F) * expf_scale).astype(F)).astype(F)
    ori = fast_atan2(dy, dx)
    mag = np.sqrt(((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F)).astype(F)
    b = np.rint((ori * F(n / 360.0)).astype(F)).astype(int)
    b = np.where(b >= n, b - n, b)
    b = np.where(b < 0, b + n, b)
    temp = np.zeros(n, F)
    np.add.at(temp, b, (wgt * mag).astype(F))  # unbuffered: sequential float32 accumulation in sample order
    hist = np.zeros(n, F)
    for i in range(n):
        a = F(F(temp[(i - 2) % n] + temp[(i + 2) % n]) * F(1.0 / 16))
        b2 = F(F(temp[(i - 1) % n] + temp[(i + 1) % n]) * F(4.0 / 16))
        hist[i] = F(F(a + b2) + F(temp[i] * F(6.0 / 16)))
    return hist, F(hist.max()) if n else F(0)


def keypoint_orientations(gauss_o, octv, kp):
    """The oriented keypoints of one refined extremum: list of (x, y, size, angle, response, octave_packed)."""
    x, y, packed, size, response, layer, r, c = kp
    scl_octv = F(F(size) * F(0.5)) / F(1 << octv)
    hist, omax = orientation_hist(gauss_o[layer], c, r, cv_round(F(ORI_RADIUS) * scl_octv), F(ORI_SIG_FCTR) * scl_octv)
    mag_thr = F(omax * F(ORI_PEAK_RATIO))
    n = ORI_HIST_BINS
    out = []
    for j in range(n):
        left, right = hist[(j - 1) % n], hist[(j + 1) % n]
        if hist[j] > left and hist[j] > right and hist[j] >= mag_thr:
            num = F(F(0.5) * F(left - right))
            den = F(F(left - F(F(2) * hist[j])) + right)
            b = F(F(j) + F(num / den))
            b = F(n + b) if b < 0 else (F(b - n) if b >= n else b)
            angle = F(F(360) - F(F(360.0 / n) * b))
            if abs(angle - F(360)) < FLT_EPSILON:
                angle = F(0)
            out.append((x, y, size, float(angle), response, packed))
    return out


def remove_duplicates_sorted(kps):
    """KeyPointsFilter::removeDuplicatedSorted on tuples (x, y, size, angle, response, octave)."""
    if not kps:
        return []
    a = np.array(kps, dtype=np.float64)
    order = np.lexsort((np.arange(len(kps)), -a[:, 5], -a[:, 4], a[:, 3], -a[:, 2], a[:, 1], a[:, 0]))
    out = []
    for i in order:
        k = kps[i]
        if out and out[-1][0] == k[0] and out[-1][1] == k[1] and out[-1][2] == k[2] and out[-1][3] == k[3]:
            continue
        out.append(k)
    return out


def unpack_octave(packed):
    octave = packed & 255
    layer = (packed >> 8) & 255
    octave = octave if octave < 128 else (-128 | octave)
    scale = F(1.0) / F(1 << octave) if octave >= 0 else F(1 << -octave)
    return octave, layer, scale


def descriptor(img, ptx, pty, ori, scl):
    """calcSIFTDescriptor with d = 4, n = 8: 128 integers in [0, 255] (float32 array)."""
    d, n = DESCR_WIDTH, DESCR_HIST_BINS
    h, w = img.shape
    px, py = cv_round(ptx), cv_round(pty)
    cos_t = F(math.cos(float(F(ori)) * (math.pi / 180)))
    sin_t = F(math.sin(float(F(ori)) * (math.pi / 180)))
    bins_per_rad = F(n / 360.0)
    exp_scale = F(-1.0) / F(d * d * 0.5)
    hist_width = F(F(DESCR_SCL_FCTR) * F(scl))
    radius = cv_round(float(hist_width) * 1.4142135623730951 * (d + 1) * 0.5)
    radius = min(radius, int(math.sqrt(float(w) * w + float(h) * h)))
    cos_t = F(cos_t / hist_width)
    sin_t = F(sin_t / hist_width)
    ii, jj = np.meshgrid(np.arange(-radius, radius + 1), np.arange(-radius, radius + 1), indexing="ij")
    ii, jj = ii.ravel(), jj.ravel()
    c_rot = (F(1) * (jj.astype(F) * cos_t).astype(F) - (ii.astype(F) * sin_t).astype(F)).astype(F)
    r_rot = ((jj.astype(F) * sin_t).astype(F) + (ii.astype(F) * cos_t).astype(F)).astype(F)
    rbin = ((r_rot + F(d // 2)).astype(F) - F(0.5)).astype(F)
    cbin = ((c_rot + F(d // 2)).astype(F) - F(0.5)).astype(F)
    r, c = py + ii, px + jj
    ok = (rbin > -1) & (rbin < d) & (cbin > -1) & (cbin < d) & (r > 0) & (r < h - 1) & (c > 0) & (c < w - 1)
    rbin, cbin, r, c, c_rot, r_rot = rbin[ok], cbin[ok], r[ok], c[ok], c_rot[ok], r_rot[ok]
    dx = (img[r, c + 1] - img[r, c - 1]).astype(F)
    dy = (img[r - 1, c] - img[r + 1, c]).astype(F)
    wgt = np.exp(((((c_rot * c_rot).astype(F) + (r_rot * r_rot).astype(F)).astype(F)) * exp_scale).astype(F)).astype(F)
    ang = fast_atan2(dy, dx)
    mag = np.sqrt(((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F)).astype(F)
    hist = np.zeros((d + 2, d + 2, n + 2), F)
    obin_all = ((ang - F(ori)).astype(F) * bins_per_rad).astype(F)
    magw = (mag * wgt).astype(F)
    r0, c0, o0 = np.floor(rbin).astype(int), np.floor(cbin).astype(int), np.floor(obin_all).astype(int)
    rb, cb, ob = (rbin - r0.astype(F)).astype(F), (cbin - c0.astype(F)).astype(F), (obin_all - o0.astype(F)).astype(F)
    o0 = np.where(o0 < 0, o0 + n, o0)
    o0 = np.where(o0 >= n, o0 - n, o0)
    v_r1 = (magw * rb).astype(F)
    v_r0 = (magw - v_r1).astype(F)
    v_rc11 = (v_r1 * cb).astype(F)
    v_rc10 = (v_r1 - v_rc11).astype(F)
    v_rc01 = (v_r0 * cb).astype(F)
    v_rc00 = (v_r0 - v_rc01).astype(F)
    v111 = (v_rc11 * ob).astype(F)
    v110 = (v_rc11 - v111).astype(F)
    v101 = (v_rc10 * ob).astype(F)
    v100 = (v_rc10 - v101).astype(F)
    v011 = (v_rc01 * ob).astype(F)
    v010 = (v_rc01 - v011).astype(F)
    v001 = (v_rc00 * ob).astype(F)
    v000 = (v_rc00 - v001).astype(F)
    base = ((r0 + 1) * (d + 2) + (c0 + 1)) * (n + 2) + o0
    step_c, step_r = n + 2, (d + 2) * (n + 2)
    idx = np.stack([base, base + 1, base + step_c, base + step_c + 1, base + step_r, base + step_r + 1,
                    base + step_r + step_c, base + step_r + step_c + 1], axis=1)
    vals = np.stack([v000, v001, v010, v011, v100, v101, v110, v111], axis=1).astype(F)
    flat = hist.reshape(-1)
    np.add.at(flat, idx.ravel(), vals.ravel())  # unbuffered: the samples' contributions are added one after the other
    dst = np.zeros(d * d * n, F)
    for i in range(d):
        for j in range(d):
            cell = hist[i + 1, j + 1]
            cell[0] = F(cell[0] + cell[n])
            cell[1] = F(cell[1] + cell[n + 1])
            dst[(i * d + j) * n : (i * d + j + 1) * n] = cell[:n]
    nrm2 = F(0)
    for v in dst:
        nrm2 = F(nrm2 + F(v * v))
    thr = F(np.sqrt(nrm2) * F(DESCR_MAG_THR))
    nrm2 = F(0)
    for k in range(dst.size):
        dst[k] = min(dst[k], thr)
        nrm2 = F(nrm2 + F(dst[k] * dst[k]))
    scale = F(F(INT_DESCR_FCTR) / max(F(np.sqrt(nrm2)), F(FLT_EPSILON)))
    return np.clip(np.rint((dst * scale).astype(F)), 0, 255).astype(F)


def detect(gray_u8, taps_fn=gaussian_taps, pyramids=None):
    """Keypoints before descriptors: list of (x, y, size, angle, response, octave_packed) in image coordinates, sorted
    and de-duplicated as detectAndCompute leaves them, plus the pyramids."""
    gauss, dog = pyramids if pyramids is not None else build_pyramids(gray_u8, taps_fn)
    kps = []
    for o in range(len(dog)):
        for layer, r, c in find_candidates(dog[o]):
            kp = adjust_local_extremum(dog[o], o, layer, r, c)
            if kp is not None:
                kps.extend(keypoint_orientations(gauss[o], o, kp))
    kps = remove_duplicates_sorted(kps)
    out = []
    for x, y, size, angle, response, packed in kps:  # firstOctave = -1: back to the coordinates of the input image
        packed = (packed & ~255) | ((packed + FIRST_OCTAVE) & 255)
        out.append((float(F(F(x) * F(0.5))), float(F(F(y) * F(0.5))), float(F(F(size) * F(0.5))), angle, response, packed))
    return out, (gauss, dog)


def describe(kps, gauss):
    desc = np.zeros((len(kps), 128), F)
    for i, (x, y, size, angle, _, packed) in enumerate(kps):
        octave, layer, scale = unpack_octave(packed)
        img = gauss[octave - FIRST_OCTAVE][layer]
        a = F(F(360) - F(angle))
        if abs(a - F(360)) < FLT_EPSILON:
            a = F(0)
        desc[i] = descriptor(img, F(F(x) * scale), F(F(y) * scale), a, F(F(F(size) * scale) * F(0.5)))
    return desc


def detect_and_compute(gray_u8, taps_fn=gaussian_taps):
    """(keypoints float32 [n][2] = (x, y), descriptors uint8 [n][128], full keypoint tuples)."""
    kps, (gauss, _) = detect(gray_u8, taps_fn)
    desc = describe(kps, gauss)
    return np.array([[k[0], k[1]] for k in kps], F).reshape(-1, 2), desc.astype(np.uint8), kps

"""Synthetic correspondence sets shaped like SIFT matches of the bundled datasets.

Used by the parity tests and by bench.py (there is no dataset download and no
OpenCV on the build or GPU machines).  All draws come from
numpy.random.default_rng(seed) so they never disturb the legacy global stream
the RANSAC index draws use.
"""
from __future__ import annotations

import numpy as np

# Intrinsics of dataset 5 (data/5/cfg.yml:15-16 in the reference) and dataset 10
# (data/10/cfg.yml:17-18, fx != fy).
K_DATASET5 = [[2489.14, 0, 968], [0, 2489.14, 648], [0, 0, 1]]
K_DATASET10 = [[689.87, 0, 380.17], [0, 691.04, 251.70], [0, 0, 1]]


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _through_f32(xy):
    """Pixel coordinates pass through float32 once and are then held as float64
    (reference: src/extract_sift.py:126-130)."""
    return xy.astype(np.float32).astype(np.float64)


def two_view_scene(n=10000, inlier_ratio=0.3, planar_ratio=0.5, seed=0, K=K_DATASET5,
                   noise_px=0.3, image_size=(1936, 1296), R=None, T=None):
    """Two-view match set: homogeneous pixel coordinates x1, x2 (3,n) float64.

    Points are x,y~U(-2,2), z~U(4,8); a fraction lies on the plane z = 6 + 0.2 x so that
    the joint E/H loop of the reference terminates.  Camera 2 is [R|T].  A fraction
    (1 - inlier_ratio) of x2 is replaced by uniform image positions.
    Returns dict(x1, x2, K, R, T, is_inlier).
    """
    rng = np.random.default_rng(seed)
    Kn = np.asarray(K, float)
    R = rot_y(0.1) if R is None else R
    T = np.array([1.0, 0.1, 0.05]) if T is None else np.asarray(T, float)
    X = np.empty((3, n))
    X[0] = rng.uniform(-2, 2, n)
    X[1] = rng.uniform(-2, 2, n)
    X[2] = rng.uniform(4, 8, n)
    planar = rng.random(n) < planar_ratio
    X[2, planar] = 6 + 0.2 * X[0, planar]
    p1 = Kn @ X
    p2 = Kn @ (R @ X + T[:, None])
    x1 = p1[:2] / p1[2] + rng.normal(0, noise_px, (2, n))
    x2 = p2[:2] / p2[2] + rng.normal(0, noise_px, (2, n))
    is_inlier = rng.random(n) < inlier_ratio
    nout = int((~is_inlier).sum())
    x2[0, ~is_inlier] = rng.uniform(0, image_size[0], nout)
    x2[1, ~is_inlier] = rng.uniform(0, image_size[1], nout)
    x1 = np.vstack([_through_f32(x1), np.ones((1, n))])
    x2 = np.vstack([_through_f32(x2), np.ones((1, n))])
    return dict(x1=x1, x2=x2, K=K, R=R, T=T, is_inlier=is_inlier, X=X)


def resection_scene(m=50000, outlier_ratio=0.5, seed=0, K=K_DATASET5, noise_px=0.5,
                    R=None, T=None):
    """2D-3D correspondences for the translation stages.

    Returns dict(X (3,m) float32, x_norm (3,m) float64, K, R, T_true) with
    x = pi(R X + T) + noise/f in normalised coordinates, a fraction replaced by outliers.
    Dtypes follow the reference's find_correspondences (src/utils.py:89-91).
    """
    rng = np.random.default_rng(seed)
    f = float(K[0][0])
    R = rot_y(0.07) @ rot_x(-0.03) if R is None else R
    T = np.array([0.4, -0.2, 0.3]) if T is None else np.asarray(T, float)
    X = np.empty((3, m))
    X[0] = rng.uniform(-2, 2, m)
    X[1] = rng.uniform(-2, 2, m)
    X[2] = rng.uniform(4, 8, m)
    X = X.astype(np.float32)
    p = R @ X.astype(np.float64) + T[:, None]
    x = p[:2] / p[2] + rng.normal(0, noise_px / f, (2, m))
    out = rng.random(m) < outlier_ratio
    x[:, out] = rng.uniform(-0.4, 0.4, (2, int(out.sum())))
    # normalised coordinates come from float32 pixel positions in the reference
    Kn = np.asarray(K, float)
    px = _through_f32(Kn[:2, :2] @ x + Kn[:2, 2:3])
    x_norm = np.linalg.inv(Kn) @ np.vstack([px, np.ones((1, m))])
    return dict(X=X, x_norm=x_norm, K=K, R=R, T_true=T, is_outlier=out)


def multi_view_scene(n_cams=5, n_points=900, init_pair=(1, 3), seed=0, K=K_DATASET5, inlier_ratio=0.7,
                     planar_ratio=0.6, noise_px=0.3, image_size=(1936, 1296)):
    """A small multi-camera scene in the shape the pipeline consumes: consecutive-pair matches
    (`x_pairs`), the initial pair with one descriptor column per match, and a synthetic 2D-3D
    matcher that stands in for the OpenCV front-end.  Descriptors are one-hot-free: column j of
    `init_pair_desc` just stores the point id j, and the matcher looks ids up.
    """
    rng = np.random.default_rng(seed)
    Kn = np.asarray(K, float)
    X = np.empty((3, n_points))
    X[0] = rng.uniform(-2, 2, n_points)
    X[1] = rng.uniform(-2, 2, n_points)
    X[2] = rng.uniform(4, 8, n_points)
    planar = rng.random(n_points) < planar_ratio
    X[2, planar] = 6 + 0.2 * X[0, planar]
    Rs = [np.eye(3)]
    Ts = [np.zeros(3)]
    for c in range(1, n_cams):
        Rs.append(Rs[-1] @ rot_y(0.05 + 0.01 * c) @ rot_x(0.01 * (-1) ** c))
        Ts.append(np.array([0.45 * c, 0.05 * c, 0.02 * c]))

    def project(c, idxs, with_outliers):
        p = Kn @ (Rs[c] @ X[:, idxs] + Ts[c][:, None])
        x = p[:2] / p[2] + rng.normal(0, noise_px, (2, len(idxs)))
        if with_outliers:
            out = rng.random(len(idxs)) > inlier_ratio
            x[0, out] = rng.uniform(0, image_size[0], int(out.sum()))
            x[1, out] = rng.uniform(0, image_size[1], int(out.sum()))
        return np.vstack([_through_f32(x), np.ones((1, len(idxs)))])

    x_pairs = []
    for c in range(n_cams - 1):
        ids = rng.choice(n_points, size=int(0.7 * n_points), replace=False)
        x_pairs += [project(c, ids, False), project(c + 1, ids, True)]
    ids0 = rng.choice(n_points, size=int(0.8 * n_points), replace=False)
    a, b = init_pair
    init_pair_dict = {
        "init_pair_indices": list(init_pair),
        "x_init": [project(a, ids0, False), project(b, ids0, True)],
        "init_pair_desc": np.tile(np.arange(len(ids0), dtype=np.float32), (128, 1)),
    }
    img_paths = [f"synthetic://camera/{c}" for c in range(n_cams)]
    obs = {}
    for c in range(n_cams):
        sel = rng.choice(len(ids0), size=int(0.6 * len(ids0)), replace=False)
        px = project(c, ids0[sel], True)
        obs[img_paths[c]] = (sel, px)

    def find_correspondences(img_path, desc_X, X0, K_):
        """Stands in for src/utils.py:65-93: returns (X float32 3xM, x_norm float64 3xM) for the
        descriptor columns present in desc_X (ids in row 0)."""
        sel, px = obs[img_path]
        present = {int(v): j for j, v in enumerate(desc_X[0])}
        keep = [k for k, s in enumerate(sel) if int(s) in present]
        cols = [present[int(sel[k])] for k in keep]
        Xc = np.float32(X0[:, cols])
        xn = np.linalg.inv(np.asarray(K_, float)) @ px[:, keep]
        return Xc, xn

    return dict(K=K, x_pairs=x_pairs, init_pair_dict=init_pair_dict, img_paths=img_paths, find_correspondences=find_correspondences,
                R_true=Rs, T_true=Ts, X_true=X)


def _blob_texture(size, n_blobs, rng):
    """A texture with structure at many scales (Gaussian blobs of random size, contrast and elongation plus fine noise),
    float64 in [0, 255]: something a blob detector finds thousands of stable features on."""
    tex = np.zeros((size, size))
    for _ in range(n_blobs):
        s = float(np.exp(rng.uniform(np.log(1.5), np.log(14.0))))
        cx, cy, a, e = rng.uniform(0, size), rng.uniform(0, size), rng.uniform(-1, 1), rng.uniform(0.6, 1.6)
        r = int(4 * s * max(e, 1 / e)) + 1
        x0, x1, y0, y1 = max(int(cx) - r, 0), min(int(cx) + r + 1, size), max(int(cy) - r, 0), min(int(cy) + r + 1, size)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        tex[y0:y1, x0:x1] += a * np.exp(-(((xx - cx) / e) ** 2 + ((yy - cy) * e) ** 2) / (2 * s * s))
    tex += rng.normal(0, 0.02, tex.shape)
    lo, hi = np.percentile(tex, [0.5, 99.5])
    return np.clip((tex - lo) / (hi - lo), 0, 1) * 255.0


def _sample(tex, u, v):
    """Bilinear lookup of tex at fractional (column u, row v); coordinates are clamped to the texture."""
    h, w = tex.shape
    u = np.clip(u, 0, w - 1.001)
    v = np.clip(v, 0, h - 1.001)
    u0, v0 = np.floor(u).astype(int), np.floor(v).astype(int)
    fu, fv = u - u0, v - v0
    return (tex[v0, u0] * (1 - fu) * (1 - fv) + tex[v0, u0 + 1] * fu * (1 - fv) + tex[v0 + 1, u0] * (1 - fu) * fv
            + tex[v0 + 1, u0 + 1] * fu * fv)


def render_views(n_cams=10, width=1936, height=1296, seed=0, K=None, texture_size=3072, n_blobs=9000):
    """Images of a piecewise planar textured scene seen by n_cams cameras moving sideways: a large tilted background
    plane z = 6 + 0.2 x and four fronto-parallel rectangles at depths 4.2 .. 5.5 in front of it (so the scene is far from
    one plane).  Returns dict(images uint8 [n][h][w], K, R_true, T_true); camera c maps X to K (R_c X + T_c).  With the
    default K the intrinsics are dataset 5's scaled to the requested image width."""
    rng = np.random.default_rng(seed)
    if K is None:
        s = width / 1936.0
        K = [[2489.14 * s, 0, 968 * s], [0, 2489.14 * s, height / 2.0], [0, 0, 1]]
    Kn = np.asarray(K, float)
    tex_bg = _blob_texture(texture_size, n_blobs, rng)
    ext_bg = 4.5  # the background texture covers x, y in [-ext, ext]
    # x0, x1, y0, y1, z of the front rectangles (nearest last, so that it is drawn over the others)
    rects = [(-0.2, 0.9, -0.1, 0.9, 5.5), (-2.1, -1.0, -0.2, 0.8, 5.0), (-1.4, -0.1, -1.0, -0.15, 4.6), (0.1, 1.2, -0.9, -0.2, 4.2)]
    tex_fg = [_blob_texture(texture_size // 3, n_blobs // 6, rng) for _ in rects]
    Rs, Ts = [np.eye(3)], [np.zeros(3)]
    for c in range(1, n_cams):
        Rs.append(Rs[-1] @ rot_y(0.022 + 0.002 * (c % 3)) @ rot_x(0.004 * (-1) ** c))
        Ts.append(np.array([-0.16 * c, 0.012 * c, 0.008 * c]))
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    rays_cam = np.linalg.inv(Kn) @ np.stack([xx.ravel(), yy.ravel(), np.ones(xx.size)])
    images = np.empty((n_cams, height, width), np.uint8)
    for c in range(n_cams):
        d = Rs[c].T @ rays_cam
        C0 = -Rs[c].T @ Ts[c]
        # background: n . X = 6 with n = (-0.2, 0, 1)
        n = np.array([-0.2, 0.0, 1.0])
        t_bg = (6.0 - n @ C0) / (n @ d)
        X = C0[:, None] + d * t_bg
        img = _sample(tex_bg, (X[0] + ext_bg) / (2 * ext_bg) * (texture_size - 1), (X[1] + ext_bg) / (2 * ext_bg) * (texture_size - 1))
        for fg, tex in zip(rects, tex_fg):
            t_fg = (fg[4] - C0[2]) / d[2]
            Xf = C0[:, None] + d * t_fg
            hit = (t_fg > 0) & (Xf[0] >= fg[0]) & (Xf[0] <= fg[1]) & (Xf[1] >= fg[2]) & (Xf[1] <= fg[3])
            ts = tex.shape[0] - 1
            img_fg = _sample(tex, (Xf[0] - fg[0]) / (fg[1] - fg[0]) * ts, (Xf[1] - fg[2]) / (fg[3] - fg[2]) * ts)
            img = np.where(hit, img_fg, img)
        img = img + rng.normal(0, 1.0, img.shape)  # sensor noise
        images[c] = np.clip(np.rint(img), 0, 255).astype(np.uint8).reshape(height, width)
    return dict(images=images, K=K, R_true=Rs, T_true=Ts)


def write_image_dataset(directory, views, init_pair=(4, 6)):
    """Writes the rendered views as PNG files with a cfg.yml in the layout of the reference's data/<dataset> folders
    (camera.focal_length, camera.principal_point, image_file_names, initial_pair).  Returns the image paths."""
    import os

    from PIL import Image

    os.makedirs(directory, exist_ok=True)
    names = [f"view_{c:02d}.png" for c in range(len(views["images"]))]
    for name, img in zip(names, views["images"]):
        Image.fromarray(img, mode="L").save(os.path.join(directory, name))
    K = views["K"]
    with open(os.path.join(directory, "cfg.yml"), "w") as f:
        f.write("camera:\n")
        f.write(f"  focal_length: [{K[0][0]}, {K[1][1]}]\n")
        f.write(f"  principal_point: [{K[0][2]}, {K[1][2]}]\n")
        f.write("image_file_names:\n")
        for name in names:
            f.write(f"  - {name}\n")
        f.write(f"initial_pair: [{init_pair[0]}, {init_pair[1]}]\n")
    return [os.path.join(directory, n) for n in names]

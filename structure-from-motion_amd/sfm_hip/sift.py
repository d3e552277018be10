"""SIFT on the GPU with the call shape of cv2.SIFT_create().detectAndCompute(gray, None), which is how the reference
obtains its features (src/extract_sift.py:104-114, src/utils.py:78-80).

The scale space, the extrema search, the sub-pixel refinement, the orientation histograms and the descriptors run in
libsfm_hip.so (csrc/sfm_sift.hip); the host only sorts the few thousand keypoints into OpenCV's order, drops the
duplicates (KeyPointsFilter::removeDuplicatedSorted) and rescales them from the doubled image (firstOctave = -1).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _capi as C

DEFAULT_CAPACITY = 1 << 17


class SiftResult:
    """keypoints float32 [n][2] (x, y), descriptors uint8 [n][128], and the full records float32 [n][6]
    (x, y, size, angle, response, packed octave bits) in OpenCV's order."""

    def __init__(self, records, descriptors):
        self.records = records
        self.keypoints = np.ascontiguousarray(records[:, :2])
        self.descriptors = descriptors


def _full_order(raw, rows):
    """OpenCV's keypoint order of the given rows: (x, y, size desc, angle, response desc, octave desc, input position)."""
    r = raw[rows]
    octave = r[:, 5].view(np.int32).astype(np.int64)
    return rows[np.lexsort((rows, -octave, -r[:, 4].astype(np.float64), r[:, 3], -r[:, 2].astype(np.float64), r[:, 1], r[:, 0]))]


def _sort_and_dedupe(raw):
    """KeyPointsFilter::removeDuplicatedSorted: order by (x, y, size desc, angle, response desc, octave desc), then
    keep the first of every run with equal (x, y, size, angle)."""
    n = raw.shape[0]
    if n == 0:
        return raw
    xy = np.ascontiguousarray(raw[:, :2])
    if np.all(xy >= 0):
        # non-negative float32 order like their bit patterns: one sort on a 64-bit (x, y) key, then only the runs that
        # share a position (a keypoint with several orientations) need the remaining keys
        bits = xy.view(np.uint32).astype(np.uint64)
        key = (bits[:, 0] << np.uint64(32)) | bits[:, 1]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        tied = np.zeros(n, bool)
        same = ks[1:] == ks[:-1]
        tied[1:] |= same
        tied[:-1] |= same
        if tied.any():
            pos = np.flatnonzero(tied)
            order[pos] = _full_order(raw, order[pos])  # runs stay where they are: the (x, y) key is their leading key
    else:
        order = _full_order(raw, np.arange(n))
    s = raw[order]
    keep = np.ones(n, bool)
    keep[1:] = np.any(s[1:, :4] != s[:-1, :4], axis=1)
    return s[keep]


def detect_and_compute(gray, capacity=DEFAULT_CAPACITY):
    """gray: uint8 [h][w] (host array or device tensor).  Returns a SiftResult."""
    torch = C.torch_cuda()
    lib = C.load_library()
    g = gray if isinstance(gray, torch.Tensor) else torch.from_numpy(np.array(gray, dtype=np.uint8, order="C")).cuda()
    if g.dtype != torch.uint8 or g.dim() != 2:
        raise ValueError("gray must be a uint8 image [h][w]")
    g = g.contiguous()
    h, w = int(g.shape[0]), int(g.shape[1])
    nbytes = int(lib.sfm_sift_workspace_bytes(w, h))
    if nbytes == 0:
        raise ValueError(f"image size {w} x {h} is not supported")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    while True:
        kps = torch.empty((capacity, 6), dtype=torch.float32, device="cuda")
        count = ctypes.c_int32(0)
        rc = lib.sfm_sift_detect(C.dptr(g), w, h, C.dptr(ws), nbytes, C.dptr(kps), capacity, ctypes.byref(count), C.stream_ptr())
        if rc == 100002 and count.value > capacity:  # SFM_ETOOLARGE: more keypoints than the buffer holds
            capacity = int(count.value) + 1024
            continue
        C.check(rc)
        break
    raw = kps[: count.value].cpu().numpy()
    rec = _sort_and_dedupe(raw)
    # firstOctave = -1: back to the coordinates of the input image (SIFT_Impl::detectAndCompute)
    rec = rec.copy()
    rec[:, 0] *= np.float32(0.5)
    rec[:, 1] *= np.float32(0.5)
    rec[:, 2] *= np.float32(0.5)
    packed = rec[:, 5].view(np.int32)
    rec[:, 5] = ((packed & ~255) | ((packed - 1) & 255)).astype(np.int32).view(np.float32)
    n = rec.shape[0]
    desc = torch.empty((n, 128), dtype=torch.uint8, device="cuda")
    if n:
        d_rec = torch.from_numpy(np.ascontiguousarray(rec)).cuda()
        C.check(lib.sfm_sift_describe(C.dptr(ws), w, h, C.dptr(d_rec), n, C.dptr(desc), C.stream_ptr()))
    return SiftResult(rec, desc.cpu().numpy())


def gaussian_taps(sigma):
    """The float32 taps the library blurs with for one sigma."""
    lib = C.load_library()
    taps = np.zeros(64, np.float32)
    n = ctypes.c_int32(0)
    C.check(lib.sfm_sift_gaussian_taps(float(sigma), C.hptr(taps), ctypes.byref(n)))
    return taps[: n.value].copy()


def read_modes_differ():
    """True when read_gray decodes with OpenCV: its two read modes may give images that differ in the last bit (colour
    conversion before or after the decoder's own), so features are then kept per (file, mode); Pillow gives one image."""
    try:
        import cv2  # noqa: F401
    except ImportError:
        return False
    return True


def read_gray(img_path, mode):
    """The grayscale image the reference feeds to SIFT.  mode "bgr2gray": cv2.imread + cvtColor(BGR2GRAY)
    (src/extract_sift.py:104-107); mode "grayscale": cv2.imread(IMREAD_GRAYSCALE) (src/utils.py:78).  Uses OpenCV when
    it is installed, otherwise Pillow (same ITU-R 601 luma weights; the JPEG decoders may differ in the last bit)."""
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None:
        if mode == "bgr2gray":
            return cv2.cvtColor(cv2.imread(img_path), cv2.COLOR_BGR2GRAY)
        if mode == "grayscale":
            return cv2.imread(img_path, cv2.IMREAD_GRAYSCALE)
        raise ValueError(mode)
    if mode not in ("bgr2gray", "grayscale"):
        raise ValueError(mode)
    from PIL import Image

    with Image.open(img_path) as im:
        return np.asarray(im.convert("L"), dtype=np.uint8)

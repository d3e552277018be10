"""Descriptor matching on the GPU for the feature front-end (src/extract_sift.py:117-123, src/utils.py:82-86).

The reference matches with ``cv2.BFMatcher().knnMatch(query, train, k=2)`` and a ratio test.  Here the 2-NN search runs
on the int8 matrix cores (exact integer squared distances); SIFT detection runs on the GPU as well (sfm_hip/sift.py).
Features of an image are kept per (path, read mode) so that SIFT runs once per image instead of once per call: the
reference detects every image again in estimate_translation and in refine_translation.
"""
from __future__ import annotations

import os

import numpy as np

from . import _capi as C
from . import ops

# optional directory for per-image feature files (<image name>.<mode>.sift.npz); unset = memory only
FEATURE_CACHE_DIR = os.environ.get("SFM_HIP_FEATURE_CACHE")
_features = {}


class DescriptorSet:
    """Descriptors resident on the GPU in the matcher's layout (int8, shifted by -128, with squared norms)."""

    def __init__(self, desc):
        torch = C.torch_cuda()
        d = np.asarray(desc)
        if d.ndim != 2 or d.shape[1] != 128:
            raise ValueError("descriptors must be [n][128]")
        if d.dtype != np.uint8:
            d = np.ascontiguousarray(d, dtype=np.float32)
        self.n = d.shape[0]
        self.host = d  # the descriptors as given (match_keypoints returns the matched ones)
        self.d8, self.norm = ops.descriptors_to_i8(torch.from_numpy(np.ascontiguousarray(d)).cuda())


def _as_set(desc):
    return desc if isinstance(desc, DescriptorSet) else DescriptorSet(desc)


def knn_match_ratio(desc_query, desc_train, ratio):
    """The good matches of ``knnMatch(desc_query, desc_train, k=2)`` under ``m.distance < ratio * n.distance``, in query
    order: (query_idx int64[G], train_idx int64[G], distance float32[G]).  Arguments: [n][128] arrays or DescriptorSets."""
    q, t = _as_set(desc_query), _as_set(desc_train)
    if q.n == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32)
    tidx, d1, _, good, _ = ops.knn2_ratio(q.d8, q.norm, t.d8, t.norm, ratio)
    qi = good.nonzero().flatten()
    return qi.cpu().numpy(), tidx[qi].to(dtype=C.torch_cuda().int64).cpu().numpy(), d1[qi].cpu().numpy()


def image_features(img_path, mode):
    """SIFT keypoints (float32 [n][2]) and descriptors (uint8 [n][128]) of one image, detected once per (path, mode).
    mode "bgr2gray": cv2.imread + cvtColor (src/extract_sift.py:104-114); mode "grayscale": cv2.imread(IMREAD_GRAYSCALE)
    (src/utils.py:78-80)."""
    if mode not in ("bgr2gray", "grayscale"):
        raise ValueError(mode)
    # the two read modes only differ inside OpenCV's decoders (whichever detector then runs on the image); Pillow gives one
    # gray image per file, and only then do the two modes share their features
    from . import sift

    per_mode = os.environ.get("SFM_HIP_SIFT", "hip") == "cv2" or sift.read_modes_differ()
    key = (os.path.abspath(img_path), mode if per_mode else "gray")
    if key in _features:
        return _features[key]
    cache = None
    if FEATURE_CACHE_DIR:
        cache = os.path.join(FEATURE_CACHE_DIR, f"{os.path.basename(img_path)}.{key[1]}.sift.npz")
        if os.path.exists(cache):
            with np.load(cache) as z:
                _features[key] = (z["keypoints"], z["descriptors"])
            return _features[key]
    feats = detect_features(img_path, mode)
    _features[key] = feats
    if cache:
        os.makedirs(FEATURE_CACHE_DIR, exist_ok=True)
        np.savez(cache, keypoints=feats[0], descriptors=feats[1])
    return feats


class _Decoder:
    """Decodes image files on host threads ahead of their use: once a list of paths is announced (prefetch), asking for
    image i makes sure images i .. i + AHEAD are being decoded.  Decoding a 2.5-megapixel file takes 15-20 ms on one
    core, several times what the GPU spends on its SIFT features."""

    AHEAD = 16

    def __init__(self):
        self.order, self.index, self.futures, self.pool = [], {}, {}, None

    def prefetch(self, paths, mode):
        self.order = [(os.path.abspath(p), mode) for p in paths]
        self.index = {k: i for i, k in enumerate(self.order)}

    def get(self, img_path, mode):
        from . import sift

        key = (os.path.abspath(img_path), mode)
        i = self.index.get(key)
        if i is None:
            return sift.read_gray(img_path, mode)
        if self.pool is None:
            from concurrent.futures import ThreadPoolExecutor

            try:  # this job's share of the host cores, where the launcher states it
                share = int(os.environ.get("OMP_NUM_THREADS", "0"))
            except ValueError:
                share = 0
            cores = share if share > 0 else (os.cpu_count() or 2) - 1
            self.pool = ThreadPoolExecutor(max_workers=max(1, min(16, cores)))
        for k in self.order[i : i + self.AHEAD]:
            if k not in self.futures:
                self.futures[k] = self.pool.submit(sift.read_gray, k[0], k[1])
        return self.futures.pop(key).result()


_decoder = _Decoder()


def prefetch_images(img_paths, mode):
    """Announces the images that image_features will be asked for, in order, so that they are decoded ahead of time."""
    _decoder.prefetch(img_paths, mode)


def detect_features(img_path, mode):
    """SIFT of one image on the GPU (sfm_hip/sift.py).  SFM_HIP_SIFT=cv2 runs OpenCV's detector instead, for users who
    have it installed and want the reference's features bit for bit."""
    from . import sift

    gray = _decoder.get(img_path, mode)
    if os.environ.get("SFM_HIP_SIFT", "hip") == "cv2":
        import cv2

        kps, desc = cv2.SIFT_create().detectAndCompute(gray, None)
        kp = np.float32([k.pt for k in kps]).reshape(-1, 2)
        return kp, (np.ascontiguousarray(desc).astype(np.uint8) if desc is not None else np.zeros((0, 128), np.uint8))
    res = sift.detect_and_compute(gray)
    return res.keypoints, res.descriptors


_sets = {}


def image_descriptor_set(img_path, mode):
    """The descriptors of image_features(img_path, mode) resident on the GPU in the matcher's layout, built once per image
    (an image is matched several times: against both neighbours, and against the reconstructed points in two stages)."""
    kp, desc = image_features(img_path, mode)
    hit = _sets.get(id(desc))
    if hit is None or hit[0] is not desc:
        if len(_sets) > 4096:
            _sets.clear()
        hit = _sets[id(desc)] = (desc, DescriptorSet(desc) if desc.shape[0] else None)
    return hit[1] if hit[1] is not None else desc


def match_keypoints(kp1, desc1, kp2, desc2, ratio=0.4):
    """The array part of compute_sift_keypoints (src/extract_sift.py:116-135): returns
    (x1 3xN float64 homogeneous, x2 3xN, descriptors of view 1 128xN float32) for the good matches.  desc1, desc2: [n][128]
    arrays or DescriptorSets."""
    qi, ti, _ = knn_match_ratio(desc1, desc2, ratio)
    one = np.ones((1, qi.size))
    x1 = np.vstack((np.float32(kp1[qi]).T.reshape(2, -1), one))
    x2 = np.vstack((np.float32(kp2[ti]).T.reshape(2, -1), one))
    host1 = desc1.host if isinstance(desc1, DescriptorSet) else np.asarray(desc1)
    return x1, x2, host1[qi].astype(np.float32).T


def correspondences_from_features(kp, desc, desc_X, X0, K, ratio=0.75):
    """The array part of find_correspondences (src/utils.py:82-93): image features against the descriptors of the
    reconstructed points; returns (X float32 3xM, x_norm float64 3xM).  desc_X: 128xN array or a DescriptorSet."""
    train = desc_X if isinstance(desc_X, DescriptorSet) else DescriptorSet(np.ascontiguousarray(np.asarray(desc_X).T))
    qi, ti, _ = knn_match_ratio(desc, train, ratio)
    x = np.float32(kp[qi]).T.reshape(2, -1)
    X = np.float32(np.asarray(X0)[:, ti])
    x_h = np.vstack((x, np.ones((1, x.shape[1]))))
    return X, np.linalg.inv(K) @ x_h

"""2-point translation RANSAC with the reference's names (src/estimate_translation.py), on the GPU.

All cameras are scored in one launch.  The index pairs come from the global legacy np.random
stream, 2 x 100000 integers per camera in camera order, exactly as the reference draws them.
"""
from __future__ import annotations

import numpy as np

from . import _capi as C
from . import legacy_rng
from . import ops
from . import utils
from .utils import log_execution_time

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover

    def tqdm(it, **kw):
        return it


NUM_ITERATIONS = 100000  # src/estimate_translation.py:57
MAX_CAMERAS_PER_LAUNCH = 64


def _pack_camera(x_norm, X, R):
    import torch

    x = np.ascontiguousarray(np.asarray(x_norm, np.float64))
    if x.shape[0] == 2:
        x = np.vstack([x, np.ones((1, x.shape[1]))])
    Xa = np.asarray(X)
    Xa = np.ascontiguousarray(Xa if Xa.dtype == np.float32 else Xa.astype(np.float64))
    return ops.pack_points(C.dev(np.asarray(R, np.float64).reshape(9)), torch.from_numpy(Xa).cuda(), C.dev(x))


def robust_estimate_T_batch(cameras, K, pixel_threshold, num_iterations=NUM_ITERATIONS):
    """cameras: list of (x_norm, X, R).  Returns a list of T (3,) or None, one per camera
    (src/estimate_translation.py:43-77 for each)."""
    import torch

    threshold = 4 * pixel_threshold / K[0][0]
    results = []
    for s in range(0, len(cameras), MAX_CAMERAS_PER_LAUNCH):
        group = cameras[s : s + MAX_CAMERAS_PER_LAUNCH]
        sizes = [int(np.asarray(c[0]).shape[1]) for c in group]
        # == np.random.randint(0, m, size=2 * num_iterations) per camera, in camera order, from the global legacy stream
        idx = np.stack([legacy_rng.randint(m, 2 * num_iterations).reshape(num_iterations, 2) for m in sizes])
        pts = torch.cat([_pack_camera(*c) for c in group])
        best, T, _ = ops.t_ransac(pts, sizes, C.dev(idx, np.int32), threshold)
        count, _ = ops.decode_best(best.cpu().numpy())
        T = T.cpu().numpy()
        for c in range(len(group)):
            results.append(T[c].copy() if count[c] > 0 else None)  # best_T stays None without inliers (:58-59,:73)
    return results


def robust_estimate_T(x_norm, X, K, R, pixel_threshold):
    """Reference: src/estimate_translation.py:43-77."""
    return robust_estimate_T_batch([(x_norm, X, R)], K, pixel_threshold)[0]


@log_execution_time
def estimate_translation(K, desc_X, X0, absolute_rotations, img_paths, pixel_threshold):
    """Reference: src/estimate_translation.py:7-40.  The 2D-3D matching of every image is the
    OpenCV front-end (utils.find_correspondences); the RANSAC of all cameras is one GPU batch."""
    cams = []
    for i in tqdm(range(len(img_paths)), desc="Estimating translation vectors"):
        X_corr, x_corr_norm = utils.find_correspondences(img_paths[i], desc_X, X0, K)
        cams.append((x_corr_norm, X_corr, absolute_rotations[i]))
    return robust_estimate_T_batch(cams, K, pixel_threshold)


def estimate_T_2point(x_norm, X, R):
    """Reference: src/estimate_translation.py:80-106 for one sample of two correspondences."""
    pts = _pack_camera(x_norm, X, R)
    T = ops.t_models(pts, C.dev(np.array([[0, 1]]), np.int32))
    return T.cpu().numpy()[0]


def count_inliers(x_norm, X, R, T_estimated, threshold):
    """Reference: src/estimate_translation.py:109-129 for one translation."""
    pts = _pack_camera(x_norm, X, R)
    c = ops.t_count(pts, C.dev(np.asarray(T_estimated, np.float64).reshape(1, 3)), threshold)
    return int(c.cpu().numpy()[0])

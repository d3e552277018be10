"""Helpers with the names and signatures of the reference's src/utils.py.

triangulate_3D_point_DLT runs on the GPU; the one-line array helpers are plain NumPy as in
the reference.  find_correspondences belongs to the feature front-end: SIFT detection (sfm_hip/sift.py) and the
brute-force 2-nearest-neighbour matching (sfm_hip/matching.py) run on the GPU.
"""
from __future__ import annotations

import logging
import os
import pickle
import time

import numpy as np

from . import _capi as C
from . import ops


def filter_3D_points(X):
    """Reference: src/utils.py:9-14 (row-wise distance filter at 5x the 90th percentile)."""
    X_mean = np.mean(X, axis=0)
    distances = np.linalg.norm(X - X_mean, axis=1)
    return X[distances <= 5 * np.quantile(distances, 0.9)]


def triangulate_3D_point_DLT(P, points):
    """Reference: src/utils.py:17-36.  P = [P1, P2] (3x4), points = [x1, x2] (>=2 x N).
    Returns the 4xN array with last row 1 (batched on the GPU, one point per lane)."""
    points1, points2 = points
    P1, P2 = P
    p1 = np.ascontiguousarray(np.asarray(points1, dtype=np.float64)[:2])
    p2 = np.ascontiguousarray(np.asarray(points2, dtype=np.float64)[:2])
    if p1.shape[1] == 0:
        return np.array([]).T
    X = ops.triangulate(C.dev(np.asarray(P1, np.float64).reshape(12)), C.dev(np.asarray(P2, np.float64).reshape(12)), C.dev(p1), C.dev(p2))
    return X.cpu().numpy()


def pflat(X):
    """Reference: src/utils.py:39-41."""
    return X / X[-1, :]


def normalize_K(K, xs):
    """Reference: src/utils.py:44-45."""
    return np.linalg.inv(K) @ xs


def cartesian_to_homogeneous(cartesian_points):
    """Reference: src/utils.py:48-53."""
    return np.vstack((cartesian_points, np.ones((1, cartesian_points.shape[1]))))


def homogeneous_to_cartesian(points):
    """Reference: src/utils.py:56-57."""
    return points[:-1, :] / points[-1, :]


def skew_symmetric_mat(v):
    """Reference: src/utils.py:60-62."""
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def find_correspondences(img_path, desc_X, X0, K):
    """Reference: src/utils.py:65-93.  SIFT on one image (once per image), 2-NN brute-force match against the stored
    descriptors, ratio 0.75, all on the GPU; returns (X float32 3xM, x_norm float64 3xM)."""
    from . import matching

    kp, _ = matching.image_features(img_path, "grayscale")
    return matching.correspondences_from_features(kp, matching.image_descriptor_set(img_path, "grayscale"), desc_X, X0, K, ratio=0.75)


def log_execution_time(func):
    """Reference: src/utils.py:96-108."""

    def wrapper(*args, **kwargs):
        t0 = time.time()
        result = func(*args, **kwargs)
        logging.info(f"Elapsed Time for {func.__name__}: {time.time() - t0:.2f} seconds")
        return result

    wrapper.__name__ = getattr(func, "__name__", "wrapper")
    wrapper.__doc__ = func.__doc__
    return wrapper


def save_x_pairs(data, filename, save_location):
    """Reference: src/utils.py:112-115."""
    with open(os.path.join(save_location, filename), "wb") as f:
        pickle.dump(data, f)


def load_x_pairs(filename, save_location):
    """Reference: src/utils.py:119-124.  Only for caches this package (or the reference run by
    the same user) wrote: unpickling executes code from the file."""
    path = os.path.join(save_location, filename)
    if os.path.exists(path):
        with open(path, "rb") as f:
            return pickle.load(f)
    return None

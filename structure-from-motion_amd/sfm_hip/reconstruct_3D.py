"""Initial reconstruction with the reference's names (src/reconstruct_3D.py)."""
from __future__ import annotations

import logging

import numpy as np

from .ransac_algorithm import run_ransac
from .utils import log_execution_time, normalize_K, triangulate_3D_point_DLT


@log_execution_time
def compute_absolute_rotations(relative_rotations):
    """Reference: src/reconstruct_3D.py:40-60: R_abs[0] = I, R_abs[i+1] = R_abs[i] @ R_rel[i]."""
    absolute_rotations = [np.eye(3)]
    for R_rel in relative_rotations:
        absolute_rotations.append(np.dot(absolute_rotations[-1], R_rel))
    return absolute_rotations


@log_execution_time
def reconstruct_initial_3D_points(init_pair_dict, K, R_init_1, pixel_threshold):
    """Reference: src/reconstruct_3D.py:63-95.  E/H RANSAC on the initial pair, then DLT
    triangulation of ALL its matches (not only the inliers), rotated into the world frame."""
    x1 = init_pair_dict["x_init"][0]
    x2 = init_pair_dict["x_init"][1]
    R, T, inliers = run_ransac(K, x1, x2, pixel_threshold)
    P1 = np.concatenate((np.eye(3), np.zeros((3, 1))), axis=1)
    P2 = np.hstack((R, T))
    X0 = triangulate_3D_point_DLT([P1, P2], [normalize_K(K, x1), normalize_K(K, x2)])
    X0 = X0[:3, :] / X0[3, :]
    return np.dot(R_init_1.T, X0), inliers


@log_execution_time
def run_reconstruction(relative_rotations, init_pair_dict, K, pixel_threshold):
    """Reference: src/reconstruct_3D.py:7-37."""
    absolute_rotations = compute_absolute_rotations(relative_rotations)
    X0, inliers = reconstruct_initial_3D_points(
        init_pair_dict, K, absolute_rotations[init_pair_dict["init_pair_indices"][0]], pixel_threshold
    )
    logging.info(f"Number of inliers for reconstruction of initial pair: {len(inliers)}")
    return X0, absolute_rotations, inliers

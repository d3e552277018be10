"""The stage sequence of the reference driver (run_sfm.py:57-80) as a function, for callers
that already hold the matched points (no OpenCV needed) and for end-to-end tests."""
from __future__ import annotations

import logging

import numpy as np

from . import estimate_translation as _et
from . import refine_translation as _rt
from .ransac_algorithm import estimate_R
from .reconstruct_3D import run_reconstruction


def run_geometry(K, x_pairs, init_pair_dict, img_paths, pixel_threshold):
    """Relative rotations -> initial reconstruction -> robust T -> refined T.

    Follows run_sfm.py:59-80 literally, including that the refinement stage receives the
    descriptors of the inlier points but the FULL point array X0 (run_sfm.py:66-67,78-80).
    Returns a dict with the keys the reference pickles (run_sfm.py:86-93) plus X0 and inliers."""
    R_list = estimate_R(K, x_pairs, pixel_threshold)
    X0, absolute_rotations, inliers = run_reconstruction(R_list, init_pair_dict, K, pixel_threshold)
    desc_X = init_pair_dict["init_pair_desc"]
    desc_X_inliers = desc_X[:, inliers]
    logging.info(f"Total number of inliers for initial reconstruction: {np.sum(inliers)}")
    estimated_Ts = _et.estimate_translation(K, desc_X, X0, absolute_rotations, img_paths, pixel_threshold)
    refined_Ts = _rt.refine_translation(K, img_paths, desc_X_inliers, X0, absolute_rotations, estimated_Ts)
    return {
        "K": K,
        "absolute_rotations": absolute_rotations,
        "refined_Ts": refined_Ts,
        "x_pairs": x_pairs,
        "nr_images": len(img_paths),
        "X0": X0,
        "inliers": inliers,
        "estimated_Ts": estimated_Ts,
    }


def triangulate_point_cloud(sfm_data_dict):
    """The point cloud the reference's visualiser draws (utils.py:140-156 there): for every
    consecutive camera pair, DLT triangulation of its matches with the PIXEL cameras K[R|T] and
    pixel coordinates, followed by filter_3D_points applied to the 4xN array as the reference
    applies it (row-wise; in practice it keeps everything).  Returns one 4xN array per pair."""
    from .utils import filter_3D_points, triangulate_3D_point_DLT

    K = np.asarray(sfm_data_dict["K"], float)
    R = sfm_data_dict["absolute_rotations"]
    T = sfm_data_dict["refined_Ts"]
    x_pairs = sfm_data_dict["x_pairs"]
    clouds = []
    for i in range(sfm_data_dict["nr_images"] - 1):
        P1 = K @ np.hstack((R[i], np.reshape(T[i], (3, 1))))
        P2 = K @ np.hstack((R[i + 1], np.reshape(T[i + 1], (3, 1))))
        X = triangulate_3D_point_DLT([P1, P2], [x_pairs[2 * i], x_pairs[2 * i + 1]])
        clouds.append(filter_3D_points(X))
    return clouds

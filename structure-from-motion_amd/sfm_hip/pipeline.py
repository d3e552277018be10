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

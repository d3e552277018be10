"""SIFT pair extraction with the reference's names (src/extract_sift.py).

This is the OpenCV feature front-end, outside the robust-estimation path: it is kept only so
that the reference's driver finds the same function and the same cache files.  Needs cv2
unless both cache files already exist next to the images.
"""
from __future__ import annotations

import logging
import os
import time

import numpy as np

from .utils import cartesian_to_homogeneous, load_x_pairs, save_x_pairs


def compute_sift_keypoints(image_path1, image_path2):
    """Reference: src/extract_sift.py:88-135: SIFT on both images, 2-NN brute-force matching,
    ratio test 0.4; returns (x1 3xN, x2 3xN, descriptors of view 1 128xN)."""
    try:
        import cv2
    except ImportError as e:  # pragma: no cover
        raise ImportError("compute_sift_keypoints needs OpenCV (cv2)") from e
    grays = [cv2.cvtColor(cv2.imread(p), cv2.COLOR_BGR2GRAY) for p in (image_path1, image_path2)]
    sift = cv2.SIFT_create()
    (kp1, d1), (kp2, d2) = [sift.detectAndCompute(g, None) for g in grays]
    good = [m for m, n in cv2.BFMatcher().knnMatch(d1, d2, k=2) if m.distance < 0.4 * n.distance]
    x1 = cartesian_to_homogeneous(np.float32([kp1[m.queryIdx].pt for m in good]).T)
    x2 = cartesian_to_homogeneous(np.float32([kp2[m.trainIdx].pt for m in good]).T)
    return x1, x2, np.array([d1[m.queryIdx] for m in good]).T


def process_sift_for_image_pairs(img_paths, init_pair, dataset):
    """Reference: src/extract_sift.py:10-85.  Consecutive pairs then the initial pair, each
    cached as a pickle in the image directory under the reference's file names."""
    x_pairs_file = f"x_pairs_dataset_{dataset}.pkl"
    init_file = f"init_pair_dataset_{dataset}.pkl"
    where = os.path.dirname(img_paths[0])
    x_pairs = load_x_pairs(x_pairs_file, where)
    if x_pairs is None:
        x_pairs = []
        for i in range(len(img_paths) - 1):
            t0 = time.time()
            x1, x2, _ = compute_sift_keypoints(img_paths[i], img_paths[i + 1])
            logging.info(f"SIFT extraction completed for pair {i}-{i+1} in {time.time() - t0:.2f} seconds")
            x_pairs.extend([x1, x2])
        save_x_pairs(x_pairs, x_pairs_file, where)
        logging.info(f"x_pairs saved to {os.path.join(where, x_pairs_file)}")
    else:
        logging.info(f"x_pairs loaded from {os.path.join(where, x_pairs_file)}")
    init_pair_dict = load_x_pairs(init_file, where)
    if init_pair_dict is None:
        t0 = time.time()
        a, b, desc = compute_sift_keypoints(img_paths[init_pair[0]], img_paths[init_pair[1]])
        logging.info(f"Initial SIFT extraction completed in {time.time() - t0:.2f} seconds")
        init_pair_dict = {"init_pair_indices": init_pair, "x_init": [a, b], "init_pair_desc": desc}
        save_x_pairs(init_pair_dict, init_file, where)
        logging.info(f"Initial pair saved to {os.path.join(where, init_file)}")
    else:
        logging.info(f"Initial pair loaded from {os.path.join(where, init_file)}")
    return x_pairs, init_pair_dict

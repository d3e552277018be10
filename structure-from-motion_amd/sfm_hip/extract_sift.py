"""SIFT pair extraction with the reference's names (src/extract_sift.py).

The feature front-end, next to the robust-estimation path: same function names and the same cache
files as the reference.  SIFT detection (sfm_hip/sift.py) and descriptor matching (sfm_hip/matching.py) run on
the GPU; OpenCV is not needed.
"""
from __future__ import annotations

import logging
import os
import time

from .utils import load_x_pairs, save_x_pairs


def compute_sift_keypoints(image_path1, image_path2):
    """Reference: src/extract_sift.py:88-135: SIFT on both images (once per image), 2-NN brute-force matching, ratio
    test 0.4, all on the GPU; returns (x1 3xN, x2 3xN, descriptors of view 1 128xN)."""
    from . import matching

    kp1, _ = matching.image_features(image_path1, "bgr2gray")
    kp2, _ = matching.image_features(image_path2, "bgr2gray")
    d1 = matching.image_descriptor_set(image_path1, "bgr2gray")
    d2 = matching.image_descriptor_set(image_path2, "bgr2gray")
    return matching.match_keypoints(kp1, d1, kp2, d2, ratio=0.4)


def process_sift_for_image_pairs(img_paths, init_pair, dataset):
    """Reference: src/extract_sift.py:10-85.  Consecutive pairs then the initial pair, each
    cached as a pickle in the image directory under the reference's file names."""
    x_pairs_file = f"x_pairs_dataset_{dataset}.pkl"
    init_file = f"init_pair_dataset_{dataset}.pkl"
    where = os.path.dirname(img_paths[0])
    x_pairs = load_x_pairs(x_pairs_file, where)
    if x_pairs is None:
        from . import matching

        matching.prefetch_images(img_paths, "bgr2gray")  # host threads decode the files while the GPU works on earlier ones
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

#!/usr/bin/env python3
"""Command-line driver with the interface and the result file of the reference's run_sfm.py
(:16-101): same flags, same stage order, same pickle keys in results/dataset_<d>/sfm_results.pkl,
so the reference's visualization.py reads the output unchanged.  Geometry runs on the GPU."""
import argparse
import logging
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from sfm_hip import pipeline  # noqa: E402
from sfm_hip.extract_sift import process_sift_for_image_pairs  # noqa: E402
from sfm_hip.io import get_data, setup_logging  # noqa: E402


LEVELS = ("DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL")
# flag, type, default, help: the reference's command line (run_sfm.py:16-34 there)
FLAGS = (
    ("--data_path", str, None, "directory that holds one sub-directory per dataset"),
    ("--dataset", str, None, "sub-directory to reconstruct (cfg.yml + images)"),
    ("--threshold", float, 1.0, "inlier threshold in pixels"),
)
RESULT_KEYS = ("K", "absolute_rotations", "refined_Ts", "x_pairs", "nr_images")  # what visualization.py reads back


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Structure from motion on an MI355X")
    for flag, kind, default, text in FLAGS:
        parser.add_argument(flag, type=kind, default=default, help=text)
    parser.add_argument("--verbosity", type=str, choices=LEVELS, default="ERROR", help="logging level")
    return parser.parse_args(argv)


def run_sfm(argv=None):
    """Front-end (cached SIFT pairs) -> geometry on the GPU -> results/dataset_<d>/sfm_results.pkl."""
    opts = parse_args(argv)
    setup_logging(opts.verbosity)
    where = os.path.join(opts.data_path, opts.dataset)
    logging.info("reconstructing %s", where)
    K, img_paths, init_pair = get_data(where)
    x_pairs, init_pair_dict = process_sift_for_image_pairs(img_paths=img_paths, init_pair=init_pair, dataset=opts.dataset)
    res = pipeline.run_geometry(K, x_pairs, init_pair_dict, img_paths, opts.threshold)
    out_dir = os.path.join("results", f"dataset_{opts.dataset}")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "sfm_results.pkl"), "wb") as f:
        pickle.dump({key: res[key] for key in RESULT_KEYS}, f)
    logging.info("results written to %s", out_dir)
    return res


if __name__ == "__main__":
    run_sfm()

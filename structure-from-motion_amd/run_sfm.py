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


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Run Structure-from-Motion pipeline.")
    p.add_argument("--data_path", type=str, help="Base path to the dataset")
    p.add_argument("--dataset", type=str, help="Name of the dataset")
    p.add_argument("--threshold", type=float, default=1.0, help="Pixel threshold for matching (default: 1.0)")
    p.add_argument("--verbosity", type=str, choices=["DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"], default="ERROR",
                   help="Set the logging level (default: ERROR)")
    return p.parse_args(argv)


def run_sfm(argv=None):
    args = parse_args(argv)
    dataset_path = os.path.join(args.data_path, args.dataset)
    setup_logging(args.verbosity)
    logging.info(f"Dataset path: {dataset_path}")
    K, img_paths, init_pair = get_data(dataset_path)
    x_pairs, init_pair_dict = process_sift_for_image_pairs(img_paths=img_paths, init_pair=init_pair, dataset=args.dataset)
    res = pipeline.run_geometry(K, x_pairs, init_pair_dict, img_paths, args.threshold)
    output_dir = f"results/dataset_{args.dataset}"
    os.makedirs(output_dir, exist_ok=True)
    with open(f"{output_dir}/sfm_results.pkl", "wb") as f:
        pickle.dump({k: res[k] for k in ("K", "absolute_rotations", "refined_Ts", "x_pairs", "nr_images")}, f)
    logging.info("Structure-from-Motion pipeline completed and results saved.")
    return res


if __name__ == "__main__":
    run_sfm()

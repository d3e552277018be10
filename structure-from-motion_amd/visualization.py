#!/usr/bin/env python3
"""Command line of the reference's visualization.py (:38-61): reads results/dataset_<d>/sfm_results.pkl
(written by run_sfm.py of this package or of the reference) and writes output/rotation_dataset_<d>.gif.
The point cloud is triangulated on the GPU."""
import argparse
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from sfm_hip.render import COLORS, visualize_sfm_results_with_rotation  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Rotating view of a reconstruction")
    parser.add_argument("--dataset", type=str, help="Name of the dataset")
    return parser.parse_args(argv)


def visualize_results(argv=None):
    """Only for result files this package (or the reference run by the same user) wrote: unpickling executes code from the file."""
    opts = parse_args(argv)
    with open(os.path.join("results", f"dataset_{opts.dataset}", "sfm_results.pkl"), "rb") as f:
        sfm_data_dict = pickle.load(f)
    return visualize_sfm_results_with_rotation(sfm_data_dict, dataset_name=opts.dataset, colors=COLORS)


if __name__ == "__main__":
    visualize_results()

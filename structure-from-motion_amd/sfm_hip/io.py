"""Dataset configuration and logging set-up with the behaviour of the reference's top-level
utils.py (get_data :16-56, setup_logging :59-75).  Plumbing, no GPU work."""
from __future__ import annotations

import logging
import os

import yaml


def get_data(path_to_cfg):
    """Reads <dir>/cfg.yml (JSON-in-YAML with image_file_names, camera.focal_length,
    camera.principal_point, initial_pair) -> (K as a 3x3 list, image paths, initial pair)."""
    cfg_path = os.path.join(path_to_cfg, "cfg.yml")
    if not os.path.isfile(cfg_path):
        raise OSError("File not found")
    with open(cfg_path, "r") as f:
        cfg = yaml.safe_load(f)
    fx, fy = cfg["camera"]["focal_length"]
    cx, cy = cfg["camera"]["principal_point"]
    K = [[fx, 0, cx], [0, fy, cy], [0, 0, 1]]
    img_paths = [os.path.join(path_to_cfg, name) for name in cfg["image_file_names"]]
    return K, img_paths, cfg["initial_pair"]


def setup_logging(verbosity=None):
    if verbosity:
        logging.basicConfig(level=getattr(logging, verbosity), format="%(asctime)s [%(levelname)s] %(message)s",
                            handlers=[logging.StreamHandler()])
    else:
        logging.disable(logging.CRITICAL)

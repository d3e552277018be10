#!/usr/bin/env python3
"""cProfile of the geometry stage of the image-level end-to-end run (features already detected)."""
import cProfile, os, pstats, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sfm_hip import _capi, matching, pipeline, synthetic
from sfm_hip.extract_sift import process_sift_for_image_pairs
from sfm_hip.io import get_data
_capi.load_library()
views = synthetic.render_views(n_cams=10, seed=0)
with tempfile.TemporaryDirectory() as tmp:
    where = os.path.join(tmp, "data", "syn")
    synthetic.write_image_dataset(where, views, init_pair=(4, 6))
    K, img_paths, init_pair = get_data(where)
    x_pairs, ipd = process_sift_for_image_pairs(img_paths=img_paths, init_pair=init_pair, dataset="syn")
    np.random.seed(1)
    pipeline.run_geometry(K, x_pairs, ipd, img_paths, 1.0)
    np.random.seed(1)
    pr = cProfile.Profile(); pr.enable()
    pipeline.run_geometry(K, x_pairs, ipd, img_paths, 1.0)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(40)

#!/usr/bin/env python3
"""Generates tests/golden/dataset5_geometry.npz: the geometry of the reference's dataset 5 on correspondences derived
from its ten images by THIS repository's own CPU restatement of the front-end (oracle/sift.py + oracle/matching.py), and
the oracle's seeded geometry on them.  Only derived numbers are stored (matched keypoint coordinates, the descriptors of
the initial pair's matches, match index lists, the oracle's results): no image and no reference source.

Runs in the build container, where /root/reference exists (about ten minutes on 8 cores):
    python tests/golden/make_dataset5_geometry.py [--data /root/reference/data/5]

What the fixture holds (all plain arrays, allow_pickle=False):
  K                                   3 x 3 intrinsics of data/5/cfg.yml
  x_pair_<k>          k = 0..17       the 2 x 9 arrays of extract_sift.process_sift_for_image_pairs (3 x N, float64 holding
                                      float32-rounded pixel coordinates): consecutive pairs, ratio test 0.4
  x_init_0, x_init_1, init_pair_desc  the initial pair (4, 6) and the 128 x N0 descriptors of its first view's matches
  corr_full_<i>_{xy,ti}   i = 0..9    find_correspondences of image i against ALL init-pair descriptors (ratio 0.75): matched
                                      keypoints (M x 2 float32 pixels) and the index of the 3-D point each one matched
  per seed s<j>_ (six seeds): seed, R_rel, abs_rot, X0, inliers, est_T, ref_T, iterations (nine pairs + the initial pair),
  next_draw = the oracle's run_geometry with np.random.seed(seed); degenerate_record_events = how many of its record events
  came from a sample with two identical matches (their models are LAPACK's arbitrary choice: not comparable);
  s<j>_corr_inl_<i>_{xy,ti}           find_correspondences against the descriptors of that run's initial-pair inliers only (what
                                      run_sfm.py hands to refine_translation, with indices into the FULL X0: the driver's own quirk)
PARITY with cv2's keypoints and matches is UNPINNED (OpenCV is not installed here, the reference holds no match fixtures):
the correspondences are those of the restated front-end, not of cv2.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def features(path):
    from PIL import Image

    from oracle import sift as osift

    with Image.open(path) as im:
        gray = np.asarray(im.convert("L"), dtype=np.uint8)  # the decoder of the product's host side when OpenCV is absent
    t0 = time.time()
    kp, desc, _ = osift.detect_and_compute(gray)
    print(f"{os.path.basename(path)}: {len(kp)} keypoints in {time.time() - t0:.0f} s", flush=True)
    return np.asarray(kp, np.float32).reshape(-1, 2), np.asarray(desc, np.uint8).reshape(-1, 128)


def homog(xy):
    """cartesian_to_homogeneous of float32 pixel coordinates: float64 3 x N (src/extract_sift.py:126-130)."""
    return np.vstack((np.float32(xy).T.reshape(2, -1), np.ones((1, xy.shape[0]))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="/root/reference/data/5")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "dataset5_geometry.npz"))
    ap.add_argument("--seeds", type=int, nargs="+", default=[5, 6, 7, 8, 9, 10])
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--feature-cache", default="", help="scratch .npz (outside the repository) to keep the detected features between runs")
    args = ap.parse_args()
    from oracle import matching as omatch
    from oracle import sfm_oracle as orc

    with open(os.path.join(args.data, "cfg.yml")) as f:
        cfg = json.load(f)  # the dataset files are JSON inside .yml
    fx, fy = cfg["camera"]["focal_length"]
    cx, cy = cfg["camera"]["principal_point"]
    K = [[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]]
    paths = [os.path.join(args.data, n) for n in cfg["image_file_names"]]
    init_pair = tuple(cfg["initial_pair"])
    if args.feature_cache and os.path.exists(args.feature_cache):
        c = np.load(args.feature_cache)
        feats = [(c[f"kp{i}"], c[f"desc{i}"]) for i in range(len(paths))]
    else:
        with mp.get_context("spawn").Pool(args.workers) as pool:
            feats = pool.map(features, paths)
        if args.feature_cache:
            np.savez(args.feature_cache, **{f"kp{i}": f[0] for i, f in enumerate(feats)}, **{f"desc{i}": f[1] for i, f in enumerate(feats)})
    out = {"K": np.asarray(K, np.float64), "init_pair": np.asarray(init_pair, np.int64),
           "keypoints_per_image": np.asarray([f[0].shape[0] for f in feats], np.int64)}

    def pair(i, j, ratio):
        idx, _, _, good = omatch.knn2_ratio(feats[i][1], feats[j][1], ratio)
        qi = np.flatnonzero(good)
        ti = idx[qi]
        return homog(feats[i][0][qi]), homog(feats[j][0][ti]), feats[i][1][qi]

    x_pairs = []
    for i in range(len(paths) - 1):
        x1, x2, _ = pair(i, i + 1, 0.4)
        x_pairs += [x1, x2]
        print(f"pair {i}-{i + 1}: {x1.shape[1]} matches", flush=True)
    a, b, desc = pair(init_pair[0], init_pair[1], 0.4)
    init_pair_dict = {"init_pair_indices": list(init_pair), "x_init": [a, b], "init_pair_desc": desc.astype(np.float32).T}
    print(f"initial pair {init_pair}: {a.shape[1]} matches", flush=True)
    for k, x in enumerate(x_pairs):
        out[f"x_pair_{k}"] = x
    out["x_init_0"], out["x_init_1"] = a, b
    out["init_pair_desc"] = desc.T.copy()  # uint8 128 x N0 (the pipeline's float32 array holds these integers)

    n0 = a.shape[1]
    Kinv = np.linalg.inv(np.asarray(K, np.float64))
    store = {}

    def find_correspondences(img_path, desc_X, X0, Kmat):
        """src/utils.py:65-93 on the restated features: image descriptors (queries) against desc_X (train), ratio 0.75."""
        i = paths.index(img_path)
        train = np.ascontiguousarray(np.asarray(desc_X).T).astype(np.uint8)
        idx, _, _, good = omatch.knn2_ratio(feats[i][1], train, 0.75)
        qi = np.flatnonzero(good)
        ti = idx[qi]
        kind = "full" if train.shape[0] == n0 else "inl"
        store[f"corr_{kind}_{i}_xy"] = feats[i][0][qi].astype(np.float32)
        store[f"corr_{kind}_{i}_ti"] = ti.astype(np.int32)
        x = np.float32(feats[i][0][qi]).T.reshape(2, -1)
        return np.float32(np.asarray(X0)[:, ti]), Kinv @ np.vstack((x, np.ones((1, x.shape[1]))))

    def degenerate(x_a, x_b, rows):
        """True when two of the sampled matches have identical coordinates (a repeated index, or a keypoint with several
        orientations matched twice): the DLT system is rank deficient and its null vector arbitrary."""
        pts = np.vstack((x_a[:2, rows], x_b[:2, rows])).T
        return len({tuple(p) for p in pts}) < len(rows)

    for s_i, seed in enumerate(args.seeds):
        store.clear()
        np.random.seed(seed)
        t0 = time.time()
        iters, R_rel, flagged = [], [], 0
        for i in range(len(x_pairs) // 2):
            tr = {}
            R, _, inl = orc.run_ransac(K, x_pairs[2 * i], x_pairs[2 * i + 1], 1.0, trace=tr)
            R_rel.append(R)
            iters.append(int(tr["iterations"]))
            for it, kind, _ in tr["events"]:
                rows = tr["idx"][it][:8] if kind == "E" else tr["idx"][it][8:]
                flagged += bool(degenerate(x_pairs[2 * i], x_pairs[2 * i + 1], rows))
            print(f"seed {seed} pair {i}: {iters[-1]} iterations, {int(inl.sum())} inliers", flush=True)
        # the initial pair's loop runs inside run_reconstruction: trace it through the module's own entry point
        tr = {}
        orig = orc.run_ransac
        orc.run_ransac = lambda *a, **k: orig(*a, trace=tr, **k)
        try:
            X0, abs_rot, inliers = orc.run_reconstruction(R_rel, init_pair_dict, K, 1.0)
        finally:
            orc.run_ransac = orig
        iters.append(int(tr["iterations"]))
        for it, kind, _ in tr["events"]:
            rows = tr["idx"][it][:8] if kind == "E" else tr["idx"][it][8:]
            flagged += bool(degenerate(a, b, rows))
        desc_X = init_pair_dict["init_pair_desc"]
        est = orc.estimate_translation(K, desc_X, X0, abs_rot, paths, 1.0, find_correspondences)
        ref = orc.refine_translation(K, paths, desc_X[:, inliers], X0, abs_rot, est, find_correspondences)
        nxt = np.int64(np.random.randint(0, 1 << 30))
        print(f"seed {seed}: oracle geometry {time.time() - t0:.0f} s; initial pair inliers {int(inliers.sum())} of {n0}; "
              f"record events on degenerate samples: {flagged}", flush=True)
        tag = f"s{s_i}_"
        out[tag + "seed"] = np.int64(seed)
        out[tag + "next_draw"] = nxt
        out[tag + "degenerate_record_events"] = np.int64(flagged)
        for k, v in store.items():  # the correspondences against the inliers depend on the run's inlier mask
            if "_inl_" in k:
                out[tag + k] = v
            else:
                out[k] = v
        out[tag + "R_rel"] = np.stack(R_rel)
        out[tag + "abs_rot"] = np.stack(abs_rot)
        out[tag + "X0"] = X0
        out[tag + "inliers"] = inliers
        out[tag + "est_T"] = np.stack([np.reshape(t, 3) for t in est])
        out[tag + "ref_T"] = np.stack([np.reshape(t, 3) for t in ref])
        out[tag + "iterations"] = np.asarray(iters, np.int64)
    out["n_seeds"] = np.int64(len(args.seeds))
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The exact E/H scorer with and without its early loops over scenes of rising inlier ratio and planar share (10 000 matches x
400 000 hypotheses): the early loops must never cost time where hypotheses from all-inlier samples are common."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sfm_hip import _capi, ops, synthetic
n, B = int(os.environ.get("AB_N", 10000)), int(os.environ.get("AB_B", 400000))
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for inl, pl in [(0.1, 0.5), (0.3, 0.5), (0.5, 0.0), (0.5, 0.5), (0.5, 1.0), (0.7, 0.5), (0.8, 0.2), (0.8, 1.0), (0.9, 0.5), (0.95, 0.5)]:
    sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=pl, seed=3)
    Kinv = np.linalg.inv(np.asarray(sc["K"], float))
    corr = _capi.dev(ops.pack_corr(Kinv @ sc["x1"], Kinv @ sc["x2"]))
    thr = 1.0 / sc["K"][0][0]
    idx = ops.index_sets(5, 0, B, n)
    ws = torch.empty(int(_capi.load_library().sfm_eh_score_mixed_workspace_bytes(B, n)), dtype=torch.uint8, device="cuda")
    out = [(torch.empty(B, dtype=torch.int32, device="cuda"), torch.empty(B, dtype=torch.int32, device="cuda")) for _ in range(2)]
    t_full = timeit(lambda: ops.eh_score_mixed(corr, idx, thr, 3 * thr, 0.0, out=out[0], workspace=ws, every_pair_full_test=True))
    t_early = timeit(lambda: ops.eh_score_mixed(corr, idx, thr, 3 * thr, 0.0, out=out[1], workspace=ws, every_pair_full_test=False))
    same = bool(torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]))
    print(f"inliers {inl:.2f} planar {pl:.1f}: full {t_full:.3f} ms  early {t_early:.3f} ms  ratio {t_early / t_full:.3f}  counts identical {same}  "
          f"mean counts E {float(out[0][0].float().mean()):.1f} H {float(out[0][1].float().mean()):.1f}", flush=True)

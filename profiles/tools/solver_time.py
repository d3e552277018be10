#!/usr/bin/env python3
"""Times the minimal solvers alone (sfm_eh_models_f64: 8-point E + 4-point H per hypothesis, 10^6 hypotheses) beside the
bound kernel that contains them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sfm_hip import _capi, ops, synthetic
n, B = 10000, 1000000
sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=0)
Kinv = np.linalg.inv(np.asarray(sc["K"], float))
corr = _capi.dev(ops.pack_corr(Kinv @ sc["x1"], Kinv @ sc["x2"]))
thr = 1.0 / sc["K"][0][0]
idx = _capi.dev(np.random.default_rng(0).integers(0, n, (B, 12)), np.int32)
cm = ops.corr_to_mfma(corr, thr)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print(f"models only: {timeit(lambda: ops.eh_models(corr, idx)):.3f} ms")
print(f"bound kernel: {timeit(lambda: ops.eh_upper_mfma(corr, cm, idx, thr, 3 * thr)):.3f} ms")
small = _capi.dev(ops.pack_corr(Kinv @ sc["x1"], Kinv @ sc["x2"])[:64].copy())
cm_s = ops.corr_to_mfma(small, thr)
idx_s = _capi.dev(np.random.default_rng(0).integers(0, 64, (B, 12)), np.int32)
print(f"bound kernel on 64 matches (solve + pack + 2 tiles): {timeit(lambda: ops.eh_upper_mfma(small, cm_s, idx_s, thr, 3 * thr)):.3f} ms")

#!/usr/bin/env python3
"""Times sfm_eh_score_mixed on the 10 000 x 1 000 000 workload and checks its counts against sfm_eh_score_f64.
SFM_HIP_EHX_SPLIT=1 scores the two models in two launches (read once per process by the library)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sfm_hip import _capi, ops, synthetic
n, B = int(os.environ.get("AB_N", 10000)), int(os.environ.get("AB_B", 1000000))
sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=0)
Kinv = np.linalg.inv(np.asarray(sc["K"], float))
corr = _capi.dev(ops.pack_corr(Kinv @ sc["x1"], Kinv @ sc["x2"]))
thr = 1.0 / sc["K"][0][0]
np.random.seed(0)
idx = _capi.dev(np.random.randint(0, n, size=12 * B).reshape(B, 12), np.int32)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr)
print("split", os.environ.get("SFM_HIP_EHX_SPLIT"), "identical", bool(torch.equal(ce, me) and torch.equal(ch, mh)))
for r in range(3):
    print(f"round {r}: mixed {timeit(lambda: ops.eh_score_mixed(corr, idx, thr, 3 * thr)):.3f} ms")

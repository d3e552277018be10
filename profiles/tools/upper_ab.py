#!/usr/bin/env python3
"""Times the matrix-core bound kernel (sfm_eh_upper_mfma_f32) on the 10 000 x 1 000 000 workload and checks its counts:
upper bounds of the float64 counts of the first 200 000 hypotheses, and a checksum to compare forms with.
SFM_HIP_EH_FLOATCOUNT=0 selects the sign-history form (read once per process by the library)."""
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
corrm = ops.corr_to_mfma(corr, thr)
ue = torch.empty(B, dtype=torch.int32, device="cuda"); uh = torch.empty_like(ue)
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
f = lambda: ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, 0.0, out=(ue, uh))
f()
S = min(B, 200000)
ce, ch = ops.eh_score(corr, idx[:S], thr, 3 * thr)
print("floatcount", os.environ.get("SFM_HIP_EH_FLOATCOUNT"), "violations", int((ue[:S] < ce).sum()), int((uh[:S] < ch).sum()),
      "mean slack", float((ue[:S] - ce).float().mean()), float((uh[:S] - ch).float().mean()), "checksum", int(ue.sum()), int(uh.sum()))
for r in range(3):
    print(f"round {r}: bound kernel {timeit(f):.3f} ms")
for part in (1, 2):
    g = lambda: ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, 0.0, out=(ue, uh), part=part)
    print(f"part {part} alone (1 = epipolar, 2 = transfer): {timeit(g):.3f} ms")

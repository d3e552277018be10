#!/usr/bin/env python3
"""Times the bulk kernels of the two-view stage on the C2 workload (10 000 matches x 10^6 hypotheses) with HIP events
and checks the bounds against the exact counts: prints mean milliseconds per launch.
    python profiles/kernel_ab.py [reps] [matches] [hypotheses]
AB_ZERO=1 runs the same instruction streams on all-zero matches (separates the power limit from issue limits);
AB_LIB=<path> loads an experimental build of the library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from sfm_hip import _capi, ops, synthetic

if os.environ.get("AB_LIB"):  # an experimental build of the library
    _capi.LIB_PATH = os.path.join(ROOT, os.environ["AB_LIB"])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=0)
thr = 1.0 / sc["K"][0][0]
Kinv = np.linalg.inv(np.asarray(sc["K"], float))
corr_h = ops.pack_corr(Kinv @ sc["x1"], Kinv @ sc["x2"])
if os.environ.get("AB_ZERO"):  # same instruction stream on all-zero operands: separates power limits from issue limits
    corr_h[:] = 0.0
corr = _capi.dev(corr_h)
corrm = ops.corr_to_mfma(corr, thr)
idx = ops.index_sets(0, 0, B, n)
ue = torch.empty(B, dtype=torch.int32, device="cuda")
uh = torch.empty(B, dtype=torch.int32, device="cuda")


def timeit(f, reps=reps):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ce, ch = ops.eh_score_mixed(corr, idx, thr, 3 * thr)
ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, out=(ue, uh))
torch.cuda.synchronize()
viol = int((ue < ce).sum()) + int((uh < ch).sum())
print(f"n={n} B={B} bound violations {viol}; mean slack E {float((ue - ce).float().mean()):.3f} H {float((uh - ch).float().mean()):.3f}; "
      f"checksum {int(ue.sum())} {int(uh.sum())}")
for rnd in range(3):
    t1 = timeit(lambda: ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, out=(ue, uh)))
    t2 = timeit(lambda: ops.eh_score_mixed(corr, idx, thr, 3 * thr, out=(ue, uh)), max(2, reps // 4))
    print(f"round {rnd}: matrix-core pre-filter {t1:.3f} ms ({2.0 * n * B / t1 / 1e9:.2f}e12 evals/s), exact mixed scorer {t2:.3f} ms")

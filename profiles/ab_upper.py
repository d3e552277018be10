#!/usr/bin/env python3
"""A/B of the float32 pre-filter variants on the C2 workload (10 000 matches x 10^6 hypotheses),
interleaved rounds in one process: the scalar-load form (default) or the LDS-staged form
(SFM_HIP_UPPER_LDS=1 selects it at first use, so run this script once per setting) beside the
float64 kernel.  Prints mean milliseconds per launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from sfm_hip import _capi, ops, synthetic

n, B = 10000, 1000000
sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=0)
thr = 1.0 / sc["K"][0][0]
Kinv = np.linalg.inv(np.asarray(sc["K"], float))
corr_h = ops.pack_corr(Kinv @ sc["x1"], Kinv @ sc["x2"])
corr = _capi.dev(corr_h)
c32 = ops.corr_to_f32(corr)
xmax = float(np.abs(corr_h).max())
np.random.seed(0)
idx = _capi.dev(np.random.randint(0, n, size=12 * B).reshape(B, 12), np.int32)
ue = torch.empty(B, dtype=torch.int32, device="cuda")
uh = torch.empty(B, dtype=torch.int32, device="cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(f):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for rnd in range(3):
    t1 = timeit(lambda: ops.eh_upper(corr, c32, idx, thr, 3 * thr, xmax, out=(ue, uh)))
    t3 = timeit(lambda: ops.eh_score(corr, idx, thr, 3 * thr, out=(ue, uh)))
    print(f"round {rnd}: lds_env={os.environ.get('SFM_HIP_UPPER_LDS')} float32 pre-filter {t1:.3f} ms, float64 {t3:.3f} ms")

#!/usr/bin/env python3
"""Times the translation RANSAC (12 cameras x 50 000 correspondences x 100 000 hypotheses) mixed against float64.
SFM_HIP_T_SPAN=<matches per split, 0 = no extra splitting> is read once per process by the library;
SFM_DIAG_LIB=<path> loads another build of the library (same-device comparisons within one job)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sfm_hip import _capi, ops, synthetic
if os.environ.get("SFM_DIAG_LIB"):
    _capi.LIB_PATH = os.environ["SFM_DIAG_LIB"]
ncam, m, B = 12, 50000, 100000
cams = [synthetic.resection_scene(m=m, outlier_ratio=0.5, seed=100 + c) for c in range(ncam)]
pts = torch.cat([ops.pack_points(_capi.dev(c["R"].ravel()), _capi.dev(c["X"]), _capi.dev(c["x_norm"])) for c in cams])
idx = _capi.dev(np.random.default_rng(0).integers(0, m, size=(ncam, B, 2)), np.int32)
thr = 4.0 / cams[0]["K"][0][0]
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
bm, Tm, cm = ops.t_ransac(pts, [m] * ncam, idx, thr, mixed=True)
bf, Tf, cf = ops.t_ransac(pts, [m] * ncam, idx, thr, mixed=False)
print("span", os.environ.get("SFM_HIP_T_SPAN"), "identical", bool(torch.equal(cm, cf) and torch.equal(bm, bf)))
for r in range(3):
    print(f"round {r}: mixed {timeit(lambda: ops.t_ransac(pts, [m] * ncam, idx, thr, mixed=True)):.3f} ms")

#!/usr/bin/env python3
"""Times the exact E/H scorer (sfm_eh_score_mixed) on the 10 000 x 1 000 000 workload (AB_N, AB_B, AB_INL change it) and
compares its counts with the float64 kernel's on the first 200 000 hypotheses.  SFM_HIP_EHX_EARLY=0 selects the form without
the early loops, SFM_HIP_EHX_PART=1 / 2 one model's launch alone (both read once per process by the library)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sfm_hip import _capi, ops, synthetic
n, B, inl = int(os.environ.get("AB_N", 10000)), int(os.environ.get("AB_B", 1000000)), float(os.environ.get("AB_INL", 0.3))
sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=0.5, seed=0)
Kinv = np.linalg.inv(np.asarray(sc["K"], float))
corr = _capi.dev(ops.pack_corr(Kinv @ sc["x1"], Kinv @ sc["x2"]))
thr = 1.0 / sc["K"][0][0]
np.random.seed(0)
idx = _capi.dev(np.random.randint(0, n, size=12 * B).reshape(B, 12), np.int32)
ce = torch.empty(B, dtype=torch.int32, device="cuda"); ch = torch.empty_like(ce)
ws = ops.eh_score_mixed_workspace(B, n) if hasattr(ops, "eh_score_mixed_workspace") else None
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
kw = {"workspace": ws} if ws is not None else {}
f = lambda: ops.eh_score_mixed(corr, idx, thr, 3 * thr, 0.0, out=(ce, ch), **kw)
f()
S = min(B, 200000)
fe, fh = ops.eh_score(corr, idx[:S], thr, 3 * thr)
part = os.environ.get("SFM_HIP_EHX_PART")
print("early", os.environ.get("SFM_HIP_EHX_EARLY"), "part", part, "inl", inl, "mismatches", int((ce[:S] != fe).sum()) if part != "2" else "-",
      int((ch[:S] != fh).sum()) if part != "1" else "-", "checksum", int(ce.sum()), int(ch.sum()))
for r in range(3):
    print(f"round {r}: exact call {timeit(f):.3f} ms")

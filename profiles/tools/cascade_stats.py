import sys, os
sys.path.insert(0, "structure-from-motion_amd"); sys.path.insert(0, ".")
import numpy as np, torch
from sfm_hip import _capi, ops, synthetic
from oracle import sfm_oracle as orc
for (inl, n, B, seed) in [(0.0, 3000, 200000, 6), (0.05, 777, 70001, 3), (0.3, 5000, 100000, 0)]:
    sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=0.5, seed=seed)
    x1 = orc.normalize_K(sc["K"], sc["x1"]); x2 = orc.normalize_K(sc["K"], sc["x2"])
    corr_h = ops.pack_corr(x1, x2); corr = _capi.dev(corr_h); c32 = ops.corr_to_f32(corr)
    xmax = float(np.abs(corr_h).max())
    idx = _capi.dev(np.random.default_rng(seed).integers(0, n, size=(B, 12)), np.int32)
    thr = 1.0 / sc["K"][0][0]
    ce, ch = [t.cpu().numpy() for t in ops.eh_score(corr, idx, thr, 3 * thr)]
    me, mh = [t.cpu().numpy() for t in ops.eh_upper_mfma(corr, ops.corr_to_mfma(corr, thr), idx, thr, 3 * thr, xmax)]
    print(inl, n, B, "best", ce.max(), ch.max(), "slack", (me - ce).mean(), (mh - ch).mean(), "viol", (me < ce).sum(), (mh < ch).sum())
    print("  ", ops.eh_best_native(corr, c32, idx, thr, 3 * thr, xmax, loose=2))

"""Randomised exactness sweep (the long form is run by hand on the GPU box; tests/test_gpu_stress.py runs a slice):

    python tests/stress_gpu.py [n_configs [seed]] > profiles/<tag>_stress.txt

For random scenes (inlier ratio, planar share, match count, noise, threshold) it checks that
  * sfm_eh_score_mixed (with its early loops and with every pair through the full tests) returns the counts of
    sfm_eh_score_f64 for every hypothesis,
  * sfm_eh_best (all three first-level forms) returns the first maximum of those counts,
  * every float32 / matrix-core bound is an upper bound of the float64 count,
  * sfm_t_ransac_mixed returns the counts of sfm_t_ransac_f64 for every hypothesis.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import sfm_oracle as orc  # noqa: E402
from sfm_hip import _capi, ops, synthetic  # noqa: E402


def main(n_cfg, seed=2024):
    _capi.load_library()
    rng = np.random.default_rng(seed)
    bad = 0
    t_start = time.time()
    pairs_eh = pairs_t = 0
    for c in range(n_cfg):
        n = int(rng.choice([40, 333, 1000, 2500, 6000, 12000]))
        B = int(rng.choice([257, 5000, 30000, 120000]))
        inl = float(rng.choice([0.0, 0.05, 0.3, 0.6, 0.95]))
        pl = float(rng.choice([0.0, 0.5, 1.0]))
        px = float(rng.choice([0.5, 1.0, 3.0]))
        sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=pl, seed=seed + 1000 + c)
        thr = px / sc["K"][0][0]
        x1n = orc.normalize_K(sc["K"], sc["x1"])
        x2n = orc.normalize_K(sc["K"], sc["x2"])
        corr_h = ops.pack_corr(x1n, x2n)
        corr = _capi.dev(corr_h)
        c32 = ops.corr_to_f32(corr)
        cm = ops.corr_to_mfma(corr, thr)
        xmax = float(np.abs(corr_h).max())
        idx = ops.index_sets(c, 0, B, n)
        ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
        want = ((int(ce.max()), int((ce == ce.max()).nonzero()[0])), (int(ch.max()), int((ch == ch.max()).nonzero()[0])))
        me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr, xmax, every_pair_full_test=False)  # with the early loops
        viol = 0 if (torch.equal(me, ce) and torch.equal(mh, ch)) else 10**9  # the mixed-precision exact counts
        me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr, xmax, every_pair_full_test=True)
        viol += 0 if (torch.equal(me, ce) and torch.equal(mh, ch)) else 10**9
        for name, (ue, uh) in [("tight", ops.eh_upper(corr, c32, idx, thr, 3 * thr, xmax, one_sided=False)),
                               ("loose", ops.eh_upper(corr, c32, idx, thr, 3 * thr, xmax, one_sided=True)),
                               ("mfma", ops.eh_upper_mfma(corr, cm, idx, thr, 3 * thr, xmax))]:
            viol += int((ue < ce).sum()) + int((uh < ch).sum())
        got = [ops.eh_best_native(corr, c32, idx, thr, 3 * thr, xmax, loose=l)[:2] for l in (0, 1, 2)]
        ok = viol == 0 and all(g == want for g in got)
        pairs_eh += 2 * n * B
        # translation stage
        ncam = int(rng.integers(1, 5))
        sizes = [int(rng.choice([3, 50, 700, 4000, 20000])) for _ in range(ncam)]
        cams = [synthetic.resection_scene(m=max(s, 2), outlier_ratio=float(rng.choice([0.0, 0.5, 0.9])), seed=5000 + 10 * c + i)
                for i, s in enumerate(sizes)]
        pts = torch.cat([ops.pack_points(_capi.dev(cm_["R"].ravel()), _capi.dev(cm_["X"]), _capi.dev(cm_["x_norm"]))[:s].contiguous()
                         for cm_, s in zip(cams, sizes)])
        Bt = int(rng.choice([100, 4097, 50000]))
        idx_t = _capi.dev(np.stack([rng.integers(0, s, size=(Bt, 2)) for s in sizes]), np.int32)
        thr_t = float(rng.choice([1.0, 4.0, 10.0])) / synthetic.K_DATASET5[0][0]
        b64, T64, c64 = ops.t_ransac(pts, sizes, idx_t, thr_t, mixed=False)
        bmx, Tmx, cmx = ops.t_ransac(pts, sizes, idx_t, thr_t, mixed=True)
        ok_t = bool(torch.equal(c64, cmx) and torch.equal(b64, bmx))
        pairs_t += sum(sizes) * Bt
        if not (ok and ok_t):
            bad += 1
        print(f"cfg {c:3d}: n={n:5d} B={B:6d} inl={inl:.2f} planar={pl:.1f} thr={px}px  bound violations {viol}  winners {'ok' if ok else 'MISMATCH'}"
              f"  | T: cams {sizes} B={Bt} {'identical' if ok_t else 'MISMATCH'}", flush=True)
    print(f"{n_cfg} configurations, {bad} failures, {pairs_eh:.3g} E/H residual tests and {pairs_t:.3g} reprojection tests compared, "
          f"{time.time() - t_start:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 2024))

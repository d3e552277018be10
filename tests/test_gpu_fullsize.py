"""Full-size runs (the configurations the benchmark quotes) checked through size-independent
properties, since the oracle cannot score 10^10 residuals in test time:
 - the two-tier winner is the first maximum of the float64 counts, and the float32 bounds hold;
 - sharding the hypothesis range and merging packed keys gives the single-range winner;
 - permuting the correspondences (with remapped index sets) leaves every count unchanged;
 - a sample of hypotheses agrees with the oracle exactly.
Also the error paths of the C ABI."""
import ctypes

import numpy as np
import pytest

from oracle import sfm_oracle as orc
from sfm_hip import synthetic
from sfm_hip.sharding import pack_key, shard_range, unpack_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from sfm_hip import _capi, ops

    _capi.load_library()
    return torch, _capi, ops


def test_c2_full_size_properties(env):
    torch, C, ops = env
    n, B = 10000, 1000000
    sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=2)
    thr = 1.0 / sc["K"][0][0]
    x1 = orc.normalize_K(sc["K"], sc["x1"])
    x2 = orc.normalize_K(sc["K"], sc["x2"])
    corr_h = ops.pack_corr(x1, x2)
    corr = C.dev(corr_h)
    c32 = ops.corr_to_f32(corr)
    xmax = float(np.abs(corr_h).max())
    np.random.seed(5)
    idx_h = np.random.randint(0, n, size=12 * B).reshape(B, 12).astype(np.int32)
    idx = C.dev(idx_h)
    ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
    ue, uh = ops.eh_upper(corr, c32, idx, thr, 3 * thr, xmax)
    assert bool((ue >= ce).all()) and bool((uh >= ch).all())
    (cE, iE), (cH, iH), _ = ops.eh_best(corr, c32, idx, thr, 3 * thr, xmax)
    ce_h, ch_h = ce.cpu().numpy(), ch.cpu().numpy()
    assert (cE, iE) == (int(ce_h.max()), int(np.argmax(ce_h)))
    assert (cH, iH) == (int(ch_h.max()), int(np.argmax(ch_h)))
    # sharded ranges + packed-key merge == single range
    for world in (2, 8):
        keys = []
        for r in range(world):
            lo, hi = shard_range(B, r, world)
            (a, ia), (b, ib), _ = ops.eh_best(corr, c32, idx[lo:hi], thr, 3 * thr, xmax)
            keys.append((pack_key(a, lo + ia), pack_key(b, lo + ib)))
        assert unpack_key(max(k[0] for k in keys)) == (cE, iE)
        assert unpack_key(max(k[1] for k in keys)) == (cH, iH)
    # permutation of the correspondences: same models, same counts
    perm = np.random.default_rng(0).permutation(n)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    sub = slice(0, 100000)
    ce_p, ch_p = ops.eh_score(C.dev(corr_h[perm]), C.dev(inv[idx_h[sub]], np.int32), thr, 3 * thr)
    np.testing.assert_array_equal(ce_p.cpu().numpy(), ce_h[sub])
    np.testing.assert_array_equal(ch_p.cpu().numpy(), ch_h[sub])
    # a sample against the oracle, including the winners
    pick = np.concatenate([[iE, iH], np.random.default_rng(1).integers(0, B, 200)])
    oE, oH, _, _ = orc.score_EH_batch(x1, x2, idx_h[pick].astype(np.int64), thr)
    ok = np.array([len(set(r[:8])) == 8 and len(set(r[8:])) == 4 for r in idx_h[pick]])
    np.testing.assert_array_equal(ce_h[pick][ok], oE[ok])
    np.testing.assert_array_equal(ch_h[pick][ok], oH[ok])


def test_c3_full_size_properties(env):
    torch, C, ops = env
    m, B = 50000, 100000
    cams = [synthetic.resection_scene(m=m, outlier_ratio=0.5, seed=40 + c) for c in range(2)]
    pts = torch.cat([ops.pack_points(C.dev(c["R"].ravel()), C.dev(c["X"]), C.dev(c["x_norm"])) for c in cams])
    idx_h = np.random.default_rng(3).integers(0, m, size=(2, B, 2)).astype(np.int32)
    thr = 4.0 / cams[0]["K"][0][0]
    best, T, counts = ops.t_ransac(pts, [m, m], C.dev(idx_h), thr)
    cnt, ind = ops.decode_best(best.cpu().numpy())
    ch = counts.cpu().numpy()
    for c in range(2):
        assert (cnt[c], ind[c]) == (ch[c].max(), int(np.argmax(ch[c])))
    # the winner's translation and count against the oracle
    for c, cam in enumerate(cams):
        Tref = orc.estimate_T_2point_batch(cam["x_norm"], cam["X"], cam["R"], idx_h[c][[ind[c]]].astype(np.int64))
        np.testing.assert_allclose(T.cpu().numpy()[c], Tref[0], rtol=1e-8, atol=1e-10)
        assert orc.count_inliers(cam["x_norm"], cam["X"], cam["R"], Tref[0], thr) == cnt[c]
    # permutation of the points of camera 0: counts unchanged
    perm = np.random.default_rng(4).permutation(m)
    inv = np.empty(m, np.int64)
    inv[perm] = np.arange(m)
    c0 = cams[0]
    pts_p = ops.pack_points(C.dev(c0["R"].ravel()), C.dev(np.ascontiguousarray(c0["X"][:, perm])), C.dev(np.ascontiguousarray(c0["x_norm"][:, perm])))
    _, _, counts_p = ops.t_ransac(pts_p, [m], C.dev(inv[idx_h[0][:20000]][None], np.int32), thr)
    np.testing.assert_array_equal(counts_p.cpu().numpy()[0], ch[0][:20000])


def test_c_abi_rejects_bad_arguments(env):
    torch, C, ops = env
    lib = C.load_library()
    corr = torch.zeros((16, 4), dtype=torch.float64, device="cuda")
    idx = torch.zeros((4, 12), dtype=torch.int32, device="cuda")
    out = torch.zeros(4, dtype=torch.int32, device="cuda")
    null = ctypes.c_void_p(0)
    EINVAL = 100001
    assert lib.sfm_eh_score_f64(null, 16, C.dptr(idx), 4, 1e-3, 3e-3, C.dptr(out), C.dptr(out), null) == EINVAL
    assert lib.sfm_eh_score_f64(C.dptr(corr), 0, C.dptr(idx), 4, 1e-3, 3e-3, C.dptr(out), C.dptr(out), null) == EINVAL
    assert lib.sfm_eh_score_f64(C.dptr(corr), 16, C.dptr(idx), -1, 1e-3, 3e-3, C.dptr(out), C.dptr(out), null) == EINVAL
    assert lib.sfm_eh_score_f64(C.dptr(corr), 16, C.dptr(idx), 0, 1e-3, 3e-3, C.dptr(out), C.dptr(out), null) == 0  # empty block: no-op
    off = np.array([0, 8, 4], np.int64)  # decreasing offsets
    pts = torch.zeros((16, 6), dtype=torch.float64, device="cuda")
    best = torch.zeros(2, dtype=torch.int64, device="cuda")
    T = torch.zeros((2, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((2, 4), dtype=torch.int32, device="cuda")
    i2 = torch.zeros((2, 4, 2), dtype=torch.int32, device="cuda")
    assert lib.sfm_t_ransac_f64(C.dptr(pts), C.hptr(off), 2, C.dptr(i2), 4, 1e-3, C.dptr(cnt), C.dptr(best), C.dptr(T), null) == EINVAL
    off = np.zeros(70, np.int64)
    assert lib.sfm_t_ransac_f64(C.dptr(pts), C.hptr(off), 69, C.dptr(i2), 4, 1e-3, C.dptr(cnt), C.dptr(best), C.dptr(T), null) == EINVAL  # > 64 cameras
    assert lib.sfm_triangulate_f64(C.dptr(T), C.dptr(T), C.dptr(pts), C.dptr(pts), 8, 4, C.dptr(pts), null) == EINVAL  # ld < m
    with pytest.raises(C.SfmHipError):
        C.check(EINVAL)
    # the library still works after rejected calls
    ce, ch = ops.eh_score(torch.rand((64, 4), dtype=torch.float64, device="cuda"), torch.randint(0, 64, (8, 12), dtype=torch.int32, device="cuda"), 0.01, 0.03)
    assert ce.shape[0] == 8


def test_degenerate_inputs_do_not_fault(env):
    """Identical matches, all-duplicate samples, NaN coordinates: the reference produces NaN/garbage
    models that count nothing; the kernels must do the same quietly."""
    torch, C, ops = env
    corr = np.zeros((300, 4))
    corr[:, 0] = 0.1
    corr[:, 1] = -0.2
    corr[:, 2] = 0.1
    corr[:, 3] = -0.2  # every match identical
    idx = np.random.default_rng(0).integers(0, 300, (512, 12)).astype(np.int32)
    ce, ch = ops.eh_score(C.dev(corr), C.dev(idx), 4e-4, 1.2e-3)
    assert ce.shape[0] == 512 and int(ce.min()) >= 0 and int(ce.max()) <= 300
    corr2 = np.random.default_rng(1).uniform(-0.4, 0.4, (300, 4))
    corr2[7] = np.nan
    ce, ch = ops.eh_score(C.dev(corr2), C.dev(idx), 4e-4, 1.2e-3)
    has_nan = (idx[:, :8] == 7).any(1)
    assert int(ce.cpu().numpy()[has_nan].max(initial=0)) == 0  # a NaN sample gives a NaN model: no inliers
    idx_dup = np.zeros((256, 12), np.int32)  # all twelve indices equal
    ce, ch = ops.eh_score(C.dev(corr2), C.dev(idx_dup), 4e-4, 1.2e-3)
    assert ce.shape[0] == 256

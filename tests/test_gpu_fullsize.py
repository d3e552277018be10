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


def test_c2_full_size_properties(env, mixed_form):
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
    # the paths the benchmark times, on the whole block: the matrix-core cascade as one library call (also split
    # into submit / collect with two blocks in flight) and the exact mixed scorer
    for loose in (2, 1, 0):
        wE, wH, _ = ops.eh_best_native(corr, c32, idx, thr, 3 * thr, 0.0, loose=loose)
        assert (wE, wH) == ((cE, iE), (cH, iH))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    half = B // 2
    jobs = [ops.eh_best_submit(corr, c32, idx[k * half : (k + 1) * half], thr, 3 * thr, 0.0, stream=streams[k]) for k in range(2)]
    parts = [ops.eh_best_collect(j) for j in jobs]
    assert unpack_key(max(pack_key(p[0][0], k * half + p[0][1]) for k, p in enumerate(parts))) == (cE, iE)
    assert unpack_key(max(pack_key(p[1][0], k * half + p[1][1]) for k, p in enumerate(parts))) == (cH, iH)
    corrm = ops.corr_to_mfma(corr, thr)
    me, mh = ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, 0.0)
    assert bool((me >= ce).all()) and bool((mh >= ch).all())
    xe, xh = ops.eh_score_mixed(corr, idx, thr, 3 * thr)
    assert torch.equal(xe, ce) and torch.equal(xh, ch)
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
    # a sample against the oracle: the winners, the 200 hypotheses with the most inliers of either model (the ones with the
    # most pairs near a threshold) and 2000 random ones
    pick = np.concatenate([[iE, iH], np.argsort(-ce_h, kind="stable")[:200], np.argsort(-ch_h, kind="stable")[:200],
                           np.random.default_rng(1).integers(0, B, 2000)])
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
    ws = torch.zeros(int(lib.sfm_eh_score_mixed_workspace_bytes(4, 16)) + 256, dtype=torch.uint8, device="cuda")
    args = (C.dptr(corr), 16, C.dptr(idx), 4, 1e-3, 3e-3, 0.0)
    assert lib.sfm_eh_score_mixed_opt(*args, 2, C.dptr(ws), ws.numel(), C.dptr(out), C.dptr(out), null) == EINVAL  # unknown option bit
    assert lib.sfm_eh_score_mixed_opt(*args, 1, C.dptr(ws), 16, C.dptr(out), C.dptr(out), null) == EINVAL  # workspace too small
    assert lib.sfm_eh_score_mixed_opt(*args, 1, C.dptr(ws), ws.numel(), C.dptr(out), C.dptr(out), null) == 0
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


def test_c5_sharded_200k_matches(env):
    """The 200 000-match configuration with device-generated (Philox) index sets: one GPU's share of the 10^8
    hypotheses (1.25 * 10^7 ids) through the cascade against the first maximum of the exact counts of that whole
    share; 8 shards scored separately and merged by packed keys against the single range; a sample of ids and the
    winners against the oracle."""
    torch, C, ops = env
    n, total, world, seed = 200000, 100000000, 8, 0
    sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=seed)
    thr = 1.0 / sc["K"][0][0]
    x1 = orc.normalize_K(sc["K"], sc["x1"])
    x2 = orc.normalize_K(sc["K"], sc["x2"])
    corr = C.dev(ops.pack_corr(x1, x2))
    c32 = ops.corr_to_f32(corr)
    lo, hi = shard_range(total, 3, world)
    assert hi - lo == 12500000
    idx = ops.index_sets(seed, lo, hi - lo, n)
    (cE, iE), (cH, iH), st = ops.eh_best_native(corr, c32, idx, thr, 3 * thr, 0.0)
    ce, ch = ops.eh_score_mixed(corr, idx, thr, 3 * thr)  # exact counts of the whole share (2.5 * 10^12 pairs)
    assert (cE, cH) == (int(ce.max()), int(ch.max()))
    assert iE == int((ce == cE).nonzero()[0].item()) and iH == int((ch == cH).nonzero()[0].item())
    # sample + winners against the oracle
    pick = np.concatenate([[iE, iH], np.random.default_rng(1).integers(0, hi - lo, 600)])
    rows = idx[torch.from_numpy(pick).cuda()].cpu().numpy().astype(np.int64)
    from oracle import philox

    np.testing.assert_array_equal(rows, np.stack([philox.index_sets(seed, lo + int(q), 1, n)[0] for q in pick]))
    oE, oH, _, _ = orc.score_EH_batch(x1, x2, rows, thr)
    ok = np.array([len(set(r[:8])) == 8 and len(set(r[8:])) == 4 for r in rows])
    np.testing.assert_array_equal(ce.cpu().numpy()[pick][ok], oE[ok])
    np.testing.assert_array_equal(ch.cpu().numpy()[pick][ok], oH[ok])
    assert ok[0] and ok[1]
    del ce, ch, idx
    # 8 shards of a smaller id range, merged, against the single range
    small = 800000
    base = 40000000
    idx_all = ops.index_sets(seed, base, small, n)
    single = ops.eh_best_native(corr, c32, idx_all, thr, 3 * thr, 0.0)
    keys_e, keys_h = [], []
    for r in range(world):
        a, b = shard_range(small, r, world)
        part = ops.index_sets(seed, base + a, b - a, n)  # generated by the shard itself, as a rank would
        assert torch.equal(part, idx_all[a:b])
        (pe, pi), (qe, qi), _ = ops.eh_best_native(corr, c32, part, thr, 3 * thr, 0.0)
        keys_e.append(pack_key(pe, base + a + pi))
        keys_h.append(pack_key(qe, base + a + qi))
    assert unpack_key(max(keys_e)) == (single[0][0], base + single[0][1])
    assert unpack_key(max(keys_h)) == (single[1][0], base + single[1][1])


def test_c3_twelve_cameras_batched(env):
    """All cameras in one launch at the quoted size (12 x 50 000 correspondences x 100 000 hypotheses): the matrix-core
    scorer against the all-float64 kernel on every count, winners and a sample of counts against the oracle."""
    torch, C, ops = env
    ncam, m, B = 12, 50000, 100000
    cams = [synthetic.resection_scene(m=m, outlier_ratio=0.5, seed=100 + c) for c in range(ncam)]
    pts = torch.cat([ops.pack_points(C.dev(c["R"].ravel()), C.dev(c["X"]), C.dev(c["x_norm"])) for c in cams])
    idx_h = np.random.default_rng(0).integers(0, m, size=(ncam, B, 2)).astype(np.int32)
    idx = C.dev(idx_h)
    thr = 4.0 / cams[0]["K"][0][0]
    best_m, T_m, counts_m = ops.t_ransac(pts, [m] * ncam, idx, thr, mixed=True)
    best_f, T_f, counts_f = ops.t_ransac(pts, [m] * ncam, idx, thr, mixed=False)
    assert torch.equal(counts_m, counts_f) and torch.equal(best_m, best_f) and torch.equal(T_m, T_f)
    cnt, ind = ops.decode_best(best_m.cpu().numpy())
    ch = counts_m.cpu().numpy()
    Tg = T_m.cpu().numpy()
    rng = np.random.default_rng(5)
    for c, cam in enumerate(cams):
        assert (cnt[c], ind[c]) == (ch[c].max(), int(np.argmax(ch[c])))
        pick = np.concatenate([[ind[c]], rng.integers(0, B, 40)])
        pairs = idx_h[c][pick].astype(np.int64)
        Ts = orc.estimate_T_2point_batch(cam["x_norm"], cam["X"], cam["R"], pairs)
        ref = orc.count_inliers_batch(cam["x_norm"], cam["X"], cam["R"], Ts, thr)
        ok = pairs[:, 0] != pairs[:, 1]
        np.testing.assert_array_equal(ch[c][pick][ok], ref[ok])
        np.testing.assert_allclose(Tg[c], Ts[0], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("sizes", [
    [100, 300, 1000, 3000, 10000, 5000, 2000, 800, 400, 200, 6000, 1500],  # 12 cameras (dataset 10 has 12 entries)
    [150, 9000, 40, 2500, 700, 4000, 1200, 10000, 333, 64, 65, 5000, 2000, 31],  # 14 cameras (dataset 4)
])
def test_c4_ragged_camera_batches(env, sizes):
    """Batched LM refinement (10 iterations, mu = 0.01, dataset-10 intrinsics with fx != fy) over ragged per-camera
    sizes in ONE launch against the oracle's per-camera loop, every iteration; DLT triangulation of the same number
    of points against the oracle."""
    torch, C, ops = env
    Kc = synthetic.K_DATASET10
    cams = [synthetic.resection_scene(m=s, outlier_ratio=0.1, seed=300 + i, K=Kc) for i, s in enumerate(sizes)]
    pts = torch.cat([ops.pack_points(C.dev(c["R"].ravel()), C.dev(c["X"]), C.dev(c["x_norm"])) for c in cams])
    T0 = np.stack([c["T_true"] + np.array([0.02, -0.01, 0.03]) for c in cams])
    T, hist = ops.lm_translation(C.dev(np.asarray(Kc, float).ravel()), pts, sizes, C.dev(T0), 10, 0.01, want_history=True)
    hist = hist.cpu().numpy()
    for j, cam in enumerate(cams):
        ref = []
        Tr = orc.levenberg_marquardt_algorithm(Kc, cam["R"], cam["X"].astype(np.float64), cam["x_norm"], T0[j], 10, 0.01, history=ref)
        ref = np.stack(ref)
        rms = np.sqrt(np.mean(orc.compute_reprojection_error(cam["X"].astype(np.float64), cam["x_norm"], Kc, cam["R"], ref[-1]) ** 2))
        assert np.abs(hist[j] - ref).max() < 1e-4 * rms  # the stated bar
        np.testing.assert_allclose(hist[j], ref, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(T.cpu().numpy()[j], Tr.ravel(), rtol=1e-8, atol=1e-11)
    # triangulation of sum(sizes) points of a two-view scene
    npts = sum(sizes)
    sc = synthetic.two_view_scene(n=npts, inlier_ratio=1.0, planar_ratio=0.3, seed=77, K=Kc, image_size=(760, 500))
    Kn = np.asarray(Kc, float)
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = np.hstack([sc["R"], sc["T"].reshape(3, 1)])
    x1 = np.linalg.inv(Kn) @ sc["x1"]
    x2 = np.linalg.inv(Kn) @ sc["x2"]
    X = ops.triangulate(C.dev(P1.ravel()), C.dev(P2.ravel()), C.dev(x1), C.dev(x2)).cpu().numpy()
    pick = np.random.default_rng(2).integers(0, npts, 1500)
    ref = orc.triangulate_3D_point_DLT([P1, P2], [x1[:, pick], x2[:, pick]])
    np.testing.assert_allclose(X[:, pick], ref, rtol=1e-7, atol=1e-9)

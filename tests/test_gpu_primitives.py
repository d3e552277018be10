"""GPU parity of every C-ABI entry point against the oracle and the golden fixtures.
Run on the MI355X box with -m gpu; all calls go through libsfm_hip.so."""
import os

import numpy as np
import pytest

from oracle import sfm_oracle as orc
from sfm_hip import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from sfm_hip import _capi, ops as o

    _capi.load_library()
    return o


@pytest.fixture(scope="module")
def dev():
    from sfm_hip import _capi

    return _capi.dev


def scene(n=1500, inl=0.5, pl=0.5, seed=11):
    sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=pl, seed=seed)
    return sc, orc.normalize_K(sc["K"], sc["x1"]), orc.normalize_K(sc["K"], sc["x2"])


def up_to_sign(a, b):
    a = a / np.linalg.norm(a)
    b = b / np.linalg.norm(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max())


def test_eh_models_and_counts_golden(ops, dev, golden):
    g = golden("eh_scoring.npz")
    sc, x1, x2 = scene()
    corr = dev(ops.pack_corr(x1, x2))
    idx = dev(g["idx"], np.int32)
    thr = float(g["thr"])
    F, E, H = [t.cpu().numpy() for t in ops.eh_models(corr, idx, want_F=True)]
    for i in range(idx.shape[0]):
        assert up_to_sign(F[i].reshape(3, 3), g["F"][i]) < 1e-9
        assert up_to_sign(E[i].reshape(3, 3), g["E"][i]) < 1e-9
        np.testing.assert_allclose(H[i].reshape(3, 3), g["H"][i], rtol=1e-7, atol=1e-9)
    ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
    np.testing.assert_array_equal(ce.cpu().numpy(), g["cE"])
    np.testing.assert_array_equal(ch.cpu().numpy(), g["cH"])
    # masks of single models
    for i in (0, 3, 7):
        m, c = ops.epipolar_mask(dev(g["E"][i].ravel()), corr, thr)
        ref = np.unpackbits(g["maskE"][i])[:1500]
        np.testing.assert_array_equal(m.cpu().numpy(), ref)
        assert int(c.cpu()) == ref.sum()
        m, c = ops.transfer_mask(dev(g["H"][i].ravel()), corr, 3 * thr)
        ref = np.unpackbits(g["maskH"][i])[:1500]
        np.testing.assert_array_equal(m.cpu().numpy(), ref)
        assert int(c.cpu()) == ref.sum()


@pytest.mark.parametrize("n,B", [(800, 5000), (257, 700), (3000, 300)])
def test_eh_counts_vs_oracle(ops, dev, n, B):
    sc, x1, x2 = scene(n=n, inl=0.6, pl=0.5, seed=3)
    rng = np.random.default_rng(n)
    idx = rng.integers(0, n, size=(B, 12)).astype(np.int32)
    thr = 1.0 / sc["K"][0][0]
    cE, cH, _, _ = orc.score_EH_batch(x1, x2, idx, thr)
    ce, ch = ops.eh_score(dev(ops.pack_corr(x1, x2)), dev(idx), thr, 3 * thr)
    ce, ch = ce.cpu().numpy(), ch.cpu().numpy()
    okE = np.array([len(set(r[:8])) == 8 for r in idx])
    okH = np.array([len(set(r[8:])) == 4 for r in idx])
    np.testing.assert_array_equal(ce[okE], cE[okE])
    np.testing.assert_array_equal(ch[okH], cH[okH])


def test_triangulation_golden(ops, dev, golden):
    t = golden("triangulate.npz")
    sc, x1, x2 = scene()
    n = int(t["n"])
    X = ops.triangulate(dev(t["P1"].ravel()), dev(t["P2"].ravel()), dev(x1[:, :n]), dev(x2[:, :n])).cpu().numpy()
    np.testing.assert_allclose(X, t["X"], rtol=1e-9, atol=1e-9)
    Kn = np.asarray(sc["K"], float)
    X = ops.triangulate(dev((Kn @ t["P1"]).ravel()), dev((Kn @ t["P2"]).ravel()), dev(sc["x1"][:, :n]), dev(sc["x2"][:, :n])).cpu().numpy()
    np.testing.assert_allclose(X, t["Xpix"], rtol=1e-7, atol=1e-7)


def test_decompositions_golden(ops, dev, golden):
    g = golden("eh_scoring.npz")
    d = golden("decomp.npz")
    sc, x1, x2 = scene()
    corr = dev(ops.pack_corr(x1, x2))
    for k, i in enumerate(d["sel"]):
        Ed = dev(g["E"][i].ravel())
        P = ops.essential_candidates(Ed)
        counts = ops.chirality4(P, corr).cpu().numpy()
        cands = [p.reshape(3, 4) for p in P.cpu().numpy()]
        assert sorted(counts) == sorted(orc.chirality_counts(cands, x1, x2))
        best = int(np.argmax(counts))
        assert counts[best] == d["n_e"][k]
        np.testing.assert_allclose(cands[best], np.hstack([d["R_e"][k], d["T_e"][k].reshape(3, 1)]), atol=1e-9)
        # masked subset, as used at homography record events
        m = np.unpackbits(g["maskE"][i])[:1500].astype(np.uint8)
        counts = ops.chirality4(P, corr, dev(m)).cpu().numpy()
        best = int(np.argmax(counts))
        assert counts[best] == d["n_m"][k]
        np.testing.assert_allclose(cands[best], np.hstack([d["R_m"][k], d["T_m"][k].reshape(3, 1)]), atol=1e-9)
        out = ops.homography_to_rt(dev(d["H"][k].ravel()), corr).cpu().numpy()
        sols = [(out[:9].reshape(3, 3), out[9:12]), (out[12:21].reshape(3, 3), out[21:24])]
        for Rr, tr in ((d["R1"][k], d["t1"][k]), (d["R2"][k], d["t2"][k])):
            assert min(max(np.abs(R - Rr).max(), np.abs(t - tr).max()) for R, t in sols) < 1e-8
        np.testing.assert_allclose(out[24:33].reshape(3, 3), orc.skew_symmetric_mat(sols[0][1]) @ sols[0][0], atol=1e-12)
        np.testing.assert_allclose(out[33:42].reshape(3, 3), orc.skew_symmetric_mat(sols[1][1]) @ sols[1][0], atol=1e-12)
    keep = sc["is_inlier"].astype(np.uint8)
    counts = ops.chirality4(dev(d["P2s"].reshape(4, 12)), corr, dev(keep)).cpu().numpy()
    np.testing.assert_array_equal(counts, d["chir_counts"])


def resection(arr, K=synthetic.K_DATASET5, **kw):
    m, out, seed = arr
    return synthetic.resection_scene(m=int(m), outlier_ratio=float(out), seed=int(seed), K=K, **kw)


def test_t_two_point_golden(ops, dev, golden):
    g = golden("t2pt.npz")
    sc = resection(g["scene"])
    pts = ops.pack_points(dev(sc["R"].ravel()), dev(sc["X"]), dev(sc["x_norm"]))
    RX = sc["R"] @ sc["X"].astype(np.float64)
    np.testing.assert_allclose(pts.cpu().numpy()[:, :3], RX.T, rtol=1e-14, atol=1e-15)
    np.testing.assert_array_equal(pts.cpu().numpy()[:, 3:], sc["x_norm"].T)
    idx = g["idx"].astype(np.int32)
    T = ops.t_models(pts, dev(idx)).cpu().numpy()
    np.testing.assert_allclose(T, g["T"], rtol=1e-8, atol=1e-10)
    best, Tb, counts = ops.t_ransac(pts, [int(g["scene"][0])], dev(idx[None]), float(g["thr"]))
    np.testing.assert_array_equal(counts.cpu().numpy()[0], g["counts"])
    cnt, ind = ops.decode_best(best.cpu().numpy())
    assert cnt[0] == g["counts"].max() and ind[0] == int(np.argmax(g["counts"]))
    np.testing.assert_allclose(Tb.cpu().numpy()[0], g["T"][ind[0]], rtol=1e-8, atol=1e-10)


def test_t_ransac_full_run_golden_batched(ops, dev, golden):
    """The 100000-iteration run of the reference (fixture) as camera 1 of a ragged 3-camera batch."""
    g = golden("t_ransac.npz")
    sc = resection(g["scene"])
    others = [synthetic.resection_scene(m=257, outlier_ratio=0.3, seed=90), synthetic.resection_scene(m=1000, outlier_ratio=0.6, seed=91)]
    cams = [others[0], sc, others[1]]
    B = 100000
    np.random.seed(int(g["seed"]))
    idx1 = np.random.randint(0, 300, size=2 * B).reshape(B, 2)
    rng = np.random.default_rng(4)
    idx = np.stack([rng.integers(0, 257, (B, 2)), idx1, rng.integers(0, 1000, (B, 2))]).astype(np.int32)
    pts = [ops.pack_points(dev(c["R"].ravel()), dev(c["X"]), dev(c["x_norm"])) for c in cams]
    import torch

    best, Tb, counts = ops.t_ransac(torch.cat(pts), [257, 300, 1000], dev(idx), 4.0 / sc["K"][0][0])
    cnt, ind = ops.decode_best(best.cpu().numpy())
    assert ind[1] == int(g["best_index"]) and cnt[1] == int(g["best_count"])
    np.testing.assert_allclose(Tb.cpu().numpy()[1], g["T"], rtol=1e-8, atol=1e-11)
    distinct = idx1[:, 0] != idx1[:, 1]
    np.testing.assert_array_equal(counts.cpu().numpy()[1][distinct], g["counts"][distinct].astype(np.int32))
    # the other cameras against the oracle
    for c, cam in ((0, cams[0]), (2, cams[2])):
        Ts = orc.estimate_T_2point_batch(cam["x_norm"], cam["X"], cam["R"], idx[c][:3000])
        ref = orc.count_inliers_batch(cam["x_norm"], cam["X"], cam["R"], Ts, 4.0 / sc["K"][0][0])
        ok = idx[c][:3000, 0] != idx[c][:3000, 1]
        np.testing.assert_array_equal(counts.cpu().numpy()[c][:3000][ok], ref[ok])


def test_lm_golden_batched(ops, dev, golden):
    g = golden("lm.npz")
    import torch

    # cameras share K per launch: group by intrinsics
    for Kc in (synthetic.K_DATASET5, synthetic.K_DATASET10):
        ks = [k for k in range(len(g["m"])) if np.allclose(g["K"][k], np.asarray(Kc, float))]
        scs = [synthetic.resection_scene(m=int(g["m"][k]), outlier_ratio=0.0, seed=int(g["seed"][k]), K=Kc, noise_px=0.5) for k in ks]
        pts = torch.cat([ops.pack_points(dev(s["R"].ravel()), dev(s["X"]), dev(s["x_norm"])) for s in scs])
        T, hist = ops.lm_translation(dev(np.asarray(Kc, float).ravel()), pts, [int(g["m"][k]) for k in ks], dev(g["T0"][ks]), 10, 0.01, want_history=True)
        for j, k in enumerate(ks):
            ref = g["hist"][k]
            rms = np.sqrt(np.mean(orc.compute_reprojection_error(scs[j]["X"], scs[j]["x_norm"], Kc, scs[j]["R"], ref[-1]) ** 2))
            # tolerance: 1e-4 of the reprojection RMS (the stated bar); fp64 lands far inside it
            assert np.abs(hist.cpu().numpy()[j] - ref).max() < 1e-4 * rms
            np.testing.assert_allclose(hist.cpu().numpy()[j], ref, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(T.cpu().numpy()[j], ref[-1], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n,B,seed", [(2000, 20000, 1), (500, 4000, 2), (10000, 3000, 3)])
def test_f32_upper_bounds_and_two_tier_winner(ops, dev, n, B, seed):
    """The float32 pre-filter never undercounts, and eh_best returns the exact first maximum."""
    sc, x1, x2 = scene(n=n, inl=0.4, pl=0.5, seed=seed)
    corr_h = ops.pack_corr(x1, x2)
    corr = dev(corr_h)
    c32 = ops.corr_to_f32(corr)
    xmax = float(np.abs(corr_h).max())
    rng = np.random.default_rng(seed)
    idx_h = rng.integers(0, n, size=(B, 12)).astype(np.int32)
    idx = dev(idx_h)
    thr = 1.0 / sc["K"][0][0]
    ce, ch = [t.cpu().numpy() for t in ops.eh_score(corr, idx, thr, 3 * thr)]
    ue, uh = [t.cpu().numpy() for t in ops.eh_upper(corr, c32, idx, thr, 3 * thr, xmax)]
    assert (ue >= ce).all() and (uh >= ch).all()
    assert (ue - ce).mean() < 0.02 * n and (uh - ch).mean() < 0.02 * n  # and they are tight
    (cE, iE), (cH, iH), stats = ops.eh_best(corr, c32, idx, thr, 3 * thr, xmax)
    assert (cE, iE) == (int(ce.max()), int(np.argmax(ce)))
    assert (cH, iH) == (int(ch.max()), int(np.argmax(ch)))
    assert stats["candidates_e"] >= 1 and stats["candidates_h"] >= 1
    # and against the oracle for the winners themselves
    oE, oH, _, _ = orc.score_EH_batch(x1, x2, idx_h[[iE, iH]], thr)
    assert oE[0] == cE and oH[1] == cH


def test_two_tier_with_many_ties(ops, dev):
    """All-outlier data: counts are tiny and tie massively; the lowest id must still win."""
    rng = np.random.default_rng(5)
    n, B = 300, 5000
    x1 = np.vstack([rng.uniform(-0.4, 0.4, (2, n)), np.ones((1, n))])
    x2 = np.vstack([rng.uniform(-0.4, 0.4, (2, n)), np.ones((1, n))])
    corr_h = ops.pack_corr(x1, x2)
    corr = dev(corr_h)
    idx = dev(rng.integers(0, n, size=(B, 12)), np.int32)
    thr = 4e-4
    ce, ch = [t.cpu().numpy() for t in ops.eh_score(corr, idx, thr, 3 * thr)]
    (cE, iE), (cH, iH), stats = ops.eh_best(corr, ops.corr_to_f32(corr), idx, thr, 3 * thr, float(np.abs(corr_h).max()))
    assert (cE, iE) == (int(ce.max()), int(np.argmax(ce)))
    assert (cH, iH) == (int(ch.max()), int(np.argmax(ch)))


def test_t_two_tier_matches_f64(ops, dev):
    """float32 bounds of the translation RANSAC never undercount; t_best equals t_ransac
    (count, first-maximum index and T) for a ragged batch incl. a tie-heavy camera."""
    import torch

    cams = [synthetic.resection_scene(m=m, outlier_ratio=o, seed=s) for m, o, s in ((700, 0.5, 1), (257, 0.2, 2), (3000, 0.7, 3), (64, 0.95, 4))]
    sizes = [c["x_norm"].shape[1] for c in cams]
    B = 20000
    rng = np.random.default_rng(0)
    idx = dev(np.stack([rng.integers(0, m, (B, 2)) for m in sizes]), np.int32)
    pts = torch.cat([ops.pack_points(dev(c["R"].ravel()), dev(c["X"]), dev(c["x_norm"])) for c in cams])
    pts32 = ops.pts_to_f32(pts)
    thr = 4.0 / cams[0]["K"][0][0]
    ph = pts.cpu().numpy()
    off = np.concatenate([[0], np.cumsum(sizes)])
    rmax = [np.abs(ph[off[c] : off[c + 1], :3]).max() for c in range(len(cams))]
    xmax = float(np.abs(ph[:, 3:5]).max())
    best, T, counts = ops.t_ransac(pts, sizes, idx, thr)
    up = ops.t_upper(pts, pts32, sizes, idx, thr, rmax, xmax)
    assert bool((up >= counts).all())
    assert float((up - counts).float().mean()) < 0.5
    cnt, ind = ops.decode_best(best.cpu().numpy())
    c2, i2, T2 = ops.t_best(pts, pts32, sizes, idx, thr, rmax, xmax)
    np.testing.assert_array_equal(c2, cnt)
    np.testing.assert_array_equal(i2, ind)
    np.testing.assert_allclose(T2, T.cpu().numpy(), rtol=0, atol=0)


@pytest.mark.parametrize("inl,pl,n,B,seed", [(0.3, 0.5, 5000, 100000, 0), (0.05, 0.5, 4000, 60000, 3), (0.0, 0.5, 3000, 40000, 6), (0.6, 0.0, 1000, 30000, 7)])
def test_loose_bound_cascade_is_exact(ops, dev, inl, pl, n, B, seed, monkeypatch):
    """The cheaper (one-sided / box) bounds hold, and the cascade (loose bounds -> tight bounds for many
    candidates -> float64) returns the first maximum of the float64 counts in regimes from a dominant
    plane to pure outliers (where thousands of hypotheses survive the loose bound)."""
    sc, x1, x2 = scene(n=n, inl=inl, pl=pl, seed=seed)
    corr_h = ops.pack_corr(x1, x2)
    corr = dev(corr_h)
    c32 = ops.corr_to_f32(corr)
    xmax = float(np.abs(corr_h).max())
    idx = dev(np.random.default_rng(seed).integers(0, n, size=(B, 12)), np.int32)
    thr = 1.0 / sc["K"][0][0]
    ce, ch = [t.cpu().numpy() for t in ops.eh_score(corr, idx, thr, 3 * thr)]
    ue, uh = [t.cpu().numpy() for t in ops.eh_upper(corr, c32, idx, thr, 3 * thr, xmax, one_sided=True)]
    te, th = [t.cpu().numpy() for t in ops.eh_upper(corr, c32, idx, thr, 3 * thr, xmax, one_sided=False)]
    assert (ue >= te).all() or (ue >= ce).all()
    assert (ue >= ce).all() and (uh >= ch).all() and (te >= ce).all() and (th >= ch).all()
    monkeypatch.setattr(ops, "TIGHTEN_ABOVE", 64)  # force the middle level to run in the sparse regimes
    for loose in (True, False):
        (cE, iE), (cH, iH), stats = ops.eh_best(corr, c32, idx, thr, 3 * thr, xmax, loose=loose)
        assert (cE, iE) == (int(ce.max()), int(np.argmax(ce))), stats
        assert (cH, iH) == (int(ch.max()), int(np.argmax(ch))), stats


@pytest.mark.parametrize("inl,pl,n,B,seed,loose", [(0.3, 0.5, 5000, 100000, 0, 1), (0.3, 0.5, 5000, 100000, 0, 0),
                                                   (0.0, 0.5, 3000, 200000, 6, 1), (0.05, 0.5, 777, 70001, 3, 1),
                                                   (0.3, 0.5, 5000, 100000, 0, 2), (0.0, 0.5, 3000, 200000, 6, 2),
                                                   (0.05, 0.5, 777, 70001, 3, 2), (0.6, 1.0, 31, 1000, 4, 2), (0.5, 0.0, 33, 257, 5, 2)])
def test_native_cascade_entry_point(ops, dev, inl, pl, n, B, seed, loose):
    """sfm_eh_best (the whole cascade inside the library, caller-provided workspace) returns the
    first maximum of the float64 counts, also when thousands of candidates go through the tight
    re-bounding level (pure-outlier scene) and for sizes that are not multiples of anything."""
    sc, x1, x2 = scene(n=n, inl=inl, pl=pl, seed=seed)
    corr_h = ops.pack_corr(x1, x2)
    corr = dev(corr_h)
    c32 = ops.corr_to_f32(corr)
    xmax = float(np.abs(corr_h).max())
    idx = dev(np.random.default_rng(seed).integers(0, n, size=(B, 12)), np.int32)
    thr = 1.0 / sc["K"][0][0]
    ce, ch = [t.cpu().numpy() for t in ops.eh_score(corr, idx, thr, 3 * thr)]
    ws = ops.eh_best_workspace(B, n)
    for _ in range(2):  # the workspace is reused: stale contents must not matter
        (cE, iE), (cH, iH), stats = ops.eh_best_native(corr, c32, idx, thr, 3 * thr, xmax, loose=loose, workspace=ws)
        assert (cE, iE) == (int(ce.max()), int(np.argmax(ce))), stats
        assert (cH, iH) == (int(ch.max()), int(np.argmax(ch))), stats
    if (inl, loose) in ((0.0, 1), (0.05, 2)):  # the scenes on which the first level of that kind leaves thousands of candidates
        assert stats["tightened"] > 4096  # the middle level really ran


def test_native_cascade_universal_ties_fall_back_to_float64(ops, dev):
    """Identical matches: every hypothesis has the same count, more candidates than the lists hold."""
    n, B = 200, 100000
    corr_h = np.tile(np.array([[0.1, -0.2, 0.12, -0.18]]), (n, 1))
    corr = dev(corr_h)
    idx = dev(np.random.default_rng(0).integers(0, n, size=(B, 12)), np.int32)
    ce, ch = [t.cpu().numpy() for t in ops.eh_score(corr, idx, 4e-4, 1.2e-3)]
    (cE, iE), (cH, iH), stats = ops.eh_best_native(corr, ops.corr_to_f32(corr), idx, 4e-4, 1.2e-3, 0.2)
    assert (cE, iE) == (int(ce.max()), int(np.argmax(ce))), stats
    assert (cH, iH) == (int(ch.max()), int(np.argmax(ch))), stats


def test_native_cascade_many_loose_survivors(ops, dev):
    """A scene whose best counts are tiny: more hypotheses survive the loose bounds than the
    candidate lists hold, so the tight bounds are applied to everything before the float64 level."""
    n, B = 3000, 600000
    sc, x1, x2 = scene(n=n, inl=0.0, pl=0.5, seed=8)
    corr_h = ops.pack_corr(x1, x2)
    corr = dev(corr_h)
    idx = dev(np.random.default_rng(8).integers(0, n, size=(B, 12)), np.int32)
    thr = 1.0 / sc["K"][0][0]
    ce, ch = [t.cpu().numpy() for t in ops.eh_score(corr, idx, thr, 3 * thr)]
    for loose in (1, 2):
        (cE, iE), (cH, iH), stats = ops.eh_best_native(corr, ops.corr_to_f32(corr), idx, thr, 3 * thr, float(np.abs(corr_h).max()), loose=loose)
        assert (cE, iE) == (int(ce.max()), int(np.argmax(ce))), stats
        assert (cH, iH) == (int(ch.max()), int(np.argmax(ch))), stats
        assert stats["candidates_e"] > 65536 and stats["tightened"] == B and stats["rescored_f64"] < B, stats


def test_device_index_sets_match_oracle_and_are_sharding_invariant(ops):
    """sfm_index_sets_i32 against the NumPy Philox restatement, and: any split of an id range gives the same sets."""
    import torch

    from oracle import philox

    for seed, first, nb, n, width in [(0, 0, 100000, 10000, 12), (2**63 + 12345, 2**32 - 5000, 10001, 200000, 12),
                                      (7, 123456789012, 7777, 50000, 2), (99, 0, 1, 1, 12)]:
        got = ops.index_sets(seed, first, nb, n, width).cpu().numpy()
        assert (got == philox.index_sets(seed, first, nb, n, width)).all()
    whole = ops.index_sets(11, 1000, 30000, 200000)
    parts = torch.cat([ops.index_sets(11, 1000, 12345, 200000), ops.index_sets(11, 1000 + 12345, 30000 - 12345, 200000)])
    assert torch.equal(whole, parts)
    assert ops.index_sets(1, 0, 0, 10).shape == (0, 12)
    with pytest.raises(Exception):
        ops.index_sets(1, 0, 10, 0)


@pytest.mark.parametrize("n,B,inl,seed", [(1500, 20000, 0.5, 1), (10000, 50000, 0.3, 0), (97, 4099, 0.8, 2), (32, 64, 0.9, 3), (5000, 3000, 0.0, 4)])
def test_matrix_core_bounds_are_upper_bounds(ops, dev, n, B, inl, seed):
    """sfm_eh_upper_mfma_f32 (s = x2' E x1, k n1 and H x1 on the bf16 matrix cores from three-piece splits): never below
    the exact float64 count; the transfer bound about as tight as the vector-ALU form of the same test, the epipolar bound
    (whose s carries the error of three chained matrix instructions) within a few counts of it."""
    sc, x1, x2 = scene(n=n, inl=inl, pl=0.5, seed=seed)
    corr_h = ops.pack_corr(x1, x2)
    corr = dev(corr_h)
    xmax = float(np.abs(corr_h).max())
    idx = ops.index_sets(seed, 0, B, n)
    thr = 1.0 / sc["K"][0][0]
    ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
    me, mh = ops.eh_upper_mfma(corr, ops.corr_to_mfma(corr, thr), idx, thr, 3 * thr, xmax)
    ve, vh = ops.eh_upper(corr, ops.corr_to_f32(corr), idx, thr, 3 * thr, xmax, one_sided=True)
    assert int((me < ce).sum()) == 0 and int((mh < ch).sum()) == 0
    assert float((me - ce).double().mean()) <= float((ve - ce).double().mean()) * 3.5 + 1.0
    assert float((mh - ch).double().mean()) <= float((vh - ch).double().mean()) * 1.2 + 0.5


def test_submit_collect_two_jobs_in_flight(ops, dev):
    """sfm_eh_best_submit / _collect: two blocks of hypotheses in flight on two streams give what the blocking call gives,
    in the usual case (few candidates, answered by the speculative level) and with thousands of candidates."""
    import torch

    for inl, n, B, seed in [(0.3, 5000, 100000, 0), (0.0, 3000, 200000, 6)]:
        sc, x1, x2 = scene(n=n, inl=inl, pl=0.5, seed=seed)
        corr_h = ops.pack_corr(x1, x2)
        corr = dev(corr_h)
        c32 = ops.corr_to_f32(corr)
        xmax = float(np.abs(corr_h).max())
        thr = 1.0 / sc["K"][0][0]
        idxs = [ops.index_sets(seed + k, 0, B, n) for k in range(2)]
        want = [ops.eh_best_native(corr, c32, i, thr, 3 * thr, xmax) for i in idxs]
        exact = [[t.cpu().numpy() for t in ops.eh_score(corr, i, thr, 3 * thr)] for i in idxs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        jobs = [ops.eh_best_submit(corr, c32, i, thr, 3 * thr, xmax, stream=s) for i, s in zip(idxs, streams)]
        got = [ops.eh_best_collect(j) for j in jobs]
        for g, w, (ce, ch) in zip(got, want, exact):
            assert g[0] == w[0] and g[1] == w[1]
            assert g[0] == (int(ce.max()), int(np.argmax(ce))) and g[1] == (int(ch.max()), int(np.argmax(ch)))
        if inl == 0.0:
            assert got[0][2]["candidates_e"] + got[0][2]["candidates_h"] > 512  # beyond the speculative level


def test_mixed_translation_ransac_counts_are_identical_to_float64(ops, dev):
    """sfm_t_ransac_mixed (matrix cores + float32, float64 inside the error band) against sfm_t_ransac_f64: every count of
    every hypothesis, the winner and its T; cameras of ragged sizes, duplicated sample indices, correspondences placed
    on and next to the threshold circle, and a degenerate camera whose samples give non-finite translations."""
    import torch

    K = synthetic.K_DATASET5
    thr = 4.0 / K[0][0]
    rng = np.random.default_rng(7)
    sizes = [5000, 33, 1, 777, 2048]
    cams = [synthetic.resection_scene(m=max(s, 2), outlier_ratio=0.5, seed=40 + i) for i, s in enumerate(sizes)]
    packs = []
    for c, s in zip(cams, sizes):
        packs.append(ops.pack_points(dev(c["R"].ravel()), dev(c["X"]), dev(c["x_norm"]))[:s].contiguous())
    # adversarial camera: projections placed at distance thr * (1 + k * 1e-7) around the true reprojection
    c = synthetic.resection_scene(m=4000, outlier_ratio=0.0, seed=99)
    p = ops.pack_points(dev(c["R"].ravel()), dev(c["X"]), dev(c["x_norm"])).cpu().numpy()
    T = c["T_true"]
    proj = (p[:, :3] + T)
    uv = proj[:, :2] / proj[:, 2:3]
    ang = rng.uniform(0, 2 * np.pi, 4000)
    k = rng.integers(-40, 41, 4000)
    rad = thr * (1.0 + k * 1e-7)
    p[:, 3] = uv[:, 0] + rad * np.cos(ang)
    p[:, 4] = uv[:, 1] + rad * np.sin(ang)
    p[:2, 3:5] = uv[:2]  # two exact correspondences, so that the sample (0, 1) reproduces T_true
    packs.append(dev(p))
    sizes.append(4000)
    # degenerate camera: all points identical -> singular 2-point systems
    q = np.tile(p[:1], (64, 1))
    packs.append(dev(q))
    sizes.append(64)
    pts = torch.cat(packs)
    B = 5000
    idx_h = np.stack([rng.integers(0, s, size=(B, 2)) for s in sizes]).astype(np.int32)
    idx_h[:, :50, 1] = idx_h[:, :50, 0]  # duplicated indices
    idx_h[5, 0] = (0, 1)
    idx = dev(idx_h)
    b64, T64, c64 = ops.t_ransac(pts, sizes, idx, thr, mixed=False)
    bmx, Tmx, cmx = ops.t_ransac(pts, sizes, idx, thr, mixed=True)
    assert torch.equal(c64, cmx)
    assert torch.equal(b64, bmx)
    assert np.array_equal(T64.cpu().numpy(), Tmx.cpu().numpy(), equal_nan=True)
    assert int(c64[5, 0]) >= 2  # the adversarial camera really has points on both sides of the circle
    assert 100 < int(c64[5, 0]) < 3900


@pytest.mark.parametrize("scale", [1e-12, 1e-7, 1e-3, 1.0, 1e4, 1e9, 1e14, 1e20])
@pytest.mark.parametrize("thr_px", [1e-6, 4.0, 2000.0])
def test_mixed_translation_ransac_in_other_scales_and_thresholds(ops, dev, scale, thr_px):
    """The same comparison with the scene (R X and hence T) scaled by powers of ten, from scales whose float32 products
    underflow to ones where they overflow (those hypotheses take the float64 test everywhere), and with thresholds from a
    millionth of a pixel to more than the image: every count equals the float64 kernel's at that scale."""
    import torch

    K = synthetic.K_DATASET5
    thr = thr_px / K[0][0]
    rng = np.random.default_rng(3)
    sizes = [3000, 65]
    packs = []
    for i, m in enumerate(sizes):
        c = synthetic.resection_scene(m=m, outlier_ratio=0.4, seed=70 + i)
        p = ops.pack_points(dev(c["R"].ravel()), dev(c["X"]), dev(c["x_norm"])).cpu().numpy()
        if i == 0 and thr_px == 4.0:  # a ring of projections next to the threshold circle of the true translation
            proj = p[:, :3] + c["T_true"]
            uv = proj[:, :2] / proj[:, 2:3]
            ang = rng.uniform(0, 2 * np.pi, m)
            rad = thr * (1.0 + rng.integers(-40, 41, m) * 1e-7)
            p[:, 3] = uv[:, 0] + rad * np.cos(ang)
            p[:, 4] = uv[:, 1] + rad * np.sin(ang)
            p[:2, 3:5] = uv[:2]
        p[:, :3] *= scale
        packs.append(dev(p))
    pts = torch.cat(packs)
    B = 3000
    idx_h = np.stack([rng.integers(0, m, size=(B, 2)) for m in sizes]).astype(np.int32)
    idx_h[0, 0] = (0, 1)
    idx = dev(idx_h)
    b64, T64, c64 = ops.t_ransac(pts, sizes, idx, thr, mixed=False)
    bmx, Tmx, cmx = ops.t_ransac(pts, sizes, idx, thr, mixed=True)
    assert torch.equal(c64, cmx)
    assert torch.equal(b64, bmx)
    assert np.array_equal(T64.cpu().numpy(), Tmx.cpu().numpy(), equal_nan=True)
    if thr_px == 4.0 and 1e-7 <= scale <= 1e9:
        assert 100 < int(c64[0, 0]) < 2900  # the ring really has points on both sides of the circle


@pytest.mark.parametrize("n,B,inl,pl,px,seed", [(1500, 20000, 0.5, 0.5, 1.0, 1), (10000, 60000, 0.3, 0.5, 1.0, 0), (97, 4099, 0.8, 1.0, 0.5, 2),
                                                (12, 64, 0.9, 0.0, 3.0, 3), (5000, 3000, 0.0, 0.5, 1.0, 4), (2048, 257, 0.95, 0.5, 1.0, 5)])
def test_mixed_eh_scoring_counts_are_identical_to_float64(ops, dev, n, B, inl, pl, px, seed, mixed_form):
    """sfm_eh_score_mixed (matrix cores + float32, float64 inside the error bands) against sfm_eh_score_f64: every count."""
    import torch

    sc, x1, x2 = scene(n=n, inl=inl, pl=pl, seed=seed)
    corr_h = ops.pack_corr(x1, x2)
    corr = dev(corr_h)
    idx = ops.index_sets(seed, 0, B, n)
    thr = px / sc["K"][0][0]
    ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
    me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr, float(np.abs(corr_h).max()))
    assert torch.equal(ce, me) and torch.equal(ch, mh)


def test_mixed_eh_scoring_on_the_threshold_and_degenerate_inputs(ops, dev, mixed_form):
    """Matches constructed to lie within 1e-7 (relative) of the transfer threshold of a known homography and of the
    epipolar threshold of a known essential matrix, duplicated sample indices, and a set of identical matches."""
    import torch

    rng = np.random.default_rng(12)
    n = 4000
    thr = 1.0 / 2400.0
    x1 = rng.uniform(-0.4, 0.4, size=(2, n))
    # planar scene under a rotation + translation -> exact homography; second view displaced by ~3*thr from H x1
    Hm = np.array([[0.995, -0.01, 0.02], [0.012, 1.003, -0.015], [0.03, -0.02, 1.0]])
    p = Hm @ np.vstack([x1, np.ones(n)])
    x2 = p[:2] / p[2]
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = 3 * thr * (1.0 + rng.integers(-40, 41, n) * 1e-7)
    x2 = x2 + rad * np.vstack([np.cos(ang), np.sin(ang)])
    x2[:, :40] = (p[:2] / p[2])[:, :40]  # exact matches, so that samples drawn from them reproduce H (and a valid E)
    corr_h = np.ascontiguousarray(np.vstack([x1, x2]).T)
    corr = dev(corr_h)
    B = 3000
    idx_h = rng.integers(0, n, size=(B, 12)).astype(np.int32)
    idx_h[:500] = rng.integers(0, 40, size=(500, 12))  # samples among the exact matches
    idx_h[500:520, 1] = idx_h[500:520, 0]  # duplicated indices
    idx = dev(idx_h)
    ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
    me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr, float(np.abs(corr_h).max()))
    assert torch.equal(ce, me) and torch.equal(ch, mh)
    assert 500 < int(ch[:500].max()) < n - 500  # the constructed matches really straddle the threshold
    same = dev(np.tile(np.array([[0.1, -0.2, 0.12, -0.18]]), (200, 1)))
    idx2 = ops.index_sets(3, 0, 1000, 200)
    a = ops.eh_score(same, idx2, 4e-4, 1.2e-3)
    b = ops.eh_score_mixed(same, idx2, 4e-4, 1.2e-3, 0.2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_mixed_eh_scoring_in_other_coordinate_scales_and_near_the_epipoles(ops, dev, mixed_form):
    """The exact scorer takes s = x2^T E x1 and the squared line norms whole from the matrix cores and decides
    s^2 (n1 + n2) < k n1 n2 with a band derived from the coordinate bound: the counts must equal the float64 kernel's for
    pixel coordinates (|x| up to 2000, thresholds in pixels), for tiny coordinates, for matches piled up around both
    epipoles (vanishing line norms), for thresholds from 0.01 to 30 px, and with a few wild coordinates in the set."""
    import torch

    sc, x1, x2 = scene(n=3000, inl=0.5, pl=0.3, seed=21)
    K = np.asarray(sc["K"], np.float64)
    B = 6000
    rng = np.random.default_rng(5)
    cases = []
    for name, scale, thr_px in (("pixels", K[0, 0], 1.0), ("pixels, tight", K[0, 0], 0.01), ("pixels, loose", K[0, 0], 30.0),
                                ("normalised", 1.0, 1.0), ("tiny", 1e-4, 1.0), ("large", 6.0, 2.0)):
        c = ops.pack_corr(x1, x2) * scale
        cases.append((name, c, thr_px * scale / K[0, 0]))
    # matches around the epipoles of the true geometry: E e1 = 0, E^T e2 = 0
    F = orc.estimate_F_DLT(x1[:, sc["is_inlier"]][:, :200], x2[:, sc["is_inlier"]][:, :200])
    E = orc.enforce_essential(F)
    u, sv, vt = np.linalg.svd(E)
    e1, e2 = vt[2] / vt[2, 2], u[:, 2] / u[2, 2]
    c = ops.pack_corr(x1, x2).copy()
    if np.all(np.isfinite(e1)) and np.all(np.isfinite(e2)) and max(np.abs(e1).max(), np.abs(e2).max()) < 50:
        c[:600, 0:2] = e1[:2] + rng.normal(0, 1e-4, (600, 2))
        c[300:900, 2:4] = e2[:2] + rng.normal(0, 1e-4, (600, 2))
    cases.append(("epipoles", c, 1.0 / K[0, 0]))
    c = ops.pack_corr(x1, x2).copy()
    c[::97] *= 300.0  # a few wild matches raise the coordinate bound of the whole set
    cases.append(("wild", c, 1.0 / K[0, 0]))
    for name, c, thr in cases:
        corr = dev(np.ascontiguousarray(c))
        idx_h = rng.integers(0, c.shape[0], size=(B, 12)).astype(np.int32)
        inl = np.flatnonzero(sc["is_inlier"])
        idx_h[:1500] = inl[rng.integers(0, len(inl), size=(1500, 12))]  # many good models: dense around the threshold
        idx = dev(idx_h)
        ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
        me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr)
        assert torch.equal(ce, me), (name, int((ce != me).sum()), int((ce - me).abs().max()))
        assert torch.equal(ch, mh), (name, int((ch != mh).sum()))
        assert int(ce.max()) > 0


def test_previous_form_of_the_exact_scorer_still_agrees(ops, dev):
    """SFM_HIP_EHX_FORM=0 keeps the previous form of the exact scorer (rows E x1 and E^T x2 from the matrix cores, the test on the vector ALU) reachable (it is read once per process: child process)."""
    import subprocess
    import sys

    code = (
        "import sys, os, numpy as np, torch\n"
        "sys.path.insert(0, os.path.join(os.getcwd(), 'structure-from-motion_amd'))\n"
        "from sfm_hip import _capi, ops, synthetic\n"
        "_capi.load_library()\n"
        "sc = synthetic.two_view_scene(n=2500, inlier_ratio=0.4, planar_ratio=0.5, seed=4)\n"
        "Kinv = np.linalg.inv(np.asarray(sc['K'], float))\n"
        "x1 = Kinv @ np.vstack((sc['x1'][:2], np.ones((1, 2500)))); x2 = Kinv @ np.vstack((sc['x2'][:2], np.ones((1, 2500))))\n"
        "corr = _capi.dev(ops.pack_corr(x1, x2)); idx = ops.index_sets(1, 0, 120000, 2500); thr = 1.0 / sc['K'][0][0]\n"
        "a = ops.eh_score(corr, idx, thr, 3 * thr); b = ops.eh_score_mixed(corr, idx, thr, 3 * thr)\n"
        "assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])\n"
        "print('same')\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SFM_HIP_EHX_FORM="0")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0 and "same" in out.stdout, out.stderr[-2000:]


def test_matrix_core_bounds_with_operands_of_another_threshold_stay_valid(ops, dev):
    """The match-side operands carry the threshold they were prepared for; with any other threshold the kernel reports the
    trivial bound (every match) instead of a wrong one."""
    sc, x1, x2 = scene(n=700, inl=0.5, pl=0.5, seed=2)
    corr_h = ops.pack_corr(x1, x2)
    corr = dev(corr_h)
    idx = ops.index_sets(2, 0, 3000, 700)
    thr = 1.0 / sc["K"][0][0]
    me, mh = ops.eh_upper_mfma(corr, ops.corr_to_mfma(corr, 2 * thr), idx, thr, 3 * thr, float(np.abs(corr_h).max()))
    assert int(me.min()) == 700 and int(mh.min()) == 700

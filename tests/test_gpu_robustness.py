"""Behaviour at the edges of the C ABI: coordinate bounds measured on the device (a wrong hint cannot void a
bound or a winner), unusable coordinates, non-positive thresholds, out-of-range sample indices, and models that do
not depend on which other hypotheses share a wave."""
import numpy as np
import pytest

from oracle import sfm_oracle as orc
from sfm_hip import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from sfm_hip import _capi, ops

    _capi.load_library()
    return torch, _capi, ops


def scene(n, inl=0.5, seed=7):
    sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=0.5, seed=seed)
    return sc, orc.normalize_K(sc["K"], sc["x1"]), orc.normalize_K(sc["K"], sc["x2"])


def test_understated_xmax_hint_still_gives_bounds_and_exact_winner(env, mixed_form):
    """The caller's xmax is a hint only: 1e-9 (absurdly small) and 0 (unknown) give the same bounds as the true
    maximum, because the library measures the maximum while it prepares the match-side operands."""
    torch, C, ops = env
    n, B = 3000, 20000
    sc, x1, x2 = scene(n)
    thr = 1.0 / sc["K"][0][0]
    corr_h = ops.pack_corr(x1, x2)
    corr = C.dev(corr_h)
    c32 = ops.corr_to_f32(corr)
    idx = C.dev(np.random.default_rng(1).integers(0, n, (B, 12)), np.int32)
    true_xmax = float(np.abs(corr_h).max())
    measured = float(c32[n, 0].item())
    assert true_xmax <= measured <= true_xmax * (1 + 1e-6)
    ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
    ce_h, ch_h = ce.cpu().numpy(), ch.cpu().numpy()
    want = ((int(ce_h.max()), int(np.argmax(ce_h))), (int(ch_h.max()), int(np.argmax(ch_h))))
    corrm = ops.corr_to_mfma(corr, thr)
    ref_m = [t.cpu().numpy() for t in ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, true_xmax)]
    ref_v = [t.cpu().numpy() for t in ops.eh_upper(corr, c32, idx, thr, 3 * thr, true_xmax, one_sided=True)]
    for hint in (1e-9, 0.0):
        um = [t.cpu().numpy() for t in ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, hint)]
        uv = [t.cpu().numpy() for t in ops.eh_upper(corr, c32, idx, thr, 3 * thr, hint, one_sided=True)]
        for a, b in zip(um + uv, ref_m + ref_v):
            np.testing.assert_array_equal(a, b)
        assert (um[0] >= ce_h).all() and (um[1] >= ch_h).all() and (uv[0] >= ce_h).all() and (uv[1] >= ch_h).all()
        for loose in (0, 1, 2):
            wE, wH, _ = ops.eh_best_native(corr, c32, idx, thr, 3 * thr, hint, loose=loose)
            assert (wE, wH) == want
        me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr, hint)
        np.testing.assert_array_equal(me.cpu().numpy(), ce_h)
        np.testing.assert_array_equal(mh.cpu().numpy(), ch_h)
    # an overstated hint is honoured (looser but valid bounds)
    ul = [t.cpu().numpy() for t in ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, 50.0)]
    assert (ul[0] >= ref_m[0]).all() and (ul[1] >= ref_m[1]).all()
    lib = C.load_library()
    out = torch.zeros(B, dtype=torch.int32, device="cuda")
    for bad in (-1.0, float("nan")):
        assert lib.sfm_eh_upper_mfma_f32(C.dptr(corr), C.dptr(corrm), n, C.dptr(idx), B, thr, 3 * thr, bad, 2.0, C.dptr(out), C.dptr(out),
                                         C.stream_ptr()) == 100001


def test_unusable_coordinates_fall_back_to_trivial_bounds_and_float64(env, mixed_form):
    """A NaN or huge coordinate leaves no usable bound: upper bounds become n, the exact mixed scorer evaluates every
    pair in float64 and still equals the float64 kernel, and the cascade still names the float64 winner."""
    torch, C, ops = env
    n, B = 700, 3000
    sc, x1, x2 = scene(n, seed=9)
    thr = 1.0 / sc["K"][0][0]
    idx_h = np.random.default_rng(2).integers(0, n, (B, 12))
    idx = C.dev(idx_h, np.int32)
    for poison in (np.nan, 3.0e7):
        corr_h = ops.pack_corr(x1, x2)
        corr_h[123, 2] = poison
        corr = C.dev(corr_h)
        c32 = ops.corr_to_f32(corr)
        corrm = ops.corr_to_mfma(corr, thr)
        ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
        ue, uh = ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, 0.0)
        assert bool((ue == n).all()) and bool((uh == n).all())
        ue, uh = ops.eh_upper(corr, c32, idx, thr, 3 * thr, 0.0, one_sided=False)
        assert bool((ue == n).all()) and bool((uh == n).all())
        me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr)
        assert torch.equal(me, ce) and torch.equal(mh, ch)
        ce_h, ch_h = ce.cpu().numpy(), ch.cpu().numpy()
        wE, wH, st = ops.eh_best_native(corr, c32, idx, thr, 3 * thr, 0.0)
        assert wE == (int(ce_h.max()), int(np.argmax(ce_h))) and wH == (int(ch_h.max()), int(np.argmax(ch_h)))


def test_matrix_core_bounds_across_coordinate_scales_and_thresholds(env):
    """The matrix-core bounds (epipolar test whole from the matrix cores, counted through the output clamp of a
    power-of-two scaled test value) stay upper bounds of the float64 counts in pixel coordinates, in tiny and in large
    coordinate scales and from a hundredth of a pixel to tens of pixels, and the cascade on top of them names the float64
    winner."""
    torch, C, ops = env
    n, B = 2000, 30000
    sc, x1, x2 = scene(n, inl=0.4, seed=21)
    f = sc["K"][0][0]
    idx = C.dev(np.random.default_rng(5).integers(0, n, (B, 12)), np.int32)
    cases = [("pixels", None, 1.0), ("pixels, 0.01 px", None, 0.01), ("pixels, 40 px", None, 40.0)]
    cases += [("scale %g" % sc_, sc_, 1.0) for sc_ in (1.0e-3, 2.0 ** -9, 2.0 ** -15, 30.0, 900.0)]
    for name, scale, px in cases:
        if scale is None:
            a, b, thr = np.array(sc["x1"], float), np.array(sc["x2"], float), px
        else:
            a, b, thr = x1.copy(), x2.copy(), px / f * scale
            a[:2] *= scale
            b[:2] *= scale
        corr_h = ops.pack_corr(a, b)
        corr = C.dev(corr_h)
        ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
        ue, uh = ops.eh_upper_mfma(corr, ops.corr_to_mfma(corr, thr), idx, thr, 3 * thr, 0.0)
        assert int((ue < ce).sum()) == 0 and int((uh < ch).sum()) == 0, name
        slack = float((ue - ce).double().mean())
        if scale is None or scale > 1.0e-4:
            assert slack < 0.25 * n, (name, slack)  # a real bound, not the trivial one
        wE, wH, _ = ops.eh_best_native(corr, ops.corr_to_f32(corr), idx, thr, 3 * thr, 0.0, loose=2)
        ce_h, ch_h = ce.cpu().numpy(), ch.cpu().numpy()
        assert wE == (int(ce_h.max()), int(np.argmax(ce_h))) and wH == (int(ch_h.max()), int(np.argmax(ch_h))), name


def test_exact_counts_at_extreme_thresholds(env, mixed_form):
    """Thresholds outside the range in which the early loop can produce its test outcome exactly (2 thr^2 at or above 0.55, or
    below 2^-45) and at the edges of that range: every pair of such a hypothesis goes to the float64 queue or the wave goes
    over to the full tests; the counts stay those of the float64 kernel."""
    torch, C, ops = env
    n, B = 1500, 6000
    sc, x1, x2 = scene(n, inl=0.5, seed=31)
    corr = C.dev(ops.pack_corr(x1, x2))
    idx = C.dev(np.random.default_rng(9).integers(0, n, (B, 12)), np.int32)
    for thr in (0.9, 0.53, 0.5, 4.0e-7, 3.0e-7, 1.0e-8, 1.0e-12):
        ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
        me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr)
        assert torch.equal(ce, me) and torch.equal(ch, mh), thr


def test_non_positive_thresholds_follow_the_reference(env):
    """thr_h <= 0 and the reprojection threshold <= 0 admit nothing (the reference compares a norm with them); a
    negative epipolar threshold behaves like its absolute value (the reference squares it)."""
    torch, C, ops = env
    n, B = 600, 1000
    sc, x1, x2 = scene(n, seed=5)
    thr = 1.0 / sc["K"][0][0]
    corr = C.dev(ops.pack_corr(x1, x2))
    idx = C.dev(np.random.default_rng(3).integers(0, n, (B, 12)), np.int32)
    ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
    for f in (ops.eh_score, ops.eh_score_mixed):
        e2, h2 = f(corr, idx, -thr, -3 * thr)
        assert torch.equal(e2, ce) and int(h2.max()) == 0
        e3, h3 = f(corr, idx, thr, 0.0)
        assert torch.equal(e3, ce) and int(h3.max()) == 0
    _, cnt = ops.transfer_mask(C.dev(np.eye(3).ravel()), C.dev(np.tile([0.1, 0.2, 0.1, 0.2], (50, 1))), -1.0)
    assert int(cnt.item()) == 0
    cam = synthetic.resection_scene(m=400, outlier_ratio=0.3, seed=8)
    pts = ops.pack_points(C.dev(cam["R"].ravel()), C.dev(cam["X"]), C.dev(cam["x_norm"]))
    pairs = C.dev(np.random.default_rng(4).integers(0, 400, (1, 500, 2)), np.int32)
    for mixed in (True, False):
        for t in (0.0, -1e-3):
            best, T, counts = ops.t_ransac(pts, [400], pairs, t, mixed=mixed)
            assert int(counts.max()) == 0 and int(best.cpu()[0]) >> 32 == 0
    # the drop-in function returns None, as the reference does when no hypothesis ever has an inlier
    from sfm_hip.estimate_translation import robust_estimate_T

    np.random.seed(0)
    assert robust_estimate_T(cam["x_norm"], cam["X"], cam["K"], cam["R"], 0.0) is None


def test_out_of_range_sample_indices_are_clamped_not_faulted(env):
    torch, C, ops = env
    n = 300
    sc, x1, x2 = scene(n, seed=6)
    thr = 1.0 / sc["K"][0][0]
    corr = C.dev(ops.pack_corr(x1, x2))
    rng = np.random.default_rng(5)
    bad = rng.integers(0, n, (512, 12))
    bad[::3, 2] = -5
    bad[1::3, 9] = n + 1000
    bad[2::3, 0] = 2**31 - 1
    good = np.clip(bad, 0, n - 1)
    for f in (ops.eh_score, ops.eh_score_mixed):
        a = f(corr, C.dev(bad, np.int32), thr, 3 * thr)
        b = f(corr, C.dev(good, np.int32), thr, 3 * thr)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    cam = synthetic.resection_scene(m=200, outlier_ratio=0.3, seed=8)
    pts = ops.pack_points(C.dev(cam["R"].ravel()), C.dev(cam["X"]), C.dev(cam["x_norm"]))
    pb = rng.integers(0, 200, (1, 300, 2))
    pb[0, ::2, 0] = -1
    pb[0, 1::2, 1] = 100000
    a = ops.t_ransac(pts, [200], C.dev(pb, np.int32), 4 * thr)
    b = ops.t_ransac(pts, [200], C.dev(np.clip(pb, 0, 199), np.int32), 4 * thr)
    assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0])


def test_models_do_not_depend_on_wave_neighbours(env):
    """The Jacobi sweeps of a lane stop with that lane's own convergence: a hypothesis solved among 63 copies of
    itself and among 63 other hypotheses gives bit-identical E and H (and therefore identical counts everywhere)."""
    torch, C, ops = env
    n = 900
    sc, x1, x2 = scene(n, seed=12)
    corr = C.dev(ops.pack_corr(x1, x2))
    rng = np.random.default_rng(6)
    rows = rng.integers(0, n, (256, 12))
    rows[5] = rows[4]  # include near-degenerate company: a repeated sample inside a row
    rows[7, 1] = rows[7, 0]
    mixed = ops.eh_models(corr, C.dev(rows, np.int32))
    Em, Hm = mixed[0].cpu().numpy(), mixed[1].cpu().numpy()
    for r in (0, 7, 63, 64, 130, 255):
        alone = ops.eh_models(corr, C.dev(np.tile(rows[r], (64, 1)), np.int32))
        Ea, Ha = alone[0].cpu().numpy(), alone[1].cpu().numpy()
        assert np.array_equal(Ea[0], Em[r], equal_nan=True) and np.array_equal(Ha[0], Hm[r], equal_nan=True)
        assert np.array_equal(Ea[0], Ea[63], equal_nan=True)


def test_large_but_finite_coordinates_keep_exact_counts(env, mixed_form):
    """Coordinates far outside the normalised range: pixel-sized values (float32 error bands grow with them), one match at
    3e5 (bounds still usable, the exact scorer already routes every pair to float64) and a translation problem whose
    points are so large that float32 squares would overflow.  Every mixed-precision count must equal the float64 count."""
    torch, C, ops = env
    n, B = 900, 4000
    sc, x1, x2 = scene(n, seed=11)
    idx = C.dev(np.random.default_rng(4).integers(0, n, (B, 12)), np.int32)
    for scale, poison, thr in ((1.0, 3.0e5, 1.0 / sc["K"][0][0]), (sc["K"][0][0], None, 1.0), (37.5, None, 37.5 / sc["K"][0][0])):
        corr_h = ops.pack_corr(x1, x2)
        corr_h *= scale
        if poison is not None:
            corr_h[55, 1] = poison
        corr = C.dev(corr_h)
        c32 = ops.corr_to_f32(corr)
        ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
        me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr)
        assert torch.equal(me, ce) and torch.equal(mh, ch)
        ue, uh = ops.eh_upper_mfma(corr, ops.corr_to_mfma(corr, thr), idx, thr, 3 * thr, 0.0)
        assert bool((ue >= ce).all()) and bool((uh >= ch).all())
        ce_h, ch_h = ce.cpu().numpy(), ch.cpu().numpy()
        wE, wH, _ = ops.eh_best_native(corr, c32, idx, thr, 3 * thr, 0.0)
        assert wE == (int(ce_h.max()), int(np.argmax(ce_h))) and wH == (int(ch_h.max()), int(np.argmax(ch_h)))
    # translation stage: one camera with ordinary points, one with a point at 1e20, one scaled by 1e9
    cams = [synthetic.resection_scene(m=500, outlier_ratio=0.4, seed=20 + i) for i in range(3)]
    cams[1]["X"] = cams[1]["X"].copy()
    cams[1]["X"][:, 7] = 1.0e20
    cams[2]["X"] = cams[2]["X"] * np.float32(1.0e9)
    pts = torch.cat([ops.pack_points(C.dev(c["R"].ravel()), C.dev(c["X"]), C.dev(c["x_norm"])) for c in cams])
    idx_t = C.dev(np.random.default_rng(5).integers(0, 500, (3, 3000, 2)), np.int32)
    thr_t = 4.0 / synthetic.K_DATASET5[0][0]
    b64, _, c64 = ops.t_ransac(pts, [500] * 3, idx_t, thr_t, mixed=False)
    bmx, _, cmx = ops.t_ransac(pts, [500] * 3, idx_t, thr_t, mixed=True)
    assert torch.equal(c64, cmx) and torch.equal(b64, bmx)

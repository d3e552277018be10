"""The NumPy oracle against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import sfm_oracle as orc
from sfm_hip import synthetic


def scene_from(arr):
    n, inl, pl, seed = arr
    sc = synthetic.two_view_scene(n=int(n), inlier_ratio=float(inl), planar_ratio=float(pl), seed=int(seed))
    x1 = orc.normalize_K(sc["K"], sc["x1"])
    x2 = orc.normalize_K(sc["K"], sc["x2"])
    return sc, x1, x2


def same_up_to_sign(a, b, tol):
    a = a / np.linalg.norm(a)
    b = b / np.linalg.norm(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < tol


def test_solvers_and_scoring(golden):
    g = golden("eh_scoring.npz")
    sc, x1, x2 = scene_from(g["scene"])
    idx = g["idx"]
    thr = float(g["thr"])
    cE, cH, E, H = orc.score_EH_batch(x1, x2, idx, thr)
    F = orc.estimate_F_DLT_batch(np.swapaxes(x1[:, idx[:, :8]], 0, 1), np.swapaxes(x2[:, idx[:, :8]], 0, 1))
    for i in range(idx.shape[0]):
        assert same_up_to_sign(F[i], g["F"][i], 1e-9)
        assert same_up_to_sign(E[i], g["E"][i], 1e-9)
        np.testing.assert_allclose(H[i], g["H"][i], rtol=1e-7, atol=1e-9)
    np.testing.assert_array_equal(cE, g["cE"])
    np.testing.assert_array_equal(cH, g["cH"])
    mE = orc.epipolar_mask(E, x1, x2, thr)
    mH = orc.get_inlier_mask_H(H, x1, x2, 3 * thr)
    np.testing.assert_array_equal(np.packbits(mE, axis=1), g["maskE"])
    np.testing.assert_array_equal(np.packbits(mH, axis=1), g["maskH"])
    assert g["cE"].max() > 300 and g["cH"].max() > 100  # fixtures span good and bad models


def test_decompositions(golden):
    g = golden("eh_scoring.npz")
    d = golden("decomp.npz")
    sc, x1, x2 = scene_from(g["scene"])
    thr = float(g["thr"])
    for k, i in enumerate(d["sel"]):
        E = g["E"][i]
        R, T, n = orc.essential_to_RT(E, x1, x2)
        assert n == d["n_e"][k]
        np.testing.assert_allclose(R, d["R_e"][k], atol=1e-9)
        np.testing.assert_allclose(T, d["T_e"][k], atol=1e-9)
        m = np.unpackbits(g["maskE"][i])[: x1.shape[1]].astype(bool)
        R, T, n = orc.essential_to_RT(E, x1[:, m], x2[:, m])
        assert n == d["n_m"][k]
        np.testing.assert_allclose(R, d["R_m"][k], atol=1e-9)
        np.testing.assert_allclose(T, d["T_m"][k], atol=1e-9)
        R1, t1, R2, t2 = orc.homography_to_RT(d["H"][k], x1, x2)
        np.testing.assert_allclose(R1, d["R1"][k], atol=1e-8)
        np.testing.assert_allclose(t1, d["t1"][k], atol=1e-8)
        np.testing.assert_allclose(R2, d["R2"][k], atol=1e-8)
        np.testing.assert_allclose(t2, d["t2"][k], atol=1e-8)
    keep = sc["is_inlier"]
    P2s = list(d["P2s"])
    assert orc.chirality_counts(P2s, x1[:, keep], x2[:, keep]) == list(d["chir_counts"])
    bestP, cnt = orc.perform_chirality_check(P2s, x1[:, keep], x2[:, keep])
    assert cnt == int(d["chir_count"])
    np.testing.assert_array_equal(bestP, d["chir_best"])


def test_triangulation(golden):
    g = golden("eh_scoring.npz")
    t = golden("triangulate.npz")
    sc, x1, x2 = scene_from(g["scene"])
    n = int(t["n"])
    X = orc.triangulate_3D_point_DLT([t["P1"], t["P2"]], [x1[:, :n], x2[:, :n]])
    np.testing.assert_allclose(X, t["X"], rtol=1e-9, atol=1e-9)
    Kn = np.asarray(sc["K"], float)
    Xp = orc.triangulate_3D_point_DLT([Kn @ t["P1"], Kn @ t["P2"]], [sc["x1"][:, :n], sc["x2"][:, :n]])
    np.testing.assert_allclose(Xp, t["Xpix"], rtol=1e-7, atol=1e-7)


def resection_from(arr, K=synthetic.K_DATASET5):
    m, out, seed = arr
    return synthetic.resection_scene(m=int(m), outlier_ratio=float(out), seed=int(seed), K=K)


def test_t_two_point(golden):
    g = golden("t2pt.npz")
    sc = resection_from(g["scene"])
    x, X, R = sc["x_norm"], sc["X"], sc["R"]
    Ts = orc.estimate_T_2point_batch(x, X, R, g["idx"])
    np.testing.assert_allclose(Ts, g["T"], rtol=1e-8, atol=1e-10)
    for i in (0, 5, 159):
        np.testing.assert_allclose(orc.estimate_T_2point(x[:, g["idx"][i]], X[:, g["idx"][i]], R), g["T"][i], rtol=1e-9, atol=1e-11)
    cnt = orc.count_inliers_batch(x, X, R, Ts, float(g["thr"]))
    np.testing.assert_array_equal(cnt, g["counts"])
    assert cnt.max() > 100


def test_t_ransac_full_run(golden):
    g = golden("t_ransac.npz")
    sc = resection_from(g["scene"])
    np.random.seed(int(g["seed"]))
    info = {}
    T = orc.robust_estimate_T(sc["x_norm"], sc["X"], sc["K"], sc["R"], 1.0, info=info)
    assert info["index"] == int(g["best_index"])
    assert info["count"] == int(g["best_count"])
    np.testing.assert_allclose(T, g["T"], rtol=1e-9, atol=1e-12)
    idx = info["idx"]
    distinct = idx[:, 0] != idx[:, 1]
    np.testing.assert_array_equal(info["counts"][distinct], g["counts"][distinct].astype(np.int64))
    # the global stream is left exactly where the reference leaves it
    np.testing.assert_array_equal(np.random.randint(0, 2**31 - 1, size=4), g["next_draws"])


def test_lm(golden):
    g = golden("lm.npz")
    for k in range(len(g["m"])):
        K = g["K"][k].tolist()
        sc = synthetic.resection_scene(m=int(g["m"][k]), outlier_ratio=0.0, seed=int(g["seed"][k]), K=K, noise_px=0.5)
        hist = []
        T = orc.levenberg_marquardt_algorithm(K, sc["R"], sc["X"], sc["x_norm"], g["T0"][k], 10, 0.01, history=hist)
        np.testing.assert_allclose(np.array(hist), g["hist"][k], rtol=1e-10, atol=1e-13)
        assert T.shape == (3, 1)
        assert np.abs(g["hist"][k][-1] - g["T0"][k]).max() > 0  # LM accepted at least one step


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_eh_ransac_seeded(golden, tag):
    g = golden("eh_ransac.npz")
    sc, _, _ = scene_from(g[f"{tag}_scene"])
    np.random.seed(int(g[f"{tag}_seed"]))
    tr = {}
    R, T, inl = orc.run_ransac(sc["K"], sc["x1"], sc["x2"], 1.0, trace=tr)
    assert tr["iterations"] == len(g[f"{tag}_cH"])
    idx = np.array(tr["idx"])
    okE = np.array([len(set(r[:8])) == 8 for r in idx])
    okH = np.array([len(set(r[8:])) == 4 for r in idx])
    np.testing.assert_array_equal(np.array(tr["cE"])[okE], g[f"{tag}_cE"][okE].astype(np.int64))
    np.testing.assert_array_equal(np.array(tr["cH"])[okH], g[f"{tag}_cH"][okH].astype(np.int64))
    np.testing.assert_allclose(R, g[f"{tag}_R"], atol=1e-9)
    np.testing.assert_allclose(T, g[f"{tag}_T"], atol=1e-9)
    np.testing.assert_array_equal(np.packbits(inl), g[f"{tag}_inliers"])
    np.testing.assert_array_equal(np.random.randint(0, 2**31 - 1, size=4), g[f"{tag}_next_draws"])


def test_reconstruction(golden):
    g = golden("reconstruction.npz")
    sc, _, _ = scene_from(g["scene"])
    d = {"init_pair_indices": [1, 3], "x_init": [sc["x1"], sc["x2"]], "init_pair_desc": np.zeros((128, 500), np.float32)}
    np.random.seed(int(g["seed"]))
    X0, abs_rot, inl = orc.run_reconstruction(list(g["rel"]), d, sc["K"], 1.0)
    np.testing.assert_allclose(np.array(abs_rot), g["abs_rot"], atol=1e-12)
    np.testing.assert_array_equal(np.packbits(inl), g["inliers"])
    np.testing.assert_allclose(X0, g["X0"], rtol=1e-7, atol=1e-8)


def test_whole_stage_sequence_against_the_reference_run(golden, check_e2e_against_reference):
    """e2e_reference.npz: the REFERENCE's own estimate_R -> run_reconstruction -> estimate_translation -> refine_translation
    with one seed on a synthetic five-camera scene (its OpenCV matcher replaced by the scene's stand-in): the oracle's
    run_geometry must give the same rotations, cloud, inlier mask, translations and stream position."""
    g = golden("e2e_reference.npz")
    n_cams, n_points, seed = (int(v) for v in g["scene"])
    sc = synthetic.multi_view_scene(n_cams=n_cams, n_points=n_points, init_pair=tuple(int(v) for v in g["init_pair"]), seed=seed)
    np.random.seed(int(g["seed"]))
    got = orc.run_geometry(sc["K"], sc["x_pairs"], sc["init_pair_dict"], sc["img_paths"], 1.0, sc["find_correspondences"])
    check_e2e_against_reference(g, got, np.random.randint(0, 2**31 - 1, size=4))


def test_dlt_over_more_than_a_minimal_sample(golden):
    """dlt_general.npz: the reference's estimate_F_DLT on 9, 20 and 150 matches and estimate_H_DLT on 5, 12 and 90 points
    (F up to its sign: the null vector's sign is LAPACK's choice)."""
    g = golden("dlt_general.npz")
    sc, x1, x2 = scene_from(g["scene"])
    for n in (9, 20, 150):
        cols = g[f"F_cols_{n}"]
        F = orc.estimate_F_DLT(x1[:, cols], x2[:, cols])
        np.testing.assert_allclose(np.sign(np.sum(F * g[f"F_{n}"])) * F, g[f"F_{n}"], atol=1e-10)
    for n in (5, 12, 90):
        H = orc.estimate_H_DLT(g["H_p1"][:, :n], g["H_p2"][:, :n])
        np.testing.assert_allclose(H, g[f"H_{n}"], rtol=1e-8, atol=1e-10)

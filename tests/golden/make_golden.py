"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Run only in the build container (needs /root/reference):

    python tests/golden/make_golden.py

The reference is imported unmodified; OpenCV is absent here, so an empty module
named ``cv2`` is registered first (the geometry functions never call into it).
Outputs are data only: inputs (or the seed/recipe that regenerates them with
``sfm_hip.synthetic``) and the reference's outputs.
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd"))
from sfm_hip import synthetic  # noqa: E402

sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, "/root/reference")
import src as ref_pkg  # noqa: E402,F401

ra = importlib.import_module("src.ransac_algorithm")
cc = importlib.import_module("src.chirality_check")
et = importlib.import_module("src.estimate_translation")
rt = importlib.import_module("src.refine_translation")
r3 = importlib.import_module("src.reconstruct_3D")
ut = importlib.import_module("src.utils")


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(name, os.path.getsize(path), "bytes")


def distinct_rows(rng, n, B, k):
    out = np.empty((B, k), np.int64)
    for i in range(B):
        out[i] = rng.choice(n, size=k, replace=False)
    return out


def gen_solvers_and_scoring():
    sc = synthetic.two_view_scene(n=1500, inlier_ratio=0.5, planar_ratio=0.5, seed=11)
    K = sc["K"]
    x1 = ut.normalize_K(K, sc["x1"])
    x2 = ut.normalize_K(K, sc["x2"])
    rng = np.random.default_rng(5)
    B = 64
    idx = np.concatenate([distinct_rows(rng, 1500, B, 8), distinct_rows(rng, 1500, B, 4)], axis=1)
    # make a third of the samples all-inlier so that counts span a wide range
    inl = np.flatnonzero(sc["is_inlier"])
    pl = np.flatnonzero(sc["is_inlier"] & (np.abs(sc["X"][2] - (6 + 0.2 * sc["X"][0])) < 1e-12))
    for i in range(0, B, 3):
        idx[i, :8] = rng.choice(inl, 8, replace=False)
        idx[i, 8:] = rng.choice(pl, 4, replace=False)
    thr = 1.0 / K[0][0]
    F = np.zeros((B, 3, 3))
    E = np.zeros((B, 3, 3))
    H = np.zeros((B, 3, 3))
    mE = np.zeros((B, 1500), bool)
    mH = np.zeros((B, 1500), bool)
    dE = np.zeros((B, 1500))
    for i in range(B):
        F[i] = ra.estimate_F_DLT(x1[:, idx[i, :8]], x2[:, idx[i, :8]])
        E[i] = ra.enforce_essential(F[i])
        d = (ra.epipolar_errors(E[i], x1, x2) ** 2 + ra.epipolar_errors(E[i].T, x2, x1) ** 2) / 2
        dE[i] = d
        mE[i] = d < thr**2
        H[i] = ra.estimate_H_DLT(x1[:, idx[i, 8:]], x2[:, idx[i, 8:]])
        mH[i] = ra.get_inlier_mask_H(H[i], x1, x2, 3 * thr)
    save(
        "eh_scoring.npz",
        scene=np.array([1500, 0.5, 0.5, 11]),
        idx=idx, F=F, E=E, H=H,
        maskE=np.packbits(mE, axis=1), maskH=np.packbits(mH, axis=1),
        cE=mE.sum(1), cH=mH.sum(1), thr=thr,
    )
    return sc, x1, x2, idx, E, H, mE, thr


def gen_decomp(sc, x1, x2, idx, E, H, mE, thr):
    sel = [0, 3, 6, 9, 1, 2]
    out = {}
    R_e, T_e, n_e = [], [], []
    for i in sel:
        R, T, n = ra.essential_to_RT(E[i], x1, x2)
        R_e.append(R); T_e.append(T); n_e.append(n)
    # masked subset call, as at H record events
    R_m, T_m, n_m = [], [], []
    for i in sel:
        R, T, n = ra.essential_to_RT(E[i], x1[:, mE[i]], x2[:, mE[i]])
        R_m.append(R); T_m.append(T); n_m.append(n)
    Hs, R1, t1, R2, t2 = [], [], [], [], []
    for i in sel:
        Hc = H[i].copy()
        a, b, c, d = ra.homography_to_RT(Hc, x1, x2)
        Hs.append(H[i]); R1.append(a); t1.append(b); R2.append(c); t2.append(d)
    # explicit chirality call with hand-made candidates
    P2s = [np.hstack((sc["R"], s * sc["T"].reshape(3, 1))) for s in (1, -1)]
    P2s += [np.hstack((sc["R"].T, s * sc["T"].reshape(3, 1))) for s in (1, -1)]
    keep = sc["is_inlier"]
    bestP, cnt = cc.perform_chirality_check(P2s, x1[:, keep], x2[:, keep])
    counts = []
    P1 = np.concatenate((np.eye(3), np.zeros((3, 1))), axis=1)
    for P2 in P2s:
        Xt = ut.triangulate_3D_point_DLT([P1, P2], [x1[:, keep], x2[:, keep]])
        counts.append(cc.count_points_in_front_of_cam(P1, P2, Xt))
    save(
        "decomp.npz", sel=np.array(sel),
        R_e=np.array(R_e), T_e=np.array(T_e), n_e=np.array(n_e),
        R_m=np.array(R_m), T_m=np.array(T_m), n_m=np.array(n_m),
        H=np.array(Hs), R1=np.array(R1), t1=np.array(t1), R2=np.array(R2), t2=np.array(t2),
        P2s=np.array(P2s), chir_best=bestP, chir_count=cnt, chir_counts=np.array(counts),
    )


def gen_triangulate(sc, x1, x2):
    P1 = np.concatenate((np.eye(3), np.zeros((3, 1))), axis=1)
    P2 = np.hstack((sc["R"], sc["T"].reshape(3, 1)))
    n = 1000
    X = ut.triangulate_3D_point_DLT([P1, P2], [x1[:, :n], x2[:, :n]])
    # pixel-space call as the visualiser does (K @ [R|T], pixel coordinates)
    Kn = np.asarray(sc["K"], float)
    Xp = ut.triangulate_3D_point_DLT([Kn @ P1, Kn @ P2], [sc["x1"][:, :n], sc["x2"][:, :n]])
    save("triangulate.npz", n=n, P1=P1, P2=P2, X=X, Xpix=Xp)


def gen_t2pt():
    sc = synthetic.resection_scene(m=400, outlier_ratio=0.4, seed=21)
    x, X, K, R = sc["x_norm"], sc["X"], sc["K"], sc["R"]
    rng = np.random.default_rng(9)
    B = 160
    idx = distinct_rows(rng, 400, B, 2)
    idx[-8:, 1] = idx[-8:, 0]  # duplicated index: rank-2 system, min-norm answer
    good = np.flatnonzero(~sc["is_outlier"])
    for i in range(0, B - 8, 2):
        idx[i] = rng.choice(good, 2, replace=False)
    thr = 4 * 1.0 / K[0][0]
    Ts = np.zeros((B, 3))
    cnt = np.zeros(B, np.int64)
    for i in range(B):
        Ts[i] = et.estimate_T_2point(x[:, idx[i]], X[:, idx[i]], R)
        cnt[i] = et.count_inliers(x, X, R, Ts[i], thr)
    save("t2pt.npz", scene=np.array([400, 0.4, 21]), idx=idx, T=Ts, counts=cnt, thr=thr)


def gen_t_ransac():
    sc = synthetic.resection_scene(m=300, outlier_ratio=0.5, seed=22)
    x, X, K, R = sc["x_norm"], sc["X"], sc["K"], sc["R"]
    counts = []
    orig = et.count_inliers

    def rec(*a):
        c = orig(*a)
        counts.append(c)
        return c

    et.count_inliers = rec
    try:
        np.random.seed(1234)
        T = et.robust_estimate_T(x, X, K, R, 1.0)
    finally:
        et.count_inliers = orig
    counts = np.array(counts)
    state_after = np.random.randint(0, 2**31 - 1, size=4)
    save("t_ransac.npz", scene=np.array([300, 0.5, 22]), seed=1234, T=T,
         counts=counts.astype(np.uint16), best_index=int(np.argmax(counts)),
         best_count=int(counts.max()), next_draws=state_after)


def gen_t_ransac_more():
    """Three more seeded 100 000-iteration robust_estimate_T runs of the reference: 2 000 correspondences with 60 % outliers,
    5 000 with 20 %, 40 with 30 % (all counts, the winner, the next draws of the global stream)."""
    out = {}
    runs = []
    for j, (m, outl, sseed, rseed) in enumerate([(2000, 0.6, 23, 4321), (5000, 0.2, 24, 99), (40, 0.3, 25, 7)]):
        sc = synthetic.resection_scene(m=m, outlier_ratio=outl, seed=sseed)
        counts = []
        orig = et.count_inliers

        def rec(*a):
            c = orig(*a)
            counts.append(c)
            return c

        et.count_inliers = rec
        try:
            np.random.seed(rseed)
            T = et.robust_estimate_T(sc["x_norm"], sc["X"], sc["K"], sc["R"], 1.0)
        finally:
            et.count_inliers = orig
        counts = np.array(counts)
        print("t_ransac_more", j, "m", m, "best", int(counts.max()), "at", int(np.argmax(counts)), flush=True)
        runs.append([m, outl, sseed, rseed])
        out.update({f"r{j}_T": T, f"r{j}_counts": counts.astype(np.uint16), f"r{j}_next_draws": np.random.randint(0, 2**31 - 1, size=4)})
    out["runs"] = np.array(runs, np.float64)  # correspondences, outlier share, scene seed, run seed
    save("t_ransac_more.npz", **out)


def gen_e2e():
    """The reference's own stage sequence (run_sfm.py:59-80: estimate_R, run_reconstruction, estimate_translation,
    refine_translation) with one seed on a synthetic five-camera scene; OpenCV's matcher inside find_correspondences is
    replaced by the scene's stand-in (bound into both modules that imported it with `from .utils import *`)."""
    gen_e2e_scene("e2e_reference.npz", dict(n_cams=5, n_points=900, seed=4), 2024)
    gen_e2e_scene("e2e_reference_b.npz", dict(n_cams=7, n_points=1500, init_pair=(2, 4), seed=12), 77)


def gen_e2e_scene(name, scene, run_seed):
    sc = synthetic.multi_view_scene(**scene)
    o_et, o_rt = et.find_correspondences, rt.find_correspondences
    et.find_correspondences = rt.find_correspondences = sc["find_correspondences"]
    try:
        np.random.seed(run_seed)
        R_list = ra.estimate_R(sc["K"], sc["x_pairs"], 1.0)
        X0, abs_rot, inl = r3.run_reconstruction(R_list, sc["init_pair_dict"], sc["K"], 1.0)
        desc_X = sc["init_pair_dict"]["init_pair_desc"]
        est = et.estimate_translation(sc["K"], desc_X, X0, abs_rot, sc["img_paths"], 1.0)
        ref = rt.refine_translation(sc["K"], sc["img_paths"], desc_X[:, inl], X0, abs_rot, est)
        nd = np.random.randint(0, 2**31 - 1, size=4)
    finally:
        et.find_correspondences, rt.find_correspondences = o_et, o_rt
    assert all(t is not None for t in est)
    ip = scene.get("init_pair", (1, 3))  # the default of synthetic.multi_view_scene
    save(name, scene=np.array([scene["n_cams"], scene["n_points"], scene["seed"]]), init_pair=np.array(ip), seed=run_seed, R_rel=np.array(R_list), abs_rot=np.array(abs_rot), X0=X0,
         inliers=np.packbits(inl), n_inliers_bits=np.array(inl.shape[0]), estimated_Ts=np.array([np.reshape(t, 3) for t in est]),
         refined_Ts=np.array([np.reshape(t, 3) for t in ref]), next_draws=nd)


def gen_dlt_general():
    """estimate_F_DLT / estimate_H_DLT of the reference on MORE than a minimal sample (they loop over every column they are
    given): inlier subsets of 9, 20 and 150 matches of the scoring scene for F, 5, 12 and 90 exactly transferred points for H."""
    sc = synthetic.two_view_scene(n=1500, inlier_ratio=0.5, planar_ratio=0.5, seed=11)
    x1 = ut.normalize_K(sc["K"], sc["x1"])
    x2 = ut.normalize_K(sc["K"], sc["x2"])
    inl = np.flatnonzero(sc["is_inlier"])
    rng = np.random.RandomState(8)
    out = {"scene": np.array([1500, 0.5, 0.5, 11])}
    for n in (9, 20, 150):
        cols = inl[rng.permutation(len(inl))[:n]]
        out[f"F_cols_{n}"] = cols
        out[f"F_{n}"] = ra.estimate_F_DLT(x1[:, cols], x2[:, cols])
    Hs = np.array([[1.02, 0.03, 0.01], [-0.02, 0.97, 0.04], [0.05, -0.03, 1.0]])
    p1 = np.vstack([rng.uniform(-0.5, 0.5, (2, 90)), np.ones((1, 90))])
    p2 = Hs @ p1
    p2 = p2 / p2[2] + np.vstack([rng.normal(0, 1e-3, (2, 90)), np.zeros((1, 90))])
    out["H_p1"], out["H_p2"] = p1, p2
    for n in (5, 12, 90):
        out[f"H_{n}"] = ra.estimate_H_DLT(p1[:, :n], p2[:, :n])
    save("dlt_general.npz", **out)


def gen_lm():
    cams = []
    for k, (Kc, seed) in enumerate([(synthetic.K_DATASET5, 31), (synthetic.K_DATASET10, 32),
                                    (synthetic.K_DATASET10, 33), (synthetic.K_DATASET5, 34)]):
        m = [500, 120, 800, 64][k]
        sc = synthetic.resection_scene(m=m, outlier_ratio=0.0, seed=seed, K=Kc, noise_px=0.5)
        T0 = sc["T_true"] + np.array([0.03, -0.02, 0.01]) * (k + 1)
        hist = []
        for it in range(1, 11):
            hist.append(rt.levenberg_marquardt_algorithm(Kc, sc["R"], sc["X"], sc["x_norm"], T0, it, 0.01).ravel())
        cams.append(dict(K=np.asarray(Kc, float), m=m, seed=seed, T0=T0, hist=np.array(hist)))
    save("lm.npz", K=np.array([c["K"] for c in cams]), m=np.array([c["m"] for c in cams]),
         seed=np.array([c["seed"] for c in cams]), T0=np.array([c["T0"] for c in cams]),
         hist=np.array([c["hist"] for c in cams]))


def traced_run_ransac(sc, seed, thr_px=1.0):
    log = []
    o_epi, o_h, o_f = ra.epipolar_errors, ra.get_inlier_mask_H, ra.estimate_F_DLT

    def w_f(*a):
        log.append(("F", None))
        return o_f(*a)

    def w_epi(*a):
        d = o_epi(*a)
        log.append(("epi", d.copy()))
        return d

    def w_h(*a):
        m = o_h(*a)
        log.append(("H", int(m.sum())))
        return m

    ra.estimate_F_DLT, ra.epipolar_errors, ra.get_inlier_mask_H = w_f, w_epi, w_h
    try:
        np.random.seed(seed)
        R, T, inl = ra.run_ransac(sc["K"], sc["x1"], sc["x2"], thr_px)
    finally:
        ra.estimate_F_DLT, ra.epipolar_errors, ra.get_inlier_mask_H = o_f, o_epi, o_h
    thr = thr_px / sc["K"][0][0]
    cE, cH = [], []
    i = 0
    while i < len(log):
        if log[i][0] == "F":
            d = (log[i + 1][1] ** 2 + log[i + 2][1] ** 2) / 2
            cE.append(int((d < thr**2).sum()))
            i += 3
        elif log[i][0] == "H":
            cH.append(log[i][1])
            i += 1
        else:
            i += 1
    next_draws = np.random.randint(0, 2**31 - 1, size=4)
    return R, T, inl, np.array(cE), np.array(cH), next_draws


def gen_eh_ransac():
    out = {}
    for tag, (n, inl, pl, sseed, rseed) in {
        "a": (600, 0.7, 0.6, 41, 7),
        "b": (1000, 0.6, 0.5, 42, 8),
        "c": (400, 0.8, 0.7, 43, 9),
    }.items():
        sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=pl, seed=sseed)
        R, T, m, cE, cH, nd = traced_run_ransac(sc, rseed)
        print("eh_ransac", tag, "iterations", len(cH), "inliers", int(m.sum()))
        out.update({f"{tag}_scene": np.array([n, inl, pl, sseed]), f"{tag}_seed": rseed,
                    f"{tag}_R": R, f"{tag}_T": T, f"{tag}_inliers": np.packbits(m),
                    f"{tag}_cE": cE.astype(np.uint16), f"{tag}_cH": cH.astype(np.uint16),
                    f"{tag}_next_draws": nd})
    save("eh_ransac.npz", **out)


def gen_eh_ransac_many():
    """Thirty seeded run_ransac runs of the reference at 300, 500 and 1500 matches (ten seeds each, scenes of varying inlier
    and planar share) with their full count traces: the sample stream follows from the seed, so a consumer can tell which
    iterations drew a repeated index (a rank-deficient system whose null vector LAPACK picks arbitrarily) and which of
    them were record events."""
    out = {}
    runs = []
    for n in (300, 500, 1500):
        for j in range(10):
            sseed, rseed = 1000 + 37 * j + n, 500 + 11 * j + n
            inl, pl = 0.55 + 0.04 * (j % 5), 0.45 + 0.05 * (j % 4)
            sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=pl, seed=sseed)
            R, T, m, cE, cH, nd = traced_run_ransac(sc, rseed)
            tag = f"r{len(runs):02d}"
            print("eh_ransac_many", tag, "n", n, "iterations", len(cH), "inliers", int(m.sum()), flush=True)
            runs.append([n, inl, pl, sseed, rseed, len(cH)])
            out.update({f"{tag}_R": R, f"{tag}_T": T, f"{tag}_inliers": np.packbits(m), f"{tag}_cE": cE.astype(np.uint16),
                        f"{tag}_cH": cH.astype(np.uint16), f"{tag}_next_draws": nd})
    out["runs"] = np.array(runs, np.float64)  # n, inlier share, planar share, scene seed, run seed, iterations
    save("eh_ransac_many.npz", **out)


def traced_run_ransac_light(sc, seed, thr_px=1.0):
    """traced_run_ransac for long runs on many matches: the two distance arrays of an iteration are reduced to the count as soon
    as the second arrives (the full trace of a 100 000-iteration run on 8 000 matches would hold gigabytes)."""
    cE, cH, pending = [], [], []
    state = {"expect": 0}
    o_epi, o_h, o_f = ra.epipolar_errors, ra.get_inlier_mask_H, ra.estimate_F_DLT
    thr = thr_px / sc["K"][0][0]

    def w_f(*a):
        state["expect"] = 2  # the loop body calls epipolar_errors twice right after estimate_F_DLT (src/ransac_algorithm.py:94-102)
        pending.clear()
        return o_f(*a)

    def w_epi(*a):
        d = o_epi(*a)
        if state["expect"] > 0:
            pending.append(d)
            state["expect"] -= 1
            if state["expect"] == 0:
                cE.append(int(((pending[0] ** 2 + pending[1] ** 2) / 2 < thr**2).sum()))
                pending.clear()
        return d

    def w_h(*a):
        m = o_h(*a)
        cH.append(int(m.sum()))
        return m

    ra.estimate_F_DLT, ra.epipolar_errors, ra.get_inlier_mask_H = w_f, w_epi, w_h
    try:
        np.random.seed(seed)
        R, T, inl = ra.run_ransac(sc["K"], sc["x1"], sc["x2"], thr_px)
    finally:
        ra.estimate_F_DLT, ra.epipolar_errors, ra.get_inlier_mask_H = o_f, o_epi, o_h
    next_draws = np.random.randint(0, 2**31 - 1, size=4)
    return R, T, inl, np.array(cE), np.array(cH), next_draws


def gen_eh_ransac_large():
    """Six seeded run_ransac runs of the reference on 4 000 and 8 000 matches at 30-50 % inliers: loops of thousands to tens of
    thousands of iterations, i.e. several doubling blocks of the drop-in loop and its draw-ahead (same layout as eh_ransac_many)."""
    out = {}
    runs = []
    for n, inl, pl, j in [(4000, 0.5, 0.5, 0), (4000, 0.4, 0.3, 1), (4000, 0.3, 0.6, 2), (8000, 0.5, 0.4, 3), (8000, 0.4, 0.6, 4), (8000, 0.33, 0.2, 5)]:
        sseed, rseed = 7000 + 41 * j + n, 900 + 13 * j + n
        sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=pl, seed=sseed)
        R, T, m, cE, cH, nd = traced_run_ransac_light(sc, rseed)
        tag = f"r{len(runs):02d}"
        print("eh_ransac_large", tag, "n", n, "iterations", len(cH), "inliers", int(m.sum()), flush=True)
        assert len(cE) == len(cH)
        runs.append([n, inl, pl, sseed, rseed, len(cH)])
        out.update({f"{tag}_R": R, f"{tag}_T": T, f"{tag}_inliers": np.packbits(m), f"{tag}_cE": cE.astype(np.uint16),
                    f"{tag}_cH": cH.astype(np.uint16), f"{tag}_next_draws": nd})
    out["runs"] = np.array(runs, np.float64)
    save("eh_ransac_large.npz", **out)


def gen_eh_ransac_thresholds():
    """Six seeded run_ransac runs of the reference at pixel thresholds of 0.25, 2.5 and 8 (1 200 matches, two seeds each);
    layout of eh_ransac_many with the threshold as a seventh column of `runs`."""
    out = {}
    runs = []
    for thr_px in (0.25, 2.5, 8.0):
        for j in range(2):
            n, inl, pl = 1200, 0.5 + 0.1 * j, 0.4 + 0.2 * j
            sseed, rseed = 3000 + 17 * j + int(100 * thr_px), 60 + 7 * j + int(10 * thr_px)
            sc = synthetic.two_view_scene(n=n, inlier_ratio=inl, planar_ratio=pl, seed=sseed)
            R, T, m, cE, cH, nd = traced_run_ransac_light(sc, rseed, thr_px)
            tag = f"r{len(runs):02d}"
            print("eh_ransac_thresholds", tag, "thr", thr_px, "iterations", len(cH), "inliers", int(m.sum()), flush=True)
            runs.append([n, inl, pl, sseed, rseed, len(cH), thr_px])
            out.update({f"{tag}_R": R, f"{tag}_T": T, f"{tag}_inliers": np.packbits(m), f"{tag}_cE": cE.astype(np.uint16),
                        f"{tag}_cH": cH.astype(np.uint16), f"{tag}_next_draws": nd})
    out["runs"] = np.array(runs, np.float64)
    save("eh_ransac_thresholds.npz", **out)


def gen_reconstruction():
    sc = synthetic.two_view_scene(n=500, inlier_ratio=0.75, planar_ratio=0.6, seed=51)
    rel = [synthetic.rot_y(0.05), synthetic.rot_x(0.02) @ synthetic.rot_y(0.04), synthetic.rot_y(-0.03)]
    d = {"init_pair_indices": [1, 3], "x_init": [sc["x1"], sc["x2"]],
         "init_pair_desc": np.zeros((128, 500), np.float32)}
    np.random.seed(77)
    X0, abs_rot, inl = r3.run_reconstruction(rel, d, sc["K"], 1.0)
    save("reconstruction.npz", scene=np.array([500, 0.75, 0.6, 51]), seed=77, rel=np.array(rel),
         X0=X0, abs_rot=np.array(abs_rot), inliers=np.packbits(inl))


if __name__ == "__main__":
    if "--only-many" in sys.argv:
        gen_eh_ransac_many()
        sys.exit(0)
    if "--only-thresholds" in sys.argv:
        gen_eh_ransac_thresholds()
        sys.exit(0)
    if "--only-dlt" in sys.argv:
        gen_dlt_general()
        sys.exit(0)
    if "--only-e2e" in sys.argv:
        gen_e2e()
        sys.exit(0)
    if "--only-t-more" in sys.argv:
        gen_t_ransac_more()
        sys.exit(0)
    if "--only-large" in sys.argv:
        gen_eh_ransac_large()
        sys.exit(0)
    pack = gen_solvers_and_scoring()
    gen_decomp(*pack)
    gen_triangulate(pack[0], pack[1], pack[2])
    gen_t2pt()
    gen_lm()
    gen_reconstruction()
    gen_eh_ransac()
    gen_eh_ransac_many()
    gen_eh_ransac_large()
    gen_eh_ransac_thresholds()
    gen_t_ransac()
    gen_t_ransac_more()
    gen_e2e()
    gen_dlt_general()

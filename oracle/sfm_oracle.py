"""CPU oracle for the SfM robust-estimation hot path (TEST INFRASTRUCTURE ONLY).

This module is a float64 NumPy restatement of the reference pipeline's
geometry functions.  It exists so that the HIP kernels can be checked on a
machine that does not have the reference checked out.  Nothing in the shipped
package imports it: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
itself (``/root/reference``, with a stub ``cv2`` module because the hot path
never touches OpenCV) and stores its outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against them.

Every function cites the reference file:line it restates.  The restatement is
vectorised over hypotheses (``*_batch``) where the reference loops, but keeps
the reference's formulas, operation order inside a formula, thresholds,
tie-breaking and quirks.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

# --------------------------------------------------------------------------
# small helpers (reference: src/utils.py:39-62)
# --------------------------------------------------------------------------


def pflat(X):
    """src/utils.py:39-41"""
    return X / X[-1, :]


def normalize_K(K, xs):
    """src/utils.py:44-45  inv(K) @ xs in float64."""
    return np.linalg.inv(K) @ xs


def cartesian_to_homogeneous(p):
    """src/utils.py:48-53"""
    return np.vstack((p, np.ones((1, p.shape[1]))))


def homogeneous_to_cartesian(p):
    """src/utils.py:56-57"""
    return p[:-1, :] / p[-1, :]


def skew_symmetric_mat(v):
    """src/utils.py:60-62"""
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def filter_3D_points(X):
    """src/utils.py:9-14 (row-wise filter; see the reference for its quirk on 4xN input)."""
    Xm = np.mean(X, axis=0)
    d = np.linalg.norm(X - Xm, axis=1)
    q = np.quantile(d, 0.9)
    return X[d <= 5 * q]


# --------------------------------------------------------------------------
# minimal solvers
# --------------------------------------------------------------------------


def dlt_rows_F(x1s, x2s):
    """Rows of the 8-point system, src/ransac_algorithm.py:240-253.

    x1s, x2s: (..., 3, n).  Returns (..., n, 9) with row
    [x1x2, x1y2, x1z2, y1x2, y1y2, y1z2, z1x2, z1y2, z1z2].
    """
    a = np.swapaxes(x1s, -1, -2)[..., :, :, None]  # (..., n, 3, 1)
    b = np.swapaxes(x2s, -1, -2)[..., :, None, :]  # (..., n, 1, 3)
    return (a * b).reshape(a.shape[:-2] + (9,))


def estimate_F_DLT_batch(x1s, x2s):
    """src/ransac_algorithm.py:216-263 for a batch: x1s, x2s (B,3,8) -> F (B,3,3).

    F = Vh[-1].reshape(3,3).T so that x2^T F x1 = 0.
    """
    A = dlt_rows_F(x1s, x2s)
    _, _, Vh = np.linalg.svd(A)
    v = Vh[..., -1, :]
    return np.swapaxes(v.reshape(v.shape[:-1] + (3, 3)), -1, -2)


def estimate_F_DLT(x1s, x2s):
    return estimate_F_DLT_batch(x1s[None], x2s[None])[0]


def enforce_essential_batch(E):
    """src/ransac_algorithm.py:176-190  U diag(1,1,0) Vt."""
    U, _, Vt = np.linalg.svd(E)
    return U[..., :, :2] @ Vt[..., :2, :]


def enforce_essential(E):
    return enforce_essential_batch(E[None])[0]


def dlt_rows_H(x1s, x2s):
    """src/ransac_algorithm.py:279-284: two rows per correspondence, (...,2n,9)."""
    x1 = x1s[..., 0, :]
    y1 = x1s[..., 1, :]
    x2 = x2s[..., 0, :]
    y2 = x2s[..., 1, :]
    z = np.zeros_like(x1)
    o = np.ones_like(x1)
    r0 = np.stack([-x1, -y1, -o, z, z, z, x2 * x1, x2 * y1, x2], axis=-1)
    r1 = np.stack([z, z, z, -x1, -y1, -o, y2 * x1, y2 * y1, y2], axis=-1)
    A = np.stack([r0, r1], axis=-2)  # (..., n, 2, 9)
    return A.reshape(A.shape[:-3] + (A.shape[-3] * 2, 9))


def estimate_H_DLT_batch(x1s, x2s):
    """src/ransac_algorithm.py:266-293 for a batch (B,3,4) -> (B,3,3), H/H[2,2]."""
    A = dlt_rows_H(x1s, x2s)
    _, _, Vh = np.linalg.svd(A)
    v = Vh[..., -1, :]
    H = v.reshape(v.shape[:-1] + (3, 3))
    with np.errstate(all="ignore"):
        return H / H[..., 2:3, 2:3]


def estimate_H_DLT(x1s, x2s):
    return estimate_H_DLT_batch(x1s[None], x2s[None])[0]


# --------------------------------------------------------------------------
# scoring
# --------------------------------------------------------------------------


def epipolar_errors(F, x1, x2):
    """src/ransac_algorithm.py:193-213. F (3,3) or (B,3,3); x1,x2 (3,N)."""
    with np.errstate(all="ignore"):
        l = F @ x1
        l = l / np.sqrt(l[..., 0:1, :] ** 2 + l[..., 1:2, :] ** 2)
        return np.abs(np.sum(l * x2, axis=-2))


def epipolar_mask(E, x1, x2, err_threshold):
    """Symmetric epipolar inlier test, src/ransac_algorithm.py:98-101."""
    Et = np.swapaxes(E, -1, -2)
    with np.errstate(all="ignore"):
        return (
            epipolar_errors(E, x1, x2) ** 2 + epipolar_errors(Et, x2, x1) ** 2
        ) / 2 < err_threshold**2


def get_inlier_mask_H(H, x1s, x2s, threshold):
    """src/ransac_algorithm.py:296-329 (one-way transfer error)."""
    with np.errstate(all="ignore"):
        t = H @ x1s
        t = t / t[..., 2:3, :]
        d = np.sqrt(np.sum((x2s[:2, :] - t[..., :2, :]) ** 2, axis=-2))
        return d < threshold


def score_EH_batch(x1, x2, idx, err_threshold, chunk=256):
    """Counts for a block of hypotheses.

    idx: (B,12) int, columns 0-7 are the 8-point sample, 8-11 the 4-point sample
    (src/ransac_algorithm.py:94,116).  Returns (cE, cH, E, H).
    """
    B = idx.shape[0]
    cE = np.zeros(B, np.int64)
    cH = np.zeros(B, np.int64)
    Es = np.zeros((B, 3, 3))
    Hs = np.zeros((B, 3, 3))
    for s in range(0, B, chunk):
        ii = idx[s : s + chunk]
        iE = ii[:, :8]
        iH = ii[:, 8:12]
        E = enforce_essential_batch(
            estimate_F_DLT_batch(
                np.swapaxes(x1[:, iE], 0, 1), np.swapaxes(x2[:, iE], 0, 1)
            )
        )
        H = estimate_H_DLT_batch(
            np.swapaxes(x1[:, iH], 0, 1), np.swapaxes(x2[:, iH], 0, 1)
        )
        cE[s : s + chunk] = epipolar_mask(E, x1, x2, err_threshold).sum(-1)
        cH[s : s + chunk] = get_inlier_mask_H(H, x1, x2, 3 * err_threshold).sum(-1)
        Es[s : s + chunk] = E
        Hs[s : s + chunk] = H
    return cE, cH, Es, Hs


# --------------------------------------------------------------------------
# triangulation and chirality
# --------------------------------------------------------------------------


def triangulate_3D_point_DLT(P, points):
    """src/utils.py:17-36: per point smallest right singular vector of the 4x4 DLT matrix."""
    P1, P2 = P
    p1, p2 = points
    P1 = np.asarray(P1, float)
    P2 = np.asarray(P2, float)
    n = p1.shape[1]
    if n == 0:
        return np.zeros((0,)).reshape(0)  # np.array([]).T in the reference
    A = np.empty((n, 4, 4))
    A[:, 0] = p1[0][:, None] * P1[2] - P1[0]
    A[:, 1] = p1[1][:, None] * P1[2] - P1[1]
    A[:, 2] = p2[0][:, None] * P2[2] - P2[0]
    A[:, 3] = p2[1][:, None] * P2[2] - P2[1]
    _, _, Vt = np.linalg.svd(A)
    X = Vt[:, -1, :]
    with np.errstate(all="ignore"):
        X = X / X[:, 3:4]
    return X.T


def chirality_counts(P2s, x1, x2):
    """Counts of src/chirality_check.py:39-60 for each candidate."""
    P1 = np.concatenate((np.eye(3), np.zeros((3, 1))), axis=1)
    counts = []
    for P2 in P2s:
        if x1.shape[1] == 0:
            counts.append(0)
            continue
        X = triangulate_3D_point_DLT([P1, P2], [x1, x2])
        with np.errstate(all="ignore"):
            z2 = (P2 @ X)[2]
            z1 = (P1 @ X)[2]
            counts.append(int(np.sum((z2 > 0) & (z1 > 0))))
    return counts


def perform_chirality_check(P2s, x1, x2):
    """src/chirality_check.py:5-36, :63-91: first strict maximum wins (start at -1)."""
    counts = chirality_counts(P2s, x1, x2)
    best, bi = -1, -1
    for i, c in enumerate(counts):
        if c > best:
            best, bi = c, i
    return P2s[bi], best


def essential_to_RT(E, x1, x2):
    """src/ransac_algorithm.py:406-446."""
    W = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    U, _, Vt = scipy.linalg.svd(E)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        Vt = -Vt
    u3 = U[:, -1].reshape(-1, 1)
    Ra = U @ W @ Vt
    Rb = U @ W.T @ Vt
    P2s = [np.hstack((Ra, u3)), np.hstack((Ra, -u3)), np.hstack((Rb, u3)), np.hstack((Rb, -u3))]
    best_P, n = perform_chirality_check(P2s, x1, x2)
    return best_P[:, :3], best_P[:, 3], n


def homography_to_RT(H, x1, x2):
    """src/ransac_algorithm.py:332-403. Works on a copy of H (the reference flips H in place)."""
    H = np.array(H, float)

    def unitize(a, b):
        den = 1.0 / np.sqrt(a**2 + b**2)
        return a * den, b * den

    N = x1.shape[1]
    positives = np.sum(np.sum(x2 * (H @ x1), axis=0) > 0)
    if positives < N / 2:
        H = -H
    with np.errstate(all="ignore"):
        U, S, Vt = np.linalg.svd(H)
        s1 = S[0] / S[1]
        s3 = S[2] / S[1]
        a1 = np.sqrt(1 - s3**2)
        b1 = np.sqrt(s1**2 - 1)
        a, b = unitize(a1, b1)
        c, d = unitize(1 + s1 * s3, a1 * b1)
        e, f = unitize(-b / s1, -a / s3)
        v1 = Vt.T[:, 0]
        v3 = Vt.T[:, 2]
        n1 = b * v1 - a * v3
        n2 = b * v1 + a * v3
        R1 = U @ np.array([[c, 0, d], [0, 1, 0], [-d, 0, c]]) @ Vt
        R2 = U @ np.array([[c, 0, -d], [0, 1, 0], [d, 0, c]]) @ Vt
        t1 = e * v1 + f * v3
        t2 = e * v1 - f * v3
        if n1[2] < 0:
            t1 = -t1
        if n2[2] < 0:
            t2 = -t2
        t1 = R1 @ t1
        t2 = R2 @ t2
    return R1, t1, R2, t2


# --------------------------------------------------------------------------
# joint E/H RANSAC (sequential state machine)
# --------------------------------------------------------------------------


def ransac_iters(alpha, eps, s):
    """src/ransac_algorithm.py:87-88 incl. the log(1)=0 -> inf behaviour."""
    with np.errstate(all="ignore"):
        return np.abs(np.log(1 - alpha) / np.log(1 - eps**s))


def estimate_E_robust(K, x1, x2, pixel_threshold, trace=None, max_iterations=None):
    """src/ransac_algorithm.py:57-173, consuming the global legacy np.random stream
    in the reference's order (8 then 4 integers per iteration).

    trace: optional dict that receives lists 'cE', 'cH', 'idx', 'events' ((iteration, 'E'|'H', accepted) per record event)
    and 'iterations'.
    max_iterations: safety stop for tests (the reference has none).
    """
    N = x1.shape[1]
    alpha = 0.95
    eps_E = eps_H = 0.1
    best_E = 0
    best_H = 0
    thr = pixel_threshold / K[0][0]
    E_iters = ransac_iters(alpha, eps_E, 8)
    H_iters = ransac_iters(alpha, eps_H, 4)
    R_best = T_best = inliers = None
    it = 0
    while it < max(E_iters, H_iters):
        if max_iterations is not None and it >= max_iterations:
            break
        iE = np.random.randint(0, N, size=8)
        E = enforce_essential(estimate_F_DLT(x1[:, iE], x2[:, iE]))
        mask = epipolar_mask(E, x1, x2, thr)
        nE = int(np.sum(mask))
        if nE > best_E:
            if trace is not None:
                trace.setdefault("events", []).append((it, "E", True))
            best_E = nE
            R_best, T_best, _ = essential_to_RT(E, x1, x2)
            eps_E = best_E / N
            E_iters = ransac_iters(alpha, eps_E, 8)
            inliers = mask
        iH = np.random.randint(0, N, size=4)
        H = estimate_H_DLT(x1[:, iH], x2[:, iH])
        mask_H = get_inlier_mask_H(H, x1, x2, thr * 3)
        nH = int(np.sum(mask_H))
        if trace is not None:
            trace.setdefault("cE", []).append(nE)
            trace.setdefault("cH", []).append(nH)
            trace.setdefault("idx", []).append(np.concatenate([iE, iH]))
        if nH > best_H:
            best_H = nH
            Ra, Ta, Rb, Tb = homography_to_RT(H, x1, x2)
            Ea = skew_symmetric_mat(Ta) @ Ra
            Eb = skew_symmetric_mat(Tb) @ Rb
            ma = epipolar_mask(Ea, x1, x2, thr)
            mb = epipolar_mask(Eb, x1, x2, thr)
            Rba, Tba, na = essential_to_RT(Ea, x1[:, ma], x2[:, ma])
            Rbb, Tbb, nb = essential_to_RT(Eb, x1[:, mb], x2[:, mb])
            if na > nb:
                pick = (Rba, Tba, na, ma) if na > nE else None
            else:
                pick = (Rbb, Tbb, nb, mb) if nb > nE else None
            if trace is not None:
                trace.setdefault("events", []).append((it, "H", pick is not None))
            if pick is not None:
                R_best, T_best, best_E, inliers = pick
                eps_H = best_H / N
                eps_E = best_E / N
                H_iters = ransac_iters(alpha, eps_H, 4)
                E_iters = ransac_iters(alpha, eps_E, 8)
        it += 1
    if trace is not None:
        trace["iterations"] = it
    return R_best, np.reshape(T_best, (3, 1)), inliers


def run_ransac(K, x1, x2, pixel_threshold, **kw):
    """src/ransac_algorithm.py:47-54"""
    return estimate_E_robust(K, normalize_K(K, x1), normalize_K(K, x2), pixel_threshold, **kw)


def estimate_R(K, x_pairs, pixel_threshold):
    """src/ransac_algorithm.py:10-44"""
    return [
        run_ransac(K, x_pairs[2 * i], x_pairs[2 * i + 1], pixel_threshold)[0]
        for i in range(len(x_pairs) // 2)
    ]


# --------------------------------------------------------------------------
# 2-point translation RANSAC
# --------------------------------------------------------------------------


def estimate_T_2point(x_norm, X, R):
    """src/estimate_translation.py:80-106 (scipy lstsq / gelsd on the stacked 6x3 system)."""
    A = []
    b = []
    for Xj, xj in zip(X.T, x_norm.T):
        S = skew_symmetric_mat(xj)
        A.append(S)
        b.append(-S @ (R @ Xj))
    A = np.vstack(A)
    b = np.vstack(b).reshape(-1)
    T = scipy.linalg.lstsq(A, b)[0]
    return T.flatten()


def estimate_T_2point_batch(x_norm, X, R, idx):
    """Vectorised 2-point solve for idx (B,2): minimum-norm least squares through pinv,
    which equals gelsd's answer for both the rank-3 and the duplicated-index rank-2 case."""
    RX = R @ X.astype(np.float64)
    xs = x_norm[:, idx]  # (3,B,2)
    rs = RX[:, idx]
    B = idx.shape[0]
    A = np.zeros((B, 6, 3))
    b = np.zeros((B, 6))
    for j in range(2):
        x, y, z = xs[0, :, j], xs[1, :, j], xs[2, :, j]
        S = np.zeros((B, 3, 3))
        S[:, 0, 1] = -z
        S[:, 0, 2] = y
        S[:, 1, 0] = z
        S[:, 1, 2] = -x
        S[:, 2, 0] = -y
        S[:, 2, 1] = x
        A[:, 3 * j : 3 * j + 3] = S
        b[:, 3 * j : 3 * j + 3] = -(S @ rs[:, :, j].T[:, :, None])[:, :, 0]
    return (np.linalg.pinv(A) @ b[:, :, None])[:, :, 0]


def count_inliers(x_norm, X, R, T, threshold):
    """src/estimate_translation.py:109-129 (no depth-sign test)."""
    with np.errstate(all="ignore"):
        p = R @ X + T[:, None]
        p = p / p[2, :]
        d = np.linalg.norm(p[:2, :] - x_norm[:2, :], axis=0)
        return int(np.sum(d < threshold))


def count_inliers_batch(x_norm, X, R, Ts, threshold, chunk=512):
    RX = R @ X.astype(np.float64)
    out = np.zeros(Ts.shape[0], np.int64)
    with np.errstate(all="ignore"):
        for s in range(0, Ts.shape[0], chunk):
            p = RX[None] + Ts[s : s + chunk, :, None]
            p = p / p[:, 2:3, :]
            d = np.sqrt(
                (p[:, 0] - x_norm[0]) ** 2 + (p[:, 1] - x_norm[1]) ** 2
            )
            out[s : s + chunk] = np.sum(d < threshold, axis=-1)
    return out


def robust_estimate_T(x_norm, X, K, R, pixel_threshold, num_iterations=100000, info=None):
    """src/estimate_translation.py:43-77, vectorised: draws 2*num_iterations integers
    from the global legacy stream (identical to the reference's per-iteration draws),
    first strict maximum wins, None if every count is zero."""
    threshold = 4 * pixel_threshold / K[0][0]
    M = x_norm.shape[1]
    idx = np.random.randint(0, M, size=2 * num_iterations).reshape(num_iterations, 2)
    Ts = estimate_T_2point_batch(x_norm, X, R, idx)
    counts = count_inliers_batch(x_norm, X, R, Ts, threshold)
    best = int(np.argmax(counts))  # first maximum
    if info is not None:
        info.update(index=best, count=int(counts[best]), counts=counts, idx=idx)
    if counts[best] <= 0:
        return None
    return Ts[best]


# --------------------------------------------------------------------------
# LM refinement of T
# --------------------------------------------------------------------------


def compute_reprojection_error(X, x_norm, K, R, T):
    """src/refine_translation.py:107-132: [dx_0..dx_{n-1}, dy_0..dy_{n-1}]."""
    p = R @ X + T[:, None]
    p = p / p[2]
    return (p[:2, :] - x_norm[:2, :]).flatten()


def lm_jacobian(X, K, R, T):
    """Closed form of src/refine_translation.py:158-200 stacked as in :89-93:
    rows interleaved [u_0, v_0, u_1, v_1, ...] and taken of the PIXEL projection K(RX+T)."""
    K = np.asarray(K, float)
    Xc = R @ X + T[:, None]
    xp = K @ Xc
    n = X.shape[1]
    J = np.zeros((2 * n, 3))
    inv = 1 / (xp[2] ** 2)
    for i in range(3):
        dk = K[:, i]
        J[0::2, i] = (xp[2] * dk[0] - xp[0] * dk[2]) * inv
        J[1::2, i] = (xp[2] * dk[1] - xp[1] * dk[2]) * inv
    return J


def levenberg_marquardt_algorithm(K, R, X, x_norm, T_initial, num_iterations, mu, history=None):
    """src/refine_translation.py:57-104 with both reference quirks: the residual vector is
    ordered [all dx | all dy] while the Jacobian rows are interleaved, and the Jacobian is of
    the pixel projection while the residual is in normalised coordinates."""
    T = np.copy(T_initial)
    for _ in range(num_iterations):
        r = compute_reprojection_error(X, x_norm, K, R, T)
        J = lm_jacobian(X, K, R, T)
        C = J.T @ J + mu * np.eye(3)
        c = J.T @ r
        delta = np.linalg.solve(-C, c)
        Tn = T + delta
        rn = compute_reprojection_error(X, x_norm, K, R, Tn)
        if np.sum(rn**2) < np.sum(r**2):
            T = Tn
        if history is not None:
            history.append(np.array(T, float).copy())
    return np.reshape(T, (3, 1))


# --------------------------------------------------------------------------
# initial reconstruction
# --------------------------------------------------------------------------


def compute_absolute_rotations(relative_rotations):
    """src/reconstruct_3D.py:40-60: R_abs[i+1] = R_abs[i] @ R_rel[i]."""
    out = [np.eye(3)]
    for Rr in relative_rotations:
        out.append(np.dot(out[-1], Rr))
    return out


def reconstruct_initial_3D_points(init_pair_dict, K, R_init_1, pixel_threshold):
    """src/reconstruct_3D.py:63-95: triangulates ALL matches, not only inliers."""
    x1 = init_pair_dict["x_init"][0]
    x2 = init_pair_dict["x_init"][1]
    R, T, inliers = run_ransac(K, x1, x2, pixel_threshold)
    P1 = np.concatenate((np.eye(3), np.zeros((3, 1))), axis=1)
    P2 = np.hstack((R, T))
    X0 = triangulate_3D_point_DLT([P1, P2], [normalize_K(K, x1), normalize_K(K, x2)])
    X0 = X0[:3, :] / X0[3, :]
    return np.dot(R_init_1.T, X0), inliers


def run_reconstruction(relative_rotations, init_pair_dict, K, pixel_threshold):
    """src/reconstruct_3D.py:7-37"""
    abs_rot = compute_absolute_rotations(relative_rotations)
    X0, inliers = reconstruct_initial_3D_points(
        init_pair_dict, K, abs_rot[init_pair_dict["init_pair_indices"][0]], pixel_threshold
    )
    return X0, abs_rot, inliers


# --------------------------------------------------------------------------
# drivers over a correspondence provider (the reference calls cv2 here)
# --------------------------------------------------------------------------


def estimate_translation(K, desc_X, X0, absolute_rotations, img_paths, pixel_threshold, find_correspondences):
    """src/estimate_translation.py:7-40 with the 2D-3D matcher passed in."""
    out = []
    for i in range(len(img_paths)):
        X_corr, x_corr = find_correspondences(img_paths[i], desc_X, X0, K)
        out.append(robust_estimate_T(x_corr, X_corr, K, absolute_rotations[i], pixel_threshold))
    return out


def refine_translation(K, img_paths, desc_X_inliers, X0, absolute_rotations, estimated_Ts, find_correspondences):
    """src/refine_translation.py:7-54 with the 2D-3D matcher passed in."""
    cams = [find_correspondences(p, desc_X_inliers, X0, K) for p in img_paths]
    return [
        levenberg_marquardt_algorithm(K, absolute_rotations[i], cams[i][0], cams[i][1], T, 10, 0.01)
        for i, T in enumerate(estimated_Ts)
    ]


def run_geometry(K, x_pairs, init_pair_dict, img_paths, pixel_threshold, find_correspondences):
    """run_sfm.py:59-80."""
    R_list = estimate_R(K, x_pairs, pixel_threshold)
    X0, abs_rot, inliers = run_reconstruction(R_list, init_pair_dict, K, pixel_threshold)
    desc_X = init_pair_dict["init_pair_desc"]
    est = estimate_translation(K, desc_X, X0, abs_rot, img_paths, pixel_threshold, find_correspondences)
    ref = refine_translation(K, img_paths, desc_X[:, inliers], X0, abs_rot, est, find_correspondences)
    return {"absolute_rotations": abs_rot, "refined_Ts": ref, "X0": X0, "inliers": inliers, "estimated_Ts": est}


def camera_centres_and_axes(P_list):
    """utils.py:92-102 of the reference's visualiser: the centre of every camera is the null vector of its 3x4
    matrix divided by its last entry, the drawn viewing direction is P[2, 0:3]."""
    from scipy.linalg import null_space

    c = np.zeros((4, len(P_list)))
    v = np.zeros((3, len(P_list)))
    for i, P in enumerate(P_list):
        ns = null_space(P)
        if ns.size:
            c[:, i] = ns[:, 0]
        v[:, i] = P[2, 0:3]
    c = c / c[3, :]
    return c[:3], v

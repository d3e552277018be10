"""Synthetic correspondence sets shaped like SIFT matches of the bundled datasets.

Used by the parity tests and by bench.py (there is no dataset download and no
OpenCV on the build or GPU machines).  All draws come from
numpy.random.default_rng(seed) so they never disturb the legacy global stream
the RANSAC index draws use.
"""
from __future__ import annotations

import numpy as np

# Intrinsics of dataset 5 (data/5/cfg.yml:15-16 in the reference) and dataset 10
# (data/10/cfg.yml:17-18, fx != fy).
K_DATASET5 = [[2489.14, 0, 968], [0, 2489.14, 648], [0, 0, 1]]
K_DATASET10 = [[689.87, 0, 380.17], [0, 691.04, 251.70], [0, 0, 1]]


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _through_f32(xy):
    """Pixel coordinates pass through float32 once and are then held as float64
    (reference: src/extract_sift.py:126-130)."""
    return xy.astype(np.float32).astype(np.float64)


def two_view_scene(n=10000, inlier_ratio=0.3, planar_ratio=0.5, seed=0, K=K_DATASET5,
                   noise_px=0.3, image_size=(1936, 1296), R=None, T=None):
    """Two-view match set: homogeneous pixel coordinates x1, x2 (3,n) float64.

    Points are x,y~U(-2,2), z~U(4,8); a fraction lies on the plane z = 6 + 0.2 x so that
    the joint E/H loop of the reference terminates.  Camera 2 is [R|T].  A fraction
    (1 - inlier_ratio) of x2 is replaced by uniform image positions.
    Returns dict(x1, x2, K, R, T, is_inlier).
    """
    rng = np.random.default_rng(seed)
    Kn = np.asarray(K, float)
    R = rot_y(0.1) if R is None else R
    T = np.array([1.0, 0.1, 0.05]) if T is None else np.asarray(T, float)
    X = np.empty((3, n))
    X[0] = rng.uniform(-2, 2, n)
    X[1] = rng.uniform(-2, 2, n)
    X[2] = rng.uniform(4, 8, n)
    planar = rng.random(n) < planar_ratio
    X[2, planar] = 6 + 0.2 * X[0, planar]
    p1 = Kn @ X
    p2 = Kn @ (R @ X + T[:, None])
    x1 = p1[:2] / p1[2] + rng.normal(0, noise_px, (2, n))
    x2 = p2[:2] / p2[2] + rng.normal(0, noise_px, (2, n))
    is_inlier = rng.random(n) < inlier_ratio
    nout = int((~is_inlier).sum())
    x2[0, ~is_inlier] = rng.uniform(0, image_size[0], nout)
    x2[1, ~is_inlier] = rng.uniform(0, image_size[1], nout)
    x1 = np.vstack([_through_f32(x1), np.ones((1, n))])
    x2 = np.vstack([_through_f32(x2), np.ones((1, n))])
    return dict(x1=x1, x2=x2, K=K, R=R, T=T, is_inlier=is_inlier, X=X)


def resection_scene(m=50000, outlier_ratio=0.5, seed=0, K=K_DATASET5, noise_px=0.5,
                    R=None, T=None):
    """2D-3D correspondences for the translation stages.

    Returns dict(X (3,m) float32, x_norm (3,m) float64, K, R, T_true) with
    x = pi(R X + T) + noise/f in normalised coordinates, a fraction replaced by outliers.
    Dtypes follow the reference's find_correspondences (src/utils.py:89-91).
    """
    rng = np.random.default_rng(seed)
    f = float(K[0][0])
    R = rot_y(0.07) @ rot_x(-0.03) if R is None else R
    T = np.array([0.4, -0.2, 0.3]) if T is None else np.asarray(T, float)
    X = np.empty((3, m))
    X[0] = rng.uniform(-2, 2, m)
    X[1] = rng.uniform(-2, 2, m)
    X[2] = rng.uniform(4, 8, m)
    X = X.astype(np.float32)
    p = R @ X.astype(np.float64) + T[:, None]
    x = p[:2] / p[2] + rng.normal(0, noise_px / f, (2, m))
    out = rng.random(m) < outlier_ratio
    x[:, out] = rng.uniform(-0.4, 0.4, (2, int(out.sum())))
    # normalised coordinates come from float32 pixel positions in the reference
    Kn = np.asarray(K, float)
    px = _through_f32(Kn[:2, :2] @ x + Kn[:2, 2:3])
    x_norm = np.linalg.inv(Kn) @ np.vstack([px, np.ones((1, m))])
    return dict(X=X, x_norm=x_norm, K=K, R=R, T_true=T, is_outlier=out)


def multi_view_scene(n_cams=5, n_points=900, init_pair=(1, 3), seed=0, K=K_DATASET5, inlier_ratio=0.7,
                     planar_ratio=0.6, noise_px=0.3, image_size=(1936, 1296)):
    """A small multi-camera scene in the shape the pipeline consumes: consecutive-pair matches
    (`x_pairs`), the initial pair with one descriptor column per match, and a synthetic 2D-3D
    matcher that stands in for the OpenCV front-end.  Descriptors are one-hot-free: column j of
    `init_pair_desc` just stores the point id j, and the matcher looks ids up.
    """
    rng = np.random.default_rng(seed)
    Kn = np.asarray(K, float)
    X = np.empty((3, n_points))
    X[0] = rng.uniform(-2, 2, n_points)
    X[1] = rng.uniform(-2, 2, n_points)
    X[2] = rng.uniform(4, 8, n_points)
    planar = rng.random(n_points) < planar_ratio
    X[2, planar] = 6 + 0.2 * X[0, planar]
    Rs = [np.eye(3)]
    Ts = [np.zeros(3)]
    for c in range(1, n_cams):
        Rs.append(Rs[-1] @ rot_y(0.05 + 0.01 * c) @ rot_x(0.01 * (-1) ** c))
        Ts.append(np.array([0.45 * c, 0.05 * c, 0.02 * c]))

    def project(c, idxs, with_outliers):
        p = Kn @ (Rs[c] @ X[:, idxs] + Ts[c][:, None])
        x = p[:2] / p[2] + rng.normal(0, noise_px, (2, len(idxs)))
        if with_outliers:
            out = rng.random(len(idxs)) > inlier_ratio
            x[0, out] = rng.uniform(0, image_size[0], int(out.sum()))
            x[1, out] = rng.uniform(0, image_size[1], int(out.sum()))
        return np.vstack([_through_f32(x), np.ones((1, len(idxs)))])

    x_pairs = []
    for c in range(n_cams - 1):
        ids = rng.choice(n_points, size=int(0.7 * n_points), replace=False)
        x_pairs += [project(c, ids, False), project(c + 1, ids, True)]
    ids0 = rng.choice(n_points, size=int(0.8 * n_points), replace=False)
    a, b = init_pair
    init_pair_dict = {
        "init_pair_indices": list(init_pair),
        "x_init": [project(a, ids0, False), project(b, ids0, True)],
        "init_pair_desc": np.tile(np.arange(len(ids0), dtype=np.float32), (128, 1)),
    }
    img_paths = [f"synthetic://camera/{c}" for c in range(n_cams)]
    obs = {}
    for c in range(n_cams):
        sel = rng.choice(len(ids0), size=int(0.6 * len(ids0)), replace=False)
        px = project(c, ids0[sel], True)
        obs[img_paths[c]] = (sel, px)

    def find_correspondences(img_path, desc_X, X0, K_):
        """Stands in for src/utils.py:65-93: returns (X float32 3xM, x_norm float64 3xM) for the
        descriptor columns present in desc_X (ids in row 0)."""
        sel, px = obs[img_path]
        present = {int(v): j for j, v in enumerate(desc_X[0])}
        keep = [k for k, s in enumerate(sel) if int(s) in present]
        cols = [present[int(sel[k])] for k in keep]
        Xc = np.float32(X0[:, cols])
        xn = np.linalg.inv(np.asarray(K_, float)) @ px[:, keep]
        return Xc, xn

    return dict(K=K, x_pairs=x_pairs, init_pair_dict=init_pair_dict, img_paths=img_paths, find_correspondences=find_correspondences,
                R_true=Rs, T_true=Ts, X_true=X)

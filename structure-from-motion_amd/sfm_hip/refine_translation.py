"""Levenberg-Marquardt refinement of T with the reference's names (src/refine_translation.py).

All cameras run in one launch, one workgroup per camera, all iterations inside the kernel.
"""
from __future__ import annotations

import numpy as np

from . import _capi as C
from . import ops
from . import utils
from .estimate_translation import MAX_CAMERAS_PER_LAUNCH, _pack_camera
from .utils import log_execution_time


def levenberg_marquardt_batch(K, cameras, num_iterations=10, mu=0.01, want_history=False):
    """cameras: list of (R, X, x_norm, T_initial).  Returns a list of (3,1) arrays
    (and, if asked, the translation after every iteration, shape (iters,3) per camera)."""
    import torch

    Kd = C.dev(np.asarray(K, np.float64).reshape(9))
    out, hists = [], []
    for s in range(0, len(cameras), MAX_CAMERAS_PER_LAUNCH):
        group = cameras[s : s + MAX_CAMERAS_PER_LAUNCH]
        sizes = [int(np.asarray(c[2]).shape[1]) for c in group]
        pts = [_pack_camera(c[2], c[1], c[0]) for c in group]
        pts = torch.cat(pts) if sum(sizes) else torch.zeros((1, 6), dtype=torch.float64, device="cuda")
        # np.copy(None) is what the reference trips over when RANSAC found no translation (:82)
        T0 = np.stack([np.asarray(np.copy(c[3]), np.float64).reshape(3) for c in group])
        res = ops.lm_translation(Kd, pts, sizes, C.dev(T0), num_iterations, mu, want_history=want_history)
        T, h = res if want_history else (res, None)
        T = T.cpu().numpy()
        for c in range(len(group)):
            out.append(T[c].reshape(3, 1).copy())
            if want_history:
                hists.append(h.cpu().numpy()[c])
    return (out, hists) if want_history else out


def levenberg_marquardt_algorithm(K, R, X, x_norm, T_initial, num_iterations, mu):
    """Reference: src/refine_translation.py:57-104 (fixed damping, accept only on decrease)."""
    return levenberg_marquardt_batch(K, [(R, X, x_norm, T_initial)], num_iterations, mu)[0]


@log_execution_time
def refine_translation(K, img_paths, desc_X_inliers, X0, absolute_rotations, estimated_Ts):
    """Reference: src/refine_translation.py:7-54."""
    cams = []
    for i in range(len(img_paths)):
        X_corr, x_corr_norm = utils.find_correspondences(img_paths[i], desc_X_inliers, X0, K)
        cams.append((X_corr, x_corr_norm))
    batch = [(absolute_rotations[i], cams[i][0], cams[i][1], T) for i, T in enumerate(estimated_Ts)]
    return levenberg_marquardt_batch(K, batch, num_iterations=10, mu=0.01)


# ---- pieces of one LM iteration, kept for callers of the reference's module API ------------
# The kernel fuses them; these NumPy forms are not used by the pipeline path.


def compute_reprojection_error(X, x_norm, K, R, T):
    """Reference: src/refine_translation.py:107-132."""
    p = R @ X + T[:, np.newaxis]
    p = p / p[2]
    return (p[:2, :] - x_norm[:2, :]).flatten()


def ComputeUpdate(error, J, mu):
    """Reference: src/refine_translation.py:135-155."""
    C_ = J.T @ J + mu * np.eye(J.shape[1])
    return np.linalg.solve(-C_, J.T @ error)


def projection_derivatives_wrt_T(X, K, R, T):
    """Reference: src/refine_translation.py:158-200 (2x3 Jacobian of the PIXEL projection)."""
    Kn = np.asarray(K, float)
    xp = Kn @ (R @ X + T)
    J = np.zeros((2, 3))
    for i in range(3):
        dk = Kn[:, i]
        inv = 1 / (xp[2] ** 2)
        J[0, i] = (xp[2] * dk[0] - xp[0] * dk[2]) * inv
        J[1, i] = (xp[2] * dk[1] - xp[1] * dk[2]) * inv
    return J

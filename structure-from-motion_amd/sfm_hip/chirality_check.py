"""Chirality check with the reference's names (src/chirality_check.py), on the GPU."""
from __future__ import annotations

import numpy as np

from . import _capi as C
from . import ops


def perform_chirality_check(P2s, x1, x2):
    """Reference: src/chirality_check.py:5-36.  Triangulates every correspondence under each
    of the candidate second cameras (first camera [I|0]; four in the pipeline, any number here) and returns the
    candidate with the most points in front of both cameras (first strict maximum) and that count."""
    if len(P2s) == 0:
        raise ValueError("no candidate cameras")
    P = [np.asarray(p, np.float64).reshape(12) for p in P2s]
    if np.asarray(x1).shape[1] == 0:
        counts = np.zeros(len(P), np.int32)
    else:
        corr = C.dev(ops.pack_corr(x1, x2))
        counts = []
        for g in range(0, len(P), 4):  # the kernel scores four candidates per launch (what the pipeline passes); pad the last group
            group = P[g : g + 4] + [P[-1]] * (4 - len(P[g : g + 4]))
            counts.extend(ops.chirality4(C.dev(np.stack(group)), corr, None).cpu().numpy()[: len(P[g : g + 4])])
        counts = np.asarray(counts, np.int32)
    best_idx, n = select_best_from_counts(counts)
    return P2s[best_idx], n


def select_best_from_counts(counts):
    """The update rule of src/chirality_check.py:79-91 applied to precomputed counts."""
    best, bi = -1, -1
    for i, c in enumerate(counts):
        if c > best:
            best, bi = int(c), i
    return bi, best


def count_points_in_front_of_cam(P1, P2, points_3d):
    """Reference: src/chirality_check.py:39-60, for already triangulated points (4xN).
    A convenience for callers of the old API; the pipeline path counts inside the GPU kernel."""
    with np.errstate(all="ignore"):
        return int(np.sum(((P2 @ points_3d)[2] > 0) & ((P1 @ points_3d)[2] > 0)))


def select_best_solution(triangulated_points, P1, P2s):
    """Reference: src/chirality_check.py:63-91."""
    return select_best_from_counts([count_points_in_front_of_cam(P1, P2s[i], X) for i, X in enumerate(triangulated_points)])

"""NumPy restatement of the descriptor matcher (TEST INFRASTRUCTURE ONLY).

The reference matches SIFT descriptors with OpenCV: ``cv2.BFMatcher().knnMatch(d_query, d_train, k=2)``
followed by ``m.distance < ratio * n.distance`` (src/extract_sift.py:117-123, ratio 0.4;
src/utils.py:82-86, ratio 0.75).  OpenCV (opencv-python, any 4.x; not installed here) is a
third-party dependency absent from the reference tree, and the reference holds no match fixtures:
PARITY UNPINNED against cv2 itself.  What is restated is BFMatcher's published behaviour for
NORM_L2 on float32 descriptors: distance = float32 sqrt of the sum of squared differences, the two
smallest distances per query in ascending order, the earlier train index first on equal distance.
For integer-valued descriptors (what cv2.SIFT emits) the sum of squares is an exact integer in any
summation order, so this restatement does not depend on OpenCV's SIMD accumulation order.
"""
import numpy as np


def knn2_ratio(desc_q, desc_t, ratio, chunk=2048):
    """desc_q [Q][128], desc_t [T][128] (integer valued).  Returns (train_idx int32[Q], dist1 f32[Q], dist2 f32[Q], good bool[Q])."""
    q = np.asarray(desc_q, dtype=np.float64)
    t = np.asarray(desc_t, dtype=np.float64)
    if t.shape[0] < 2:
        raise ValueError("need at least two train descriptors")
    tn = (t * t).sum(1)
    idx = np.empty(q.shape[0], np.int32)
    d1 = np.empty(q.shape[0], np.float32)
    d2 = np.empty(q.shape[0], np.float32)
    for s in range(0, q.shape[0], chunk):
        qq = q[s : s + chunk]
        d = (qq * qq).sum(1)[:, None] + tn[None, :] - 2.0 * (qq @ t.T)  # exact: integers below 2^53
        i1 = d.argmin(1)  # first occurrence = lowest train index
        rows = np.arange(qq.shape[0])
        best = d[rows, i1].copy()
        d[rows, i1] = np.inf
        second = d.min(1)
        idx[s : s + chunk] = i1
        d1[s : s + chunk] = np.sqrt(best.astype(np.float32))
        d2[s : s + chunk] = np.sqrt(second.astype(np.float32))
    good = d1.astype(np.float64) < ratio * d2.astype(np.float64)
    return idx, d1, d2, good


def synthetic_descriptors(n_shared, n_only_q, n_only_t, noise=(2.0, 45.0), seed=0):
    """SIFT-like uint8 descriptor sets: `n_shared` features seen in both sets (the train copy perturbed with a
    per-feature noise level drawn from `noise` = (lo, hi), so the distance ratios straddle the usual thresholds, and
    re-quantised), the rest unrelated; both sets shuffled.  Returns (desc_q, desc_t, truth) with truth[i] = train
    index of query i or -1."""
    rng = np.random.default_rng(seed)

    def raw(n):
        v = rng.gamma(0.6, 1.0, size=(n, 128))
        v = np.minimum(v / np.linalg.norm(v, axis=1, keepdims=True), 0.2)
        v = v / np.linalg.norm(v, axis=1, keepdims=True)
        return np.clip(np.rint(512.0 * v), 0, 255)

    shared = raw(n_shared)
    moved = np.clip(np.rint(shared + rng.normal(0.0, 1.0, shared.shape) * rng.uniform(noise[0], noise[1], (n_shared, 1))), 0, 255)
    dq = np.concatenate([shared, raw(n_only_q)])
    dt = np.concatenate([moved, raw(n_only_t)])
    pq = rng.permutation(dq.shape[0])
    pt = rng.permutation(dt.shape[0])
    inv_t = np.empty_like(pt)
    inv_t[pt] = np.arange(pt.size)
    truth = np.full(dq.shape[0], -1, np.int64)
    truth[:n_shared] = inv_t[:n_shared]
    return dq[pq].astype(np.uint8), dt[pt].astype(np.uint8), truth[pq]

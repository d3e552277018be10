"""Two-view E/H RANSAC with the names and signatures of the reference's
src/ransac_algorithm.py, running on the GPU.

estimate_E_robust reproduces the reference's sequential state machine (adaptive trip count,
E and H record events, the H branch that may overwrite and even lower the best E count) by
replaying it over inlier counts that the GPU computes for blocks of pre-drawn index sets.
The index sets come from the global legacy np.random stream in the reference's order (8 then
4 integers per iteration); on return the stream is exactly where the reference leaves it.
"""
from __future__ import annotations

import logging
import math
import time

import numpy as np

from . import _capi as C
from . import legacy_rng
from . import ops
from .utils import normalize_K

try:  # progress bars are a side effect of the reference too (src/ransac_algorithm.py:25)
    from tqdm import tqdm
except Exception:  # pragma: no cover

    def tqdm(it, **kw):
        return it


ALPHA = 0.95  # src/ransac_algorithm.py:77
EPS0 = 0.1  # :78-79
S_E, S_H = 8, 4  # :75-76


def ransac_iters(eps, s):
    """|log(1-alpha)/log(1-eps^s)| of src/ransac_algorithm.py:87-88, :112, :156-157; like the
    reference it is inf when 1-eps^s rounds to 1."""
    with np.errstate(all="ignore"):
        return np.abs(np.log(1 - ALPHA) / np.log(1 - np.float64(eps) ** s))


class ReplayState:
    """The reference's loop state (src/ransac_algorithm.py:74-90)."""

    def __init__(self, n):
        self.n = n
        self.best_E = 0
        self.best_H = 0
        self.E_iters = ransac_iters(EPS0, S_E)
        self.H_iters = ransac_iters(EPS0, S_H)
        self.it = 0
        self.R = None
        self.T = None
        self.inliers = None
        self.pending_E = None  # index row of the latest E record event whose R, T, inliers have not been computed yet
        self.events = []  # (iteration, 'E'|'H', accepted) for logging and tests
        self.event_rows = []  # the index row of each of them

    def limit(self):
        return max(self.E_iters, self.H_iters)


def replay_block(st, idx, cE, cH, backend, exact=None):
    """Advances the state machine over one block of hypotheses.

    idx: (nb,12) index sets, cE/cH: their inlier counts.  Returns the number of iterations of
    the block the reference would have executed (== nb unless the loop ended inside it).
    exact (optional bool (nb,)): where False, cE/cH hold UPPER BOUNDS of the counts.  The loop only looks at counts
    that exceed the running bests, so a bound at or below them is as good as the count; a bound above them is
    replaced by the exact count (backend.refine) before it is used."""
    nb = idx.shape[0]
    if hasattr(backend, "prefetch_h_records"):
        # best_H only ever rises (:124-125), so the H record events of this block are the strict running maxima of cH
        # above the incoming best_H, whatever the E branch does: they can all be computed before the replay starts
        run = np.maximum.accumulate(np.concatenate([[st.best_H], cH]))
        backend.prefetch_h_records(idx, np.flatnonzero(cH > run[:-1]))
    j = 0
    while j < nb:
        lim = st.limit()
        remaining = lim - st.it
        if np.isfinite(lim):
            k_stop = j + (int(math.ceil(remaining)) if remaining > 0 else 0)
        else:
            k_stop = nb + 1
        ev = np.flatnonzero((cE[j:] > st.best_E) | (cH[j:] > st.best_H))
        if exact is not None and ev.size and not exact[j + ev[0]]:
            # (only after an accepted H record has LOWERED the best E count below the floor the bounds were filtered
            # with: :150-154 / :161-165) every bound that now exceeds a running best becomes an exact count, in one call
            todo = j + ev[~exact[j + ev]]
            cE[todo], cH[todo] = backend.refine(todo)
            exact[todo] = True
            continue
        k_event = j + int(ev[0]) if ev.size else nb
        k_next = min(k_event, k_stop, nb)
        st.it += k_next - j
        j = k_next
        if j >= nb:
            break
        if k_stop <= k_event:
            return j  # `while iterations < max(E_iters, H_iters)` is false here (:91)
        nE = int(cE[j])
        if nE > st.best_E:  # :105-113
            st.best_E = nE
            if len(set(idx[j][:S_E].tolist())) < S_E:
                # a sample drawn with a repeated match makes the 8-point system rank deficient: its null vector is an
                # arbitrary one, here as in LAPACK, so this record need not coincide with the reference's
                logging.info("E record at iteration %d comes from a sample with a repeated match", st.it)
            # R, T and the inlier mask of an E record are only ever read after the loop, and any later record
            # overwrites them: the O(N) work (essential_to_RT over all matches, :110) is done once, for the last one
            st.pending_E = np.array(idx[j])
            st.E_iters = ransac_iters(st.best_E / st.n, S_E)
            st.events.append((st.it, "E", True))
            st.event_rows.append(np.array(idx[j]))
        nH = int(cH[j])
        if nH > st.best_H:  # :124-169
            st.best_H = nH
            if len(set(idx[j][S_E:].tolist())) < S_H:
                logging.info("H record at iteration %d comes from a sample with a repeated match", st.it)
            (Ra, Ta, na, ma), (Rb, Tb, nb_, mb) = backend.h_record(idx[j])
            pick = None
            if na > nb_:
                if na > nE:
                    pick = (Ra, Ta, na, ma)
            else:
                if nb_ > nE:
                    pick = (Rb, Tb, nb_, mb)
            if pick is not None:
                st.pending_E = None
                st.R, st.T, st.best_E, st.inliers = pick
                st.H_iters = ransac_iters(st.best_H / st.n, S_H)
                st.E_iters = ransac_iters(st.best_E / st.n, S_E)
            st.events.append((st.it, "H", pick is not None))
            st.event_rows.append(np.array(idx[j]))
        st.it += 1
        j += 1
    return j


def _block_size(it, lim, chunk, max_iterations):
    """Size of the next block for a loop that stands at iteration `it` with trip count `lim` (0: nothing left to draw)."""
    if not it < lim:
        return 0
    nb = chunk if not np.isfinite(lim) else int(min(chunk, max(1, math.ceil(lim - it))))
    if max_iterations is not None:
        nb = min(nb, max_iterations - it)
    return max(nb, 0)


def _draw(n, nb):
    """(stream position before the draw, index sets): == np.random.randint(0, n, size=12 * nb), same stream position after."""
    mark = legacy_rng.mark()
    return mark, legacy_rng.randint(n, 12 * nb).reshape(nb, 12)


AHEAD_MIN_PAIRS = 1 << 24  # draw the next block during the scoring of this one only when scoring takes longer than drawing


def run_replay(n, backend, chunk0=512, chunk_max=1 << 18, max_iterations=None, ahead_min_pairs=AHEAD_MIN_PAIRS):
    """Draws index blocks from np.random, scores them through `backend` and replays the state
    machine; rewinds the stream so that exactly 12 integers per executed iteration are consumed.

    A backend with submit/collect scores a block asynchronously: the next block's index sets are drawn on the host
    while the GPU works (sized as if the block in flight ran to its end; a loop that ends earlier just leaves more
    to rewind).  Drawing a sample costs the host about 50 ns, scoring it against a thousand matches costs the GPU about
    1 ns: on small problems the draws ARE the loop, so the first block is small (loops on real image pairs end after
    60-900 iterations) and nothing is drawn ahead until a block holds ahead_min_pairs (hypothesis, match) pairs."""
    st = ReplayState(n)
    chunk = chunk0
    nb = _block_size(st.it, st.limit(), chunk, max_iterations)
    nxt = _draw(n, nb) if nb else None
    while nxt is not None:
        rng_state, idx = nxt
        nb = idx.shape[0]
        handle = backend.submit(idx, st.best_E, st.best_H) if hasattr(backend, "submit") else None
        chunk = min(chunk * 2, chunk_max)
        nb_ahead = _block_size(st.it + nb, st.limit(), chunk, max_iterations) if handle is not None and nb * n >= ahead_min_pairs else 0
        nxt = _draw(n, nb_ahead) if nb_ahead else None  # speculative: overlaps with the scoring of this block
        exact = None
        if handle is not None:
            cE, cH, exact = backend.collect(handle, st.best_E, st.best_H)
        else:
            bounded = backend.score_bounded(idx, st.best_E, st.best_H) if hasattr(backend, "score_bounded") else None
            if bounded is not None:
                cE, cH, exact = bounded
            else:
                cE, cH = backend.score(idx)
        used = replay_block(st, idx, cE, cH, backend, exact)
        ended = used < nb or not st.it < st.limit() or (max_iterations is not None and st.it >= max_iterations)
        if ended:
            if used == nb and nxt is not None:
                legacy_rng.rewind(nxt[0])  # takes back the block drawn ahead
            elif used < nb:
                legacy_rng.rewind(rng_state)  # also takes back the block drawn ahead
                if used > 0:
                    legacy_rng.randint(n, 12 * used)
            break
        if nxt is None:  # the trip count grew past what was planned (an accepted H record may lower the best E count)
            nb = _block_size(st.it, st.limit(), chunk, max_iterations)
            nxt = _draw(n, nb) if nb else None
    if st.pending_E is not None:
        st.R, st.T, st.inliers = backend.e_record(st.pending_E)
        st.pending_E = None
    return st


def first_strict_max(counts):
    """select_best_solution of src/chirality_check.py:79-91: start at -1, strict '>'."""
    best, bi = -1, -1
    for i, c in enumerate(counts):
        if c > best:
            best, bi = int(c), i
    return bi, best


class GpuBackend:
    """Device side of the replay: block scoring and the two kinds of record event."""

    def __init__(self, x1, x2, err_threshold):
        self.thr = float(err_threshold)
        corr_h = ops.pack_corr(x1, x2)
        self.corr = C.dev(corr_h)
        self.n = self.corr.shape[0]
        self.scratch = ops.RecordScratch(self.n)

    MAX_PREFETCH = 64
    BOUNDED_MIN_PAIRS = 1 << 28  # below this a block is scored exactly: the launch, not the arithmetic, is what costs

    def submit(self, idx, best_E, best_H):
        """Uploads a block and launches its scoring without waiting for it: matrix-core UPPER bounds when the block is
        large and both running bests exist (the loop only ever looks at hypotheses whose counts exceed them), otherwise
        the exact counts of every hypothesis."""
        nb = idx.shape[0]
        d_idx = C.dev(idx, np.int32)
        if nb * self.n < self.BOUNDED_MIN_PAIRS or best_E <= 0 or best_H <= 0:
            return ("exact", d_idx) + ops.eh_score_mixed(self.corr, d_idx, self.thr, 3 * self.thr)
        if getattr(self, "corrm", None) is None:
            self.corrm = ops.corr_to_mfma(self.corr, self.thr)
        return ("bounds", d_idx) + ops.eh_upper_mfma(self.corr, self.corrm, d_idx, self.thr, 3 * self.thr)

    def collect(self, handle, best_E, best_H):
        """(cE, cH, exact) of a submitted block; exact is None when every count is exact, else a mask (see replay_block)."""
        import torch

        kind, d_idx, ue, uh = handle
        self.d_idx = d_idx
        self.h_cache = {}
        if kind == "exact":
            return ue.cpu().numpy(), uh.cpu().numpy(), None
        nb = d_idx.shape[0]
        cand = torch.nonzero((ue > best_E) | (uh > best_H)).flatten()
        if cand.shape[0] > nb // 8:  # the bests are still low, nearly everything is a candidate: exact counts for the whole block
            ce, ch = ops.eh_score_mixed(self.corr, d_idx, self.thr, 3 * self.thr)
            return ce.cpu().numpy(), ch.cpu().numpy(), None
        exact = np.zeros(nb, bool)
        if cand.shape[0]:
            xe, xh = ops.eh_score(self.corr, d_idx.index_select(0, cand), self.thr, 3 * self.thr)
            ue[cand] = xe
            uh[cand] = xh
            exact[cand.cpu().numpy()] = True
        return ue.cpu().numpy(), uh.cpu().numpy(), exact

    def score_bounded(self, idx, best_E, best_H):
        """Synchronous form of submit + collect for large blocks; None when exact scoring of the whole block is the
        better choice (small block, or no running bests yet)."""
        if idx.shape[0] * self.n < self.BOUNDED_MIN_PAIRS or best_E <= 0 or best_H <= 0:
            return None
        cE, cH, exact = self.collect(self.submit(idx, best_E, best_H), best_E, best_H)
        return (cE, cH, exact) if exact is not None else (cE, cH, np.ones(idx.shape[0], bool))

    def refine(self, positions):
        """Exact counts of the given positions of the block scored last."""
        import torch

        xe, xh = ops.eh_score(self.corr, self.d_idx.index_select(0, torch.from_numpy(np.asarray(positions)).cuda()), self.thr, 3 * self.thr)
        return xe.cpu().numpy(), xh.cpu().numpy()

    def score(self, idx):
        d_idx = self.d_idx = C.dev(idx, np.int32)
        self.h_cache = {}
        # float64 counts; the residual tests run on the matrix cores + float32 with float64 inside the error band
        # (the library measures the coordinate range itself and evaluates everything in float64 when it is unusable)
        ce, ch = ops.eh_score_mixed(self.corr, d_idx, self.thr, 3 * self.thr)
        return ce.cpu().numpy(), ch.cpu().numpy()

    def e_record(self, row):
        """R, T of essential_to_RT over ALL points (src/ransac_algorithm.py:110) and the inlier mask: one library call."""
        return ops.e_record(self.corr, row, self.thr, self.scratch)

    def prefetch_h_records(self, idx, positions):
        """All H record events of the block just scored, in one launch and one read-back."""
        import torch

        positions = positions[: self.MAX_PREFETCH]
        if positions.size == 0:
            return
        rows = self.d_idx.index_select(0, torch.from_numpy(positions).cuda())
        for p, res in zip(positions, ops.h_record_batch(self.corr, rows, self.thr)):
            self.h_cache[idx[p].tobytes()] = res

    def h_record(self, row):
        """Both homography solutions scored as src/ransac_algorithm.py:128-147 (prefetched with the block, else one call)."""
        hit = self.h_cache.get(np.asarray(row).tobytes()) if hasattr(self, "h_cache") else None
        return hit if hit is not None else ops.h_record(self.corr, row, self.thr, self.scratch)


def estimate_E_robust(K, x1, x2, pixel_threshold):
    """Reference: src/ransac_algorithm.py:57-173.  x1, x2: K-normalised 3xN.
    Returns (R (3,3), T (3,1), inliers bool (N,))."""
    err_threshold = pixel_threshold / K[0][0]
    backend = GpuBackend(x1, x2, err_threshold)
    st = run_replay(x1.shape[1], backend)
    if st.R is None:
        # the reference fails the same way when no hypothesis ever has an inlier (:173)
        raise UnboundLocalError("local variable 'R_best' referenced before assignment")
    inliers = st.inliers.cpu().numpy().astype(bool)
    logging.debug("E/H RANSAC: %d iterations, events %s", st.it, st.events)
    return st.R, np.reshape(st.T, (3, 1)), inliers


def run_ransac(K, x1, x2, pixel_threshold):
    """Reference: src/ransac_algorithm.py:47-54."""
    return estimate_E_robust(K, normalize_K(K, x1), normalize_K(K, x2), pixel_threshold)


def estimate_R(K, x_pairs, pixel_threshold):
    """Reference: src/ransac_algorithm.py:10-44: relative rotation of each consecutive pair."""
    R_list = []
    for i in tqdm(range(len(x_pairs) // 2), desc="Running RANSAC"):
        t0 = time.time()
        logging.info(f"Processing pair {i+1}/{len(x_pairs) // 2}")
        R, _, inliers = run_ransac(K, x_pairs[2 * i], x_pairs[2 * i + 1], pixel_threshold)
        logging.info(f"RANSAC completed for pair {i+1} in {time.time() - t0:.2f} seconds with {len(inliers)} inliers")
        R_list.append(R)
    return R_list


# ---- single-model functions of the reference module, each one GPU call --------------------


def _sample_corr(x1s, x2s, least):
    x1s = np.asarray(x1s, np.float64)
    x2s = np.asarray(x2s, np.float64)
    if x1s.shape[1] < least:
        raise ValueError(f"at least {least} correspondences are needed, got {x1s.shape[1]}")
    return ops.pack_corr(x1s, x2s)


def estimate_F_DLT(x1s, x2s):
    """Reference: src/ransac_algorithm.py:216-263 (the sign of F is arbitrary there as well: it is a singular vector).
    Eight correspondences go through the in-register solver of the scoring kernels, more through sfm_dlt_general_f64."""
    corr = _sample_corr(x1s, x2s, 8)
    if corr.shape[0] > 8:
        return ops.dlt_general(C.dev(corr), False).cpu().numpy().reshape(3, 3)
    idx = np.concatenate([np.arange(8), np.arange(4)])[None, :]
    F, _, _ = ops.eh_models(C.dev(corr), C.dev(idx, np.int32), want_F=True)
    return F.cpu().numpy().reshape(3, 3)


def enforce_essential(E):
    """Reference: src/ransac_algorithm.py:176-190."""
    return ops.enforce_essential(C.dev(np.asarray(E, np.float64).reshape(9))).cpu().numpy().reshape(3, 3)


def estimate_H_DLT(x1s, x2s):
    """Reference: src/ransac_algorithm.py:266-293 (four correspondences: the in-register solver; more: sfm_dlt_general_f64)."""
    corr = _sample_corr(x1s, x2s, 4)
    if corr.shape[0] > 4:
        return ops.dlt_general(C.dev(corr), True).cpu().numpy().reshape(3, 3)
    corr8 = np.concatenate([corr, corr], axis=0)
    idx = np.concatenate([np.arange(8), np.arange(4)])[None, :]
    _, H = ops.eh_models(C.dev(corr8), C.dev(idx, np.int32))
    return H.cpu().numpy().reshape(3, 3)


def epipolar_errors(F, x1, x2):
    """Reference: src/ransac_algorithm.py:193-213."""
    d = ops.epipolar_errors(C.dev(np.asarray(F, np.float64).reshape(9)), C.dev(x1, np.float64), C.dev(x2, np.float64))
    return d.cpu().numpy()


def get_inlier_mask_H(H, x1s, x2s, threshold):
    """Reference: src/ransac_algorithm.py:296-329."""
    corr = C.dev(ops.pack_corr(x1s, x2s))
    mask, _ = ops.transfer_mask(C.dev(np.asarray(H, np.float64).reshape(9)), corr, threshold)
    return mask.cpu().numpy().astype(bool)


def homography_to_RT(H, x1, x2):
    """Reference: src/ransac_algorithm.py:332-403.  Like the reference, H is negated IN PLACE
    when fewer than half of the points have x2 . (H x1) > 0 (:359-361)."""
    corr = C.dev(ops.pack_corr(x1, x2))
    Hd = C.dev(np.asarray(H, np.float64).reshape(9))
    out = ops.homography_to_rt(Hd, corr).cpu().numpy()
    if isinstance(H, np.ndarray) and out[42] != 0.0:
        H *= -1  # the kernel's sign vote said so
    return out[:9].reshape(3, 3), out[9:12].copy(), out[12:21].reshape(3, 3), out[21:24].copy()


def essential_to_RT(E, x1, x2):
    """Reference: src/ransac_algorithm.py:406-446."""
    corr = C.dev(ops.pack_corr(x1, x2)) if np.asarray(x1).shape[1] else None
    P = ops.essential_candidates(C.dev(np.asarray(E, np.float64).reshape(9)))
    if corr is None:
        counts = np.zeros(4, np.int32)
    else:
        counts = ops.chirality4(P, corr, None).cpu().numpy()
    bi, n = first_strict_max(counts)
    Pb = P.cpu().numpy()[bi].reshape(3, 4)
    return Pb[:, :3].copy(), Pb[:, 3].copy(), n

"""Host logic of the E/H RANSAC replay (sfm_hip.ransac_algorithm.run_replay) with a backend
built from the oracle: the block-wise replay must reproduce the reference's sequential loop
(fixtures) for any block size, and leave np.random where the reference leaves it.  CPU only."""
import numpy as np
import pytest

from oracle import sfm_oracle as orc
from sfm_hip import synthetic
from sfm_hip.ransac_algorithm import first_strict_max, ransac_iters, run_replay


class OracleBackend:
    def __init__(self, x1, x2, thr):
        self.x1, self.x2, self.thr = x1, x2, thr
        self.blocks = []

    def score(self, idx):
        self.blocks.append(idx.shape[0])
        cE, cH, _, _ = orc.score_EH_batch(self.x1, self.x2, idx, self.thr)
        return cE, cH

    def e_record(self, row):
        E = orc.enforce_essential(orc.estimate_F_DLT(self.x1[:, row[:8]], self.x2[:, row[:8]]))
        R, T, _ = orc.essential_to_RT(E, self.x1, self.x2)
        return R, T, orc.epipolar_mask(E, self.x1, self.x2, self.thr)

    def h_record(self, row):
        H = orc.estimate_H_DLT(self.x1[:, row[8:]], self.x2[:, row[8:]])
        Ra, Ta, Rb, Tb = orc.homography_to_RT(H, self.x1, self.x2)
        out = []
        for R, T in ((Ra, Ta), (Rb, Tb)):
            E = orc.skew_symmetric_mat(T) @ R
            m = orc.epipolar_mask(E, self.x1, self.x2, self.thr)
            Rr, Tr, n = orc.essential_to_RT(E, self.x1[:, m], self.x2[:, m])
            out.append((Rr, Tr, n, m))
        return out[0], out[1]


@pytest.mark.parametrize("tag,chunk0", [("a", 4096), ("a", 7), ("b", 64), ("c", 1), ("c", 1000)])
def test_replay_reproduces_reference_runs(golden, tag, chunk0):
    g = golden("eh_ransac.npz")
    n, inl, pl, seed = g[f"{tag}_scene"]
    sc = synthetic.two_view_scene(n=int(n), inlier_ratio=float(inl), planar_ratio=float(pl), seed=int(seed))
    x1 = orc.normalize_K(sc["K"], sc["x1"])
    x2 = orc.normalize_K(sc["K"], sc["x2"])
    np.random.seed(int(g[f"{tag}_seed"]))
    be = OracleBackend(x1, x2, 1.0 / sc["K"][0][0])
    st = run_replay(int(n), be, chunk0=chunk0)
    assert st.it == len(g[f"{tag}_cH"])
    np.testing.assert_allclose(st.R, g[f"{tag}_R"], atol=1e-9)
    np.testing.assert_allclose(np.reshape(st.T, (3, 1)), g[f"{tag}_T"], atol=1e-9)
    np.testing.assert_array_equal(np.packbits(st.inliers), g[f"{tag}_inliers"])
    np.testing.assert_array_equal(np.random.randint(0, 2**31 - 1, size=4), g[f"{tag}_next_draws"])
    assert any(k == "E" for _, k, _ in st.events) and any(k == "H" for _, k, _ in st.events)


def test_iteration_formula_edge_cases():
    assert np.isinf(ransac_iters(1e-3, 8))  # 1 - eps^8 rounds to 1: the reference divides by log(1) = 0
    assert abs(ransac_iters(0.1, 4) - 29955.8) < 1.0
    assert ransac_iters(1.0, 8) == 0.0  # log(0) = -inf
    assert first_strict_max([0, 0, 0, 0]) == (0, 0)
    assert first_strict_max([3, 5, 5, 1]) == (1, 5)


def test_replay_never_draws_past_a_finite_limit():
    """Once both limits are finite the block size shrinks to what is left."""
    sc = synthetic.two_view_scene(n=300, inlier_ratio=0.9, planar_ratio=0.8, seed=5)
    x1 = orc.normalize_K(sc["K"], sc["x1"])
    x2 = orc.normalize_K(sc["K"], sc["x2"])
    np.random.seed(3)
    ref = {}
    orc.estimate_E_robust(sc["K"], x1, x2, 1.0, trace=ref)
    after_ref = np.random.randint(0, 1000, 3)
    np.random.seed(3)
    be = OracleBackend(x1, x2, 1.0 / sc["K"][0][0])
    st = run_replay(300, be, chunk0=16)
    assert st.it == ref["iterations"]
    np.testing.assert_array_equal(np.random.randint(0, 1000, 3), after_ref)
    assert sum(be.blocks) >= st.it


def test_record_event_candidate_choice_is_the_first_strict_maximum():
    """The host half of a record event: among the four [R|t] candidates the first strict maximum of the chirality counts
    wins and an all-zero vote keeps candidate 0 (src/chirality_check.py:72-91: best count starts at -1)."""
    from sfm_hip import ops

    host = np.zeros(256)
    P = np.arange(48, dtype=np.float64).reshape(4, 3, 4)
    host[18:66] = P.ravel()
    for counts, want in [([5, 9, 9, 2], 1), ([0, 0, 0, 0], 0), ([3, 3, 3, 3], 0), ([1, 2, 3, 4], 3)]:
        host[66:68] = np.array(counts, np.int32).view(np.float64)
        R, T, n = ops._pick_candidate(host, 18, 66)
        assert n == counts[want]
        assert (R == P[want][:, :3]).all() and (T == P[want][:, 3]).all()


class BoundedOracleBackend(OracleBackend):
    """Block scoring driven by bounds, as the GPU backend does for large blocks: upper bounds (exact count + random
    slack) everywhere, exact counts only where a bound exceeds a running best."""

    def __init__(self, *a):
        super().__init__(*a)
        self.rng = np.random.default_rng(0)
        self.refined = 0
        self.bounded_blocks = 0

    def score_bounded(self, idx, best_E, best_H):
        if best_E <= 0 or best_H <= 0:
            return None
        self.bounded_blocks += 1
        self.cur = orc.score_EH_batch(self.x1, self.x2, idx, self.thr)[:2]
        ue = self.cur[0] + self.rng.integers(0, 40, idx.shape[0])
        uh = self.cur[1] + self.rng.integers(0, 40, idx.shape[0])
        exact = (ue > best_E) | (uh > best_H)
        ue[exact] = self.cur[0][exact]
        uh[exact] = self.cur[1][exact]
        return ue, uh, exact

    def refine(self, positions):
        self.refined += len(positions)
        return self.cur[0][positions], self.cur[1][positions]


@pytest.mark.parametrize("tag,chunk0", [("a", 64), ("b", 16), ("c", 5), ("c", 32)])
def test_bound_driven_replay_reproduces_reference_runs(golden, tag, chunk0):
    g = golden("eh_ransac.npz")
    n, inl, pl, seed = g[f"{tag}_scene"]
    sc = synthetic.two_view_scene(n=int(n), inlier_ratio=float(inl), planar_ratio=float(pl), seed=int(seed))
    x1 = orc.normalize_K(sc["K"], sc["x1"])
    x2 = orc.normalize_K(sc["K"], sc["x2"])
    np.random.seed(int(g[f"{tag}_seed"]))
    be = BoundedOracleBackend(x1, x2, 1.0 / sc["K"][0][0])
    st = run_replay(int(n), be, chunk0=chunk0)
    assert be.bounded_blocks > 0
    assert st.it == len(g[f"{tag}_cH"])
    np.testing.assert_allclose(st.R, g[f"{tag}_R"], atol=1e-9)
    np.testing.assert_allclose(np.reshape(st.T, (3, 1)), g[f"{tag}_T"], atol=1e-9)
    np.testing.assert_array_equal(np.packbits(st.inliers), g[f"{tag}_inliers"])
    np.testing.assert_array_equal(np.random.randint(0, 2**31 - 1, size=4), g[f"{tag}_next_draws"])


class AsyncOracleBackend(BoundedOracleBackend):
    """The submit / collect form the GPU backend offers: run_replay draws the next block while this one is 'in flight'
    and must still leave the stream exactly where the reference leaves it."""

    def submit(self, idx, best_E, best_H):
        self.submitted = getattr(self, "submitted", 0) + 1
        return (idx, best_E > 0 and best_H > 0)

    def discard(self, handle):
        self.discarded = getattr(self, "discarded", 0) + 1

    def collect(self, handle, best_E, best_H):
        idx, bounded = handle
        self.collected = getattr(self, "collected", 0) + 1
        if bounded:
            return self.score_bounded(idx, best_E, best_H)
        return self.score(idx) + (None,)


@pytest.mark.parametrize("tag,chunk0,chunk_max", [("a", 64, 1 << 18), ("b", 16, 64), ("c", 5, 1 << 18), ("c", 32, 32), ("a", 100000, 100000)])
def test_draw_ahead_replay_reproduces_reference_runs(golden, tag, chunk0, chunk_max):
    g = golden("eh_ransac.npz")
    n, inl, pl, seed = g[f"{tag}_scene"]
    sc = synthetic.two_view_scene(n=int(n), inlier_ratio=float(inl), planar_ratio=float(pl), seed=int(seed))
    x1 = orc.normalize_K(sc["K"], sc["x1"])
    x2 = orc.normalize_K(sc["K"], sc["x2"])
    np.random.seed(int(g[f"{tag}_seed"]))
    be = AsyncOracleBackend(x1, x2, 1.0 / sc["K"][0][0])
    st = run_replay(int(n), be, chunk0=chunk0, chunk_max=chunk_max, ahead_min_pairs=0)
    assert be.submitted > 0
    assert be.submitted == be.collected + getattr(be, "discarded", 0)  # every block in flight is collected or waited for
    assert st.it == len(g[f"{tag}_cH"])
    np.testing.assert_allclose(st.R, g[f"{tag}_R"], atol=1e-9)
    np.testing.assert_allclose(np.reshape(st.T, (3, 1)), g[f"{tag}_T"], atol=1e-9)
    np.testing.assert_array_equal(np.packbits(st.inliers), g[f"{tag}_inliers"])
    np.testing.assert_array_equal(np.random.randint(0, 2**31 - 1, size=4), g[f"{tag}_next_draws"])


def test_draw_ahead_replay_honours_an_iteration_cap():
    sc = synthetic.two_view_scene(n=400, inlier_ratio=0.2, planar_ratio=0.5, seed=2)
    x1 = orc.normalize_K(sc["K"], sc["x1"])
    x2 = orc.normalize_K(sc["K"], sc["x2"])
    for cap in (1, 63, 64, 65, 200):
        np.random.seed(9)
        st = run_replay(400, AsyncOracleBackend(x1, x2, 1.0 / sc["K"][0][0]), chunk0=64, max_iterations=cap, ahead_min_pairs=0)
        after = np.random.randint(0, 1 << 30, 2)
        np.random.seed(9)
        np.random.randint(0, 400, size=12 * st.it)
        np.testing.assert_array_equal(after, np.random.randint(0, 1 << 30, 2))
        assert st.it <= cap


def test_bound_driven_replay_refines_after_the_best_E_count_was_lowered():
    """An accepted H record may LOWER the best E count (the reference compares with the current iteration's E count);
    bounds that were filtered against the old best must then be replaced by exact counts before the loop reads them."""
    from sfm_hip.ransac_algorithm import ReplayState, replay_block

    class Scripted:
        def __init__(self, true_e, true_h):
            self.true_e, self.true_h, self.refined = true_e, true_h, []

        def e_record(self, row):
            return "R_E", "T_E", ("mask", int(row[0]))

        def h_record(self, row):
            return ("Ra", "Ta", 30, "ma"), ("Rb", "Tb", 10, "mb")  # solution a with 30 inliers wins if 30 > nE of that iteration

        def refine(self, pos):
            self.refined.extend(int(p) for p in pos)
            return self.true_e[pos], self.true_h[pos]

    def run(bounded):
        st = ReplayState(1000)
        st.best_E, st.best_H = 100, 50
        st.E_iters = st.H_iters = 1e9
        true_e = np.array([5, 3, 20, 45, 60, 7, 31, 2])
        true_h = np.array([1, 60, 2, 3, 4, 5, 6, 7])  # position 1: H record; its E count 3 < 30: accepted, best_E drops to 30
        idx = np.arange(8 * 12).reshape(8, 12)
        be = Scripted(true_e, true_h)
        if bounded:
            ue = true_e + np.array([0, 0, 5, 40, 3, 10, 0, 90])  # bounds, all at or below the old best 100
            exact = np.array([False, True, False, False, False, False, False, False])
            used = replay_block(st, idx, ue.copy(), true_h.copy(), be, exact)
        else:
            used = replay_block(st, idx, true_e.copy(), true_h.copy(), be)
        return st, used, be

    a, ua, _ = run(False)
    b, ub, be = run(True)
    assert (ua, a.it, a.best_E, a.best_H, a.events) == (ub, b.it, b.best_E, b.best_H, b.events)
    assert a.best_E == 60 and [e[1] for e in a.events] == ["H", "E", "E"]
    assert be.refined  # bounds above 30 had to be made exact
    assert a.pending_E is not None and (a.pending_E == b.pending_E).all()

"""Thirty seeded run_ransac runs of the REFERENCE (tests/golden/eh_ransac_many.npz, made by tests/golden/make_golden.py:
300, 500 and 1500 matches, ten seeds each, full count traces) against the GPU path.

Samples are drawn with replacement (src/ransac_algorithm.py:94,116), so a few percent of them repeat an index; such a
system is rank deficient and its null vector is whatever LAPACK (there) or the in-register solver (here) happens to return,
so its counts and models are not comparable.  The test therefore
  * compares the counts of EVERY iteration of every run that drew no repeated index with the reference's trace (exactly),
  * counts the repeated-index iterations, the count differences on them and the record events that fall on one,
  * asserts that every run without such a record event reproduces the reference's rotation, translation, inlier mask,
    iteration count and position of the global sample stream -- and, as measured on this fixture, that all thirty do
    (seeded_runs_report, also used by profiles/tools/seeded_runs_report.py)."""
import os

import numpy as np
import pytest

from sfm_hip import synthetic

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def seeded_runs_report(fixture="eh_ransac_many.npz"):
    from oracle import sfm_oracle as orc
    from sfm_hip import _capi as C
    from sfm_hip import ops
    from sfm_hip import ransac_algorithm as ra

    C.load_library()
    g = np.load(os.path.join(HERE, "golden", fixture), allow_pickle=False)
    rows = []
    for r, row in enumerate(g["runs"]):
        n, inl, pl, sseed, rseed, iters = row[:6]
        thr_px = float(row[6]) if len(row) > 6 else 1.0  # a seventh column holds the pixel threshold of the run
        n, sseed, rseed, iters = int(n), int(sseed), int(rseed), int(iters)
        tag = f"r{r:02d}"
        sc = synthetic.two_view_scene(n=n, inlier_ratio=float(inl), planar_ratio=float(pl), seed=sseed)
        x1, x2 = orc.normalize_K(sc["K"], sc["x1"]), orc.normalize_K(sc["K"], sc["x2"])
        thr = thr_px / sc["K"][0][0]
        np.random.seed(rseed)
        idx = np.random.randint(0, n, size=12 * iters).reshape(iters, 12)
        repE = (np.diff(np.sort(idx[:, :8], axis=1), axis=1) == 0).any(axis=1)  # a sample that holds an index twice
        repH = (np.diff(np.sort(idx[:, 8:], axis=1), axis=1) == 0).any(axis=1)
        ce, ch = ops.eh_score(C.dev(ops.pack_corr(x1, x2)), C.dev(idx, np.int32), thr, 3 * thr)
        ce, ch = ce.cpu().numpy(), ch.cpu().numpy()
        cE, cH = g[f"{tag}_cE"].astype(np.int64), g[f"{tag}_cH"].astype(np.int64)
        clean_mismatch = int((ce[~repE] != cE[~repE]).sum() + (ch[~repH] != cH[~repH]).sum())
        rep_mismatch = int((ce[repE] != cE[repE]).sum() + (ch[repH] != cH[repH]).sum())
        # the drop-in run itself
        np.random.seed(rseed)
        st = ra.run_replay(n, ra.GpuBackend(x1, x2, thr))
        nxt = np.random.randint(0, 2**31 - 1, size=4)
        ev_on_rep = sum(1 for it, kind, _ in st.events if it < iters and (repE[it] if kind == "E" else repH[it]))
        same = bool(st.it == iters and np.array_equal(nxt, g[f"{tag}_next_draws"]) and st.R is not None
                    and np.allclose(st.R, g[f"{tag}_R"], atol=1e-9) and np.allclose(np.reshape(st.T, (3, 1)), g[f"{tag}_T"], atol=1e-9)
                    and np.array_equal(np.packbits(st.inliers.cpu().numpy().astype(bool)), g[f"{tag}_inliers"]))
        rows.append({"run": tag, "matches": n, "iterations": iters, "iterations_with_repeated_index": int((repE | repH).sum()),
                     "count_mismatches_without_repeated_index": clean_mismatch, "count_differences_on_repeated_index": rep_mismatch,
                     "record_events_on_repeated_index": int(ev_on_rep), "flagged": bool(rep_mismatch or ev_on_rep), "identical_to_reference": same})
    return rows


def test_thirty_seeded_runs_against_the_reference():
    rows = seeded_runs_report()
    assert len(rows) == 30
    checked = sum(r["iterations"] for r in rows)
    assert checked > 100000
    assert sum(r["count_mismatches_without_repeated_index"] for r in rows) == 0  # every comparable iteration of every run
    same = sum(r["identical_to_reference"] for r in rows)
    on_rep = sum(r["record_events_on_repeated_index"] > 0 for r in rows)
    print(f"seeded runs identical to the reference: {same} of 30; iterations compared: {checked}, "
          f"{sum(r['iterations_with_repeated_index'] for r in rows)} of them with a repeated index "
          f"({sum(r['count_differences_on_repeated_index'] for r in rows)} count differences there), runs with a record event on one: {on_rep}")
    # a run without a record event on a repeated-index sample must reproduce the reference; on this fixture all thirty do
    assert all(r["identical_to_reference"] for r in rows if r["record_events_on_repeated_index"] == 0)
    assert same == 30


def test_six_long_seeded_runs_on_thousands_of_matches_against_the_reference():
    """tests/golden/eh_ransac_large.npz: six seeded run_ransac runs of the reference on 4 000 and 8 000 matches at 30-50 %
    inliers, 5 900 to 161 000 iterations each: the drop-in loop goes through its doubling blocks up to the largest, draws
    ahead while the GPU scores (blocks of 2^24 pairs and more) and rewinds the global stream when the loop ends inside a
    block; every count comes from the exact scorer with its early loops."""
    rows = seeded_runs_report("eh_ransac_large.npz")
    assert len(rows) == 6
    checked = sum(r["iterations"] for r in rows)
    assert checked > 300000
    assert sum(r["count_mismatches_without_repeated_index"] for r in rows) == 0
    same = sum(r["identical_to_reference"] for r in rows)
    print(f"long seeded runs identical to the reference: {same} of 6; iterations compared: {checked}, "
          f"{sum(r['iterations_with_repeated_index'] for r in rows)} of them with a repeated index, runs with a record event on one: "
          f"{sum(r['record_events_on_repeated_index'] > 0 for r in rows)}")
    assert all(r["identical_to_reference"] for r in rows if r["record_events_on_repeated_index"] == 0)


def test_seeded_runs_at_other_pixel_thresholds_against_the_reference():
    """tests/golden/eh_ransac_thresholds.npz: six seeded run_ransac runs of the reference at thresholds of 0.25, 2.5 and 8 pixels."""
    rows = seeded_runs_report("eh_ransac_thresholds.npz")
    assert len(rows) == 6
    assert sum(r["count_mismatches_without_repeated_index"] for r in rows) == 0
    same = sum(r["identical_to_reference"] for r in rows)
    print(f"runs at other thresholds identical to the reference: {same} of 6; iterations compared: {sum(r['iterations'] for r in rows)}, "
          f"runs with a record event on a repeated-index sample: {sum(r['record_events_on_repeated_index'] > 0 for r in rows)}")
    assert all(r["identical_to_reference"] for r in rows if r["record_events_on_repeated_index"] == 0)

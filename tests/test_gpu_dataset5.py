"""The geometry of the reference's dataset 5 (ten 1936 x 1296 photographs, data/5/cfg.yml) on REAL-image correspondences:
tests/golden/dataset5_geometry.npz holds the matches that this repository's CPU restatement of the front-end
(oracle/sift.py + oracle/matching.py, run by tests/golden/make_dataset5_geometry.py in the build container) derived from
the images, and the oracle's seeded geometry on them for six seeds.

Real matches contain identical correspondences under different indices (a keypoint with several orientations is matched
once per orientation: about 9 % of the matches here), so a minimal sample holds the same point twice far more often than
through a repeated index alone.  Such a DLT system is rank deficient and its null vector is LAPACK's arbitrary choice in
the reference and the in-register solver's here: a record event fired by such a sample is not comparable.  The test
therefore asserts, per seed, the FULL outcome (rotations, initial cloud, inlier mask, both sets of translations, the
iteration count of every loop through the position of the global sample stream) whenever no record event of either side
came from a degenerate sample, requires that at least two seeds are of that kind, and reports the rest.
Parity with cv2's own matches is unpinned (OpenCV is not installed where the fixture was made)."""
import os

import numpy as np
import pytest

from oracle import sfm_oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def load_fixture():
    import importlib.util

    spec = importlib.util.spec_from_file_location("dataset5_fixture", os.path.join(HERE, "golden", "dataset5_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.load_fixture()


def degenerate(x_a, x_b, rows):
    pts = np.vstack((x_a[:2, rows], x_b[:2, rows])).T
    return len({tuple(p) for p in pts}) < len(rows)


def test_fixture_is_consistent_with_the_oracle_on_the_first_pair():
    """Cheap, on the CPU: the oracle's loop on pair 0 stops after the stored number of iterations with the stored rotation."""
    g, K, x_pairs, init, paths, matcher = load_fixture()
    np.random.seed(int(g["s0_seed"]))
    tr = {}
    R, _, inl = orc.run_ransac(K, x_pairs[0], x_pairs[1], 1.0, trace=tr)
    assert tr["iterations"] == int(g["s0_iterations"][0])
    np.testing.assert_allclose(R, g["s0_R_rel"][0], atol=1e-12)


def dataset5_report():
    from sfm_hip import _capi, pipeline, utils
    from sfm_hip import ransac_algorithm as ra

    _capi.load_library()
    g, K, x_pairs, init, paths, matcher = load_fixture()
    rows = []
    orig_fc, orig_replay = utils.find_correspondences, ra.run_replay
    try:
        for j in range(int(g["n_seeds"])):
            tag = f"s{j}_"
            states = []

            def spy(n, backend, *a, **k):
                st = orig_replay(n, backend, *a, **k)
                states.append(st)
                return st

            ra.run_replay = spy
            utils.find_correspondences = matcher(tag)
            np.random.seed(int(g[tag + "seed"]))
            got = pipeline.run_geometry(K, x_pairs, init, paths, 1.0)
            stream_ok = int(np.random.randint(0, 1 << 30)) == int(g[tag + "next_draw"])
            assert len(states) == 10  # nine consecutive pairs, then the initial pair
            gpu_deg = 0
            for p, st in enumerate(states):
                xa, xb = (x_pairs[2 * p], x_pairs[2 * p + 1]) if p < 9 else init["x_init"]
                for (it, kind, _), row in zip(st.events, st.event_rows):
                    gpu_deg += bool(degenerate(xa, xb, row[:8] if kind == "E" else row[8:]))
            its = [int(st.it) for st in states]
            same = bool(stream_ok and its == [int(v) for v in g[tag + "iterations"]]
                        and all(np.allclose(a, b, atol=1e-9) for a, b in zip(got["absolute_rotations"], g[tag + "abs_rot"]))
                        and np.array_equal(got["inliers"], g[tag + "inliers"]) and np.allclose(got["X0"], g[tag + "X0"], rtol=1e-6, atol=1e-7)
                        and all(np.allclose(np.reshape(a, 3), b, rtol=1e-7, atol=1e-9) for a, b in zip(got["estimated_Ts"], g[tag + "est_T"]))
                        and all(np.allclose(np.reshape(a, 3), b, rtol=1e-7, atol=1e-9) for a, b in zip(got["refined_Ts"], g[tag + "ref_T"])))
            first_diff = next((p for p, (a, b) in enumerate(zip(its, g[tag + "iterations"])) if a != int(b)), None)
            rows.append({"seed": int(g[tag + "seed"]), "identical_to_oracle": same, "iterations_gpu": its,
                         "iterations_oracle": [int(v) for v in g[tag + "iterations"]], "first_loop_that_differs": first_diff,
                         "degenerate_record_events_oracle": int(g[tag + "degenerate_record_events"]), "degenerate_record_events_gpu": int(gpu_deg),
                         "initial_pair_inliers": int(np.sum(got["inliers"]))})
    finally:
        utils.find_correspondences, ra.run_replay = orig_fc, orig_replay
    return rows


def test_dataset5_geometry_on_real_image_correspondences():
    rows = dataset5_report()
    clean = [r for r in rows if r["degenerate_record_events_oracle"] == 0 and r["degenerate_record_events_gpu"] == 0]
    print("dataset-5 seeds:", [(r["seed"], r["identical_to_oracle"], r["degenerate_record_events_oracle"], r["degenerate_record_events_gpu"]) for r in rows])
    assert all(r["identical_to_oracle"] for r in clean), [r for r in clean if not r["identical_to_oracle"]]
    assert len(clean) >= 2
    # and a seed that differs does so from a loop on whose record events a degenerate sample sits, never before
    for r in rows:
        if not r["identical_to_oracle"]:
            assert r["degenerate_record_events_oracle"] + r["degenerate_record_events_gpu"] > 0

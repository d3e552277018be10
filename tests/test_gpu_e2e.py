"""End-to-end geometry (the stage sequence of run_sfm.py:59-80) on a synthetic multi-camera
scene, GPU package vs the oracle with the same seed: every rotation, the initial point cloud,
the inlier mask and every translation must agree, and the global index stream must end at the
same position.  The OpenCV front-end is replaced by a synthetic matcher on both sides."""
import numpy as np
import pytest

from oracle import sfm_oracle as orc
from sfm_hip import synthetic

pytestmark = pytest.mark.gpu


def test_pipeline_matches_oracle_end_to_end(monkeypatch):
    from sfm_hip import _capi, pipeline, utils

    _capi.load_library()
    sc = synthetic.multi_view_scene(n_cams=5, n_points=900, seed=4)
    monkeypatch.setattr(utils, "find_correspondences", sc["find_correspondences"])
    np.random.seed(2024)
    got = pipeline.run_geometry(sc["K"], sc["x_pairs"], sc["init_pair_dict"], sc["img_paths"], 1.0)
    after_gpu = np.random.randint(0, 2**31 - 1, 4)
    np.random.seed(2024)
    ref = orc.run_geometry(sc["K"], sc["x_pairs"], sc["init_pair_dict"], sc["img_paths"], 1.0, sc["find_correspondences"])
    after_ref = np.random.randint(0, 2**31 - 1, 4)
    np.testing.assert_array_equal(after_gpu, after_ref)
    assert got["nr_images"] == 5 and len(got["absolute_rotations"]) == 5
    for a, b in zip(got["absolute_rotations"], ref["absolute_rotations"]):
        np.testing.assert_allclose(a, b, atol=1e-9)
    np.testing.assert_array_equal(got["inliers"], ref["inliers"])
    np.testing.assert_allclose(got["X0"], ref["X0"], rtol=1e-6, atol=1e-7)  # stated bar: 1e-5 relative
    for a, b in zip(got["estimated_Ts"], ref["estimated_Ts"]):
        assert (a is None) == (b is None)
        if a is not None:
            np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-9)
    for a, b in zip(got["refined_Ts"], ref["refined_Ts"]):
        assert a.shape == (3, 1)
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-9)
    # sanity against ground truth: relative rotations are recovered
    for c in range(1, 5):
        Rrel_true = sc["R_true"][c] @ sc["R_true"][c - 1].T
        Rrel = got["absolute_rotations"][c - 1].T @ got["absolute_rotations"][c]
        ang = np.arccos(np.clip((np.trace(Rrel_true.T @ Rrel) - 1) / 2, -1, 1))
        assert ang < 0.02


@pytest.mark.parametrize("fixture", ["e2e_reference.npz", "e2e_reference_b.npz"])
def test_pipeline_matches_the_reference_run_end_to_end(monkeypatch, golden, check_e2e_against_reference, fixture):
    """tests/golden/e2e_reference*.npz hold what the REFERENCE's own stage sequence (estimate_R, run_reconstruction,
    estimate_translation, refine_translation; its OpenCV matcher replaced by the scene's stand-in) returns for one seed on
    a synthetic five-camera and on a seven-camera scene: the GPU package must return the same, down to the position of the
    global stream."""
    from sfm_hip import _capi, pipeline, utils

    _capi.load_library()
    g = golden(fixture)
    n_cams, n_points, seed = (int(v) for v in g["scene"])
    sc = synthetic.multi_view_scene(n_cams=n_cams, n_points=n_points, init_pair=tuple(int(v) for v in g["init_pair"]), seed=seed)
    monkeypatch.setattr(utils, "find_correspondences", sc["find_correspondences"])
    np.random.seed(int(g["seed"]))
    got = pipeline.run_geometry(sc["K"], sc["x_pairs"], sc["init_pair_dict"], sc["img_paths"], 1.0)
    check_e2e_against_reference(g, got, np.random.randint(0, 2**31 - 1, size=4))


def test_cli_driver_writes_reference_compatible_results(tmp_path, monkeypatch):
    """run_sfm.py counterpart on a synthetic dataset directory: cfg.yml + the reference's two cache
    files (so no OpenCV is needed), result pickle with the reference's keys, and the visualiser's
    point cloud against the oracle."""
    import json
    import pickle
    import sys as _sys

    from sfm_hip import _capi, pipeline, utils
    from sfm_hip.utils import save_x_pairs

    _capi.load_library()
    sc = synthetic.multi_view_scene(n_cams=4, n_points=700, init_pair=(1, 2), seed=9)
    d = tmp_path / "data" / "syn"
    d.mkdir(parents=True)
    names = [f"img{c}.png" for c in range(4)]
    (d / "cfg.yml").write_text(json.dumps({"image_file_names": names, "camera": {"focal_length": [2489.14, 2489.14], "principal_point": [968, 648]},
                                           "initial_pair": [1, 2]}))
    save_x_pairs(sc["x_pairs"], "x_pairs_dataset_syn.pkl", str(d))
    save_x_pairs(sc["init_pair_dict"], "init_pair_dataset_syn.pkl", str(d))
    obs_by_name = {str(d / n): sc["img_paths"][c] for c, n in enumerate(names)}
    matcher = lambda p, desc, X0, K_: sc["find_correspondences"](obs_by_name[p], desc, X0, K_)
    monkeypatch.setattr(utils, "find_correspondences", matcher)
    monkeypatch.chdir(tmp_path)
    _sys.path.insert(0, str(__import__("pathlib").Path(pipeline.__file__).resolve().parents[1]))
    import run_sfm as cli

    np.random.seed(31)
    cli.run_sfm(["--data_path", str(tmp_path / "data"), "--dataset", "syn"])
    with open(tmp_path / "results" / "dataset_syn" / "sfm_results.pkl", "rb") as f:
        res = pickle.load(f)  # written by this test run
    assert sorted(res) == ["K", "absolute_rotations", "nr_images", "refined_Ts", "x_pairs"]
    assert res["nr_images"] == 4 and len(res["refined_Ts"]) == 4 and res["refined_Ts"][0].shape == (3, 1)
    np.random.seed(31)
    ref = orc.run_geometry(res["K"], sc["x_pairs"], sc["init_pair_dict"], [str(d / n) for n in names], 1.0, matcher)
    for a, b in zip(res["refined_Ts"], ref["refined_Ts"]):
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-9)
    clouds = pipeline.triangulate_point_cloud(res)
    Kn = np.asarray(res["K"], float)
    for i, X in enumerate(clouds):
        P1 = Kn @ np.hstack((ref["absolute_rotations"][i], ref["refined_Ts"][i]))
        P2 = Kn @ np.hstack((ref["absolute_rotations"][i + 1], ref["refined_Ts"][i + 1]))
        Xr = orc.filter_3D_points(orc.triangulate_3D_point_DLT([P1, P2], [sc["x_pairs"][2 * i], sc["x_pairs"][2 * i + 1]]))
        assert X.shape == Xr.shape
        np.testing.assert_allclose(X, Xr, rtol=1e-5, atol=1e-6)  # stated bar for 3D points: 1e-5 relative
    # the visualiser's command line reads the same file and writes the animation (the reference's 120 frames cut to 3 here)
    from PIL import Image

    import visualization as vis
    from sfm_hip import render

    monkeypatch.setattr(render, "FRAME_STEP_DEGREES", 120)
    gif = vis.visualize_results(["--dataset", "syn"])
    assert gif == "output/rotation_dataset_syn.gif"
    with Image.open(tmp_path / gif) as im:
        assert im.n_frames == 3

"""The whole pipeline from image files: rendered views of a textured two-plane scene are written as PNG files with a
cfg.yml in the reference's dataset layout, and the command-line driver (run_sfm.py counterpart) reads them, detects and
matches SIFT features on the GPU, and reconstructs.  Ground truth is the renderer's: relative rotations are recovered
and the triangulated cloud lies where the scene is."""
import os
import pickle
import sys

import numpy as np
import pytest

from sfm_hip import synthetic

pytestmark = pytest.mark.gpu


def test_cli_reconstructs_a_rendered_scene_from_image_files(tmp_path, monkeypatch):
    from sfm_hip import _capi, matching, pipeline

    _capi.load_library()
    views = synthetic.render_views(n_cams=5, width=726, height=486, seed=3, texture_size=1536, n_blobs=3000)
    paths = synthetic.write_image_dataset(str(tmp_path / "data" / "syn"), views, init_pair=(1, 3))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(matching, "_features", {})
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(pipeline.__file__))))
    import run_sfm as cli

    np.random.seed(5)
    res = cli.run_sfm(["--data_path", str(tmp_path / "data"), "--dataset", "syn"])
    # the front-end ran on the GPU: every image was detected once, each pair has plenty of matches at ratio 0.4
    assert len(matching._features) >= 5
    for i in range(4):
        assert res["x_pairs"][2 * i].shape[0] == 3 and res["x_pairs"][2 * i].shape[1] > 150
    assert os.path.exists(tmp_path / "data" / "syn" / "x_pairs_dataset_syn.pkl")  # the reference's cache files
    with open(tmp_path / "results" / "dataset_syn" / "sfm_results.pkl", "rb") as f:
        saved = pickle.load(f)  # written by this test run
    assert sorted(saved) == ["K", "absolute_rotations", "nr_images", "refined_Ts", "x_pairs"]
    R = res["absolute_rotations"]
    for c in range(1, 5):
        Rrel_true = views["R_true"][c] @ views["R_true"][c - 1].T
        Rrel = R[c] @ R[c - 1].T
        ang = np.degrees(np.arccos(np.clip((np.trace(Rrel_true.T @ Rrel) - 1) / 2, -1, 1)))
        assert ang < 2.0, (c, ang)  # an 8-point model from the best MINIMAL sample (never refitted), 1 px threshold at f = 933 px
    clouds = pipeline.triangulate_point_cloud(res)
    assert len(clouds) == 4
    # scale and frame are the reconstruction's own (first camera of the initial pair is the origin, baseline 1), so look
    # at a scale-free property: the cloud of every pair is in front of both of its cameras
    for i, X in enumerate(clouds):
        for c in (i, i + 1):
            depth = (R[c] @ X[:3] + np.reshape(res["refined_Ts"][c], (3, 1)))[2]
            assert (depth > 0).mean() > 0.9

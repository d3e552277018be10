"""The geometry of the reference's dataset 5 (ten 1936 x 1296 photographs, data/5/cfg.yml) on REAL-image correspondences:
tests/golden/dataset5_geometry.npz holds the matches that this repository's CPU restatement of the front-end
(oracle/sift.py + oracle/matching.py, run by tests/golden/make_dataset5_geometry.py in the build container) derived from
the images, and the oracle's seeded geometry on them.  The GPU pipeline must reproduce it: rotations, the initial cloud,
the inlier mask, both sets of translations, the iteration counts (through the position of the global sample stream).
Parity with cv2's own matches is unpinned (OpenCV is not installed where the fixture was made)."""
import os

import numpy as np
import pytest

from oracle import sfm_oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def load_fixture():
    g = np.load(os.path.join(HERE, "golden", "dataset5_geometry.npz"), allow_pickle=False)
    K = [[float(v) for v in row] for row in g["K"]]
    x_pairs = [g[f"x_pair_{k}"] for k in range(18)]
    desc = g["init_pair_desc"].astype(np.float32)
    init = {"init_pair_indices": [int(v) for v in g["init_pair"]], "x_init": [g["x_init_0"], g["x_init_1"]], "init_pair_desc": desc}
    paths = [f"image_{i}" for i in range(10)]
    n0 = desc.shape[1]
    Kinv = np.linalg.inv(np.asarray(K, np.float64))

    def find_correspondences(img_path, desc_X, X0, Kmat):
        """The stored matches of find_correspondences (src/utils.py:65-93) for this image: against all descriptors of the
        initial pair, or against those of its inliers (the caller is told apart by the size of desc_X)."""
        i = paths.index(img_path)
        kind = "full" if np.asarray(desc_X).shape[1] == n0 else "inl"
        xy, ti = g[f"corr_{kind}_{i}_xy"], g[f"corr_{kind}_{i}_ti"]
        x = np.float32(xy).T.reshape(2, -1)
        return np.float32(np.asarray(X0)[:, ti]), Kinv @ np.vstack((x, np.ones((1, x.shape[1]))))

    return g, K, x_pairs, init, paths, find_correspondences


def test_fixture_is_consistent_with_the_oracle_on_the_first_pair():
    """CPU-cheap part, repeated here so that the GPU run starts from a checked fixture: the oracle's loop on pair 0 stops
    after the stored number of iterations with the stored rotation."""
    g, K, x_pairs, init, paths, fc = load_fixture()
    np.random.seed(int(g["seed"]))
    tr = {}
    R, _, inl = orc.run_ransac(K, x_pairs[0], x_pairs[1], 1.0, trace=tr)
    assert tr["iterations"] == int(g["iterations"][0])
    np.testing.assert_allclose(R, g["R_rel"][0], atol=1e-12)


def test_dataset5_geometry_on_real_image_correspondences(monkeypatch):
    from sfm_hip import _capi, pipeline, utils

    _capi.load_library()
    g, K, x_pairs, init, paths, fc = load_fixture()
    assert min(x.shape[1] for x in x_pairs) >= 200 and max(x.shape[1] for x in x_pairs) > 5000
    monkeypatch.setattr(utils, "find_correspondences", fc)
    np.random.seed(int(g["seed"]))
    got = pipeline.run_geometry(K, x_pairs, init, paths, 1.0)
    assert int(np.random.randint(0, 1 << 30)) == int(g["next_draw"])  # every loop consumed exactly the reference's share of the stream
    for a, b in zip(got["absolute_rotations"], g["abs_rot"]):
        np.testing.assert_allclose(a, b, atol=1e-9)
    np.testing.assert_array_equal(got["inliers"], g["inliers"])
    np.testing.assert_allclose(got["X0"], g["X0"], rtol=1e-6, atol=1e-7)  # stated bar: 1e-5 relative
    for a, b in zip(got["estimated_Ts"], g["est_T"]):
        np.testing.assert_allclose(np.reshape(a, 3), b, rtol=1e-7, atol=1e-9)
    for a, b in zip(got["refined_Ts"], g["ref_T"]):
        assert a.shape == (3, 1)
        np.testing.assert_allclose(np.reshape(a, 3), b, rtol=1e-7, atol=1e-9)

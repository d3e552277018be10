"""Host side of the rotating point-cloud animation (the reference's utils.py:78-179): camera centres against the
null-space restatement, and a short GIF written from given clouds (the GPU triangulation is checked elsewhere)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _scene(n_img=4, n_pts=300, seed=0):
    from sfm_hip import synthetic

    rng = np.random.default_rng(seed)
    K = [list(r) for r in synthetic.K_DATASET5]
    Rs, Ts = [], []
    for i in range(n_img):
        a = 0.05 * i
        Rs.append(np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]))
        Ts.append(np.array([[0.3 * i], [0.02 * i], [0.01 * i]]))
    X = np.vstack([rng.uniform(-2, 2, (2, n_pts)), rng.uniform(4, 8, (1, n_pts)), np.ones((1, n_pts))])
    x_pairs = []
    for i in range(n_img - 1):
        for j in (i, i + 1):
            x = np.asarray(K) @ (Rs[j] @ X[:3] + Ts[j])
            x_pairs.append(x / x[2])
    data = {"K": K, "absolute_rotations": Rs, "refined_Ts": Ts, "x_pairs": x_pairs, "nr_images": n_img}
    return data, [X.copy() for _ in range(n_img - 1)]


def test_camera_centres_equal_null_space_restatement():
    from oracle import sfm_oracle as orc
    from sfm_hip import render

    data, _ = _scene()
    P = render.pixel_cameras(data)
    c, v = render.camera_centres_and_axes(P)
    c0, v0 = orc.camera_centres_and_axes(P)
    np.testing.assert_allclose(c, c0, rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(v, v0)
    for i in range(len(P)):  # the centre is what the camera maps to zero
        np.testing.assert_allclose(P[i] @ np.append(c[:, i], 1.0), 0, atol=1e-9)


def test_short_animation_is_written(tmp_path):
    from PIL import Image

    from sfm_hip import render

    data, clouds = _scene()
    path = render.visualize_sfm_results_with_rotation(data, "unit", clouds=clouds, frames=[0, 90, 180, 270], out_dir=str(tmp_path))
    assert path == os.path.join(str(tmp_path), "rotation_dataset_unit.gif")
    with Image.open(path) as im:
        assert im.n_frames == 4
        assert im.size == (10 * render.GIF_DPI, 8 * render.GIF_DPI)
        assert im.info.get("duration") == render.FRAME_INTERVAL_MS
    assert len(np.arange(0, 360, render.FRAME_STEP_DEGREES)) == 120  # the reference's frame count


CODE = r'''
import sys, types
sys.modules["cv2"] = types.ModuleType("cv2")
sys.path[:0] = [PKG, REF, TESTS]
import matplotlib
matplotlib.use("Agg")
import numpy as np
import utils as ref_utils                     # the reference's top-level utils.py, unmodified
from sfm_hip import render
from test_render_host import _scene

class Recorder:
    def __init__(self): self.calls = []
    def scatter(self, *a, **k): self.calls.append(("scatter", [np.asarray(x) for x in a], k))
    def quiver(self, *a, **k): self.calls.append(("quiver", [np.asarray(x) for x in a], k))

data, clouds = _scene()
P = render.pixel_cameras(data)
a, b = Recorder(), Recorder()
ref_utils.plot_3d_points_and_cameras(clouds[0], [P[0], P[1]], a, "blue")
render.plot_3d_points_and_cameras(clouds[0], [P[0], P[1]], b, "blue")
assert [c[0] for c in a.calls] == [c[0] for c in b.calls] == ["scatter", "quiver"]
for (_, xa, ka), (_, xb, kb) in zip(a.calls, b.calls):
    assert ka == kb, (ka, kb)
    for u, w in zip(xa, xb):
        np.testing.assert_allclose(u, w, rtol=1e-9, atol=1e-9)
print("OK")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present on this machine")
def test_drawing_calls_equal_the_reference_visualiser():
    code = (CODE.replace("PKG", repr(os.path.join(ROOT, "structure-from-motion_amd"))).replace("REF", repr(REF))
            .replace("TESTS", repr(os.path.join(ROOT, "tests"))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().endswith("OK")

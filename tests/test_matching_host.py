"""Host side of the feature front-end without a GPU: features are detected once per (image, read mode), kept in
memory and, when a cache directory is set, as .npz files; OpenCV is replaced by a stand-in module that counts calls."""
import sys
import types

import numpy as np
import pytest


@pytest.fixture
def fake_cv2(monkeypatch):
    calls = {"detect": 0, "imread": []}
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_GRAYSCALE = 0
    cv2.COLOR_BGR2GRAY = 6

    class KP:
        def __init__(self, x, y):
            self.pt = (x, y)

    class Sift:
        def detectAndCompute(self, gray, mask):
            calls["detect"] += 1
            rng = np.random.default_rng(int(gray.sum()) % 1000)
            n = 17
            return [KP(float(i), float(2 * i) + 0.5) for i in range(n)], rng.integers(0, 256, size=(n, 128)).astype(np.float32)

    def imread(path, flag=1):
        calls["imread"].append((path, flag))
        return np.full((4, 4) if flag == 0 else (4, 4, 3), len(path) % 7, np.uint8)

    cv2.imread = imread
    cv2.cvtColor = lambda im, code: im[..., 0]
    cv2.SIFT_create = lambda: Sift()
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    return calls


def test_features_are_detected_once_per_image_and_mode(fake_cv2, tmp_path, monkeypatch):
    from sfm_hip import matching

    monkeypatch.setenv("SFM_HIP_SIFT", "cv2")  # the OpenCV detector, for users who have it and want the reference's features
    monkeypatch.setattr(matching, "_features", {})
    monkeypatch.setattr(matching, "FEATURE_CACHE_DIR", str(tmp_path / "cache"))
    kp, d = matching.image_features("/data/img_001.jpg", "grayscale")
    assert kp.shape == (17, 2) and kp.dtype == np.float32 and d.shape == (17, 128) and d.dtype == np.uint8
    kp2, d2 = matching.image_features("/data/img_001.jpg", "grayscale")
    assert kp2 is kp and fake_cv2["detect"] == 1  # second call: memory
    matching.image_features("/data/img_001.jpg", "bgr2gray")  # the other read mode is a different image for OpenCV
    assert fake_cv2["detect"] == 2 and fake_cv2["imread"][-1][1] == 1
    # a new process (empty memory) finds the files
    monkeypatch.setattr(matching, "_features", {})
    kp3, d3 = matching.image_features("/data/img_001.jpg", "grayscale")
    assert fake_cv2["detect"] == 2 and (kp3 == kp).all() and (d3 == d).all()
    assert sorted(p.name for p in (tmp_path / "cache").iterdir()) == ["img_001.jpg.bgr2gray.sift.npz", "img_001.jpg.grayscale.sift.npz"]
    with pytest.raises(ValueError):
        matching.image_features("/data/img_002.jpg", "other")


def test_gpu_detector_is_the_default_and_runs_once_per_file(tmp_path, monkeypatch):
    """Without SFM_HIP_SIFT=cv2 the GPU detector runs (stubbed here: no GPU on this machine), and the two read modes of the
    reference share one feature set per file, because only OpenCV's decoders tell them apart."""
    from sfm_hip import matching, sift

    calls = {"read": [], "detect": 0}

    def read_gray(path, mode):
        calls["read"].append((path, mode))
        return np.zeros((8, 8), np.uint8)

    def detect(gray, capacity=0):
        calls["detect"] += 1
        return sift.SiftResult(np.zeros((3, 6), np.float32), np.zeros((3, 128), np.uint8))

    monkeypatch.delenv("SFM_HIP_SIFT", raising=False)
    monkeypatch.setattr(matching, "_features", {})
    monkeypatch.setattr(matching, "FEATURE_CACHE_DIR", None)
    monkeypatch.setattr(sift, "read_gray", read_gray)
    monkeypatch.setattr(sift, "detect_and_compute", detect)
    kp, d = matching.image_features("/data/a.png", "bgr2gray")
    kp2, _ = matching.image_features("/data/a.png", "grayscale")
    assert kp2 is kp and calls["detect"] == 1 and kp.shape == (3, 2) and d.shape == (3, 128)
    matching.image_features("/data/b.png", "grayscale")
    assert calls["detect"] == 2


def test_matcher_refuses_to_run_without_gpu(fake_cv2):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sfm_hip import _capi, matching

    with pytest.raises(_capi.SfmHipError):
        matching.DescriptorSet(np.zeros((4, 128), np.uint8))


def test_read_gray_without_opencv_handles_colour_and_gray_files(tmp_path, monkeypatch):
    """Without cv2 the image files are decoded with Pillow: colour files become ITU-R 601 luma (the weights of
    cv2.COLOR_BGR2GRAY and of IMREAD_GRAYSCALE), gray files are taken as they are, both read modes give the same array."""
    from PIL import Image

    from sfm_hip import sift

    monkeypatch.setitem(sys.modules, "cv2", None)  # importing cv2 raises ImportError
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (20, 30, 3)).astype(np.uint8)
    gray = rng.integers(0, 256, (20, 30)).astype(np.uint8)
    Image.fromarray(rgb).save(tmp_path / "c.png")
    Image.fromarray(gray).save(tmp_path / "g.png")
    Image.fromarray(rgb).save(tmp_path / "c.jpg", quality=95)
    a = sift.read_gray(str(tmp_path / "c.png"), "bgr2gray")
    b = sift.read_gray(str(tmp_path / "c.png"), "grayscale")
    assert a.dtype == np.uint8 and a.shape == (20, 30) and (a == b).all()
    f = rgb.astype(float)
    luma = (f[..., 0] * 299 + f[..., 1] * 587 + f[..., 2] * 114) / 1000.0
    assert np.abs(a.astype(float) - luma).max() <= 1.0
    assert (sift.read_gray(str(tmp_path / "g.png"), "grayscale") == gray).all()
    assert sift.read_gray(str(tmp_path / "c.jpg"), "bgr2gray").shape == (20, 30)
    with pytest.raises(ValueError):
        sift.read_gray(str(tmp_path / "c.png"), "other")

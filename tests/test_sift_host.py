"""CPU checks of the SIFT restatement (oracle/sift.py) and of the host half of the GPU implementation: the restatement
behaves like SIFT on images whose answer is known, and the keypoint ordering / duplicate removal on the host equals the
restatement's."""
import numpy as np

from oracle import sift as osift


def test_isolated_blobs_are_found_at_their_position_and_scale():
    h, w = 128, 160
    yy, xx = np.mgrid[0:h, 0:w]
    blobs = [(40.0, 40.0, 3.0), (110.0, 70.0, 5.0), (60.3, 95.2, 2.2)]
    img = np.full((h, w), 60.0)
    for cx, cy, s in blobs:
        img += 150 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    kps, _ = osift.detect(np.clip(img, 0, 255).astype(np.uint8))
    assert len(kps) >= 3
    for cx, cy, s in blobs:
        near = [k for k in kps if abs(k[0] - cx) < 1.0 and abs(k[1] - cy) < 1.0]
        assert near, (cx, cy)
        # a Gaussian blob of std s is an extremum of the scale-normalised Laplacian at scale s; size = 2 * scale
        best = min(near, key=lambda k: abs(k[2] - 2 * s))
        assert 0.75 < best[2] / (2 * s) < 1.3, (s, best[2])


def test_descriptor_is_normalised_clipped_and_rotation_covariant():
    rng = np.random.default_rng(0)
    h = w = 96
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w))
    for _ in range(40):
        cx, cy, s, a = rng.uniform(10, 86), rng.uniform(10, 86), rng.uniform(1.5, 5), rng.uniform(-1, 1)
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    img = (255 * (img - img.min()) / (img.max() - img.min())).astype(np.uint8)
    kp, desc, full = osift.detect_and_compute(img)
    kp90, desc90, full90 = osift.detect_and_compute(np.ascontiguousarray(np.rot90(img)))  # exact 90 degree rotation
    # the default (not 'precise') upscale is not symmetric about the image centre, so a few border keypoints come and go
    assert len(full) > 10 and abs(len(full) - len(full90)) <= 0.25 * len(full)
    assert desc.dtype == np.uint8 and desc.max() <= 255
    norms = np.linalg.norm(desc.astype(float), axis=1)
    assert np.all(norms > 400) and np.all(norms < 560)  # 512 up to clipping and rounding
    # rot90 maps (x, y) -> (y, w - 1 - x); descriptors of corresponding keypoints stay close.  Reported positions carry the
    # quarter-pixel bias of the default upscale (dst X samples src at X / 2 - 0.25: reported = true + 0.25 on both axes),
    # which the rotation turns into + 0.5 on the flipped axis.
    hits = 0
    for i, k in enumerate(full):
        target = np.array([k[1], w - 1 - k[0] + 0.5])
        d = np.linalg.norm(kp90 - target, axis=1)
        j = int(np.argmin(d))
        if d[j] < 0.3 and abs(full90[j][2] - k[2]) < 0.1 * k[2]:
            dist = np.linalg.norm(desc[i].astype(float) - desc90[j].astype(float))
            if dist < 120:
                hits += 1
    assert hits > 0.6 * len(full)


def test_host_ordering_and_duplicate_removal_equal_the_restatement():
    from sfm_hip import sift

    rng = np.random.default_rng(1)
    n = 400
    raw = np.zeros((n, 6), np.float32)
    raw[:, 0] = rng.integers(0, 12, n) * 0.5
    raw[:, 1] = rng.integers(0, 12, n) * 0.25
    raw[:, 2] = rng.choice([2.0, 3.5, 4.0], n)
    raw[:, 3] = rng.choice([0.0, 90.0, 181.5], n)
    raw[:, 4] = rng.choice([0.01, 0.02, 0.03], n)
    raw[:, 5] = rng.integers(0, 1 << 20, n).astype(np.int32).view(np.float32)
    got = sift._sort_and_dedupe(raw)
    tuples = [(float(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4]), int(r[5:6].view(np.int32)[0])) for r in raw]
    want = osift.remove_duplicates_sorted(tuples)
    assert got.shape[0] == len(want) < n
    for g, t in zip(got, want):
        assert tuple(float(v) for v in g[:5]) == t[:5] and int(g[5:6].view(np.int32)[0]) == t[5]

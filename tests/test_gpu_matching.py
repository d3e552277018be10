"""GPU parity of the descriptor matcher (int8 matrix cores) against the NumPy restatement of
cv2.BFMatcher().knnMatch(k=2) + ratio test.  OpenCV is not installed here and the reference holds no match
fixtures, so parity against cv2 itself is unpinned; for integer-valued descriptors the restated result does not
depend on a summation order, and everything below is compared bit for bit."""
import numpy as np
import pytest

from oracle import matching as omatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    from sfm_hip import _capi, matching

    _capi.load_library()
    return matching


def run(m, dq, dt, ratio):
    from sfm_hip import ops

    q, t = m.DescriptorSet(dq), m.DescriptorSet(dt)
    tidx, d1, d2, good, cnt = ops.knn2_ratio(q.d8, q.norm, t.d8, t.norm, ratio)
    return tidx.cpu().numpy(), d1.cpu().numpy(), d2.cpu().numpy(), good.cpu().numpy().astype(bool), cnt


@pytest.mark.parametrize("shared,only_q,only_t,seed", [(3000, 2000, 1500, 1), (1, 0, 1, 2), (10, 21, 23, 3), (700, 331, 0, 4),
                                                      (5000, 15000, 10000, 5), (31, 1, 33, 6), (64, 64, 0, 7)])
@pytest.mark.parametrize("ratio", [0.4, 0.75])
def test_knn2_ratio_matches_oracle(m, shared, only_q, only_t, seed, ratio):
    dq, dt, truth = omatch.synthetic_descriptors(shared, only_q, only_t, seed=seed)
    want = omatch.knn2_ratio(dq, dt, ratio)
    got = run(m, dq, dt, ratio)
    assert (got[0] == want[0]).all()
    assert (got[1] == want[1]).all() and (got[2] == want[2]).all()  # float32 distances, bit for bit
    assert (got[3] == want[3]).all() and got[4] == int(want[3].sum())
    if shared >= 1000:
        assert want[3].sum() > 0.2 * shared and (want[0][want[3]] == truth[want[3]]).mean() > 0.99


def test_float32_and_uint8_inputs_agree_and_invalid_descriptors_are_rejected(m):
    dq, dt, _ = omatch.synthetic_descriptors(200, 100, 100, seed=9)
    a = run(m, dq, dt, 0.75)
    b = run(m, dq.astype(np.float32), dt.astype(np.float32), 0.75)
    for x, y in zip(a[:4], b[:4]):
        assert (x == y).all()
    bad = dq.astype(np.float32)
    bad[17, 5] += 0.5
    with pytest.raises(ValueError):
        m.DescriptorSet(bad)
    bad = dq.astype(np.float32)
    bad[3, 100] = 256.0
    with pytest.raises(ValueError):
        m.DescriptorSet(bad)
    bad[3, 100] = -1.0
    with pytest.raises(ValueError):
        m.DescriptorSet(bad)
    with pytest.raises(ValueError):
        run(m, dq, dt[:1], 0.75)  # knnMatch with k=2 needs two train descriptors


def test_ties_extremes_and_exact_duplicates(m):
    rng = np.random.default_rng(3)
    dt = rng.integers(0, 256, size=(500, 128), dtype=np.uint8)
    dt[400] = dt[7]  # duplicated train descriptor: the lower index must win and the ratio test must fail
    dt[123] = 0
    dt[321] = 255
    dq = np.concatenate([dt[[7, 400, 50]], np.full((1, 128), 255, np.uint8), np.zeros((1, 128), np.uint8),
                         rng.integers(0, 256, size=(40, 128), dtype=np.uint8)])
    want = omatch.knn2_ratio(dq, dt, 0.75)
    got = run(m, dq, dt, 0.75)
    for x, y in zip(got[:4], want):
        assert (x == y).all()
    assert got[0][0] == 7 and got[0][1] == 7 and got[1][0] == 0.0 and got[2][0] == 0.0 and not got[3][0]
    assert got[0][2] == 50 and got[1][2] == 0.0 and got[3][2]
    assert got[0][3] == 321 and got[0][4] == 123
    # the largest possible distance: all-0 against all-255 descriptors, 128 * 255^2
    far = run(m, np.zeros((3, 128), np.uint8), np.full((2, 128), 255, np.uint8), 0.75)
    assert (far[1] == np.sqrt(np.float32(128 * 255 * 255))).all() and (far[0] == 0).all()


def test_front_end_array_functions(m):
    """The array parts of compute_sift_keypoints (ratio 0.4) and find_correspondences (ratio 0.75) against the same
    steps written with the oracle matcher."""
    rng = np.random.default_rng(5)
    dq, dt, _ = omatch.synthetic_descriptors(1500, 700, 900, seed=11)
    kp1 = rng.uniform(0, 1900, size=(dq.shape[0], 2)).astype(np.float32)
    kp2 = rng.uniform(0, 1900, size=(dt.shape[0], 2)).astype(np.float32)
    x1, x2, desc = m.match_keypoints(kp1, dq.astype(np.float32), kp2, dt.astype(np.float32), ratio=0.4)
    idx, _, _, good = omatch.knn2_ratio(dq, dt, 0.4)
    qi = np.nonzero(good)[0]
    assert x1.shape == (3, qi.size) and x1.dtype == np.float64 and desc.shape == (128, qi.size) and desc.dtype == np.float32
    assert (x1[:2] == kp1[qi].T).all() and (x2[:2] == kp2[idx[qi]].T).all() and (x1[2] == 1).all() and (desc == dq[qi].T).all()

    K = [[2393.95, 0, 932.38], [0, 2398.12, 628.26], [0, 0, 1]]
    X0 = rng.normal(size=(3, dt.shape[0]))
    X, x_norm = m.correspondences_from_features(kp1, dq.astype(np.float32), dt.astype(np.float32).T, X0, K)
    idx, _, _, good = omatch.knn2_ratio(dq, dt, 0.75)
    qi = np.nonzero(good)[0]
    assert X.dtype == np.float32 and (X == np.float32(X0[:, idx[qi]])).all()
    want = np.linalg.inv(K) @ np.vstack((kp1[qi].T, np.ones((1, qi.size))))
    assert x_norm.dtype == np.float64 and (x_norm == want).all()
    # a prepared train set is reused across images
    ds = m.DescriptorSet(dt)
    X2, x2n = m.correspondences_from_features(kp1, dq, ds, X0, K)
    assert (X2 == X).all() and (x2n == x_norm).all()

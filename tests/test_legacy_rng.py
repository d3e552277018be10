"""The in-place sample draws (sfm_hip.legacy_rng, C side sfm_legacy_randint_fill) against np.random.randint itself:
same integers, same generator state afterwards, for ranges around powers of two, across the 624-word regeneration
boundary, from arbitrary starting positions.  CPU only (the function is host code of the library)."""
import numpy as np
import pytest

from sfm_hip import legacy_rng


def _same_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


@pytest.mark.parametrize("seed", [0, 1, 12345])
def test_randint_equals_numpy_and_leaves_the_same_state(seed):
    assert legacy_rng._fast_path_ok(), "the in-place generator is not active: NumPy's MT19937 state was not reachable"
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    np.random.randint(0, 9, size=int(rng.integers(0, 1500)))  # arbitrary position inside a 624-word block
    for n in [1, 2, 3, 4, 5, 255, 256, 257, 4200, 10000, 65535, 65536, 65537, 200000, 2**24 + 1, 2**31 - 1]:
        count = int(rng.integers(1, 5000))
        s0 = np.random.get_state()
        want = np.random.randint(0, n, size=count)
        s1 = np.random.get_state()
        np.random.set_state(s0)
        got = legacy_rng.randint(n, count)
        assert got.dtype == np.int32
        np.testing.assert_array_equal(got, want)
        assert _same_state(np.random.get_state(), s1)
    # interleaving with NumPy's own calls (other distributions keep their cached state untouched)
    np.random.seed(seed)
    a = [np.random.randint(0, 777, size=13), np.random.normal(), np.random.randint(0, 10**6, size=1000), np.random.random()]
    np.random.seed(seed)
    b = [legacy_rng.randint(777, 13), np.random.normal(), legacy_rng.randint(10**6, 1000), np.random.random()]
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_block_draw_equals_sequential_8_and_4_draws():
    """One block draw of 12 * B integers is the reference's B iterations of randint(0, N, size=8) then size=4
    (src/ransac_algorithm.py:94,116)."""
    np.random.seed(4)
    seq = np.stack([np.concatenate([np.random.randint(0, 4200, size=8), np.random.randint(0, 4200, size=4)]) for _ in range(300)])
    nxt = np.random.randint(0, 100, 5)
    np.random.seed(4)
    blk = legacy_rng.randint(4200, 12 * 300).reshape(300, 12)
    np.testing.assert_array_equal(blk, seq)
    np.testing.assert_array_equal(np.random.randint(0, 100, 5), nxt)


def test_mark_and_rewind():
    np.random.seed(9)
    m = legacy_rng.mark()
    first = legacy_rng.randint(1000, 5000)
    legacy_rng.rewind(m)
    np.testing.assert_array_equal(legacy_rng.randint(1000, 5000), first)
    legacy_rng.rewind(m)
    np.testing.assert_array_equal(np.random.randint(0, 1000, size=5000), first)


def test_bad_arguments_are_rejected():
    from sfm_hip import _capi as C

    lib = C.load_library()
    key = np.zeros(624, np.uint32)
    out = np.zeros(4, np.int32)
    import ctypes

    pos = ctypes.c_int32(700)  # outside [0, 624]
    assert lib.sfm_legacy_randint_fill(C.hptr(key), ctypes.byref(pos), 10, 4, C.hptr(out)) == 100001
    pos = ctypes.c_int32(0)
    assert lib.sfm_legacy_randint_fill(C.hptr(key), ctypes.byref(pos), 0, 4, C.hptr(out)) == 100001
    assert lib.sfm_legacy_randint_fill(C.hptr(key), ctypes.byref(pos), 2**31, 4, C.hptr(out)) == 100001


def test_randint_fills_a_caller_buffer_identically():
    from sfm_hip import legacy_rng

    np.random.seed(77)
    want = np.random.randint(0, 4321, size=5000)
    after = np.random.randint(0, 1 << 30, 3)
    np.random.seed(77)
    buf = np.empty(5000, np.int32)
    got = legacy_rng.randint(4321, 5000, out=buf)
    assert got is buf
    np.testing.assert_array_equal(buf, want)
    np.testing.assert_array_equal(np.random.randint(0, 1 << 30, 3), after)
    with pytest.raises(ValueError):
        legacy_rng.randint(10, 7, out=np.empty(8, np.int32))
    with pytest.raises(ValueError):
        legacy_rng.randint(10, 8, out=np.empty(8, np.int64))

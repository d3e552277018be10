"""Sample-index draws from NumPy's global legacy generator, in place and fast.

The reference draws every minimal sample with ``np.random.randint(0, N, size=k)`` from the global
``np.random`` stream (src/ransac_algorithm.py:94,116; src/estimate_translation.py:64).  ``randint`` here returns
the same integers and leaves the global generator in the same state, but generates them with the library's own
MT19937 + masked-rejection loop working directly on NumPy's state memory (about 1 ns per integer instead of
5-25 ns).  ``mark`` / ``rewind`` give the E/H loop its "un-draw": the state is small (624 words + a position), so
a mark is a copy of it.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _capi as C


class _MtState(ctypes.Structure):
    # layout of NumPy's mt19937_state (numpy/random/src/mt19937/mt19937.h): key[624], pos
    _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int)]


def _global_state():
    """ctypes view of the MT19937 state behind the legacy global functions, or None when it is not reachable
    (another bit generator installed, or a NumPy whose layout this file does not know)."""
    try:
        rs = np.random.mtrand._rand
        bg = rs._bit_generator
        if type(bg).__name__ != "MT19937":
            return None
        st = _MtState.from_address(bg.ctypes.state_address)
        if not 0 <= st.pos <= 624:
            return None
        return st, bg
    except Exception:
        return None


_checked = None


def _fast_path_ok():
    """One-time self check: the in-place generator must reproduce np.random.randint on a private copy of the state."""
    global _checked
    if _checked is None:
        _checked = False
        got = _global_state()
        if got is not None:
            st, bg = got
            saved = np.random.get_state()
            try:
                np.random.seed(987654321)
                ref_state = np.random.get_state()
                want = np.concatenate([np.random.randint(0, n, size=700) for n in (1, 2, 3, 1000, 4097, 2**31 - 1)])
                end_state = np.random.get_state()
                np.random.set_state(ref_state)
                have = np.concatenate([_fill(st, bg, n, 700) for n in (1, 2, 3, 1000, 4097, 2**31 - 1)])
                now = np.random.get_state()
                _checked = bool(np.array_equal(want, have) and np.array_equal(now[1], end_state[1]) and now[2] == end_state[2])
            finally:
                np.random.set_state(saved)
    return _checked


def _fill(st, bg, n, count, out=None):
    out = np.empty(int(count), np.int32) if out is None else out
    with bg.lock:
        pos = ctypes.c_int32(st.pos)
        C.check(C.load_library().sfm_legacy_randint_fill(ctypes.addressof(st.key), ctypes.byref(pos), int(n), int(count), C.hptr(out)))
        st.pos = pos.value
    return out


def randint(n, count, out=None):
    """np.random.randint(0, n, size=count) as int32, consuming the global legacy stream identically.  `out`: an optional
    contiguous int32 array of `count` elements to fill instead of a new one (page-locked memory, for instance)."""
    if out is not None and (out.dtype != np.int32 or out.size != count or not out.flags.c_contiguous):
        raise ValueError("out must be a contiguous int32 array of `count` elements")
    if count > 0 and 0 < n < 2**31 and _fast_path_ok():
        st, bg = _global_state()
        return _fill(st, bg, n, count, out)
    drawn = np.random.randint(0, n, size=count).astype(np.int32)
    if out is None:
        return drawn
    out[...] = drawn.reshape(out.shape)
    return out


def mark():
    """Snapshot of the global legacy generator (to be handed to rewind)."""
    return np.random.get_state()


def rewind(state):
    np.random.set_state(state)

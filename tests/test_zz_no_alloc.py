"""The entry point that allocates nothing (sfm_eh_score_noalloc) against the entry point it is a variant of
(sfm_eh_score_mixed with a caller workspace): identical counts, bit for bit, on two calls in a row, the second with a
smaller input, so that whatever the first call left in the library's fixed scratch array cannot leak into the second.
Both are called through the C ABI.  Collected last (file name), because the round's end runs pytest -x."""
import numpy as np
import pytest

from oracle import sfm_oracle as orc
from sfm_hip import synthetic

pytestmark = pytest.mark.gpu


def _scene(n, seed):
    sc = synthetic.two_view_scene(n=n, inlier_ratio=0.4, planar_ratio=0.5, seed=seed)
    x1 = orc.normalize_K(sc["K"], sc["x1"])
    x2 = orc.normalize_K(sc["K"], sc["x2"])
    return sc, x1, x2


def _both(torch, C, ops, lib, corr, idx, thr):
    n, nb = corr.shape[0], idx.shape[0]
    ce = torch.full((nb,), -7, dtype=torch.int32, device="cuda")
    ch = torch.full((nb,), -7, dtype=torch.int32, device="cuda")
    C.check(lib.sfm_eh_score_noalloc(C.dptr(corr), n, C.dptr(idx), nb, thr, 3 * thr, C.dptr(ce), C.dptr(ch), C.stream_ptr()))
    ws = torch.empty(int(lib.sfm_eh_score_mixed_workspace_bytes(nb, n)), dtype=torch.uint8, device="cuda")
    re = torch.full((nb,), -9, dtype=torch.int32, device="cuda")
    rh = torch.full((nb,), -9, dtype=torch.int32, device="cuda")
    C.check(lib.sfm_eh_score_mixed(C.dptr(corr), n, C.dptr(idx), nb, thr, 3 * thr, 0.0, C.dptr(ws), ws.numel(), C.dptr(re), C.dptr(rh),
                                   C.stream_ptr()))
    torch.cuda.synchronize()
    return ce.cpu().numpy(), ch.cpu().numpy(), re.cpu().numpy(), rh.cpu().numpy()


def test_noalloc_entry_point_matches_workspace_entry_point_twice():
    import torch

    from sfm_hip import _capi as C
    from sfm_hip import ops

    lib = C.load_library()
    rng = np.random.default_rng(11)
    # first call: a large input (many matches: the scratch array is filled far beyond what the second call uses;
    # few hypotheses: the launch splits the match range and accumulates with atomics)
    sc, x1, x2 = _scene(20000, 3)
    thr = 1.0 / sc["K"][0][0]
    corr = C.dev(ops.pack_corr(x1, x2))
    idx = C.dev(rng.integers(0, 20000, (3000, 12)), np.int32)
    ce, ch, re, rh = _both(torch, C, ops, lib, corr, idx, thr)
    np.testing.assert_array_equal(ce, re)
    np.testing.assert_array_equal(ch, rh)
    assert ce.max() > 100 and ch.max() > 100
    # second call: smaller input (ragged match count: the last tile of 32 is partial)
    sc2, y1, y2 = _scene(1237, 4)
    corr2 = C.dev(ops.pack_corr(y1, y2))
    idx2_h = rng.integers(0, 1237, (700, 12))
    idx2 = C.dev(idx2_h, np.int32)
    ce2, ch2, re2, rh2 = _both(torch, C, ops, lib, corr2, idx2, thr)
    np.testing.assert_array_equal(ce2, re2)
    np.testing.assert_array_equal(ch2, rh2)
    # and both equal the oracle's float64 counts on samples without a repeated index
    oE, oH, _, _ = orc.score_EH_batch(y1, y2, idx2_h.astype(np.int64), thr)
    ok = np.array([len(set(r[:8])) == 8 and len(set(r[8:])) == 4 for r in idx2_h])
    np.testing.assert_array_equal(ce2[ok], oE[ok])
    np.testing.assert_array_equal(ch2[ok], oH[ok])
    # on a stream of the caller's own (the scratch array is looked up for the device that owns the stream)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ce3, ch3, re3, rh3 = _both(torch, C, ops, lib, corr2, idx2, thr)
    np.testing.assert_array_equal(ce3, re2)
    np.testing.assert_array_equal(ch3, rh2)


def test_noalloc_entry_point_rejects_oversize_and_bad_arguments():
    import torch

    from sfm_hip import _capi as C

    lib = C.load_library()
    corr = torch.zeros((8, 4), dtype=torch.float64, device="cuda")
    idx = torch.zeros((4, 12), dtype=torch.int32, device="cuda")
    out = torch.zeros(4, dtype=torch.int32, device="cuda")
    too_many = 32768 + 1
    rc = lib.sfm_eh_score_noalloc(C.dptr(corr), too_many, C.dptr(idx), 4, 1e-3, 3e-3, C.dptr(out), C.dptr(out), C.stream_ptr())
    assert rc == 100002 and b"SFM_NOALLOC_MAX_MATCHES" in lib.sfm_hip_error_string(rc)
    assert lib.sfm_eh_score_noalloc(C.dptr(corr), 0, C.dptr(idx), 4, 1e-3, 3e-3, C.dptr(out), C.dptr(out), C.stream_ptr()) == 100001
    assert lib.sfm_eh_score_noalloc(None, 8, C.dptr(idx), 4, 1e-3, 3e-3, C.dptr(out), C.dptr(out), C.stream_ptr()) == 100001

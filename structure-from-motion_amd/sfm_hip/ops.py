"""Device-level operators: thin Python wrappers, one per C-ABI entry point.

Inputs and outputs are torch CUDA tensors (device memory) unless a name ends in ``_host``.
Everything is enqueued on torch's current stream; callers synchronise by reading results.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _capi as C


def pack_corr(x1, x2):
    """(3,N) K-normalised homogeneous points of two views -> host array [N][4] float64.

    The kernels assume the third homogeneous coordinate is exactly 1 (it always is after
    normalize_K of pixel coordinates whose third row is 1; reference src/utils.py:44-45)."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    if x1.shape[0] == 2:
        x1 = np.vstack([x1, np.ones((1, x1.shape[1]))])
    if x2.shape[0] == 2:
        x2 = np.vstack([x2, np.ones((1, x2.shape[1]))])
    if x1.shape != x2.shape or x1.shape[0] != 3:
        raise ValueError("x1 and x2 must both be 3xN")
    if not (np.all(x1[2] == 1.0) and np.all(x2[2] == 1.0)):
        raise ValueError("the HIP path needs homogeneous points with third coordinate exactly 1")
    return np.ascontiguousarray(np.stack([x1[0], x1[1], x2[0], x2[1]], axis=1))


def _empty(shape, dtype):
    torch = C.torch_cuda()
    return torch.empty(shape, dtype=dtype, device="cuda")


def eh_score(corr, idx, thr_e, thr_h, out=None):
    """corr: float64 [N][4]; idx: int32 [B][12] -> (count_e, count_h) int32 [B]."""
    torch = C.torch_cuda()
    lib = C.load_library()
    n = corr.shape[0]
    nb = idx.shape[0]
    assert corr.dtype == torch.float64 and corr.is_contiguous() and corr.shape[1] == 4
    assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.shape[1] == 12
    if out is None:
        ce = _empty((nb,), torch.int32)
        ch = _empty((nb,), torch.int32)
    else:
        ce, ch = out
    C.check(lib.sfm_eh_score_f64(C.dptr(corr), n, C.dptr(idx), nb, float(thr_e), float(thr_h), C.dptr(ce), C.dptr(ch), C.stream_ptr()))
    return ce, ch


EVERY_PAIR_FULL_TEST = False  # default of eh_score_mixed's every_pair_full_test (the tests run both forms through it)


def eh_score_mixed(corr, idx, thr_e, thr_h, xmax=0.0, out=None, workspace=None, every_pair_full_test=None):
    """The counts of eh_score (identical for every hypothesis) with the residual tests on the matrix cores + float32 and a
    float64 re-evaluation of the pairs inside the error band.  xmax: optional hint of max |coordinate| over corr; the
    library measures it on the device and uses the larger value.  every_pair_full_test: no early loops (a pair that fails a
    cheaper implied test is otherwise counted out without the full test); the counts are the same."""
    torch = C.torch_cuda()
    lib = C.load_library()
    nb, n = idx.shape[0], corr.shape[0]
    if out is None:
        ce = _empty((nb,), torch.int32)
        ch = _empty((nb,), torch.int32)
    else:
        ce, ch = out
    if workspace is None:
        workspace = _empty((max(256, int(lib.sfm_eh_score_mixed_workspace_bytes(nb, n))),), torch.uint8)
    if every_pair_full_test is None:
        every_pair_full_test = EVERY_PAIR_FULL_TEST
    C.check(lib.sfm_eh_score_mixed_opt(C.dptr(corr), n, C.dptr(idx), nb, float(thr_e), float(thr_h), float(xmax), 1 if every_pair_full_test else 0,
                                       C.dptr(workspace), workspace.numel(), C.dptr(ce), C.dptr(ch), C.stream_ptr()))
    return ce, ch


def corr_to_f32(corr):
    """float32 copy [n + 1][4] of the matches: row n is a trailer whose first word is the measured max |coordinate|."""
    torch = C.torch_cuda()
    lib = C.load_library()
    out = _empty((int(lib.sfm_corr_f32_bytes(corr.shape[0])) // 16, 4), torch.float32)
    C.check(lib.sfm_corr_to_f32(C.dptr(corr), corr.shape[0], C.dptr(out), C.stream_ptr()))
    return out


def eh_upper(corr, corr32, idx, thr_e, thr_h, xmax, safety=2.0, out=None, one_sided=False):
    """float32 pre-filter: upper bounds (up_e, up_h) int32 [B] of the exact counts.
    xmax: hint of max |coordinate| over corr (0 = unknown; the measured value in corr32 is used when larger).
    one_sided: cheaper, looser bound for E."""
    torch = C.torch_cuda()
    lib = C.load_library()
    nb = idx.shape[0]
    if out is None:
        ue = _empty((nb,), torch.int32)
        uh = _empty((nb,), torch.int32)
    else:
        ue, uh = out
    C.check(lib.sfm_eh_upper_f32(C.dptr(corr), C.dptr(corr32), corr.shape[0], C.dptr(idx), nb, float(thr_e), float(thr_h),
                                 float(xmax), float(safety), int(bool(one_sided)), C.dptr(ue), C.dptr(uh), C.stream_ptr()))
    return ue, uh


def corr_to_mfma(corr, thr_e):
    """Match-side operands of the matrix-core pre-filter for the epipolar threshold thr_e (bf16 piece vectors of
    (x1, y1, 1) + float32 (x2, y2) divided by the square root of the slackened threshold constant)."""
    torch = C.torch_cuda()
    lib = C.load_library()
    n = corr.shape[0]
    corrm = _empty((int(lib.sfm_corr_mfma_bytes(n)),), torch.uint8)
    C.check(lib.sfm_corr_to_mfma(C.dptr(corr), n, float(thr_e), C.dptr(corrm), C.stream_ptr()))
    return corrm


def eh_upper_mfma(corr, corrm, idx, thr_e, thr_h, xmax=0.0, safety=2.0, out=None, part=0):
    """Loose float32 upper bounds like eh_upper(one_sided=True), with E x1 and H x1 on the bf16 matrix cores.
    part=1 / 2 (measurement aid): only the epipolar / only the transfer test."""
    torch = C.torch_cuda()
    lib = C.load_library()
    nb = idx.shape[0]
    if out is None:
        ue = _empty((nb,), torch.int32)
        uh = _empty((nb,), torch.int32)
    else:
        ue, uh = out
    if part:
        C.check(lib.sfm_eh_upper_mfma_part_f32(C.dptr(corr), C.dptr(corrm), corr.shape[0], C.dptr(idx), nb, float(thr_e), float(thr_h),
                                               float(xmax), float(safety), int(part), C.dptr(ue), C.dptr(uh), C.stream_ptr()))
        return ue, uh
    C.check(lib.sfm_eh_upper_mfma_f32(C.dptr(corr), C.dptr(corrm), corr.shape[0], C.dptr(idx), nb, float(thr_e), float(thr_h),
                                      float(xmax), float(safety), C.dptr(ue), C.dptr(uh), C.stream_ptr()))
    return ue, uh


def eh_best_workspace(nb, n):
    """Device scratch for eh_best_native over nb hypotheses and n matches (uint8 tensor)."""
    torch = C.torch_cuda()
    lib = C.load_library()
    return _empty((int(lib.sfm_eh_best_workspace_bytes(int(nb), int(n))),), torch.uint8)


def eh_best_native(corr, corr32, idx, thr_e, thr_h, xmax, safety=2.0, loose=2, workspace=None):
    """The bound cascade as ONE library call (sfm_eh_best).  Same result as eh_best below.
    loose: first level = 0 tight float32 bounds, 1 loose bounds on the vector ALU, 2 loose bounds with the
    3x3 products on the bf16 matrix cores (default, fastest)."""
    lib = C.load_library()
    nb = idx.shape[0]
    if workspace is None:
        workspace = eh_best_workspace(nb, corr.shape[0])
    res = np.zeros(8, np.int64)
    C.check(lib.sfm_eh_best(C.dptr(corr), C.dptr(corr32), corr.shape[0], C.dptr(idx), nb, float(thr_e), float(thr_h), float(xmax), float(safety),
                            int(loose), C.dptr(workspace), workspace.numel(), C.hptr(res), C.stream_ptr()))
    stats = {"candidates_e": int(res[4]), "candidates_h": int(res[5]), "tightened": int(res[6]), "rescored_f64": int(res[7])}
    return (int(res[0]), int(res[1])), (int(res[2]), int(res[3])), stats


class EhBestJob:
    """One block of hypotheses in flight: eh_best_submit enqueues all levels of the cascade on `stream` (no host
    synchronisation), eh_best_collect waits and returns what eh_best_native returns.  Keep one workspace per job."""

    def __init__(self, args, workspace, pending, stream):
        self.args, self.workspace, self.pending, self.stream = args, workspace, pending, stream


def eh_best_pending():
    """Page-locked host buffer that receives the outcome of one job."""
    torch = C.torch_cuda()
    return torch.empty(int(C.load_library().sfm_eh_best_pending_bytes()), dtype=torch.uint8, pin_memory=True)


def eh_best_submit(corr, corr32, idx, thr_e, thr_h, xmax, safety=2.0, loose=2, workspace=None, stream=None, pending=None):
    torch = C.torch_cuda()
    lib = C.load_library()
    nb = idx.shape[0]
    if workspace is None:
        workspace = eh_best_workspace(nb, corr.shape[0])
    if pending is None:
        pending = eh_best_pending()
    sp = C.stream_ptr() if stream is None else stream.cuda_stream
    args = (C.dptr(corr), C.dptr(corr32), corr.shape[0], C.dptr(idx), nb, float(thr_e), float(thr_h), float(xmax), float(safety),
            int(loose), C.dptr(workspace), workspace.numel())
    C.check(lib.sfm_eh_best_submit(*args, pending.data_ptr(), sp))
    job = EhBestJob(args, workspace, pending, sp)
    job.keep = (corr, corr32, idx)  # the device buffers must outlive the job
    return job


def eh_best_collect(job):
    lib = C.load_library()
    res = np.zeros(8, np.int64)
    C.check(lib.sfm_eh_best_collect(*job.args, job.pending.data_ptr(), C.hptr(res), job.stream))
    stats = {"candidates_e": int(res[4]), "candidates_h": int(res[5]), "tightened": int(res[6]), "rescored_f64": int(res[7])}
    return (int(res[0]), int(res[1])), (int(res[2]), int(res[3])), stats


TIGHTEN_ABOVE = 4096  # candidates beyond this count go through the tight float32 bound before float64


def eh_best(corr, corr32, idx, thr_e, thr_h, xmax, safety=2.0, scratch=None, loose=True):
    """Exact first-maximum winners of a block of hypotheses by a cascade of upper bounds:

      1. float32 bounds for every hypothesis (loose=True: one-sided epipolar + box transfer test,
         26 instead of 38 instructions per residual pair);
      2. exact float64 count of the hypothesis with the largest bound -> a floor;
      3. every hypothesis whose bound reaches the floor is a candidate; if there are many they are
         first re-bounded with the tight float32 forms;
      4. the remaining candidates are scored in float64; the first maximum wins.

    Every level only discards hypotheses whose upper bound is below an exact count that exists,
    so the result equals arg-max over sfm_eh_score_f64's counts with lowest-index tie-break.
    Returns ((count_e, index_e), (count_h, index_h), stats) with indices local to `idx`."""
    torch = C.torch_cuda()
    ue, uh = eh_upper(corr, corr32, idx, thr_e, thr_h, xmax, safety, out=scratch, one_sided=loose)
    keys = torch.zeros((2,), dtype=torch.int64, device="cuda")
    argmax_count(ue, 0, keys[0:1])
    argmax_count(uh, 0, keys[1:2])
    lead = 0xFFFFFFFF - (keys & 0xFFFFFFFF)  # ids of the largest upper bounds, still on the device
    ce, ch = eh_score(corr, idx.index_select(0, lead), thr_e, thr_h)
    floor = torch.stack([ce[0], ch[1]]).to(torch.int64)  # exact counts of the two leaders
    mask_e = ue >= floor[0]
    mask_h = uh >= floor[1]
    # one host round trip: leaders, their exact counts and the number of candidates
    info = torch.cat([lead, floor, torch.stack([mask_e.sum(), mask_h.sum()])]).cpu().numpy()
    le, lh, fe, fh, ne, nh = (int(v) for v in info)
    stats = {"candidates_e": ne, "candidates_h": nh, "tightened": 0}
    win_e, win_h = (fe, le), (fh, lh)
    if ne > 1 or nh > 1:
        # several hypotheses could still beat (or tie with a lower id than) a leader
        cand_e = torch.nonzero(mask_e).flatten() if ne > 1 else lead[0:1]  # ascending ids
        cand_h = torch.nonzero(mask_h).flatten() if nh > 1 else lead[1:2]
        if loose and cand_e.shape[0] + cand_h.shape[0] > TIGHTEN_ABOVE:
            cand = torch.cat([cand_e, cand_h])
            te, th = eh_upper(corr, corr32, idx.index_select(0, cand), thr_e, thr_h, xmax, safety, one_sided=False)
            k = cand_e.shape[0]
            cand_e = cand_e[te[:k] >= floor[0]]
            cand_h = cand_h[th[k:] >= floor[1]]
            stats["tightened"] = int(cand.shape[0])
        cand = torch.cat([cand_e, cand_h])
        ce, ch = eh_score(corr, idx.index_select(0, cand), thr_e, thr_h)
        k = cand_e.shape[0]
        res = torch.cat([cand, ce.to(torch.int64), ch.to(torch.int64)]).cpu().numpy()
        m = cand.shape[0]
        ids, xe, xh = res[:m], res[m : 2 * m], res[2 * m :]
        ie = int(np.argmax(xe[:k]))  # ascending ids: the first maximum is the lowest id
        ih = int(np.argmax(xh[k:]))
        win_e = (int(xe[ie]), int(ids[ie]))
        win_h = (int(xh[k + ih]), int(ids[k + ih]))
        stats["rescored_f64"] = int(m)
    return win_e, win_h, stats


def eh_models(corr, idx, want_F=False):
    torch = C.torch_cuda()
    lib = C.load_library()
    nb = idx.shape[0]
    F = _empty((nb, 9), torch.float64) if want_F else None
    E = _empty((nb, 9), torch.float64)
    H = _empty((nb, 9), torch.float64)
    C.check(lib.sfm_eh_models_f64(C.dptr(corr), corr.shape[0], C.dptr(idx), nb, C.dptr(F), C.dptr(E), C.dptr(H), C.stream_ptr()))
    return (F, E, H) if want_F else (E, H)


def dlt_general(corr, homography):
    """estimate_F_DLT / estimate_H_DLT over all rows of corr (float64 [n][4], n >= 8 / 4): float64 [9], row-major."""
    torch = C.torch_cuda()
    lib = C.load_library()
    out = _empty((9,), torch.float64)
    C.check(lib.sfm_dlt_general_f64(C.dptr(corr), corr.shape[0], 1 if homography else 0, C.dptr(out), C.stream_ptr()))
    return out


def epipolar_mask(E, corr, thr):
    """E: float64 [9] device -> (mask uint8 [N], count int32 [1])."""
    torch = C.torch_cuda()
    lib = C.load_library()
    n = corr.shape[0]
    mask = _empty((n,), torch.uint8)
    count = _empty((1,), torch.int32)
    C.check(lib.sfm_epipolar_mask_f64(C.dptr(E), C.dptr(corr), n, float(thr), C.dptr(mask), C.dptr(count), C.stream_ptr()))
    return mask, count


def transfer_mask(H, corr, thr):
    torch = C.torch_cuda()
    lib = C.load_library()
    n = corr.shape[0]
    mask = _empty((n,), torch.uint8)
    count = _empty((1,), torch.int32)
    C.check(lib.sfm_transfer_mask_f64(C.dptr(H), C.dptr(corr), n, float(thr), C.dptr(mask), C.dptr(count), C.stream_ptr()))
    return mask, count


def epipolar_errors(F, x1, x2):
    """F float64 [9]; x1, x2 float64 [3][N] -> d float64 [N]."""
    torch = C.torch_cuda()
    lib = C.load_library()
    n = x1.shape[1]
    d = _empty((n,), torch.float64)
    C.check(lib.sfm_epipolar_errors_f64(C.dptr(F), C.dptr(x1), C.dptr(x2), n, x1.stride(0), C.dptr(d), C.stream_ptr()))
    return d


def enforce_essential(F):
    torch = C.torch_cuda()
    lib = C.load_library()
    E = _empty((9,), torch.float64)
    C.check(lib.sfm_enforce_essential_f64(C.dptr(F), C.dptr(E), C.stream_ptr()))
    return E


def essential_candidates(E):
    torch = C.torch_cuda()
    lib = C.load_library()
    P = _empty((4, 12), torch.float64)
    C.check(lib.sfm_essential_candidates_f64(C.dptr(E), C.dptr(P), C.stream_ptr()))
    return P


def chirality4(P2s, corr, mask=None):
    torch = C.torch_cuda()
    lib = C.load_library()
    counts = _empty((4,), torch.int32)
    C.check(lib.sfm_chirality4_f64(C.dptr(P2s), C.dptr(corr), C.dptr(mask), corr.shape[0], C.dptr(counts), C.stream_ptr()))
    return counts


def homography_to_rt(H, corr):
    torch = C.torch_cuda()
    lib = C.load_library()
    out = _empty((43,), torch.float64)
    C.check(lib.sfm_homography_to_rt_f64(C.dptr(H), C.dptr(corr), corr.shape[0], C.dptr(out), C.stream_ptr()))
    return out


def triangulate(P1, P2, x1, x2):
    """P1,P2: float64 [12]; x1,x2: float64 [>=2][M] contiguous rows -> X float64 [4][M]."""
    torch = C.torch_cuda()
    lib = C.load_library()
    m = x1.shape[1]
    X = _empty((4, m), torch.float64)
    C.check(lib.sfm_triangulate_f64(C.dptr(P1), C.dptr(P2), C.dptr(x1), C.dptr(x2), m, x1.stride(0), C.dptr(X), C.stream_ptr()))
    return X


def pack_points(R, X, x):
    """R: float64 [9]; X: float32 or float64 [3][M]; x: float64 [3][M] -> pts float64 [M][6]."""
    torch = C.torch_cuda()
    lib = C.load_library()
    m = x.shape[1]
    assert X.dtype in (torch.float32, torch.float64) and X.is_contiguous() and x.is_contiguous()
    pts = _empty((m, 6), torch.float64)
    C.check(lib.sfm_pack_points_f64(C.dptr(R), C.dptr(X), int(X.dtype == torch.float64), C.dptr(x), m, C.dptr(pts), C.stream_ptr()))
    return pts


def _offsets(cam_sizes):
    off = np.zeros(len(cam_sizes) + 1, np.int64)
    off[1:] = np.cumsum(cam_sizes)
    return off


def t_ransac(pts, cam_sizes, idx, thr, mixed=True, workspace=None):
    """pts: float64 [sum M][6]; idx: int32 [ncam][B][2] -> (best uint64->int64 view [ncam], T_best [ncam][3], counts [ncam][B]).
    mixed=True: the residual tests run on the matrix cores + float32 with a float64 re-evaluation of the pairs inside the
    error band (identical counts, about 2x faster); mixed=False: every test in float64."""
    torch = C.torch_cuda()
    lib = C.load_library()
    ncam, nb = idx.shape[0], idx.shape[1]
    off = _offsets(cam_sizes)
    counts = _empty((ncam, nb), torch.int32)
    best = _empty((ncam,), torch.int64)
    T = _empty((ncam, 3), torch.float64)
    if mixed:
        ws_bytes = int(lib.sfm_t_ransac_mixed_workspace_bytes(C.hptr(off), ncam))
        ws = workspace if workspace is not None else _empty((ws_bytes,), torch.uint8)
        C.check(lib.sfm_t_ransac_mixed(C.dptr(pts), C.hptr(off), ncam, C.dptr(idx), nb, float(thr), C.dptr(ws), ws.numel(), C.dptr(counts),
                                       C.dptr(best), C.dptr(T), C.stream_ptr()))
    else:
        C.check(lib.sfm_t_ransac_f64(C.dptr(pts), C.hptr(off), ncam, C.dptr(idx), nb, float(thr), C.dptr(counts), C.dptr(best), C.dptr(T),
                                     C.stream_ptr()))
    return best, T, counts


def pts_to_f32(pts):
    torch = C.torch_cuda()
    lib = C.load_library()
    out = _empty((pts.shape[0], 8), torch.float32)
    C.check(lib.sfm_pts_to_f32(C.dptr(pts), pts.shape[0], C.dptr(out), C.stream_ptr()))
    return out


def t_upper(pts, pts32, cam_sizes, idx, thr, rmax, xmax, safety=2.0):
    """float32 pre-filter of t_ransac: upper bounds int32 [ncam][B]."""
    torch = C.torch_cuda()
    lib = C.load_library()
    ncam, nb = idx.shape[0], idx.shape[1]
    off = _offsets(cam_sizes)
    rm = np.ascontiguousarray(rmax, dtype=np.float64)
    up = _empty((ncam, nb), torch.int32)
    C.check(lib.sfm_t_upper_f32(C.dptr(pts), C.dptr(pts32), C.hptr(off), ncam, C.dptr(idx), nb, float(thr), C.hptr(rm), float(xmax),
                                float(safety), C.dptr(up), C.stream_ptr()))
    return up


def argmax_count_batched(counts):
    """counts int32 [R][B] -> packed keys int64 [R] (first maximum of every row)."""
    torch = C.torch_cuda()
    lib = C.load_library()
    best = _empty((counts.shape[0],), torch.int64)
    C.check(lib.sfm_argmax_count_batched_i32(C.dptr(counts), counts.shape[1], counts.shape[0], C.dptr(best), C.stream_ptr()))
    return best


def t_best(pts, pts32, cam_sizes, idx, thr, rmax, xmax, safety=2.0):
    """Exact winners of the translation RANSAC of every camera, two tiers (float32 bounds for all
    hypotheses, float64 re-score of the candidates).  Returns (count [ncam], index [ncam], T [ncam][3])
    as host arrays, equal to what t_ransac returns."""
    torch = C.torch_cuda()
    ncam, nb = idx.shape[0], idx.shape[1]
    up = t_upper(pts, pts32, cam_sizes, idx, thr, rmax, xmax, safety)
    lead = 0xFFFFFFFF - (argmax_count_batched(up) & 0xFFFFFFFF)  # [ncam] ids of the largest bounds
    rows = torch.arange(ncam, device="cuda")
    lead_idx = idx[rows, lead].reshape(ncam, 1, 2).contiguous()
    best, T, _ = t_ransac(pts, cam_sizes, lead_idx, thr)
    floor = best >> 32  # exact count of every leader
    mask = up >= floor.to(torch.int32)[:, None]
    ncand = mask.sum(dim=1)
    info = torch.cat([lead, floor, ncand]).cpu().numpy()
    lead_h, floor_h, ncand_h = info[:ncam], info[ncam : 2 * ncam], info[2 * ncam :]
    count = floor_h.astype(np.int64).copy()
    index = lead_h.astype(np.int64).copy()
    T_out = T
    kmax = int(ncand_h.max())
    if kmax > 1:
        # candidate ids per camera, ascending, padded with the leader (already present, so a padded
        # duplicate can never beat the first occurrence)
        order = torch.argsort(mask.to(torch.int8), dim=1, descending=True, stable=True)[:, :kmax]
        pos = torch.arange(kmax, device="cuda")[None, :]
        cand = torch.where(pos < ncand[:, None], order, lead[:, None])
        cand_idx = idx[rows[:, None], cand].contiguous()
        best2, T_out, _ = t_ransac(pts, cam_sizes, cand_idx, thr)
        c2, p2 = decode_best(best2.cpu().numpy())
        cand_h = cand.cpu().numpy()
        count = c2
        index = cand_h[np.arange(ncam), p2].astype(np.int64)
    return count, index, T_out.cpu().numpy()


def t_count(pts, T, thr):
    """pts float64 [M][6]; T float64 [B][3] -> counts int32 [B]."""
    torch = C.torch_cuda()
    lib = C.load_library()
    nb = T.shape[0]
    counts = _empty((nb,), torch.int32)
    C.check(lib.sfm_t_count_f64(C.dptr(pts), pts.shape[0], C.dptr(T), nb, float(thr), C.dptr(counts), C.stream_ptr()))
    return counts


def argmax_count(counts, index_base=0, best=None):
    """counts int32 [B] -> best int64 [1] holding the packed (count, lowest index) key."""
    torch = C.torch_cuda()
    lib = C.load_library()
    if best is None:
        best = torch.zeros((1,), dtype=torch.int64, device="cuda")
    C.check(lib.sfm_argmax_count_i32(C.dptr(counts), counts.shape[0], int(index_base), C.dptr(best), C.stream_ptr()))
    return best


def decode_best(best_host):
    """uint64 keys (as int64) -> (count, index) arrays."""
    k = np.asarray(best_host).astype(np.uint64)
    count = (k >> np.uint64(32)).astype(np.int64)
    index = (np.uint64(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF))).astype(np.int64)
    return count, index


def index_sets(seed, first_id, nb, n, width=12, out=None):
    """Device-generated index sets int32[nb][width] of the hypothesis ids [first_id, first_id + nb): a pure function
    of (seed, id, n), so every sharding of an id range sees the same sets (the reference draws them from the global
    np.random stream: src/ransac_algorithm.py:92,116, src/estimate_translation.py:65)."""
    torch = C.torch_cuda()
    lib = C.load_library()
    idx = out if out is not None else _empty((nb, width), torch.int32)
    if tuple(idx.shape) != (nb, width) or idx.dtype != torch.int32 or not idx.is_contiguous():
        raise ValueError("out must be a contiguous int32 [nb][width] device tensor")
    C.check(lib.sfm_index_sets_i32(int(seed) & (2**64 - 1), int(first_id), nb, int(n), width, C.dptr(idx), C.stream_ptr()))
    return idx


def t_models(pts, idx):
    torch = C.torch_cuda()
    lib = C.load_library()
    nb = idx.shape[0]
    T = _empty((nb, 3), torch.float64)
    C.check(lib.sfm_t_models_f64(C.dptr(pts), pts.shape[0], C.dptr(idx), nb, C.dptr(T), C.stream_ptr()))
    return T


def lm_translation(K, pts, cam_sizes, T0, iters, mu, want_history=False):
    torch = C.torch_cuda()
    lib = C.load_library()
    ncam = len(cam_sizes)
    off = _offsets(cam_sizes)
    work = _empty((max(1, 2 * int(off[-1])),), torch.float64)
    T = _empty((ncam, 3), torch.float64)
    hist = _empty((ncam, iters, 3), torch.float64) if want_history else None
    C.check(lib.sfm_lm_translation_f64(C.dptr(K), C.dptr(pts), C.hptr(off), ncam, C.dptr(T0), int(iters), float(mu), C.dptr(work), C.dptr(T), C.dptr(hist), C.stream_ptr()))
    return (T, hist) if want_history else T


# ---- record events of the E/H replay ----------------------------------------------------------


class RecordScratch:
    """Device and host scratch of e_record / h_record (allocate once per image pair)."""

    def __init__(self, n):
        torch = C.torch_cuda()
        k = int(C.load_library().sfm_record_scratch_doubles())
        self.dev = _empty((k,), torch.float64)
        self.host = np.zeros(k, np.float64)
        self.n = n


def _pick_candidate(host, p_off, c_off):
    counts = host[c_off : c_off + 2].view(np.int32)[:4]
    best, n = 0, -1
    for i in range(4):  # first strict maximum (src/chirality_check.py:72-91)
        if counts[i] > n:
            best, n = i, int(counts[i])
    P = host[p_off + 12 * best : p_off + 12 * best + 12].reshape(3, 4)
    return P[:, :3].copy(), P[:, 3].copy(), n


def e_record(corr, row, thr, scratch):
    """src/ransac_algorithm.py:105-113 in one call: returns (R, T, mask uint8 device tensor)."""
    torch = C.torch_cuda()
    lib = C.load_library()
    n = corr.shape[0]
    mask = _empty((n,), torch.uint8)
    r = np.ascontiguousarray(row, np.int32)
    C.check(lib.sfm_e_record_f64(C.dptr(corr), n, C.hptr(r), float(thr), C.dptr(mask), C.dptr(scratch.dev), C.hptr(scratch.host), C.stream_ptr()))
    R, T, _ = _pick_candidate(scratch.host, 18, 66)
    return R, T, mask


def h_record(corr, row, thr, scratch):
    """src/ransac_algorithm.py:121-147 in one call: ((R, T, count, mask), (R, T, count, mask)) for the two solutions."""
    torch = C.torch_cuda()
    lib = C.load_library()
    n = corr.shape[0]
    ma, mb = _empty((n,), torch.uint8), _empty((n,), torch.uint8)
    r = np.ascontiguousarray(row, np.int32)
    C.check(lib.sfm_h_record_f64(C.dptr(corr), n, C.hptr(r), float(thr), C.dptr(ma), C.dptr(mb), C.dptr(scratch.dev), C.hptr(scratch.host),
                                 C.stream_ptr()))
    a = _pick_candidate(scratch.host, 18, 66) + (ma,)
    b = _pick_candidate(scratch.host, 113, 161) + (mb,)
    return a, b


def h_record_batch(corr, rows, thr):
    """h_record for every row of `rows` (device int32 [k][12]) in one launch: a list of what h_record returns."""
    torch = C.torch_cuda()
    lib = C.load_library()
    n, k = corr.shape[0], rows.shape[0]
    if k == 0:
        return []
    ma, mb = _empty((k, n), torch.uint8), _empty((k, n), torch.uint8)
    nd = int(lib.sfm_record_scratch_doubles())
    dev = _empty((k, nd), torch.float64)
    host = np.zeros((k, nd), np.float64)
    assert rows.dtype == torch.int32 and rows.is_contiguous() and rows.shape[1] == 12
    C.check(lib.sfm_h_record_batch_f64(C.dptr(corr), n, C.dptr(rows), k, float(thr), C.dptr(ma), C.dptr(mb), C.dptr(dev), C.hptr(host),
                                       C.stream_ptr()))
    return [(_pick_candidate(host[e], 18, 66) + (ma[e],), _pick_candidate(host[e], 113, 161) + (mb[e],)) for e in range(k)]


# ---- descriptor matching ---------------------------------------------------------------------


def descriptors_to_i8(desc):
    """desc: float32 or uint8 device tensor [m][128], integer valued in [0, 255] (what cv2.SIFT emits).
    Returns (desc8 int8[m][128] = value - 128, norm int32[m]); raises ValueError on any other content."""
    torch = C.torch_cuda()
    lib = C.load_library()
    if desc.dim() != 2 or desc.shape[1] != 128 or desc.dtype not in (torch.float32, torch.uint8) or not desc.is_contiguous():
        raise ValueError("descriptors must be a contiguous float32 or uint8 [m][128] device tensor")
    m = desc.shape[0]
    d8 = _empty((m, 128), torch.int8)
    norm = _empty((m,), torch.int32)
    bad = _empty((1,), torch.int32)
    C.check(lib.sfm_descriptors_to_i8(C.dptr(desc), int(desc.dtype == torch.uint8), m, C.dptr(d8), C.dptr(norm), C.dptr(bad), C.stream_ptr()))
    if m and int(bad.item()) != 0:
        raise ValueError("descriptor components must be integers in [0, 255] (SIFT descriptors as OpenCV emits them)")
    return d8, norm


def knn2_ratio(q8, qnorm, t8, tnorm, ratio):
    """2-nearest-neighbour match of every query against the train set + ratio test (cv2.BFMatcher().knnMatch(q, t, k=2)
    and `m.distance < ratio * n.distance`: src/extract_sift.py:117-123, src/utils.py:82-86).
    Returns (train_idx int32[nq], dist1 f32[nq], dist2 f32[nq], good uint8[nq], good_count int)."""
    torch = C.torch_cuda()
    lib = C.load_library()
    nq, nt = q8.shape[0], t8.shape[0]
    if nt < 2:
        raise ValueError("need at least two train descriptors (knnMatch with k=2 yields pairs only then)")
    ws_bytes = int(lib.sfm_knn2_workspace_bytes(nq, nt))
    ws = _empty((max(16, ws_bytes),), torch.uint8)
    tidx = _empty((nq,), torch.int32)
    d1 = _empty((nq,), torch.float32)
    d2 = _empty((nq,), torch.float32)
    good = _empty((nq,), torch.uint8)
    cnt = _empty((1,), torch.int32)
    C.check(lib.sfm_knn2_ratio_i8(C.dptr(q8), C.dptr(qnorm), nq, C.dptr(t8), C.dptr(tnorm), nt, float(ratio), C.dptr(ws), ws_bytes,
                                  C.dptr(tidx), C.dptr(d1), C.dptr(d2), C.dptr(good), C.dptr(cnt), C.stream_ptr()))
    return tidx, d1, d2, good, int(cnt.item())

#!/usr/bin/env python3
"""Benchmark of the robust-estimation hot path on MI355X.

Workload (N=1): the synthetic two-view E+H RANSAC configuration: 10 000 SIFT-like matches,
30 % inliers, 10^6 hypotheses per GPU.  One step = one pass of the fused solver + scoring
kernel over all hypotheses of this rank (every hypothesis: 8-point E and 4-point H model,
both scored against all matches) plus the arg-max of the inlier counts and, for N>1, the
merge of the per-rank winners by one all-reduce(max) of packed (count, index) keys.

Metric: hypothesis x correspondence residual evaluations per second (an E residual and an H
residual each count as one evaluation).  Inputs (matches and index sets) are resident in HBM
before the timed region.  Steps are independent blocks of hypotheses; two are kept in flight on
two HIP streams (sfm_eh_best_submit / _collect), every step is collected and merged inside the
timed region (--no-overlap runs them strictly one after the other).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FLOPS_PER_E_EVAL = 36  # algorithmic cost of one symmetric-epipolar residual test
FLOPS_PER_H_EVAL = 21  # one transfer-error residual test
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X vector FP64 (half of the FP32 vector rate)
PMC_TRAFFIC_BYTES = (2 * 26441.5 + 7812.5) * 1024  # FETCH_SIZE x2 + WRITE_SIZE of the dominant kernel, KB -> bytes


def cpu_baseline(x1n, x2n, thr, n_matches, idx):
    """The NumPy oracle (a port of the reference loop body) on a bounded sample (the head of this
    rank's index sets) of the same workload, one host thread."""
    from oracle import sfm_oracle as orc

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    sample_hyps = idx.shape[0]
    t0 = time.perf_counter()
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            cE, cH, _, _ = orc.score_EH_batch(x1n, x2n, idx, thr)
    else:
        cE, cH, _, _ = orc.score_EH_batch(x1n, x2n, idx, thr)
    dt = time.perf_counter() - t0
    return {
        "value": 2.0 * sample_hyps * n_matches / dt,
        "unit": "evals/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sample_hyps} hypotheses x {n_matches} matches of the same workload (8-pt E + 4-pt H solve and scoring), "
        f"NumPy float64 oracle, {dt:.1f} s",
    }, idx, cE, cH


def _cpu_worker(job):
    """One worker of the all-cores CPU baseline: scores its slice of the sample with the NumPy oracle."""
    n, seed, lo, hi = job
    from oracle import sfm_oracle as orc
    from sfm_hip import synthetic

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=seed)
    thr = 1.0 / sc["K"][0][0]
    x1n = orc.normalize_K(sc["K"], sc["x1"])
    x2n = orc.normalize_K(sc["K"], sc["x2"])
    idx = np.random.default_rng(seed + lo).integers(0, n, size=(hi - lo, 12))
    t0 = time.perf_counter()
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            orc.score_EH_batch(x1n, x2n, idx, thr)
    else:
        orc.score_EH_batch(x1n, x2n, idx, thr)
    return time.perf_counter() - t0


def cpu_baseline_all_cores(n, seed, workers, hyps_per_worker):
    """The same oracle on `workers` processes, each with its own range of hypotheses (the reference itself is single
    threaded; this is what sharding its loop over the host cores would give)."""
    import multiprocessing as mp

    jobs = [(n, seed, w * hyps_per_worker, (w + 1) * hyps_per_worker) for w in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(workers) as pool:
        pool.map(_cpu_worker, [(n, seed, 0, 64)] * workers)  # start the workers, import NumPy
        t1 = time.perf_counter()
        busy = pool.map(_cpu_worker, jobs)
        dt = time.perf_counter() - t1
    return {"value": 2.0 * workers * hyps_per_worker * n / dt, "unit": "evals/s", "cores": workers, "kind": "port",
            "sample": f"{workers} processes x {hyps_per_worker} hypotheses x {n} matches, NumPy float64 oracle, {dt:.1f} s wall "
            f"(slowest worker {max(busy):.1f} s, pool start {t1 - t0:.1f} s not counted)"}


def secondary_measurements(torch, ops, _capi, synthetic, seed):
    """Timings of the other rows of the path (not part of `value`): the batched 2-point
    translation RANSAC (12 cameras x 50 000 correspondences x 100 000 hypotheses) and the
    batched LM refinement + DLT triangulation over 12 cameras of dataset-10-like sizes."""
    out = {}
    K = synthetic.K_DATASET5
    ncam, m, B = 12, 50000, 100000
    rng = np.random.default_rng(seed)
    cams = [synthetic.resection_scene(m=m, outlier_ratio=0.5, seed=seed + 100 + c) for c in range(ncam)]
    pts = torch.cat([ops.pack_points(_capi.dev(c["R"].ravel()), _capi.dev(c["X"]), _capi.dev(c["x_norm"])) for c in cams])
    idx = _capi.dev(rng.integers(0, m, size=(ncam, B, 2)), np.int32)
    thr = 4.0 / K[0][0]
    reps = 3
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.t_ransac(pts, [m] * ncam, idx, thr, mixed=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        best_m, T_m, counts_m = ops.t_ransac(pts, [m] * ncam, idx, thr, mixed=True)
    b.record()
    torch.cuda.synchronize()
    ms_mixed = a.elapsed_time(b) / reps
    ops.t_ransac(pts, [m] * ncam, idx, thr, mixed=False)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        best, T, counts = ops.t_ransac(pts, [m] * ncam, idx, thr, mixed=False)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    cnt, ind = ops.decode_best(best.cpu().numpy())
    # two-tier form of the same step: float32 bounds + float64 re-score of the candidates
    pts32 = ops.pts_to_f32(pts)
    ph = pts.cpu().numpy()
    rmax = [float(np.abs(ph[c * m : (c + 1) * m, :3]).max()) for c in range(ncam)]
    xmax = float(np.abs(ph[:, 3:5]).max())
    c2, i2, T2 = ops.t_best(pts, pts32, [m] * ncam, idx, thr, rmax, xmax)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c2, i2, T2 = ops.t_best(pts, pts32, [m] * ncam, idx, thr, rmax, xmax)
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / reps * 1e3
    out["t_ransac"] = {
        "workload": f"{ncam} cameras x {m} 2D-3D correspondences x {B} hypotheses, one launch",
        "mixed_ms": ms_mixed,
        "mixed_evals_per_s": ncam * float(m) * B / (ms_mixed * 1e-3),
        "mixed_counts_identical_to_f64": bool(torch.equal(counts_m, counts) and torch.equal(best_m, best)),
        "f64_ms": ms,
        "f64_evals_per_s": ncam * float(m) * B / (ms * 1e-3),
        "two_tier_ms": ms2,
        "two_tier_evals_per_s": ncam * float(m) * B / (ms2 * 1e-3),
        "two_tier_tflops_algorithmic_12_per_eval": 12.0 * ncam * m * B / (ms2 * 1e-3) / 1e12,
        "two_tier_equals_f64": bool((c2 == cnt).all() and (i2 == ind).all() and (T2 == T.cpu().numpy()).all()),
        "best_counts": [int(c) for c in cnt],
    }
    # C4-like: LM (10 iterations, mu = 0.01) and triangulation with per-camera sizes 1e2..1e4
    sizes = [100, 300, 1000, 3000, 10000, 5000, 2000, 800, 400, 200, 6000, 1500]
    Kc = synthetic.K_DATASET10
    cams = [synthetic.resection_scene(m=s, outlier_ratio=0.0, seed=seed + 200 + i, K=Kc) for i, s in enumerate(sizes)]
    pts = torch.cat([ops.pack_points(_capi.dev(c["R"].ravel()), _capi.dev(c["X"]), _capi.dev(c["x_norm"])) for c in cams])
    T0 = _capi.dev(np.stack([c["T_true"] + 0.02 for c in cams]))
    Kd = _capi.dev(np.asarray(Kc, float).ravel())
    ops.lm_translation(Kd, pts, sizes, T0, 10, 0.01)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        ops.lm_translation(Kd, pts, sizes, T0, 10, 0.01)
    b.record()
    torch.cuda.synchronize()
    out["lm"] = {"workload": f"12 cameras, {sum(sizes)} correspondences in total, 10 iterations, one launch", "ms": a.elapsed_time(b) / reps}
    sc = synthetic.two_view_scene(n=30000, inlier_ratio=1.0, planar_ratio=0.3, seed=seed + 7)
    P1 = _capi.dev(np.hstack([np.eye(3), np.zeros((3, 1))]).ravel())
    P2 = _capi.dev(np.hstack([sc["R"], sc["T"].reshape(3, 1)]).ravel())
    Kn = np.asarray(sc["K"], float)
    x1 = _capi.dev(np.linalg.inv(Kn) @ sc["x1"])
    x2 = _capi.dev(np.linalg.inv(Kn) @ sc["x2"])
    ops.triangulate(P1, P2, x1, x2)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        ops.triangulate(P1, P2, x1, x2)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    out["triangulation"] = {"workload": "30000 points, one launch", "ms": ms, "points_per_s": 30000 / (ms * 1e-3),
                            "hbm_bytes": 30000 * (32 + 32)}
    # descriptor matching (feature front-end): 2-NN + ratio test on the int8 matrix cores
    out["descriptor_matching"] = []
    for nq, nt in [(8000, 8000), (50000, 50000)]:
        g = torch.Generator(device="cuda").manual_seed(seed + nq)
        q8, qn = ops.descriptors_to_i8(torch.randint(0, 256, (nq, 128), dtype=torch.uint8, device="cuda", generator=g))
        t8, tn = ops.descriptors_to_i8(torch.randint(0, 256, (nt, 128), dtype=torch.uint8, device="cuda", generator=g))
        ops.knn2_ratio(q8, qn, t8, tn, 0.75)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            ops.knn2_ratio(q8, qn, t8, tn, 0.75)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        out["descriptor_matching"].append({"workload": f"{nq} query x {nt} train SIFT descriptors (uint8 x 128), k=2 + ratio test",
                                           "ms": ms, "int8_tops": 2.0 * 128 * nq * nt / (ms * 1e-3) / 1e12,
                                           "pairs_per_s": float(nq) * nt / (ms * 1e-3)})
    return out


def seed_sweep(torch, ops, _capi, synthetic, orc, n, B):
    """The headline step on scenes generated with seeds 0..4 (two steps in flight, 6 timed steps each): evals/s per
    seed and their median."""
    rates, winners = [], []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ws = [ops.eh_best_workspace(B, n) for _ in range(2)]
    pend = [ops.eh_best_pending() for _ in range(2)]
    for seed in range(5):
        sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=seed)
        thr = 1.0 / sc["K"][0][0]
        corr_h = ops.pack_corr(orc.normalize_K(sc["K"], sc["x1"]), orc.normalize_K(sc["K"], sc["x2"]))
        corr = _capi.dev(corr_h)
        corr32 = ops.corr_to_f32(corr)
        xmax = float(np.abs(corr_h).max())
        idx = ops.index_sets(seed, 0, B, n)
        torch.cuda.synchronize()

        def run(k):
            prev, out = None, None
            for i in range(k):
                job = ops.eh_best_submit(corr, corr32, idx, thr, 3 * thr, xmax, workspace=ws[i % 2], stream=streams[i % 2], pending=pend[i % 2])
                if prev is not None:
                    out = ops.eh_best_collect(prev)
                prev = job
            return ops.eh_best_collect(prev) if prev is not None else out

        run(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wE, wH, _ = run(6)
        torch.cuda.synchronize()
        rates.append(2.0 * n * B * 6 / (time.perf_counter() - t0))
        winners.append([list(wE), list(wH)])
    return {"seeds": [0, 1, 2, 3, 4], "evals_per_s": rates, "median_evals_per_s": float(np.median(rates)), "winners_E_H": winners}


def sweep(torch, ops, _capi, synthetic, orc, seed):
    """Throughput of the two-tier E+H scoring step over match-set sizes (one GPU)."""
    rows = []
    for n, B in [(2000, 1000000), (10000, 1000000), (50000, 400000), (200000, 100000), (200000, 12500000)]:
        sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=seed)
        thr = 1.0 / sc["K"][0][0]
        x1n = orc.normalize_K(sc["K"], sc["x1"])
        x2n = orc.normalize_K(sc["K"], sc["x2"])
        corr_h = ops.pack_corr(x1n, x2n)
        corr = _capi.dev(corr_h)
        corr32 = ops.corr_to_f32(corr)
        xmax = float(np.abs(corr_h).max())
        idx = ops.index_sets(seed, 0, B, n)  # generated on the device from (seed, hypothesis id)
        ce = torch.empty(B, dtype=torch.int32, device="cuda")
        ch = torch.empty(B, dtype=torch.int32, device="cuda")
        ws = ops.eh_best_workspace(B, n)
        for _ in range(3 if B * n < 5e11 else 1):
            ops.eh_best_native(corr, corr32, idx, thr, 3 * thr, xmax, workspace=ws)
        torch.cuda.synchronize()
        reps = 5 if B * n < 5e11 else 1
        t0 = time.perf_counter()
        for _ in range(reps):
            wE, wH, st = ops.eh_best_native(corr, corr32, idx, thr, 3 * thr, xmax, workspace=ws)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        rows.append({"matches": n, "hypotheses": B, "ms_per_step": dt * 1e3, "evals_per_s": 2.0 * n * B / dt,
                     "tflops_algorithmic": 57.0 * n * B / dt / 1e12, "winner_E": list(wE), "winner_H": list(wH), **st})
        del idx, ce, ch, ws
    return rows


def e2e(torch, synthetic, orc, seed, with_cpu):
    """End-to-end geometry (run_sfm.py:59-80) on a synthetic scene shaped like dataset 5:
    10 cameras, dataset-5 intrinsics, a few thousand matches per consecutive pair.  The OpenCV
    front-end is replaced by the scene's stand-in matcher on both sides; SIFT time is not part
    of either number."""
    from sfm_hip import pipeline, utils

    sc = synthetic.multi_view_scene(n_cams=10, n_points=6000, init_pair=(4, 6), seed=seed + 5)
    warm = synthetic.multi_view_scene(n_cams=3, n_points=400, init_pair=(0, 1), seed=seed + 6)
    utils.find_correspondences = warm["find_correspondences"]
    np.random.seed(seed + 10)
    pipeline.run_geometry(warm["K"], warm["x_pairs"], warm["init_pair_dict"], warm["img_paths"], 1.0)  # library load, allocator
    utils.find_correspondences = sc["find_correspondences"]
    np.random.seed(seed + 11)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = pipeline.run_geometry(sc["K"], sc["x_pairs"], sc["init_pair_dict"], sc["img_paths"], 1.0)
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    out = {"workload": "10 cameras, 6000 scene points, 4200 matches per consecutive pair, 4800 in the initial pair, 1 px threshold",
           "gpu_seconds": gpu_s, "inliers_initial_pair": int(got["inliers"].sum())}
    if with_cpu:
        np.random.seed(seed + 11)
        t0 = time.perf_counter()
        ref = orc.run_geometry(sc["K"], sc["x_pairs"], sc["init_pair_dict"], sc["img_paths"], 1.0, sc["find_correspondences"])
        out["cpu_oracle_seconds"] = time.perf_counter() - t0
        out["max_abs_diff_R"] = float(max(np.abs(a - b).max() for a, b in zip(got["absolute_rotations"], ref["absolute_rotations"])))
        out["max_abs_diff_T"] = float(max(np.abs(a - b).max() for a, b in zip(got["refined_Ts"], ref["refined_Ts"])))
        out["max_rel_diff_X0"] = float(np.abs(got["X0"] - ref["X0"]).max() / np.abs(ref["X0"]).max())
        out["inlier_masks_equal"] = bool((got["inliers"] == ref["inliers"]).all())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--matches", type=int, default=10000)
    ap.add_argument("--hyps", type=int, default=1000000, help="hypotheses per GPU")
    ap.add_argument("--cpu-sample", type=int, default=50000, help="hypotheses of the CPU baseline sample (0 = skip)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--index-source", choices=["auto", "legacy", "philox"], default="auto",
                    help="legacy: np.random.randint on the host, as the reference draws its samples; philox: generated on the device "
                    "from (seed, global hypothesis id), identical under any sharding; auto = legacy on one GPU, philox on several")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N>1: nccl (= RCCL over xGMI, one GPU per rank); gloo only to rehearse the "
                    "multi-rank path on a box with fewer GPUs than ranks (ranks then share GPUs)")
    ap.add_argument("--no-overlap", action="store_true", help="one step at a time on one stream (no overlap of a step's tail with the next step)")
    ap.add_argument("--cpu-all-cores", type=int, default=0, metavar="WORKERS",
                    help="also time the NumPy oracle on WORKERS host processes (hypothesis-range sharding) before the GPU is touched")
    ap.add_argument("--no-secondary", action="store_true", help="skip the T-RANSAC / LM / triangulation timings")
    ap.add_argument("--e2e", action="store_true", help="only run the synthetic dataset-5-shaped end-to-end geometry and print it")
    ap.add_argument("--e2e-cpu", action="store_true", help="with --e2e: also time the NumPy oracle end to end (minutes) and compare")
    ap.add_argument("--sweep", action="store_true", help="only run the match-count sweep (2k..200k matches) and print it")
    args = ap.parse_args()

    all_cores = None
    if args.cpu_all_cores > 0 and int(os.environ.get("RANK", "0")) == 0:
        # worker processes are started before this process initialises the GPU
        all_cores = cpu_baseline_all_cores(args.matches, args.seed, args.cpu_all_cores, 20000)

    import torch
    import torch.distributed as dist

    from oracle import sfm_oracle as orc
    from sfm_hip import _capi, ops, sharding, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    key_device = "cuda" if args.backend == "nccl" else "cpu"
    _capi.load_library()

    if args.e2e:
        print(json.dumps({"e2e": e2e(torch, synthetic, orc, args.seed, args.e2e_cpu)}))
        return
    if args.sweep:
        print(json.dumps({"sweep": sweep(torch, ops, _capi, synthetic, orc, args.seed)}))
        return

    # ---- workload: every rank holds the same matches and its own hypothesis range -----------
    n, B = args.matches, args.hyps
    sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=args.seed)
    thr = 1.0 / sc["K"][0][0]
    x1n = orc.normalize_K(sc["K"], sc["x1"])
    x2n = orc.normalize_K(sc["K"], sc["x2"])
    corr = _capi.dev(ops.pack_corr(x1n, x2n))
    # this rank scores the global hypothesis ids [rank*B, (rank+1)*B)
    base = rank * B
    index_source = args.index_source if args.index_source != "auto" else ("legacy" if world == 1 else "philox")
    if index_source == "legacy":
        np.random.seed(args.seed + 1000 * rank)
        idx_host = np.random.randint(0, n, size=12 * B).reshape(B, 12).astype(np.int32)
        idx = _capi.dev(idx_host)
    else:
        idx = ops.index_sets(args.seed, base, B, n)
        idx_host = idx.cpu().numpy() if world == 1 else None  # only the single-GPU run re-checks a sample on the host
    corr32 = ops.corr_to_f32(corr)
    xmax = float(np.abs(np.concatenate([x1n[:2], x2n[:2]])).max())
    ce = torch.empty(B, dtype=torch.int32, device="cuda")
    ch = torch.empty(B, dtype=torch.int32, device="cuda")
    best = torch.zeros(2, dtype=torch.int64, device=key_device)
    last_stats = {}
    workspace = ops.eh_best_workspace(B, n)

    # Steps are independent blocks of hypotheses, so two are kept in flight on two streams: the short tail of a step
    # (arg-max, leader and candidate re-scores, read-back) runs in the shadow of the next step's bulk kernel.
    depth = 1 if args.no_overlap else 2
    streams = [torch.cuda.Stream() for _ in range(depth)]
    workspaces = [workspace] + [ops.eh_best_workspace(B, n) for _ in range(depth - 1)]
    pendings = [ops.eh_best_pending() for _ in range(depth)]
    torch.cuda.synchronize()

    def finish(job):
        (cE, iE), (cH, iH), st = ops.eh_best_collect(job)
        last_stats.update(st)
        best.copy_(torch.tensor([sharding.pack_key(cE, base + iE), sharding.pack_key(cH, base + iH)], dtype=torch.int64))
        sharding.merge_best(best)  # RCCL all-reduce(max) of the packed winners when world > 1

    def run_steps(k):
        # every step: float32 bounds for all hypotheses -> candidates -> float64 re-score -> exact winners -> merge
        prev = None
        for i in range(k):
            job = ops.eh_best_submit(corr, corr32, idx, thr, 3 * thr, xmax, workspace=workspaces[i % depth], stream=streams[i % depth],
                                     pending=pendings[i % depth])
            if prev is not None:
                finish(prev)
            prev = job
        if prev is not None:
            finish(prev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=key_device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- kernel time of the dominant kernel, HIP events on the launch stream -----------------
    corrm = ops.corr_to_mfma(corr, thr)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, xmax, out=(ce, ch))
        b.record()
    torch.cuda.synchronize()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    ue_host = ce[: args.cpu_sample].cpu().numpy() if args.cpu_sample > 0 else None
    uh_host = ch[: args.cpu_sample].cpu().numpy() if args.cpu_sample > 0 else None
    # the two tests separately (each launch still solves both models): epipolar only, transfer only
    part_ms = {}
    for part, name in ((1, "epipolar"), (2, "transfer")):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in ev:
            a.record()
            ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, xmax, out=(ce, ch), part=part)
            b.record()
        torch.cuda.synchronize()
        part_ms[name] = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # the same loose bounds entirely on the vector ALU (the first level before the matrix-core form existed)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for a, b in ev:
        a.record()
        ops.eh_upper(corr, corr32, idx, thr, 3 * thr, xmax, out=(ce, ch), one_sided=True)
        b.record()
    torch.cuda.synchronize()
    loose_valu_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # the tight float32 bounds (symmetric epipolar + disc transfer test) on the same workload
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for a, b in ev:
        a.record()
        ops.eh_upper(corr, corr32, idx, thr, 3 * thr, xmax, out=(ce, ch), one_sided=False)
        b.record()
    torch.cuda.synchronize()
    tight_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # the all-float64 kernel on the same workload (exact counts for every hypothesis)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for a, b in ev:
        a.record()
        ops.eh_score(corr, idx, thr, 3 * thr, out=(ce, ch))
        b.record()
    torch.cuda.synchronize()
    f64_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # the same exact counts from the matrix cores + float32 with float64 inside the error bands
    ce2 = torch.empty_like(ce)
    ch2 = torch.empty_like(ch)
    mixed_ws = torch.empty(int(_capi.load_library().sfm_eh_score_mixed_workspace_bytes(B, n)), dtype=torch.uint8, device="cuda")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for a, b in ev:
        a.record()
        ops.eh_score_mixed(corr, idx, thr, 3 * thr, xmax, out=(ce2, ch2), workspace=mixed_ws)
        b.record()
    torch.cuda.synchronize()
    mixed_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    mixed_identical = bool(torch.equal(ce, ce2) and torch.equal(ch, ch2))
    del mixed_ws, ce2, ch2

    if rank == 0:
        evals_per_step = 2.0 * B * n * world
        flops_per_launch = float(B) * n * (FLOPS_PER_E_EVAL + FLOPS_PER_H_EVAL)
        achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12
        cnt, ind = ops.decode_best(best.cpu().numpy())
        out = {
            "metric": "RANSAC hyp x corr residual evals/sec",
            "value": evals_per_step * args.steps / dt,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32+f64",  # float32 pre-filter; float64 solvers and exact re-score of the candidates
            "data": "synthetic",
            "config": {
                "workload": f"synthetic two-view E+H RANSAC: {n} matches, 30% inliers, {B} hypotheses per GPU "
                "(8-pt E + 4-pt H solved and scored per hypothesis; exact winner = first maximum of the inlier counts), "
                "winners merged by all-reduce(max)",
                "matches": n,
                "hypotheses_per_gpu": B,
                "index_sets": "np.random.randint on the host (legacy stream)" if index_source == "legacy"
                else "Philox4x32-10 on the device, keyed by (seed, global hypothesis id)",
                "pixel_threshold": 1.0,
                "winner_E": {"count": int(cnt[0]), "index": int(ind[0])},
                "winner_H": {"count": int(cnt[1]), "index": int(ind[1])},
                "f64_rescored_candidates": dict(last_stats),
            },
            "roofline": {
                "bound": "valu_fp32",
                "kernel": "eh_upper_mfma_kernel (float32 bounds: one-sided epipolar + box transfer test; E x1 and H x1 on the bf16 "
                "matrix cores from three-piece splits, the rest on the vector ALU)",
                "achieved": achieved,
                "peak": PEAK_FP32_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                # HBM bytes per launch from rocprofv3 PMC passes of this same command (the latest summary under profiles/):
                # FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B) + WRITE_SIZE; algorithmic bytes are
                # 48 MB of index sets + 0.5 MB of matches read, 8 MB of counts written.
                "traffic": PMC_TRAFFIC_BYTES if (n, B) == (10000, 1000000) else None,
                "traffic_unit": "bytes per launch (PMC, measured in a separate rocprofv3 run)",
                "kernel_ms": kern_ms,
                "flops_per_launch": flops_per_launch,
                "note": "not HBM-bound: each match is reused by every hypothesis of a workgroup. `achieved` uses the "
                "algorithmic cost of the reference's tests (36 flops per E residual + 21 per H residual) and is priced against "
                "the float32 vector peak, which bounds the part that is not GEMM shaped; the kernel reaches the exact winner with "
                "cheaper implied tests, moves the two 3x3 products (24 of its 43 flops per pair) to the matrix cores and "
                "executes 12 vector instructions per pair (see executed_valu_tflops). profiles/microbench/valu_rates measures "
                "131.8 TFLOP/s for a pure v_fma_f32 stream on this chip",
                "executed_valu_tflops": 18.0 * B * n / (kern_ms * 1e-3) / 1e12,
                # frac can exceed 1 against the vector peak alone because 24 of the 57 algorithmic flops per pair run on the
                # matrix cores.  Ceiling with the matrix cores counted: the GEMM-shaped part of the reference's tests
                # (E x1, E^T x2, H x1 = 32 flops per pair) hidden entirely behind the vector work, the other 25 flops at the
                # float32 vector peak.
                "combined_ceiling": {"peak": PEAK_FP32_VALU_TFLOPS * 57.0 / 25.0, "frac": achieved / (PEAK_FP32_VALU_TFLOPS * 57.0 / 25.0),
                                     "valu_issue_frac_at_2p4GHz": 12.0 * B * n / 64 / 1024 / (kern_ms * 1e-3) / 1.2e9,
                                     "mfma_busy_frac_at_2p4GHz": 6.0 * B * n / 1024 / 1024 * 32 / 2.4e9 / (kern_ms * 1e-3)},
                "per_test": {"epipolar_only_kernel_ms": part_ms["epipolar"], "epipolar_evals_per_s": float(B) * n / (part_ms["epipolar"] * 1e-3),
                             "transfer_only_kernel_ms": part_ms["transfer"], "transfer_evals_per_s": float(B) * n / (part_ms["transfer"] * 1e-3),
                             "note": "each launch still solves both models"},
                "loose_valu_kernel": {"kernel": "eh_upper_kernel<true>", "kernel_ms": loose_valu_ms,
                                      "achieved": flops_per_launch / (loose_valu_ms * 1e-3) / 1e12,
                                      "frac": flops_per_launch / (loose_valu_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS},
                "tight_f32_kernel": {"kernel": "eh_upper_kernel<false>", "kernel_ms": tight_ms,
                                     "achieved": flops_per_launch / (tight_ms * 1e-3) / 1e12,
                                     "frac": flops_per_launch / (tight_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS},
                "f64_kernel": {"kernel": "eh_score_kernel", "kernel_ms": f64_ms, "achieved": flops_per_launch / (f64_ms * 1e-3) / 1e12,
                               "peak": PEAK_FP64_VALU_TFLOPS, "frac": flops_per_launch / (f64_ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS},
                "exact_mixed_kernel": {"kernel": "eh_score_mfma_kernel (the float64 counts of every hypothesis from matrix cores + float32, "
                                       "float64 inside the error bands)", "kernel_ms": mixed_ms,
                                       "achieved": flops_per_launch / (mixed_ms * 1e-3) / 1e12,
                                       "counts_identical_to_f64_kernel": mixed_identical},
            },
        }
        if args.cpu_sample > 0 and world == 1:
            base_res, idx_s, cE_s, cH_s = cpu_baseline(x1n, x2n, thr, n, idx_host[: args.cpu_sample])
            out["cpu_baseline"] = base_res
            if all_cores is not None:
                out["cpu_baseline_all_cores"] = all_cores
            # the CPU sample is the head of this rank's index stream: reuse it as a parity check
            ok = np.array([len(set(r[:8])) == 8 and len(set(r[8:])) == 4 for r in idx_s])
            gE = ce[: args.cpu_sample].cpu().numpy()
            gH = ch[: args.cpu_sample].cpu().numpy()
            iw = [int(ind[0]) - base, int(ind[1]) - base]
            wE, wH, _, _ = orc.score_EH_batch(x1n, x2n, idx_host[iw].astype(np.int64), thr)
            out["parity"] = {
                "checked_hypotheses": int(ok.sum()),
                "f64_count_mismatches_E": int((gE[ok] != cE_s[ok]).sum()),
                "f64_count_mismatches_H": int((gH[ok] != cH_s[ok]).sum()),
                "f32_upper_bound_violations_E": int((ue_host[ok] < cE_s[ok]).sum()),
                "f32_upper_bound_violations_H": int((uh_host[ok] < cH_s[ok]).sum()),
                "winner_counts_match_oracle": bool(wE[0] == cnt[0] and wH[1] == cnt[1]),
                "winner_is_first_max_of_f64_counts": bool(
                    int(ind[0]) - base == int((ce == ce.max()).nonzero()[0].item()) and int(cnt[0]) == int(ce.max().item())
                    and int(ind[1]) - base == int((ch == ch.max()).nonzero()[0].item()) and int(cnt[1]) == int(ch.max().item())),
            }
        if world == 1 and not args.no_secondary:
            try:  # the headline line must print even if a side measurement fails
                out["secondary"] = secondary_measurements(torch, ops, _capi, synthetic, args.seed)
                out["secondary"]["seed_sweep"] = seed_sweep(torch, ops, _capi, synthetic, orc, n, B)
            except Exception as e:  # noqa: BLE001
                out["secondary"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

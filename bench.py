#!/usr/bin/env python3
"""Benchmark of the robust-estimation hot path on MI355X.

Workload, one GPU (--workload c2, the default for N=1): the synthetic two-view E+H RANSAC configuration: 10 000
SIFT-like matches, 30 % inliers, 10^6 hypotheses.  One step = every hypothesis solved (8-point E and 4-point H model)
and BOTH of the reference's residual tests decided for every (hypothesis, match) pair (sfm_eh_score_mixed_opt with
SFM_EHX_EVERY_PAIR_FULL_TEST: the float64 inlier counts of every hypothesis, every pair through the full symmetric-epipolar
and disc tests), then the arg-max of the two count arrays (sfm_argmax_count_i32) and the merge of the winners.  `value`
counts those decided residual tests per second; `roofline` prices the same kernel with the algorithmic 57 flops per pair.
Two faster steps that return the same winners are reported beside it, never as `value`: the library's default form of the
same call, whose early loops count a pair out as soon as a cheaper implied test proves it to be no inlier
(`exact_counts_early_out_value`: still the float64 counts of every hypothesis), and the bound-driven arg-max step
(sfm_eh_best: matrix-core bounds for every pair, float64 only for the candidates; `effective_value`).

Workload, several GPUs (--workload c5, the default for N>1; also runs on one GPU): 200 000 matches x 10^8
hypotheses, STRONG scaling: the hypothesis ids [0, 10^8) are split into contiguous ranges, one per rank
(sfm_hip.sharding.shard_range); every rank generates the index sets of its ids on its own GPU (Philox keyed by
(seed, id): identical under any split), finds the exact winners of its range and the ranks merge them with one
all-reduce(max) of packed (count, index) keys per step.  A step is one pass over all 10^8 hypotheses.

Metric: hypothesis x correspondence residual evaluations per second (an E residual and an H
residual each count as one evaluation).  Inputs (matches and index sets) are resident in HBM
before the timed region.  Steps are independent blocks of hypotheses; two are kept in flight on
two HIP streams, every step is collected and merged inside the timed region (--no-overlap runs them strictly one
after the other).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FLOPS_PER_E_EVAL = 36  # algorithmic cost of one symmetric-epipolar residual test
FLOPS_PER_H_EVAL = 21  # one transfer-error residual test
FLOPS_PER_T_EVAL = 12  # one reprojection residual test (R X precomputed)
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X vector FP64 (half of the FP32 vector rate)
SPEC_CLOCK_HZ = 2.4e9
N_SIMD = 1024  # 256 CUs x 4
# a pure v_fma_f32 stream on this chip, 4 waves per SIMD, waves running to a common deadline: 2.04 cycles per instruction
# per SIMD at a held clock of 2.02 GHz (profiles/microbench/valu_issue_probe_mi355x.txt, pattern 0, waves/SIMD 4)
SUSTAINED_NS_PER_VALU_INSTRUCTION = 2.04 / 2.018


def load_profile_counters(matches, hyps):
    """Per-launch counters of the newest profiles/*_counters.json (written by profiles/summarize.py from separate
    rocprofv3 --pmc passes of this same command on the default workload).  Instruction counts and HBM traffic per
    launch are properties of kernel + workload, so they combine with the durations measured live here."""
    import glob

    if (matches, hyps) != (10000, 1000000):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def load_traffic():
    """The newest profiles/r*_traffic.json (profiles/tools/traffic_probe.py: the scoring calls of the default workloads,
    one launch at a time, under separate rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes): {kernel + grid: bytes}."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def traffic_of(table, prefix, suffix=None):
    """Memory-side bytes per CALL of the kernels whose name starts with prefix: the probe makes each call on one shape only,
    so every (instantiation, grid) row of that name is one launch of the call (the exact scorer launches one model at a
    time and the nearly empty last round of a grid apart): (sum over the rows, {row: bytes})."""
    rows = {k: v["bytes_per_launch"] for k, v in table.items() if k.startswith(prefix) and (suffix is None or suffix in k)}
    return (sum(rows.values()), rows) if rows else (None, None)


def kernel_counters(table, needle, per_call=False, kernel_ms=None):
    """per_call: counters summed over every launch of the kernel per full-size launch (a call of the dominant kernel may
    cut its last, nearly empty round off as a second small launch; the duration measured here covers both).
    kernel_ms: the launch duration measured in this run; one kernel serves several workloads of the profiled command, so the
    counters are taken from the group of profiled launches whose duration is within 1.4x of it, and there are none
    ({}: no issue figures rather than those of another workload) when the profile holds no such group."""
    if not table:
        return {}
    for name, c in table.items():
        if needle in name and "SQ_INSTS_VALU" in c:
            c = dict(c)
            if per_call:
                for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "FETCH_SIZE", "WRITE_SIZE"):
                    if k + "__per_call" in c:
                        c[k] = c[k + "__per_call"]
            elif kernel_ms is not None:
                groups = c.get("__clusters")
                if groups is None:  # a counters file without duration groups: usable only if its full-size launches match
                    ok = 1 / 1.4 <= c.get("SQ_INSTS_VALU__dur_us", 0.0) / (kernel_ms * 1e3) <= 1.4
                    return c if ok else {}
                for k, rows in groups.items():
                    near = min(rows, key=lambda r: abs(math.log(r[0] / (kernel_ms * 1e3))))
                    if not 1 / 1.4 <= near[0] / (kernel_ms * 1e3) <= 1.4:
                        return {}
                    c[k], c[k + "__dur_us"] = near[1], near[0]
            return c
    return {}


def kernel_counters_sum(table, needle, kernel_ms, also=None):
    """Counters of a library call that is made of one launch of each of several instantiations of a kernel (the exact E/H
    scorer runs one launch per model on large blocks).  Per instantiation that has launches of that size at all: the
    group of profiled launches, among those lasting between 0.15 and 1.0 x kernel_ms, with the most launches (the timing
    loop of this workload repeats its launch; other workloads of the profiled command do not).  The sums are used only
    if the summed durations agree with kernel_ms within 1.4x ({} otherwise, and when fewer than two instantiations qualify:
    kernel_counters handles a single one)."""
    if not table:
        return {}
    lo, hi = 0.15 * kernel_ms * 1e3, 1.0 * kernel_ms * 1e3
    entries = []
    for name, c in table.items():
        groups = c.get("__clusters", {})
        if also is not None and also not in name:  # `also`: a second part of the name (template arguments) that must be there too
            continue
        if needle in name and any(lo <= r[0] <= hi for r in groups.get("SQ_INSTS_VALU", [])):
            entries.append(groups)
    if len(entries) < 2:
        return {}
    total, dur = {}, 0.0
    for groups in entries:
        ref = max((r for r in groups["SQ_INSTS_VALU"] if lo <= r[0] <= hi), key=lambda r: (r[2], r[0]))
        for k, rows in groups.items():
            best = min(rows, key=lambda r: abs(math.log(r[0] / ref[0])))
            if not 1 / 1.4 <= best[0] / ref[0] <= 1.4:
                return {}
            total[k] = total.get(k, 0.0) + best[1]
            if k == "GRBM_GUI_ACTIVE":
                total["GRBM_GUI_ACTIVE__dur_us"] = total.get("GRBM_GUI_ACTIVE__dur_us", 0.0) + best[0]
        dur += ref[0]
    if not 1 / 1.4 <= dur / (kernel_ms * 1e3) <= 1.4:
        return {}
    return total


def issue_roofline(kernel_ms, c):
    """Counter-derived utilisation of the vector issue slots: every VALU wave-instruction occupies a SIMD for 2 cycles
    (64 lanes over a 32-lane SIMD), every bf16 32x32x16 MFMA occupies the matrix pipe for 32."""
    if not c or "SQ_INSTS_VALU" not in c:
        return None
    t = kernel_ms * 1e-3
    out = {
        "valu_instructions_per_launch": c["SQ_INSTS_VALU"],
        "valu_issue_frac_at_spec_clock": c["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * SPEC_CLOCK_HZ * t),
        "fma_equivalent_tflops": c["SQ_INSTS_VALU"] * 128.0 / t / 1e12,
        "valu_issue_frac_of_sustained_fma_stream": c["SQ_INSTS_VALU"] * SUSTAINED_NS_PER_VALU_INSTRUCTION * 1e-9 / (N_SIMD * t),
    }
    if "SQ_INSTS_MFMA" in c:
        out["mfma_instructions_per_launch"] = c["SQ_INSTS_MFMA"]
        out["mfma_pipe_busy_frac_at_spec_clock"] = c["SQ_INSTS_MFMA"] * 32.0 / (N_SIMD * SPEC_CLOCK_HZ * t)
    if "GRBM_GUI_ACTIVE" in c and c.get("GRBM_GUI_ACTIVE__dur_us"):
        out["clock_ghz_under_the_profiler"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (c["GRBM_GUI_ACTIVE__dur_us"] * 1e-6) / 1e9
    return out


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


class ExactScorer:
    """The exact step: sfm_eh_score_mixed (float64 inlier counts of EVERY hypothesis of a block, every (hypothesis, match)
    pair decided by the reference's two residual tests) + sfm_argmax_count_i32 on both count arrays + the read-back of
    the two packed winners.  Blocks are submitted to `depth` slots in turn (own stream, count arrays, workspace and
    page-locked result words per slot), so `depth` blocks can be in flight; collect(slot) waits for that slot's block."""

    def __init__(self, torch, _capi, ops, corr, thr, xmax, max_block, n, depth, full_test=True):
        self.torch, self.ops = torch, ops
        # full_test: every pair through the full symmetric-epipolar and disc tests (the unit of `value`); False: the library's
        # default, in which a pair that fails a cheaper implied test is counted out without the full test (same counts)
        self.full_test = full_test
        self.corr, self.thr, self.xmax = corr, thr, xmax
        lib = _capi.load_library()
        self.depth = depth
        self.streams = [torch.cuda.Stream() for _ in range(depth)]
        self.events = [torch.cuda.Event() for _ in range(depth)]
        self.ce = [torch.empty(max_block, dtype=torch.int32, device="cuda") for _ in range(depth)]
        self.ch = [torch.empty(max_block, dtype=torch.int32, device="cuda") for _ in range(depth)]
        self.ws = [torch.empty(int(lib.sfm_eh_score_mixed_workspace_bytes(max_block, n)), dtype=torch.uint8, device="cuda") for _ in range(depth)]
        self.best = [torch.zeros(2, dtype=torch.int64, device="cuda") for _ in range(depth)]
        self.host = [torch.zeros(2, dtype=torch.int64).pin_memory() for _ in range(depth)]

    def submit(self, slot, idx, first_id):
        """Queues one block of index sets (global hypothesis ids first_id ...) on the slot's stream; returns at once."""
        nb = idx.shape[0]
        with self.torch.cuda.stream(self.streams[slot]):
            ce, ch = self.ce[slot][:nb], self.ch[slot][:nb]
            self.best[slot].zero_()
            self.ops.eh_score_mixed(self.corr, idx, self.thr, 3 * self.thr, self.xmax, out=(ce, ch), workspace=self.ws[slot],
                                    every_pair_full_test=self.full_test)
            self.ops.argmax_count(ce, first_id, self.best[slot][0:1])
            self.ops.argmax_count(ch, first_id, self.best[slot][1:2])
            self.host[slot].copy_(self.best[slot], non_blocking=True)
            self.events[slot].record()

    def collect(self, slot):
        """Waits for the slot's block: its two packed winners (count << 32 | 0xFFFFFFFF - id) as Python ints."""
        self.events[slot].synchronize()
        return int(self.host[slot][0]), int(self.host[slot][1])

    def run(self, blocks):
        """All blocks [(idx, first_id), ...] with `depth` in flight; returns the merged winners of the two models."""
        keys = [0, 0]
        pending = []
        for i, (idx, first_id) in enumerate(blocks):
            slot = i % self.depth
            if len(pending) == self.depth:
                k = self.collect(pending.pop(0))
                keys = [max(keys[0], k[0]), max(keys[1], k[1])]
            self.submit(slot, idx, first_id)
            pending.append(slot)
        while pending:
            k = self.collect(pending.pop(0))
            keys = [max(keys[0], k[0]), max(keys[1], k[1])]
        return keys


def secondary_measurements(torch, ops, _capi, synthetic, seed, table=None):
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
        "roofline": {
            "bound": "valu_fp32",
            "kernel": "t_score_mfma_kernel (dx - thr*pz, dy and dx + thr*pz on the bf16 matrix cores, float32 sign test of their product form, float64 inside the error band; "
            "evaluates the reference's reprojection test for every pair); kernel_ms is the whole call (operand preparation, arg-max and winner "
            "kernels included: microseconds)",
            "kernel_ms": ms_mixed,
            "achieved": FLOPS_PER_T_EVAL * ncam * float(m) * B / (ms_mixed * 1e-3) / 1e12,
            "peak": PEAK_FP32_VALU_TFLOPS,
            "unit": "TFLOP/s",
            "frac": FLOPS_PER_T_EVAL * ncam * float(m) * B / (ms_mixed * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
            "flops_per_eval": FLOPS_PER_T_EVAL,
            "issue": issue_roofline(ms_mixed, kernel_counters(table, "t_score_mfma_kernel", kernel_ms=ms_mixed)),
            "traffic": traffic_of(load_traffic()[0], "t_score_mfma_kernel")[0],
            "traffic_unit": "bytes per launch, FETCH_SIZE x2 + WRITE_SIZE (profiles/tools/traffic_probe.py on this workload): the index pairs are "
            "read once per split of a camera's correspondence range (8 x 9.6 MB), the counts are added atomically once per split (38 MB), every XCD "
            "gathers sample points from all cameras' float64 points (8 x 29 MB) and streams its eighth of the match-side pieces (58 MB)",
        },
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


def replay_throughput(torch, synthetic, orc, seed):
    """The drop-in loop itself (sfm_hip.ransac_algorithm.run_replay = estimate_E_robust without the final unpacking) on a
    long run: 200 000 matches, 15 % inliers (the adaptive limit stays far away), capped at 2 * 10^6 iterations.  Index sets
    come from the global legacy np.random stream as in the reference; large blocks are scored through the matrix-core
    bounds, exact counts only where a bound exceeds a running best.  The same run with exact scoring of every hypothesis
    beside it; both must end in the same state."""
    from sfm_hip import ransac_algorithm as ra

    n, cap = 200000, 2000000
    sc = synthetic.two_view_scene(n=n, inlier_ratio=0.15, planar_ratio=0.5, seed=seed + 3)
    x1, x2 = orc.normalize_K(sc["K"], sc["x1"]), orc.normalize_K(sc["K"], sc["x2"])
    thr = 1.0 / sc["K"][0][0]
    out = {"workload": f"{n} matches, 15% inliers, {cap} iterations of the E/H loop, index sets from the legacy np.random stream"}
    states = []
    saved = ra.GpuBackend.BOUNDED_MIN_PAIRS
    try:
        for name, min_pairs in (("bound_driven", saved), ("exact_every_hypothesis", 1 << 62)):
            ra.GpuBackend.BOUNDED_MIN_PAIRS = min_pairs
            backend = ra.GpuBackend(x1, x2, thr)
            np.random.seed(seed + 21)
            ra.run_replay(n, backend, max_iterations=20000)  # warm-up: allocator, first launches
            np.random.seed(seed + 22)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = ra.run_replay(n, backend, max_iterations=cap)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name] = {"seconds": dt, "iterations": int(st.it), "evals_per_s": 2.0 * n * st.it / dt, "best_E": int(st.best_E),
                         "best_H": int(st.best_H), "record_events": len(st.events)}
            states.append((int(st.it), int(st.best_E), int(st.best_H), list(st.events), np.random.randint(0, 1 << 30)))
    finally:
        ra.GpuBackend.BOUNDED_MIN_PAIRS = saved
    out["identical_final_state_and_stream_position"] = bool(states[0] == states[1])
    return out


def seed_sweep(torch, ops, _capi, synthetic, orc, n, B):
    """The headline (exact) step on scenes generated with seeds 0..4 (two steps in flight, 6 timed steps each): evals/s per
    seed and their median."""
    rates, winners = [], []
    for seed in range(5):
        sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=seed)
        thr = 1.0 / sc["K"][0][0]
        corr_h = ops.pack_corr(orc.normalize_K(sc["K"], sc["x1"]), orc.normalize_K(sc["K"], sc["x2"]))
        corr = _capi.dev(corr_h)
        scorer = ExactScorer(torch, _capi, ops, corr, thr, float(np.abs(corr_h).max()), B, n, 2)
        idx = ops.index_sets(seed, 0, B, n)
        torch.cuda.synchronize()
        scorer.run([(idx, 0)] * 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keys = scorer.run([(idx, 0)] * 6)
        torch.cuda.synchronize()
        rates.append(2.0 * n * B * 6 / (time.perf_counter() - t0))
        cnt, ind = ops.decode_best(np.array(keys, dtype=np.int64))
        winners.append([[int(cnt[0]), int(ind[0])], [int(cnt[1]), int(ind[1])]])
        del scorer
    return {"seeds": [0, 1, 2, 3, 4], "evals_per_s": rates, "median_evals_per_s": float(np.median(rates)), "winners_E_H": winners}


def sweep(torch, ops, _capi, synthetic, orc, seed):
    """Throughput over match-set sizes (one GPU, one step at a time): the exact step (every pair decided by the reference's
    tests: sfm_eh_score_mixed_opt without early loops + arg-max), the same exact counts with the library's early loops, and the
    bound-driven arg-max step (sfm_eh_best); all three must name the same winners."""
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
        big = B * n >= 5e11
        reps = 1 if big else 5
        scorer = ExactScorer(torch, _capi, ops, corr, thr, xmax, B, n, 1)
        scorer.run([(idx, 0)] * (1 if big else 3))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keys = scorer.run([(idx, 0)] * reps)
        torch.cuda.synchronize()
        dte = (time.perf_counter() - t0) / reps
        cnt, ind = ops.decode_best(np.array(keys, dtype=np.int64))
        del scorer
        scorer = ExactScorer(torch, _capi, ops, corr, thr, xmax, B, n, 1, full_test=False)  # the same counts with the early loops
        scorer.run([(idx, 0)] * (1 if big else 3))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keys_early = scorer.run([(idx, 0)] * reps)
        torch.cuda.synchronize()
        dtee = (time.perf_counter() - t0) / reps
        del scorer
        ws = ops.eh_best_workspace(B, n)
        for _ in range(1 if big else 3):
            ops.eh_best_native(corr, corr32, idx, thr, 3 * thr, xmax, workspace=ws)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            wE, wH, st = ops.eh_best_native(corr, corr32, idx, thr, 3 * thr, xmax, workspace=ws)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        rows.append({"matches": n, "hypotheses": B,
                     "exact_ms_per_step": dte * 1e3, "exact_evals_per_s": 2.0 * n * B / dte, "exact_tflops_algorithmic": 57.0 * n * B / dte / 1e12,
                     "exact_frac_of_fp32_vector_peak": 57.0 * n * B / dte / 1e12 / PEAK_FP32_VALU_TFLOPS,
                     "exact_early_out_ms_per_step": dtee * 1e3, "exact_early_out_pairs_per_s": 2.0 * n * B / dtee,
                     "exact_early_out_winners_equal": bool(list(keys_early) == list(keys)),
                     "bound_driven_ms_per_step": dt * 1e3, "bound_driven_bounded_pairs_per_s": 2.0 * n * B / dt,
                     "winner_E": [int(cnt[0]), int(ind[0])], "winner_H": [int(cnt[1]), int(ind[1])],
                     "bound_driven_winners_equal": bool(list(wE) == [int(cnt[0]), int(ind[0])] and list(wH) == [int(cnt[1]), int(ind[1])]), **st})
        del idx, ws
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
    matcher = {"s": 0.0}

    def timed_matcher(*a):
        t = time.perf_counter()
        r = sc["find_correspondences"](*a)
        matcher["s"] += time.perf_counter() - t
        return r

    utils.find_correspondences = timed_matcher
    best = None
    for _ in range(3):
        matcher["s"] = 0.0
        np.random.seed(seed + 11)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = pipeline.run_geometry(sc["K"], sc["x_pairs"], sc["init_pair_dict"], sc["img_paths"], 1.0)
        torch.cuda.synchronize()
        gpu_s = time.perf_counter() - t0
        if best is None or gpu_s < best[0]:
            best = (gpu_s, matcher["s"])
    gpu_s, match_s = best
    out = {"workload": "10 cameras, 6000 scene points, 4200 matches per consecutive pair, 4800 in the initial pair, 1 px threshold",
           "gpu_seconds": gpu_s, "stand_in_matcher_seconds": match_s, "geometry_seconds": gpu_s - match_s,
           "note": "best of 3 seeded repetitions (identical results); the stand-in matcher is host Python of the test scene, not part of the path",
           "inliers_initial_pair": int(got["inliers"].sum())}
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


def sift_full_size_vs_oracle(torch, synthetic, seed):
    """One rendered 1936 x 1296 image through the GPU SIFT and through the NumPy restatement (about a minute on one core):
    the CPU time beside the GPU time, and the comparison at full size."""
    from oracle import sift as osift
    from sfm_hip import sift

    img = synthetic.render_views(n_cams=1, seed=seed)["images"][0]
    g = torch.from_numpy(img).cuda()
    sift.detect_and_compute(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sift.detect_and_compute(g)
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    kps, (gauss, _) = osift.detect(img, taps_fn=sift.gaussian_taps)
    desc = osift.describe(kps, gauss)
    cpu_s = time.perf_counter() - t0
    out = {"image": "1936 x 1296 rendered", "gpu_seconds": gpu_s, "cpu_oracle_seconds_one_core": cpu_s, "keypoints_gpu": int(res.records.shape[0]),
           "keypoints_oracle": len(kps)}
    if len(kps) == res.records.shape[0]:
        ko = np.array([k[:5] for k in kps], np.float64)
        out["positions_and_responses_identical"] = bool(np.array_equal(res.records[:, :2].astype(np.float64), ko[:, :2])
                                                        and np.array_equal(res.records[:, 4].astype(np.float64), ko[:, 4]))
        dang = np.abs(res.records[:, 3].astype(np.float64) - ko[:, 3])
        out["max_angle_difference_degrees"] = float(np.minimum(dang, 360 - dang).max())
        diff = np.abs(res.descriptors.astype(int) - desc.astype(int))
        out["max_descriptor_difference"] = int(diff.max())
        out["descriptor_entries_that_differ"] = float((diff != 0).mean())
    return out


def e2e_images(torch, synthetic, seed, n_cams=10, width=1936, height=1296):
    """The whole pipeline from image FILES, shaped like dataset 5 (10 images of 1936 x 1296, its intrinsics): rendered
    views of a textured two-plane scene are written as PNG files + cfg.yml, then the command-line driver's stages run:
    image decoding, SIFT detection + description (GPU), descriptor matching (GPU), geometry (GPU).  The reference's own
    dataset cannot be used (its images may not be copied and do not exist on the GPU box) and OpenCV is absent, so this is
    the measurable stand-in for 'dataset-5 end-to-end seconds'; rotations are checked against the renderer's ground truth."""
    import tempfile

    from sfm_hip import matching, pipeline, sift
    from sfm_hip.extract_sift import process_sift_for_image_pairs
    from sfm_hip.io import get_data

    t0 = time.perf_counter()
    views = synthetic.render_views(n_cams=n_cams, width=width, height=height, seed=seed)
    render_s = time.perf_counter() - t0
    out = {"workload": f"{n_cams} rendered images of {width} x {height} (dataset-5 intrinsics), PNG files + cfg.yml, 1 px threshold",
           "render_seconds_not_counted": render_s}
    with tempfile.TemporaryDirectory() as tmp:
        where = os.path.join(tmp, "data", "syn")
        synthetic.write_image_dataset(where, views, init_pair=(4, 6) if n_cams >= 7 else (0, 1))
        sift.detect_and_compute(views["images"][0][:256, :256])  # warm-up: library load, allocator
        best = None
        for rep in range(2):
            matching._features.clear()
            for f in os.listdir(where):
                if f.endswith(".pkl"):
                    os.remove(os.path.join(where, f))
            stage = {"decode": 0.0, "sift": 0.0, "images_detected": 0}
            orig_read, orig_detect = sift.read_gray, sift.detect_and_compute

            def timed_read(*a, **k):
                t = time.perf_counter()
                r = orig_read(*a, **k)
                stage["decode"] += time.perf_counter() - t
                return r

            def timed_detect(*a, **k):
                t = time.perf_counter()
                r = orig_detect(*a, **k)
                torch.cuda.synchronize()
                stage["sift"] += time.perf_counter() - t
                stage["images_detected"] += 1
                return r

            sift.read_gray, sift.detect_and_compute = timed_read, timed_detect
            try:
                np.random.seed(seed + 31)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                K, img_paths, init_pair = get_data(where)
                x_pairs, init_pair_dict = process_sift_for_image_pairs(img_paths=img_paths, init_pair=init_pair, dataset="syn")
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                front = dict(stage)
                res = pipeline.run_geometry(K, x_pairs, init_pair_dict, img_paths, 1.0)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            finally:
                sift.read_gray, sift.detect_and_compute = orig_read, orig_detect
            if best is None or t2 - t0 < best["total_seconds"]:
                feats = [v[0].shape[0] for v in matching._features.values()]
                best = {"total_seconds": t2 - t0, "front_end_seconds": t1 - t0, "image_decode_cpu_seconds_on_host_threads": front["decode"],
                        "sift_seconds": front["sift"], "images_detected": stage["images_detected"],
                        "front_end_seconds_outside_sift": t1 - t0 - front["sift"],
                        "geometry_seconds_incl_2d3d_matching": t2 - t1,
                        "sift_seconds_inside_geometry": stage["sift"] - front["sift"], "keypoints_per_image_min_max": [int(min(feats)), int(max(feats))],
                        "matches_per_pair_min_max": [int(min(x.shape[1] for x in x_pairs)), int(max(x.shape[1] for x in x_pairs))]}
        R = res["absolute_rotations"]
        errs = []
        for c in range(1, n_cams):
            Rt = views["R_true"][c] @ views["R_true"][c - 1].T
            Rr = R[c] @ R[c - 1].T
            errs.append(float(np.degrees(np.arccos(np.clip((np.trace(Rt.T @ Rr) - 1) / 2, -1, 1)))))
        best["relative_rotation_errors_degrees"] = [round(e, 3) for e in errs]
    out.update(best)
    out["note"] = "best of 2 repetitions; every image is decoded (on host threads, ahead of its use) and detected once; the reference detects each image three times"
    return out


def e2e_dataset5_correspondences(torch, reps=3):
    """Seconds of the geometry stages (pipeline.run_geometry: nine relative poses by E/H RANSAC, the initial pair's cloud,
    ten 2-point translation RANSACs, Levenberg-Marquardt, triangulation of every pair) on the REAL-image correspondences of
    the reference's dataset 5: tests/golden/dataset5_geometry.npz holds the matches that this repository's CPU front-end
    derived from the ten photographs where they lie beside the reference (the images themselves cannot travel); the 2D-3D
    matches are the stored ones, so descriptor matching is not inside the time.  The outcome of the seed is checked against
    the oracle's stored one (a seed none of whose record events sits on a degenerate sample)."""
    import importlib.util

    from sfm_hip import pipeline, utils

    spec = importlib.util.spec_from_file_location("dataset5_fixture", os.path.join(ROOT, "tests", "golden", "dataset5_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g, K, x_pairs, init, paths, matcher = mod.load_fixture()
    clean = [j for j in range(int(g["n_seeds"])) if int(g[f"s{j}_degenerate_record_events"]) == 0]
    j = clean[0]
    tag = f"s{j}_"
    orig = utils.find_correspondences
    utils.find_correspondences = matcher(tag)
    try:
        times = []
        for _ in range(reps + 1):  # the first run is the warm-up
            np.random.seed(int(g[tag + "seed"]))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = pipeline.run_geometry(K, x_pairs, init, paths, 1.0)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        same = bool(int(np.random.randint(0, 1 << 30)) == int(g[tag + "next_draw"])
                    and all(np.allclose(a, b, atol=1e-9) for a, b in zip(got["absolute_rotations"], g[tag + "abs_rot"]))
                    and np.array_equal(got["inliers"], g[tag + "inliers"])
                    and all(np.allclose(np.reshape(a, 3), b, rtol=1e-7, atol=1e-9) for a, b in zip(got["refined_Ts"], g[tag + "ref_T"])))
    finally:
        utils.find_correspondences = orig
    return {"workload": "geometry of dataset 5 (10 photographs of 1936 x 1296) on stored real-image correspondences: "
                        f"{min(x.shape[1] for x in x_pairs)} to {max(x.shape[1] for x in x_pairs)} matches per consecutive pair, "
                        f"{init['x_init'][0].shape[1]} on the initial pair, 1 px threshold, seed {int(g[tag + 'seed'])}",
            "geometry_seconds": min(times[1:]), "geometry_seconds_all_runs": [round(t, 4) for t in times[1:]],
            "ransac_iterations_of_the_ten_loops": [int(v) for v in g[tag + "iterations"]],
            "identical_to_oracle_outcome": same,
            "front_end": "not inside this time: features and matches are the fixture's (oracle/sift.py + oracle/matching.py on the CPU); "
                         "e2e_rendered_images times this library's own GPU front-end on files of the same shape",
            "correspondences_from": "this repository's CPU SIFT restatement, not cv2: parity with the reference's own matches is unpinned"}


C5_MATCHES, C5_HYPOTHESES, C5_BLOCK = 200000, 100000000, 12500000


def run_c5(args, torch, dist, orc, _capi, ops, sharding, synthetic, world, rank, key_device):
    """200 000 matches x 10^8 hypotheses, strong scaling over the ranks (see the module docstring).  --c5-mode exact
    (default): every pair decided by the reference's tests (the unit of `value` on one GPU as well); --c5-mode bound: the
    bound-driven arg-max step (matrix-core bounds for every pair, float64 for the candidates; the same winners)."""
    n, total = args.matches or C5_MATCHES, args.hyps or C5_HYPOTHESES
    sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=args.seed)
    thr = 1.0 / sc["K"][0][0]
    x1n = orc.normalize_K(sc["K"], sc["x1"])
    x2n = orc.normalize_K(sc["K"], sc["x2"])
    corr = _capi.dev(ops.pack_corr(x1n, x2n))
    corr32 = ops.corr_to_f32(corr)
    lo, hi = sharding.shard_range(total, rank, world)
    # this rank's ids in blocks of at most C5_BLOCK hypotheses; the index sets are generated here, on this GPU, and stay
    # resident (48 bytes per hypothesis: 4.8 GB for all 10^8 on one GPU)
    blocks = [(a, min(hi, a + C5_BLOCK)) for a in range(lo, hi, C5_BLOCK)]
    idx = [ops.index_sets(args.seed, a, b - a, n) for a, b in blocks]
    depth = 1 if args.no_overlap else 2
    bmax = max(b - a for a, b in blocks)
    exact = args.c5_mode in ("exact", "early")
    best = torch.zeros(2, dtype=torch.int64, device=key_device)
    stats = {"candidates_e": 0, "candidates_h": 0, "tightened": 0, "rescored_f64": 0}
    if exact:
        scorer = ExactScorer(torch, _capi, ops, corr, thr, 0.0, bmax, n, depth, full_test=args.c5_mode == "exact")
    else:
        streams = [torch.cuda.Stream() for _ in range(depth)]
        workspaces = [ops.eh_best_workspace(bmax, n) for _ in range(depth)]
        pendings = [ops.eh_best_pending() for _ in range(depth)]
    torch.cuda.synchronize()

    def step():
        if exact:
            keys = scorer.run([(idx[i], a) for i, (a, b) in enumerate(blocks)])  # keys carry global hypothesis ids
        else:
            keys = [0, 0]
            jobs = []

            def finish(job, base):
                (cE, iE), (cH, iH), st = ops.eh_best_collect(job)
                for k in stats:
                    stats[k] += st[k]
                keys[0] = max(keys[0], sharding.pack_key(cE, base + iE))
                keys[1] = max(keys[1], sharding.pack_key(cH, base + iH))

            for i, (a, b) in enumerate(blocks):
                job = ops.eh_best_submit(corr, corr32, idx[i], thr, 3 * thr, 0.0, workspace=workspaces[i % depth], stream=streams[i % depth],
                                         pending=pendings[i % depth])
                jobs.append((job, a))
                if len(jobs) == depth:
                    finish(*jobs.pop(0))
            while jobs:
                finish(*jobs.pop(0))
        best.copy_(torch.tensor(keys, dtype=torch.int64))
        sharding.merge_best(best)  # RCCL all-reduce(max) of the packed winners

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def progress(what, i, of):  # a step of this workload takes seconds: a line per step on stderr shows that the run is alive
        if rank == 0:
            print(f"[bench c5] {what} step {i + 1} of {of} submitted, {time.perf_counter() - t_start:.1f} s", file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    for i in range(args.warmup):
        step()
        progress("warm-up", i, args.warmup)
    sync()
    for k in stats:
        stats[k] = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        progress("timed", i, args.steps)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=key_device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank != 0:
        return
    cnt, ind = ops.decode_best(best.cpu().numpy())
    value = 2.0 * n * total * args.steps / dt
    full = args.c5_mode == "exact"
    how = ("every (hypothesis, match) pair decided by the reference's two residual tests (sfm_eh_score_mixed_opt without early loops: the float64 "
           "inlier counts of every hypothesis), arg-max of the counts" if full else
           "the float64 inlier counts of every hypothesis with the library's early loops (sfm_eh_score_mixed: a pair that fails a cheaper implied "
           "test skips the full test; pairs decided, not full residual evaluations), arg-max of the counts" if exact else
           "residual evaluations counted as bounded pairs: every (hypothesis, match) pair takes the matrix-core bound tests, candidates are re-scored "
           "in float64; exact winner = first maximum of the float64 inlier counts")
    out = {
        "metric": "RANSAC hyp x corr residual evals/sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32+f64",
        "data": "synthetic",
        "config": {
            "workload": f"synthetic two-view E+H RANSAC: {n} matches, 30% inliers, {total} hypotheses in total split into contiguous id ranges over "
            f"{world} GPU(s) (8-pt E + 4-pt H solved per hypothesis; {how}), winners merged by one all-reduce(max) of packed keys per step",
            "mode": args.c5_mode,
            "matches": n,
            "hypotheses_total": total,
            "hypotheses_this_rank": hi - lo,
            "blocks_per_rank": len(blocks),
            "index_sets": "Philox4x32-10 on the device, keyed by (seed, global hypothesis id)",
            "pixel_threshold": 1.0,
            "winner_E": {"count": int(cnt[0]), "index": int(ind[0])},
            "winner_H": {"count": int(cnt[1]), "index": int(ind[1])},
        },
        "roofline": {"bound": "valu_fp32", "kernel": "eh_score_mfma2_kernel" if exact else "eh_upper_mfma2_kernel",
                     "achieved": (FLOPS_PER_E_EVAL + FLOPS_PER_H_EVAL) / 2.0 * value / world / 1e12 if full else None,
                     "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                     "frac": (FLOPS_PER_E_EVAL + FLOPS_PER_H_EVAL) / 2.0 * value / world / 1e12 / PEAK_FP32_VALU_TFLOPS if full else None,
                     "traffic": None,
                     "note": "per GPU, from the whole-step rate (57 algorithmic flops per pair); the kernel's own event-timed duration, its counters "
                     "and the CPU baseline are those of the one-GPU default run (`python bench.py`): this line is the scaling measurement"},
    }
    if not exact:
        out["config"]["f64_rescored_candidates_per_step"] = {k: v / args.steps for k, v in stats.items()}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 60 for c2, 6 for c5, whose steps take seconds)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default: 15 for c2, 2 for c5)")
    ap.add_argument("--workload", choices=["auto", "c2", "c5"], default="auto",
                    help="c2: 10 000 matches x 10^6 hypotheses per GPU (weak scaling); c5: 200 000 matches x 10^8 hypotheses in total, split over "
                    "the GPUs (strong scaling); auto = c2 on one GPU, c5 on several")
    ap.add_argument("--c5-mode", choices=["exact", "early", "bound"], default="exact",
                    help="c5 only: exact = every pair decided by the reference's tests (the unit of `value`); early = the same exact counts with the "
                    "library's early loops (pairs that fail a cheaper implied test skip the full test); bound = the bound-driven arg-max step")
    ap.add_argument("--matches", type=int, default=0, help="0 = the workload's own size")
    ap.add_argument("--hyps", type=int, default=0, help="c2: hypotheses per GPU; c5: hypotheses in total; 0 = the workload's own size")
    ap.add_argument("--cpu-sample", type=int, default=50000, help="hypotheses of the CPU baseline sample (0 = skip)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--index-source", choices=["auto", "legacy", "philox"], default="auto",
                    help="legacy: np.random.randint on the host, as the reference draws its samples; philox: generated on the device "
                    "from (seed, global hypothesis id), identical under any sharding; auto = legacy on one GPU, philox on several")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N>1: nccl (= RCCL over xGMI, one GPU per rank); gloo only to rehearse the "
                    "multi-rank path on a box with fewer GPUs than ranks (ranks then share GPUs)")
    ap.add_argument("--no-overlap", action="store_true", help="one step at a time on one stream (no overlap of a step's tail with the next step)")
    ap.add_argument("--cpu-all-cores", type=int, default=-1, metavar="WORKERS",
                    help="also time the NumPy oracle on WORKERS host processes (hypothesis-range sharding) before the GPU is touched; "
                    "-1 = this job's share of the host cores (OMP_NUM_THREADS when set, else os.cpu_count()), 0 = skip")
    ap.add_argument("--no-secondary", action="store_true", help="skip the T-RANSAC / LM / triangulation timings")
    ap.add_argument("--e2e", action="store_true", help="only run the synthetic dataset-5-shaped end-to-end geometry and print it")
    ap.add_argument("--e2e-cpu", action="store_true", help="with --e2e: also time the NumPy oracle end to end (minutes) and compare")
    ap.add_argument("--e2e-images", action="store_true",
                    help="only run the pipeline from rendered image files shaped like dataset 5 (SIFT + matching + geometry) and print it")
    ap.add_argument("--sift-vs-oracle", action="store_true",
                    help="only run one full-size image through the GPU SIFT and the NumPy restatement (about a minute) and print it")
    ap.add_argument("--sweep", action="store_true", help="only run the match-count sweep (2k..200k matches) and print it")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    workload = args.workload if args.workload != "auto" else ("c2" if world_env == 1 else "c5")
    if args.steps is None:
        args.steps = 60 if workload == "c2" else 6
    if args.warmup is None:
        args.warmup = 15 if workload == "c2" else 2
    all_cores = None
    side_runs = args.e2e or args.sweep or args.e2e_images or args.sift_vs_oracle
    if args.cpu_all_cores < 0:
        try:
            share = int(os.environ.get("OMP_NUM_THREADS", "0"))
        except ValueError:
            share = 0
        args.cpu_all_cores = min(os.cpu_count() or 1, share) if share > 0 else (os.cpu_count() or 1)
    if workload == "c2" and world_env == 1 and not side_runs and args.cpu_sample > 0 and args.cpu_all_cores > 0:
        # worker processes are started before this process initialises the GPU
        all_cores = cpu_baseline_all_cores(args.matches or 10000, args.seed, args.cpu_all_cores, 20000)

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
    if args.sift_vs_oracle:
        print(json.dumps({"sift_vs_oracle": sift_full_size_vs_oracle(torch, synthetic, args.seed)}))
        return
    if args.e2e_images:
        print(json.dumps({"e2e_images": e2e_images(torch, synthetic, args.seed)}))
        return
    if args.sweep:
        print(json.dumps({"sweep": sweep(torch, ops, _capi, synthetic, orc, args.seed)}))
        return

    if workload == "c5":
        run_c5(args, torch, dist, orc, _capi, ops, sharding, synthetic, world, rank, key_device)
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- workload: every rank holds the same matches and its own hypothesis range -----------
    n, B = args.matches or 10000, args.hyps or 1000000
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
    depth = 1 if args.no_overlap else 2

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: exact steps.  Steps are independent blocks of hypotheses, so two are kept in flight on two
    # streams: the short tail of a step (arg-max kernels, read-back) and the ramp-down of its scoring kernel run in the
    # shadow of the next step's scoring kernel.  Every step is collected and merged inside the timed region.
    scorer = ExactScorer(torch, _capi, ops, corr, thr, xmax, B, n, depth)

    def merge(keys):
        best.copy_(torch.tensor(keys, dtype=torch.int64))
        sharding.merge_best(best)  # RCCL all-reduce(max) of the packed winners when world > 1

    def run_exact_steps(k, d):
        pending = []
        for i in range(k):
            if len(pending) == d:
                merge(scorer.collect(pending.pop(0)))
            scorer.submit(i % d, idx, base)
            pending.append(i % d)
        while pending:
            merge(scorer.collect(pending.pop(0)))

    run_exact_steps(args.warmup, depth)
    sync()
    t0 = time.perf_counter()
    run_exact_steps(args.steps, depth)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=key_device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    exact_winner = ops.decode_best(best.cpu().numpy())

    # the same exact steps strictly one after the other on one stream
    no_overlap_ms = None
    if depth > 1 and world == 1:
        run_exact_steps(2, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_exact_steps(args.steps, 1)
        torch.cuda.synchronize()
        no_overlap_ms = (time.perf_counter() - t1) / args.steps * 1e3

    # ---- kernel time of the dominant kernel: HIP events around sfm_eh_score_mixed alone, on the stream it is launched on
    mixed_ws = torch.empty(int(_capi.load_library().sfm_eh_score_mixed_workspace_bytes(B, n)), dtype=torch.uint8, device="cuda")
    ce2 = torch.empty_like(ce)
    ch2 = torch.empty_like(ch)
    ops.eh_score_mixed(corr, idx, thr, 3 * thr, xmax, out=(ce2, ch2), workspace=mixed_ws, every_pair_full_test=True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        ops.eh_score_mixed(corr, idx, thr, 3 * thr, xmax, out=(ce2, ch2), workspace=mixed_ws, every_pair_full_test=True)
        b.record()
    torch.cuda.synchronize()
    mixed_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # ---- the same exact counts with the library's early loops (not `value`): a pair that fails a cheaper implied test (one-sided
    # epipolar distance; x part of the disc test) is proven to be no float64 inlier and skips the full test
    ce3 = torch.empty_like(ce)
    ch3 = torch.empty_like(ch)
    ops.eh_score_mixed(corr, idx, thr, 3 * thr, xmax, out=(ce3, ch3), workspace=mixed_ws, every_pair_full_test=False)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        ops.eh_score_mixed(corr, idx, thr, 3 * thr, xmax, out=(ce3, ch3), workspace=mixed_ws, every_pair_full_test=False)
        b.record()
    torch.cuda.synchronize()
    early_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    early_identical = bool(torch.equal(ce3, ce2) and torch.equal(ch3, ch2))
    del ce3, ch3
    early_scorer = ExactScorer(torch, _capi, ops, corr, thr, xmax, B, n, depth, full_test=False)
    early_best = torch.zeros(2, dtype=torch.int64, device=key_device)

    def run_early_steps(k, d):
        pending = []
        for i in range(k):
            if len(pending) == d:
                early_best.copy_(torch.tensor(early_scorer.collect(pending.pop(0)), dtype=torch.int64))
                sharding.merge_best(early_best)
            early_scorer.submit(i % d, idx, base)
            pending.append(i % d)
        while pending:
            early_best.copy_(torch.tensor(early_scorer.collect(pending.pop(0)), dtype=torch.int64))
            sharding.merge_best(early_best)

    run_early_steps(args.warmup, depth)
    sync()
    t1 = time.perf_counter()
    run_early_steps(args.steps, depth)
    sync()
    early_dt = time.perf_counter() - t1
    early_winner = ops.decode_best(early_best.cpu().numpy())
    del early_scorer

    # ---- the bound-driven arg-max step (not `value`): matrix-core bounds for every pair -> candidates -> float64 re-score
    # -> the same exact winners; two steps in flight as above
    last_stats = {}
    eff_best = torch.zeros(2, dtype=torch.int64, device=key_device)
    streams = [torch.cuda.Stream() for _ in range(depth)]
    workspaces = [ops.eh_best_workspace(B, n) for _ in range(depth)]
    pendings = [ops.eh_best_pending() for _ in range(depth)]
    torch.cuda.synchronize()

    def finish(job):
        (cE, iE), (cH, iH), st = ops.eh_best_collect(job)
        last_stats.update(st)
        eff_best.copy_(torch.tensor([sharding.pack_key(cE, base + iE), sharding.pack_key(cH, base + iH)], dtype=torch.int64))
        sharding.merge_best(eff_best)

    def run_bound_steps(k, d):
        prev = None
        for i in range(k):
            job = ops.eh_best_submit(corr, corr32, idx, thr, 3 * thr, xmax, workspace=workspaces[i % d], stream=streams[i % d], pending=pendings[i % d])
            if prev is not None:
                finish(prev)
            prev = job
        if prev is not None:
            finish(prev)

    run_bound_steps(max(2, args.warmup // 2), depth)
    sync()
    t1 = time.perf_counter()
    run_bound_steps(args.steps, depth)
    sync()
    eff_dt = time.perf_counter() - t1
    tmax = torch.tensor([eff_dt], dtype=torch.float64, device=key_device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    eff_dt = float(tmax.item())
    eff_no_overlap_ms = None
    if depth > 1 and world == 1:
        run_bound_steps(2, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_bound_steps(args.steps, 1)
        torch.cuda.synchronize()
        eff_no_overlap_ms = (time.perf_counter() - t1) / args.steps * 1e3

    # ---- the other kernels of the family on the same workload, each event-timed alone ----------
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
    # the two launches of that call separately: epipolar only, transfer only
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
    mixed_identical = bool(torch.equal(ce, ce2) and torch.equal(ch, ch2))
    del mixed_ws, ce2, ch2

    if rank == 0:
        evals_per_step = 2.0 * B * n * world
        flops_per_launch = float(B) * n * (FLOPS_PER_E_EVAL + FLOPS_PER_H_EVAL)
        value = evals_per_step * args.steps / dt
        cnt, ind = exact_winner
        ecnt, eind = ops.decode_best(eff_best.cpu().numpy())
        table, counters_file = load_profile_counters(n, B)
        traffic_table, traffic_file = load_traffic()
        default_shape = (n, B) == (10000, 1000000)
        # the instantiations without early loops end in ", false>" (a traffic file from before the early loops existed has no such suffix)
        traffic, traffic_parts = (traffic_of(traffic_table, "eh_score_mfma2_kernel", ", false> grid") if default_shape else (None, None))
        if default_shape and traffic is None and not any(", true> grid" in k for k in traffic_table if k.startswith("eh_score_mfma2_kernel")):
            traffic, traffic_parts = traffic_of(traffic_table, "eh_score_mfma2_kernel")
        bound_issue = issue_roofline(kern_ms, kernel_counters_sum(table, "eh_upper_mfma2_kernel", kern_ms)
                                     or kernel_counters(table, "eh_upper_mfma_kernel<true, true", per_call=True))
        # the instantiations without early loops are <..., false>; a counters file from before the early loops existed has no such argument
        has_forms = bool(table) and any("eh_score_mfma2_kernel" in k and ", true>(" in k for k in table)
        exact_issue = issue_roofline(mixed_ms, kernel_counters_sum(table, "eh_score_mfma2_kernel", mixed_ms, ", false>(" if has_forms else None)
                                     or kernel_counters(table, "eh_score_mfma2_kernel", kernel_ms=mixed_ms))
        if bound_issue is not None and default_shape:
            bound_issue["memory_bytes_per_call"] = (traffic_of(traffic_table, "eh_upper_mfma2_kernel")[0]
                                                    or traffic_of(traffic_table, "eh_upper_mfma_kernel<true, true, true>")[0])
        achieved = flops_per_launch / (mixed_ms * 1e-3) / 1e12
        roofline = {
            "bound": "valu_fp32",
            "kernel": "eh_score_mfma2_kernel (one launch per model: <true, false, 3, false> for E and <false, true, 4, false> for H, the instantiations "
            "WITHOUT early loops; the bilinear form "
            "s = x2^T E x1, the squared line norms and the row H x1 on the bf16 matrix cores from three-piece splits, the reference's symmetric "
            "epipolar and transfer tests in float32 for every pair, float64 for the pairs inside the rigorous error band; float64 model solve "
            "included)",
            "achieved": achieved,
            "peak": PEAK_FP32_VALU_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_VALU_TFLOPS,
            "frac_from_value": value / world * (FLOPS_PER_E_EVAL + FLOPS_PER_H_EVAL) / 2.0 / 1e12 / PEAK_FP32_VALU_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "bytes per call (all its launches): FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 --pmc passes over "
            "profiles/tools/traffic_probe.py (this library call on this shape, one launch at a time; the file named in traffic_from).  Each "
            "model's launch reads the 48 MB of index rows (whole 128-byte lines are fetched although a model uses 32 or 16 of a row's 48 "
            "bytes) and the 2 MB of match operands once per XCD (8 L2s); 8 MB of counts are written; the last, nearly empty round of the E "
            "grid is launched apart in 11 pieces (its 67 tiles read their index rows 11 times)",
            "traffic_per_launch": traffic_parts,
            "traffic_from": traffic_file,
            "counters_from": counters_file,
            "kernel_ms": mixed_ms,
            "definition": "achieved = 57 algorithmic flops (36 per symmetric-epipolar + 21 per transfer residual of the reference's formulas) x "
            "hypotheses x matches / the event-timed duration of one sfm_eh_score_mixed call (operand preparation + the scoring launches); "
            "frac = achieved / 157.3 TFLOP/s; frac_from_value is the same quotient from the whole-step rate `value`",
            "counts_identical_to_f64_kernel": mixed_identical,
            "issue": exact_issue,
            "issue_note": "share of vector issue slots, from SQ_INSTS_VALU at 2 cycles each; v_alignbit, v_min3, v_min/v_max, v_cmp and v_cndmask issue at "
            "4 cycles and v_rsq at 8 on gfx950 (profiles/microbench/valu_issue_probe), so the slots really occupied are above this figure",
            "f64_kernel": {"kernel": "eh_score_kernel (every test in float64)", "kernel_ms": f64_ms,
                           "achieved": flops_per_launch / (f64_ms * 1e-3) / 1e12, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                           "frac": flops_per_launch / (f64_ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS},
        }
        effective = {
            "what": "the bound-driven arg-max step sfm_eh_best: matrix-core bound tests (one-sided epipolar test s^2 < k1 n1 + c whole from the "
            "matrix cores, 2 vector instructions + 4/1024 matrix instructions per pair; box transfer test, 6 + 3/1024; one launch per test) for "
            "every pair, exact float64 counts only for the candidates whose bound reaches "
            "the best exact count; returns the same winners as the exact step, but does NOT evaluate the reference's tests for every pair, so its "
            "rate of bounded pairs is not `value`",
            "winners_equal_exact_step": bool(int(ecnt[0]) == int(cnt[0]) and int(eind[0]) == int(ind[0]) and int(ecnt[1]) == int(cnt[1])
                                             and int(eind[1]) == int(ind[1])),
            "f64_rescored_candidates": dict(last_stats),
            "bound_kernel": {"kernel": "eh_upper_mfma2_kernel (<true, false, true> for the epipolar, <false, true, true> for the transfer test)",
                             "kernel_ms": kern_ms, "issue": bound_issue,
                             "per_test": {"epipolar_only_kernel_ms": part_ms["epipolar"], "transfer_only_kernel_ms": part_ms["transfer"],
                                          "note": "the two launches of one call, each timed alone (a launch solves its own model only)"}},
            "loose_valu_kernel": {"kernel": "eh_upper_kernel<true>", "kernel_ms": loose_valu_ms},
            "tight_f32_kernel": {"kernel": "eh_upper_kernel<false>", "kernel_ms": tight_ms},
        }
        out = {
            "metric": "RANSAC hyp x corr residual evals/sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32+f64",  # float32 tests from matrix-core rows, float64 solvers and float64 inside the error bands
            "data": "synthetic",
            "config": {
                "workload": f"synthetic two-view E+H RANSAC: {n} matches, 30% inliers, {B} hypotheses per GPU (8-pt E + 4-pt H solved per "
                "hypothesis; every (hypothesis, match) pair decided by the reference's symmetric-epipolar and transfer tests: sfm_eh_score_mixed_opt "
                "with SFM_EHX_EVERY_PAIR_FULL_TEST, i.e. without the library's early loops, gives the float64 inlier counts of every hypothesis; "
                "then sfm_argmax_count_i32 on both count arrays), winners merged by all-reduce(max)",
                "matches": n,
                "hypotheses_per_gpu": B,
                "index_sets": "np.random.randint on the host (legacy stream)" if index_source == "legacy"
                else "Philox4x32-10 on the device, keyed by (seed, global hypothesis id)",
                "pixel_threshold": 1.0,
                "winner_E": {"count": int(cnt[0]), "index": int(ind[0])},
                "winner_H": {"count": int(cnt[1]), "index": int(ind[1])},
            },
            "roofline": roofline,
            "effective_value": evals_per_step * args.steps / eff_dt,
            "effective_ms_per_step": eff_dt / args.steps * 1e3,
            "effective": effective,
            "exact_counts_early_out_value": evals_per_step * args.steps / early_dt,
            "exact_counts_early_out_ms_per_step": early_dt / args.steps * 1e3,
            "exact_counts_early_out": {
                "what": "the library's default form of the exact step (sfm_eh_score_mixed + sfm_argmax_count_i32, what estimate_E_robust's block "
                "scoring calls): the float64 inlier counts of EVERY hypothesis, identical to the headline step's, but a pair that fails a cheaper "
                "implied test -- the one-sided epipolar test s^2 < k1 n1 + c (3 MFMAs per 32 x 32 pairs + 2 vector issue slots per pair instead of "
                "5 + 10), the x strip dx^2 < (rt hz)^2 + c of the disc test (3 + 2 instead of 3 + 8) -- is proven to be no float64 inlier and skips the full "
                "test; the few pairs per thousand that pass take the float64 test.  Pairs decided per second, NOT `value`: most pairs do not go "
                "through the reference's full residual arithmetic, so its algorithmic rate may exceed the vector peak",
                "kernel_ms": early_ms,
                "counts_identical_to_headline_step": early_identical,
                "winners_equal_headline_step": bool(int(early_winner[0][0]) == int(cnt[0]) and int(early_winner[1][0]) == int(ind[0])
                                                    and int(early_winner[0][1]) == int(cnt[1]) and int(early_winner[1][1]) == int(ind[1])),
                "algorithmic_tflops": flops_per_launch / (early_ms * 1e-3) / 1e12,
            },
        }
        if no_overlap_ms is not None:
            out["no_overlap_ms_per_step"] = no_overlap_ms
            out["no_overlap_value"] = evals_per_step / (no_overlap_ms * 1e-3)
        if eff_no_overlap_ms is not None:
            effective["no_overlap_ms_per_step"] = eff_no_overlap_ms
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
                out["secondary"] = secondary_measurements(torch, ops, _capi, synthetic, args.seed, table)
                out["secondary"]["seed_sweep"] = seed_sweep(torch, ops, _capi, synthetic, orc, n, B)
                out["secondary"]["drop_in_replay"] = replay_throughput(torch, synthetic, orc, args.seed)
            except Exception as e:  # noqa: BLE001
                out["secondary"] = {"error": f"{type(e).__name__}: {e}"}
            try:  # the second half of the metric: end-to-end seconds on a dataset-5-shaped set of image files
                import contextlib
                import io

                with contextlib.redirect_stderr(io.StringIO()):  # progress bars
                    out["secondary"]["e2e_rendered_images"] = e2e_images(torch, synthetic, args.seed)
            except Exception as e:  # noqa: BLE001
                out["secondary"]["e2e_rendered_images"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                import contextlib
                import io

                with contextlib.redirect_stderr(io.StringIO()):
                    out["secondary"]["e2e_dataset5_correspondences"] = e2e_dataset5_correspondences(torch)
            except Exception as e:  # noqa: BLE001
                out["secondary"]["e2e_dataset5_correspondences"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""In-kernel clock of the three matrix-core kernels on their benchmark workloads, beside the pure-FMA probe
(profiles/microbench/clock_probe.hip) on the same device in the same lease.

Uses a DIAGNOSTIC build of the library (profiles/tools/build_diag.sh: -DSFM_CLOCK_STAMPS -> profiles/exp/libsfm_hip_stamps.so)
in which every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its tile loop into a buffer that
no kernel reads.  Each kernel is launched back to back for about two seconds on the benchmark's random data before the
stamps of the last launches are read:  held clock = cycles / ticks x 100 MHz (median over waves).

    python profiles/tools/clock_diag.py [--seconds 2.0]
"""
import argparse, ctypes, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sfm_hip import _capi

_capi.LIB_PATH = os.environ.get("SFM_DIAG_LIB", os.path.join(ROOT, "profiles", "exp", "libsfm_hip_stamps.so"))
from sfm_hip import ops, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--only", default="")
ap.add_argument("--dump", default="", help="directory for the raw stamp records (.npy)")
args = ap.parse_args()
lib = _capi.load_library()
lib.sfm_debug_clock_stamps.restype = ctypes.c_int
lib.sfm_debug_clock_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
SLOTS = 16384


def stamps(clear=True):
    buf = np.zeros((SLOTS, 8), np.uint64)
    _capi.check(lib.sfm_debug_clock_stamps(buf.ctypes.data_as(ctypes.c_void_p), SLOTS, 1 if clear else 0))
    return buf[buf[:, 1] > 0]


def measure(name, fn, pairs_per_tile_wave, valu_per_tile, mfma_per_tile):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    one = a.elapsed_time(b)
    warm = max(3, int(args.seconds * 1e3 / one))
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    stamps(clear=True)
    reps = 10
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    raw = stamps(clear=True)
    if args.dump:
        np.save(os.path.join(args.dump, name.split()[0].replace("<", "_").replace(">", "").replace(",", "_") + ".npy"), raw)
    # only the records of the LAST launch (the slots are overwritten launch after launch; a shorter launch leaves older records behind)
    last_entry = raw[:, 6].max()
    raw = raw[raw[:, 6].astype(np.int64) > int(last_entry) - int(ms * 1e5 * 1.5)]
    s = raw.astype(np.float64)
    s = s[s[:, 2] > 8]  # waves that walked a real share of the tiles
    clk = s[:, 0] / s[:, 1] * 0.1
    cyc_tile = s[:, 0] / s[:, 2]
    # the waves of one SIMD (XCC id, and the SE / SH / CU / SIMD fields of HW_ID): how many of them are inside the tile loop at a time
    key = (raw[:, 5] >> np.uint64(32)) * np.uint64(65536) + ((raw[:, 5] & np.uint64(0xFFFF)) >> np.uint64(4))
    t_lo, t_hi = float(raw[:, 6].min()), float(raw[:, 4].max())
    simds = np.unique(key)
    in_loop = float((raw[:, 4] - raw[:, 3]).sum()) / ((t_hi - t_lo) * len(simds))
    in_kernel = float((raw[:, 4] - raw[:, 6]).sum()) / ((t_hi - t_lo) * len(simds))
    # per SIMD: time-weighted distribution of the number of waves inside their tile loops
    hist = np.zeros(10)
    for k in simds[:: max(1, len(simds) // 128)]:
        r = raw[key == k]
        ev = sorted([(int(a), 1) for a in r[:, 3]] + [(int(b), -1) for b in r[:, 4]])
        cur, prev = 0, int(t_lo)
        for t, d in ev:
            hist[min(cur, 9)] += t - prev
            cur, prev = cur + d, t
        hist[0] += int(t_hi) - prev
    hist /= hist.sum()
    out = {"kernel": name, "ms_per_launch": round(ms, 4), "waves_sampled": int(len(s)), "simds_seen": int(len(simds)),
           "held_clock_ghz_median": round(float(np.median(clk)), 4), "held_clock_ghz_p5_p95": [round(float(np.percentile(clk, 5)), 4), round(float(np.percentile(clk, 95)), 4)],
           "shader_cycles_per_tile_and_wave_median": round(float(np.median(cyc_tile)), 1),
           "last_launch_span_ms": round((t_hi - t_lo) * 1e-5, 4),
           "waves_per_simd_inside_the_tile_loop_time_average": round(in_loop, 3), "waves_per_simd_resident_time_average": round(in_kernel, 3),
           "share_of_time_with_k_waves_of_a_simd_in_the_tile_loop": [round(float(h), 3) for h in hist[:6]],
           "solve_phase_us_median": round(float(np.median((raw[:, 3] - raw[:, 6]).astype(np.float64))) * 1e-2, 2),
           "tile_loop_us_median": round(float(np.median((raw[:, 4] - raw[:, 3]).astype(np.float64))) * 1e-2, 2),
           "counted_lanes_per_wave_and_tile_mean": round(float(raw[:, 7].sum()) / max(1.0, float(raw[:, 2].sum())), 4),
           "pairs_per_tile_and_wave": pairs_per_tile_wave, "valu_instructions_per_tile_and_wave": valu_per_tile, "mfma_per_tile_and_wave": mfma_per_tile}
    print(json.dumps(out), flush=True)
    return out


res = []
n, B = 10000, 1000000
sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=0)
Kinv = np.linalg.inv(np.asarray(sc["K"], float))
corr = _capi.dev(ops.pack_corr(Kinv @ sc["x1"], Kinv @ sc["x2"]))
thr = 1.0 / sc["K"][0][0]
np.random.seed(0)
idx = _capi.dev(np.random.randint(0, n, size=12 * B).reshape(B, 12), np.int32)
xmax = 0.0
ce = torch.empty(B, dtype=torch.int32, device="cuda"); ch = torch.empty_like(ce)
corrm = ops.corr_to_mfma(corr, thr)
if not args.only or "upper" in args.only:
    res.append(measure("eh_upper_mfma_kernel<true,true> 10^4 x 10^6", lambda: ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr, xmax, out=(ce, ch)), 2048, 395, 12))
ws = torch.empty(int(lib.sfm_eh_score_mixed_workspace_bytes(B, n)), dtype=torch.uint8, device="cuda")
if not args.only or "exact" in args.only:
    res.append(measure("eh_score_mfma_kernel (as sfm_eh_score_mixed launches it) 10^4 x 10^6", lambda: ops.eh_score_mixed(corr, idx, thr, 3 * thr, xmax, out=(ce, ch), workspace=ws), 2048, None, 16))
if not args.only or "t" in args.only.split(","):
    ncam, m, Bt = 12, 50000, 100000
    cams = [synthetic.resection_scene(m=m, outlier_ratio=0.5, seed=100 + c) for c in range(ncam)]
    pts = torch.cat([ops.pack_points(_capi.dev(c["R"].ravel()), _capi.dev(c["X"]), _capi.dev(c["x_norm"])) for c in cams])
    idt = _capi.dev(np.random.default_rng(0).integers(0, m, size=(ncam, Bt, 2)), np.int32)
    thrt = 4.0 / cams[0]["K"][0][0]
    res.append(measure("t_score_mfma_kernel 12 x 5*10^4 x 10^5", lambda: ops.t_ransac(pts, [m] * ncam, idt, thrt, mixed=True), 2048, None, 6))
print(json.dumps({"clock_diag": res}))

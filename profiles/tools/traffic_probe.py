"""Memory-side traffic of the scoring kernels against bytes that are known, one launch at a time.

    python3 profiles/tools/traffic_probe.py run [matches:hypotheses ...]
        launches, per shape, the exact E/H scorer without and with its early loops (sfm_eh_score_mixed_opt: one launch per model), the bound
        kernel (sfm_eh_upper_mfma) and, once, the exact translation scorer on the benchmark's secondary workload
        (12 cameras x 50 000 correspondences x 100 000 hypotheses), with a device synchronisation after every call.
        Run it under `rocprofv3 --pmc FETCH_SIZE --output-format csv` and again under `--pmc WRITE_SIZE`.
    python3 profiles/tools/traffic_probe.py collect <fetch dir> <write dir> <out.json>
        per kernel and grid size: launches, FETCH_SIZE and WRITE_SIZE (KB, as reported), and bytes = 2 x FETCH + WRITE
        (this chip tallies a 128-byte read request as 64 bytes; profiles/microbench/fetch_size_probe.hip shows that every
        read pattern of these kernels, partial rows and 32-byte gathers included, is fetched as whole 128-byte lines).
The default shape is the benchmark's (10 000 matches x 1 000 000 hypotheses): 48 MB of index rows are read by each launch
that solves a model from them, 4 MB of counts are written per model."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd"))


def run(args):
    import numpy as np
    import torch

    from sfm_hip import _capi, ops, synthetic

    _capi.load_library()
    shapes = [tuple(int(v) for v in a.split(":")) for a in args] or [(10000, 1000000)]
    for n, nb in shapes:
        sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=3)
        Kinv = np.linalg.inv(np.asarray(sc["K"], np.float64))
        x1 = Kinv @ np.vstack((sc["x1"][:2], np.ones((1, n))))
        x2 = Kinv @ np.vstack((sc["x2"][:2], np.ones((1, n))))
        thr = 1.0 / sc["K"][0][0]
        corr = _capi.dev(ops.pack_corr(x1, x2))
        corrm = ops.corr_to_mfma(corr, thr)
        idx = ops.index_sets(7, 0, nb, n)
        torch.cuda.synchronize()
        for rep in range(3):
            ops.eh_score_mixed(corr, idx, thr, 3 * thr, every_pair_full_test=True)  # instantiations <..., false>: what bench.py's `value` times
            torch.cuda.synchronize()
            ops.eh_score_mixed(corr, idx, thr, 3 * thr, every_pair_full_test=False)  # instantiations <..., true>: with the early loops
            torch.cuda.synchronize()
            ops.eh_upper_mfma(corr, corrm, idx, thr, 3 * thr)
            torch.cuda.synchronize()
        print(f"shape {n} matches x {nb} hypotheses: index rows {48 * nb} bytes per launch, counts {4 * nb} bytes per model, "
              f"matches as pieces {64 * n} + xy {8 * n} + float64 {32 * n} bytes", flush=True)
    ncam, m, B = 12, 50000, 100000
    rng = np.random.default_rng(0)
    cams = [synthetic.resection_scene(m=m, outlier_ratio=0.5, seed=100 + c) for c in range(ncam)]
    pts = torch.cat([ops.pack_points(_capi.dev(c["R"].ravel()), _capi.dev(c["X"]), _capi.dev(c["x_norm"])) for c in cams])
    idx = _capi.dev(rng.integers(0, m, size=(ncam, B, 2)), np.int32)
    for rep in range(3):
        ops.t_ransac(pts, [m] * ncam, idx, 4.0 / synthetic.K_DATASET5[0][0], mixed=True)
        torch.cuda.synchronize()
    print(f"translation: {ncam} cameras x {m} correspondences x {B} hypotheses: index pairs {8 * ncam * B} bytes per split of the correspondence "
          f"range, counts {4 * ncam * B} bytes per split (atomic adds), points {48 * ncam * m} bytes float64 + {96 * ncam * m} bytes as pieces", flush=True)


def collect(fetch_dir, write_dir, out):
    table = {}
    for key, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if r["Counter_Name"] != key or "mfma" not in name or "_kernel" not in name or "corr_to" in name or "t_prep" in name:
                    continue
                short = re.search(r"(\w+_kernel(<[^>]*>)?)", name).group(1)
                table.setdefault(f"{short} grid {r['Grid_Size']}", {}).setdefault(key, []).append(float(r["Counter_Value"]))
    res = {}
    for k, c in table.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f_, w_ = c["FETCH_SIZE"][1:] or c["FETCH_SIZE"], c["WRITE_SIZE"][1:] or c["WRITE_SIZE"]  # the first launch of a kernel also fetches its code
            fa, wa = sum(f_) / len(f_), sum(w_) / len(w_)
            res[k] = {"launches": len(c["FETCH_SIZE"]), "FETCH_SIZE_KB": fa, "WRITE_SIZE_KB": wa, "bytes_per_launch": (2.0 * fa + wa) * 1024.0}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        print(f"{k:70s} {v['launches']} launches  fetch {v['FETCH_SIZE_KB']:10.0f} KB (x2)  write {v['WRITE_SIZE_KB']:9.0f} KB  -> {v['bytes_per_launch'] / 1e6:8.1f} MB per launch")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "collect":
        collect(*sys.argv[2:5])
    else:
        run(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "run" else sys.argv[1:])

#!/usr/bin/env python3
"""cProfile of pipeline.run_geometry on the stored real-image correspondences of dataset 5 (tests/golden/dataset5_geometry.npz):
where the host spends the 14-15 ms that the ten RANSAC loops, the translation stage and LM take."""
import cProfile, importlib.util, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sfm_hip import pipeline, utils
spec = importlib.util.spec_from_file_location("dataset5_fixture", os.path.join(ROOT, "tests", "golden", "dataset5_fixture.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
g, K, x_pairs, init, paths, matcher = mod.load_fixture()
clean = [j for j in range(int(g["n_seeds"])) if int(g[f"s{j}_degenerate_record_events"]) == 0]
tag = f"s{clean[0]}_"
utils.find_correspondences = matcher(tag)
def run():
    np.random.seed(int(g[tag + "seed"]))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipeline.run_geometry(K, x_pairs, init, paths, 1.0)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for _ in range(3): print("run", round(run() * 1e3, 2), "ms")
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])

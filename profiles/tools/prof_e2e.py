import cProfile, pstats, sys, os, time
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from sfm_hip import pipeline, utils, synthetic, _capi
_capi.load_library()
sc = synthetic.multi_view_scene(n_cams=10, n_points=6000, init_pair=(4, 6), seed=5)
warm = synthetic.multi_view_scene(n_cams=3, n_points=400, init_pair=(0, 1), seed=6)
utils.find_correspondences = warm["find_correspondences"]
np.random.seed(10)
pipeline.run_geometry(warm["K"], warm["x_pairs"], warm["init_pair_dict"], warm["img_paths"], 1.0)
utils.find_correspondences = sc["find_correspondences"]
for rep in range(2):
    np.random.seed(11)
    torch.cuda.synchronize(); t0=time.perf_counter()
    got = pipeline.run_geometry(sc["K"], sc["x_pairs"], sc["init_pair_dict"], sc["img_paths"], 1.0)
    torch.cuda.synchronize(); print("e2e seconds", time.perf_counter()-t0)
np.random.seed(11)
pr = cProfile.Profile(); pr.enable()
got = pipeline.run_geometry(sc["K"], sc["x_pairs"], sc["init_pair_dict"], sc["img_paths"], 1.0)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)

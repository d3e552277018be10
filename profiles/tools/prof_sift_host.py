import cProfile, pstats, sys, os
sys.path.insert(0,'/root/repo/structure-from-motion_amd'); sys.path.insert(0,'/root/repo')
import numpy as np, torch
from sfm_hip import _capi, sift, synthetic
_capi.load_library()
img = synthetic.render_views(n_cams=1, seed=0)["images"][0]
g = torch.from_numpy(img).cuda()
for _ in range(3): sift.detect_and_compute(g)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): sift.detect_and_compute(g)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

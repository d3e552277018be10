#!/usr/bin/env python3
"""How often the seeded reference runs of tests/golden/eh_ransac_many.npz and eh_ransac_large.npz go through the H branch of
the loop (src/ransac_algorithm.py:124-169): H record events, accepted ones (pose, inliers and best E count overwritten), and
runs whose final pose comes from a homography."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import numpy as np
from oracle import sfm_oracle as orc
from sfm_hip import _capi, synthetic
from sfm_hip import ransac_algorithm as ra
_capi.load_library()
for fixture in ("eh_ransac_many.npz", "eh_ransac_large.npz"):
    g = np.load(os.path.join(ROOT, "tests", "golden", fixture), allow_pickle=False)
    tot = {"runs": 0, "h_records": 0, "accepted": 0, "final_from_h": 0, "e_records": 0}
    for r, (n, inl, pl, sseed, rseed, iters) in enumerate(g["runs"]):
        sc = synthetic.two_view_scene(n=int(n), inlier_ratio=float(inl), planar_ratio=float(pl), seed=int(sseed))
        x1, x2 = orc.normalize_K(sc["K"], sc["x1"]), orc.normalize_K(sc["K"], sc["x2"])
        np.random.seed(int(rseed))
        st = ra.run_replay(int(n), ra.GpuBackend(x1, x2, 1.0 / sc["K"][0][0]))
        h = [e for e in st.events if e[1] == "H"]
        acc = [e for e in h if e[2]]
        last_e = max([e[0] for e in st.events if e[1] == "E"], default=-1)
        last_acc = max([e[0] for e in acc], default=-1)
        tot["runs"] += 1; tot["h_records"] += len(h); tot["accepted"] += len(acc); tot["e_records"] += sum(e[1] == "E" for e in st.events)
        tot["final_from_h"] += int(last_acc >= 0 and last_acc >= last_e)
    print(fixture, tot)

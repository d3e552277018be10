import sys, os, json, time
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd")); sys.path.insert(0, ROOT)
import importlib.util
spec=importlib.util.spec_from_file_location('bench',os.path.join(ROOT,'bench.py')); b=importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import numpy as np, torch
from oracle import sfm_oracle as orc
from sfm_hip import synthetic, _capi
_capi.load_library()
print(json.dumps(b.replay_throughput(torch, synthetic, orc, 0), indent=1))
print(json.dumps(b.e2e(torch, synthetic, orc, 0, False), indent=1))

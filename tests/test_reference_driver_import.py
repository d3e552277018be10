"""Drop-in check at the reference's own boundary (CPU, build container only): with this
repository's package directory ahead of the reference checkout on sys.path, the UNCHANGED
reference driver (run_sfm.py) and its top-level utils.py bind to the MI355X implementations."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CODE = r'''
import sys, types
sys.modules["cv2"] = types.ModuleType("cv2")          # OpenCV is not installed here; the driver only imports it
sys.path[:0] = [PKG, REF]
import matplotlib
matplotlib.use("Agg")
import importlib.util
spec = importlib.util.spec_from_file_location("reference_run_sfm", REF + "/run_sfm.py")
run_sfm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(run_sfm)                        # the reference's file, unmodified
import utils as ref_utils                               # the reference's top-level utils.py
assert run_sfm.__file__.startswith(REF), run_sfm.__file__
for name in ("estimate_R", "run_reconstruction", "estimate_translation", "refine_translation", "process_sift_for_image_pairs"):
    fn = getattr(run_sfm, name)
    mod = sys.modules[fn.__module__] if fn.__module__ in sys.modules else None
    assert "sfm_hip" in (getattr(fn, "__module__", "") or "") or "sfm_hip" in getattr(fn, "__wrapped_module__", "sfm_hip"), (name, fn.__module__)
assert ref_utils.triangulate_3D_point_DLT.__module__ == "sfm_hip.utils"
assert ref_utils.pflat.__module__ == "sfm_hip.utils"
K, img_paths, init_pair = ref_utils.get_data(REF + "/data/5")
assert K[0][0] == 2489.14 and len(img_paths) == 10 and init_pair == [4, 6]
print("OK")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present on this machine")
def test_unchanged_reference_driver_binds_to_this_package():
    code = CODE.replace("PKG", repr(os.path.join(ROOT, "structure-from-motion_amd"))).replace("REF", repr(REF))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().endswith("OK")

"""A/B of the two forms of the exact E/H scorer (SFM_HIP_EHX_FORM is read once per process: run once per form).
Prints the event-timed duration of sfm_eh_score_mixed on 10^4 x 10^6 and whether every count equals the float64 kernel's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "structure-from-motion_amd"))
import numpy as np  # noqa: E402

from sfm_hip import _capi  # noqa: E402

if os.environ.get("SFM_DIAG_LIB"):  # an experimental build of the library
    _capi.LIB_PATH = os.environ["SFM_DIAG_LIB"]
from sfm_hip import ops, synthetic  # noqa: E402


def main():
    import torch

    _capi.load_library()
    n, nb = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (10000, 1000000)))
    sc = synthetic.two_view_scene(n=n, inlier_ratio=0.3, planar_ratio=0.5, seed=3)
    Kinv = np.linalg.inv(np.asarray(sc["K"], np.float64))
    x1 = Kinv @ np.vstack((sc["x1"][:2], np.ones((1, n))))
    x2 = Kinv @ np.vstack((sc["x2"][:2], np.ones((1, n))))
    thr = 1.0 / sc["K"][0][0]
    corr = _capi.dev(ops.pack_corr(x1, x2))
    idx = ops.index_sets(7, 0, nb, n)
    ce, ch = ops.eh_score(corr, idx, thr, 3 * thr)
    me, mh = ops.eh_score_mixed(corr, idx, thr, 3 * thr)
    torch.cuda.synchronize()
    print("form", os.environ.get("SFM_HIP_EHX_FORM", "default"), "identical E", bool(torch.equal(ce, me)), "H", bool(torch.equal(ch, mh)),
          "mismatches E", int((ce != me).sum()), "max |diff|", int((ce - me).abs().max()), "best", int(ce.max()), int(ch.max()), flush=True)
    ws = torch.empty(int(_capi.load_library().sfm_eh_score_mixed_workspace_bytes(nb, n)), dtype=torch.uint8, device="cuda")
    for r in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.eh_score_mixed(corr, idx, thr, 3 * thr, out=(me, mh), workspace=ws)
        b.record()
        torch.cuda.synchronize()
        print(f"round {r}: mixed {a.elapsed_time(b) / 10:.3f} ms", flush=True)


if __name__ == "__main__":
    main()

"""sfm_hip: MI355X implementation of the Structure-from-Motion robust-estimation path.

The submodules keep the reference's module and function names (ransac_algorithm,
chirality_check, estimate_translation, refine_translation, reconstruct_3D, utils); the
arithmetic runs in libsfm_hip.so (HIP, gfx950) through the C ABI declared in
include/sfm_hip.h.  Importing the package does not touch the GPU; the first call does.
"""
__all__ = [
    "estimate_R", "run_ransac", "estimate_E_robust", "run_reconstruction", "estimate_translation",
    "robust_estimate_T", "refine_translation", "levenberg_marquardt_algorithm", "perform_chirality_check",
    "triangulate_3D_point_DLT", "filter_3D_points", "pflat", "homogeneous_to_cartesian", "normalize_K",
    "skew_symmetric_mat", "cartesian_to_homogeneous",
]


def __getattr__(name):
    # lazy: `import sfm_hip.synthetic` must not pull in torch
    import importlib

    table = {
        "estimate_R": "ransac_algorithm", "run_ransac": "ransac_algorithm", "estimate_E_robust": "ransac_algorithm",
        "run_reconstruction": "reconstruct_3D", "estimate_translation": "estimate_translation",
        "robust_estimate_T": "estimate_translation", "refine_translation": "refine_translation",
        "levenberg_marquardt_algorithm": "refine_translation", "perform_chirality_check": "chirality_check",
        "triangulate_3D_point_DLT": "utils", "filter_3D_points": "utils", "pflat": "utils",
        "homogeneous_to_cartesian": "utils", "normalize_K": "utils", "skew_symmetric_mat": "utils",
        "cartesian_to_homogeneous": "utils",
    }
    if name in table:
        return getattr(importlib.import_module("." + table[name], __name__), name)
    raise AttributeError(name)

"""Drop-in for the reference's `src` package (src/__init__.py:1-11 there): with this directory's parent on
PYTHONPATH ahead of the reference checkout, an unchanged run_sfm.py / utils.py resolve the names they import from
`src` to the MI355X implementations.  Names are resolved on first use, so importing the package needs no GPU."""
import importlib

_HOME = {
    "process_sift_for_image_pairs": "extract_sift",
    "estimate_R": "ransac_algorithm",
    "run_reconstruction": "reconstruct_3D",
    "estimate_translation": "estimate_translation",
    "refine_translation": "refine_translation",
    "filter_3D_points": "utils",
    "triangulate_3D_point_DLT": "utils",
    "pflat": "utils",
    "homogeneous_to_cartesian": "utils",
}
__all__ = sorted(_HOME)


def __getattr__(name):
    if name not in _HOME:
        raise AttributeError(f"module 'src' has no attribute {name!r}")
    value = getattr(importlib.import_module("sfm_hip." + _HOME[name]), name)
    globals()[name] = value
    return value


def __dir__():
    return __all__

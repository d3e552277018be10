"""Drop-in for the reference's `src` package (src/__init__.py:1-11 there): put this
directory's parent on PYTHONPATH ahead of the reference checkout and an unchanged
run_sfm.py / utils.py import the MI355X implementations under the reference's names."""
from sfm_hip.extract_sift import process_sift_for_image_pairs
from sfm_hip.ransac_algorithm import estimate_R
from sfm_hip.reconstruct_3D import run_reconstruction
from sfm_hip.estimate_translation import estimate_translation
from sfm_hip.refine_translation import refine_translation
from sfm_hip.utils import (
    filter_3D_points,
    triangulate_3D_point_DLT,
    pflat,
    homogeneous_to_cartesian,
)

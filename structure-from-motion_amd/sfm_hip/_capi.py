"""ctypes binding of libsfm_hip.so (declared in include/sfm_hip.h).

PyTorch is used only as the owner of device memory and streams: tensors are allocated with
torch on the GPU and their data pointers are handed to the C ABI.  There is no CPU
fallback: importing this module without the built library, or calling it without a GPU,
raises.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFM_HIP_LIBRARY=<path> loads another build of the library (same-device comparisons of two builds within one job)
LIB_PATH = os.environ.get("SFM_HIP_LIBRARY") or os.path.join(os.path.dirname(_HERE), "lib", "libsfm_hip.so")

_c = ctypes
_P = _c.c_void_p
_SIGS = {
    "sfm_hip_abi_version": (_c.c_int, []),
    "sfm_eh_score_f64": (_c.c_int, [_P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _P, _P, _P]),
    "sfm_eh_score_mixed_workspace_bytes": (_c.c_int64, [_c.c_int64, _c.c_int32]),
    "sfm_eh_score_mixed": (_c.c_int, [_P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _c.c_double, _P, _c.c_int64, _P, _P, _P]),
    "sfm_eh_score_mixed_opt": (_c.c_int, [_P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _c.c_double, _c.c_int32, _P, _c.c_int64, _P, _P,
                               _P]),
    "sfm_eh_score_noalloc": (_c.c_int, [_P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _P, _P, _P]),
    "sfm_corr_f32_bytes": (_c.c_int64, [_c.c_int32]),
    "sfm_corr_to_f32": (_c.c_int, [_P, _c.c_int32, _P, _P]),
    "sfm_eh_upper_f32": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int32, _P, _P, _P]),
    "sfm_eh_best_workspace_bytes": (_c.c_int64, [_c.c_int64, _c.c_int32]),
    "sfm_eh_best": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int32, _P,
                               _c.c_int64, _P, _P]),
    "sfm_eh_best_pending_bytes": (_c.c_int64, []),
    "sfm_eh_best_submit": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int32, _P,
                                      _c.c_int64, _P, _P]),
    "sfm_eh_best_collect": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int32,
                                       _P, _c.c_int64, _P, _P, _P]),
    "sfm_eh_models_f64": (_c.c_int, [_P, _c.c_int32, _P, _c.c_int64, _P, _P, _P, _P]),
    "sfm_dlt_general_f64": (_c.c_int, [_P, _c.c_int32, _c.c_int32, _P, _P]),
    "sfm_epipolar_mask_f64": (_c.c_int, [_P, _P, _c.c_int32, _c.c_double, _P, _P, _P]),
    "sfm_transfer_mask_f64": (_c.c_int, [_P, _P, _c.c_int32, _c.c_double, _P, _P, _P]),
    "sfm_epipolar_errors_f64": (_c.c_int, [_P, _P, _P, _c.c_int32, _c.c_int64, _P, _P]),
    "sfm_enforce_essential_f64": (_c.c_int, [_P, _P, _P]),
    "sfm_essential_candidates_f64": (_c.c_int, [_P, _P, _P]),
    "sfm_chirality4_f64": (_c.c_int, [_P, _P, _P, _c.c_int32, _P, _P]),
    "sfm_homography_to_rt_f64": (_c.c_int, [_P, _P, _c.c_int32, _P, _P]),
    "sfm_triangulate_f64": (_c.c_int, [_P, _P, _P, _P, _c.c_int32, _c.c_int64, _P, _P]),
    "sfm_pack_points_f64": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int32, _P, _P]),
    "sfm_t_ransac_f64": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int64, _c.c_double, _P, _P, _P, _P]),
    "sfm_t_count_f64": (_c.c_int, [_P, _c.c_int32, _P, _c.c_int64, _c.c_double, _P, _P]),
    "sfm_argmax_count_i32": (_c.c_int, [_P, _c.c_int64, _c.c_uint64, _P, _P]),
    "sfm_pts_to_f32": (_c.c_int, [_P, _c.c_int64, _P, _P]),
    "sfm_t_upper_f32": (_c.c_int, [_P, _P, _P, _c.c_int32, _P, _c.c_int64, _c.c_double, _P, _c.c_double, _c.c_double, _P, _P]),
    "sfm_argmax_count_batched_i32": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _P, _P]),
    "sfm_t_models_f64": (_c.c_int, [_P, _c.c_int32, _P, _c.c_int64, _P, _P]),
    "sfm_lm_translation_f64": (_c.c_int, [_P, _P, _P, _c.c_int32, _P, _c.c_int32, _c.c_double, _P, _P, _P, _P]),
    "sfm_index_sets_i32": (_c.c_int, [_c.c_uint64, _c.c_uint64, _c.c_int64, _c.c_int32, _c.c_int32, _P, _P]),
    "sfm_corr_mfma_bytes": (_c.c_int64, [_c.c_int32]),
    "sfm_corr_to_mfma": (_c.c_int, [_P, _c.c_int32, _c.c_double, _P, _P]),
    "sfm_eh_upper_mfma_f32": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _P, _P, _P]),
    "sfm_t_ransac_mixed_workspace_bytes": (_c.c_int64, [_P, _c.c_int32]),
    "sfm_t_ransac_mixed": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int64, _c.c_double, _P, _c.c_int64, _P, _P, _P, _P]),
    "sfm_record_scratch_doubles": (_c.c_int64, []),
    "sfm_e_record_f64": (_c.c_int, [_P, _c.c_int32, _P, _c.c_double, _P, _P, _P, _P]),
    "sfm_h_record_f64": (_c.c_int, [_P, _c.c_int32, _P, _c.c_double, _P, _P, _P, _P, _P]),
    "sfm_h_record_batch_f64": (_c.c_int, [_P, _c.c_int32, _P, _c.c_int32, _c.c_double, _P, _P, _P, _P, _P]),
    "sfm_eh_upper_mfma_part_f32": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int64, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int32, _P,
                                              _P, _P]),
    "sfm_legacy_randint_fill": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _P]),
    "sfm_descriptors_to_i8": (_c.c_int, [_P, _c.c_int32, _c.c_int64, _P, _P, _P, _P]),
    "sfm_knn2_workspace_bytes": (_c.c_int64, [_c.c_int32, _c.c_int32]),
    "sfm_knn2_ratio_i8": (_c.c_int, [_P, _P, _c.c_int32, _P, _P, _c.c_int32, _c.c_double, _P, _c.c_int64, _P, _P, _P, _P, _P, _P]),
    "sfm_sift_workspace_bytes": (_c.c_size_t, [_c.c_int32, _c.c_int32]),
    "sfm_sift_detect": (_c.c_int, [_P, _c.c_int32, _c.c_int32, _P, _c.c_size_t, _P, _c.c_int32, _P, _P]),
    "sfm_sift_describe": (_c.c_int, [_P, _c.c_int32, _c.c_int32, _P, _c.c_int32, _P, _P]),
    "sfm_sift_gaussian_taps": (_c.c_int, [_c.c_double, _P, _P]),
    "sfm_sift_octaves": (_c.c_int, [_c.c_int32, _c.c_int32, _P, _P, _P, _P]),
}

EXPORTED_SYMBOLS = sorted(list(_SIGS) + ["sfm_hip_error_string"])

_lib = None


class SfmHipError(RuntimeError):
    pass


def load_library():
    """Loads libsfm_hip.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SfmHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # torch ships its own libamdhip64.so.7; it must be in the process BEFORE this library is
    # loaded so that the dynamic linker binds us to that same runtime instance.  Loaded the
    # other way round, /opt/rocm's copy would come in as a second HIP runtime that owns no
    # device ("no ROCm-capable device is detected" at the first launch).
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib.sfm_hip_error_string.restype = ctypes.c_char_p
    lib.sfm_hip_error_string.argtypes = [ctypes.c_int]
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load_library().sfm_hip_error_string(code).decode()
        raise SfmHipError(f"libsfm_hip call failed with code {code}: {msg}")


_torch = None


def torch_cuda():
    """torch, after checking once that a GPU is visible (there is no CPU fallback)."""
    global _torch
    if _torch is None:
        import torch

        if not torch.cuda.is_available():
            raise SfmHipError("no GPU visible: the HIP path needs an MI355X and has no CPU fallback")
        _torch = torch
    return _torch


def stream_ptr():
    torch = torch_cuda()
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a, dtype=None):
    """Host array -> contiguous device tensor."""
    torch = torch_cuda()
    a = np.ascontiguousarray(a, dtype=dtype)
    return torch.from_numpy(a).cuda()


def dptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def hptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)

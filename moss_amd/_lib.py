"""ctypes binding of libmoss_raster.so (the C ABI declared in include/moss_raster.h).

There is NO fallback: if the HIP library is missing the import of any op fails loudly.  PyTorch is used for
device memory and streams only; every tensor crosses the boundary as a raw device pointer.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# "lib" = the product build.  MOSS_AMD_LIB_DIR=lib_diag selects the DIAGNOSTIC build (python -m moss_amd.build --diag: -DMOSS_DIAG,
# environment knobs that pick kernel variants, stamp buffers) -- for the A/B scripts under scripts/, never for results.
_LIB_DIR = os.path.join(_HERE, os.environ.get("MOSS_AMD_LIB_DIR", "lib"))
LIB_PATH = os.path.join(_LIB_DIR, "libmoss_raster.so")
EXT_PATH = os.path.join(_LIB_DIR, "_moss_C.so")          # the compiled PyTorch extension (csrc/torch_binding.cpp) over the same C ABI
ABI_VERSION = 6                                          # include/moss_raster.h MOSS_ABI_VERSION this binding was written against

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

_lib = None
_lock = threading.Lock()

ERR_NAMES = {-1: "invalid argument", -2: "HIP error", -3: "allocation failed", -4: "prefiltered point culled", -5: "unsupported"}

_f = C.c_float
_d = C.c_double
_i = C.c_int
_p = C.c_void_p


def _declare(lib):
    lib.moss_abi_version.restype = _i
    lib.moss_last_error.restype = C.c_char_p
    lib.moss_raster_geometry_bytes.restype = C.c_size_t
    lib.moss_raster_geometry_bytes.argtypes = [_i]
    lib.moss_raster_image_bytes.restype = C.c_size_t
    lib.moss_raster_image_bytes.argtypes = [_i, _i]
    lib.moss_raster_binning_bytes.restype = C.c_size_t
    lib.moss_raster_binning_bytes.argtypes = [_i]
    lib.moss_raster_binning_bytes_forward_only.restype = C.c_size_t
    lib.moss_raster_binning_bytes_forward_only.argtypes = [_i]
    lib.moss_raster_frame_state_bytes.restype = C.c_size_t
    lib.moss_raster_frame_state_bytes.argtypes = [_i, _i]
    lib.moss_build_has_diagnostics.restype = _i
    lib.moss_adamw_state_bytes.restype = C.c_size_t
    lib.moss_raster_forward.restype = _i
    lib.moss_raster_forward.argtypes = [
        ALLOC_FN, _p, ALLOC_FN, _p, ALLOC_FN, _p,          # geometry / binning / image allocators
        _i, _i, _i,                                        # P, D, M
        _p, _i, _i,                                        # background, width, height
        _p, _p, _p, _p,                                    # means3D, shs, colors_precomp, opacities
        _p, _f, _p, _p,                                    # scales, scale_modifier, rotations, cov3D_precomp
        _p, _p, _p,                                        # viewmatrix, projmatrix, cam_pos
        _f, _f, _i,                                        # tan_fovx, tan_fovy, prefiltered
        _p, _p, _p, _p, _i, _p]                            # out_color, out_depth, out_alpha, radii, debug, stream
    _fwd = list(lib.moss_raster_forward.argtypes)
    lib.moss_raster_forward_async.restype = _i
    lib.moss_raster_forward_async.argtypes = _fwd[:-1] + [_p, _p]                         # debug -> capacity (both int), frame_state, stream
    lib.moss_raster_forward_tf.restype = _i
    lib.moss_raster_forward_tf.argtypes = _fwd[:-1] + [_p, _p]                            # cov3D_precomp -> transforms, debug -> capacity, frame_state
    lib.moss_raster_backward_tf.restype = _i
    lib.moss_raster_backward_tf.argtypes = [
        _i, _i, _i, _i,                                    # P, D, M, R
        _p, _i, _i,                                        # background, width, height
        _p, _p, _p,                                        # means3D, shs, colors_precomp
        _p, _f, _p, _p,                                    # scales, scale_modifier, rotations, transforms
        _p, _p, _p, _f, _f,                                # viewmatrix, projmatrix, campos, tan_fovx, tan_fovy
        _p, _p, _p,                                        # geom, binning, image buffers
        _p, _p, _p,                                        # dL_dpix, dL_ddepths, dL_dalphas
        _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,            # dL_dmean2D .. dL_drot, dL_dtransforms
        _p]                                                # stream
    lib.moss_raster_read_status.restype = _i
    lib.moss_raster_read_status.argtypes = [_p, _p, _p]
    lib.moss_raster_backward.restype = _i
    lib.moss_raster_backward.argtypes = [
        _i, _i, _i, _i,                                    # P, D, M, R
        _p, _i, _i,                                        # background, width, height
        _p, _p, _p, _p,                                    # means3D, shs, colors_precomp, alphas
        _p, _f, _p, _p,                                    # scales, scale_modifier, rotations, cov3D_precomp
        _p, _p, _p, _f, _f,                                # viewmatrix, projmatrix, campos, tan_fovx, tan_fovy
        _p, _p, _p, _p,                                    # radii, geom, binning, image buffers
        _p, _p, _p,                                        # dL_dpix, dL_ddepths, dL_dalphas
        _p, _p, _p, _p, _p, _p, _p, _p, _p,                # dL_dmean2D .. dL_drot
        _i, _p]                                            # debug, stream
    lib.moss_raster_forward_raw.restype = _i
    lib.moss_raster_forward_raw.argtypes = [
        ALLOC_FN, _p, ALLOC_FN, _p, ALLOC_FN, _p, _i, _i, _i, _p, _i, _i,
        _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _f, _f, _i,
        _p, _p, _p, _p, _i, _i, _p, _p]                     # ..., radii, raw_flags, capacity, frame_state, stream
    lib.moss_raster_backward_raw.restype = _i
    lib.moss_raster_backward_raw.argtypes = [
        _i, _i, _i, _i, _p, _i, _i,
        _p, _p, _p, _p, _p, _f, _p, _p,                     # means3D, shs, colors, opacities, scales, mod, rotations, transforms
        _p, _p, _p, _f, _f,
        _p, _p, _p, _p, _p, _p,
        _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]     # 10 gradient outputs, raw_flags, stream
    lib.moss_raster_mark_visible.restype = _i
    lib.moss_raster_backward_raw_adamw.restype = _i
    lib.moss_raster_backward_raw_adamw.argtypes = list(lib.moss_raster_backward_raw.argtypes[:-3]) + [_p] + list(lib.moss_raster_backward_raw.argtypes[-3:])
    lib.moss_raster_mark_visible.argtypes = [_i, _p, _p, _p, _p, _p]
    lib.moss_knn_workspace_bytes.restype = C.c_size_t
    lib.moss_knn_workspace_bytes.argtypes = [_i]
    lib.moss_knn_dist2.restype = _i
    lib.moss_knn_dist2.argtypes = [_i, _p, _p, _p, C.c_size_t, _p]
    lib.moss_knn_query.restype = _i
    lib.moss_knn_query.argtypes = [_i, _i, _i, _p, _p, _p, _p, _p]
    lib.moss_knn_grid_workspace_bytes.restype = C.c_size_t
    lib.moss_knn_grid_workspace_bytes.argtypes = [_i]
    lib.moss_knn_grid_build.restype = _i
    lib.moss_knn_grid_build.argtypes = [_i, _p, _p, C.c_size_t, _p]
    lib.moss_knn_grid_query.restype = _i
    lib.moss_knn_grid_query.argtypes = [_i, _i, _i, _p, C.c_size_t, _p, _p, _p, _p]
    lib.moss_densify_stats.restype = _i
    lib.moss_densify_stats.argtypes = [_i, _p, _p, _i, _p, _p, _p, _p]
    lib.moss_neighbour_kl.restype = _i
    lib.moss_neighbour_kl.argtypes = [_i, _i, _p, _p, _p, _p, _p, _p]
    lib.moss_loss_workspace_bytes.restype = C.c_size_t
    lib.moss_loss_workspace_bytes.argtypes = [_i, _i, _i]
    lib.moss_photometric_loss.restype = _i
    lib.moss_photometric_loss.argtypes = [_i, _i, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, C.c_size_t, _p]
    lib.moss_photometric_loss_weighted.restype = _i
    lib.moss_photometric_loss_weighted.argtypes = [_i, _i, _i, _p, _p, _p, _p, _f, _f, _f, _p, _p, _p, _p, C.c_size_t, _p]
    lib.moss_adamw_multi.restype = _i
    lib.moss_adamw_multi.argtypes = [_p, _p]
    lib.moss_photometric_loss_roi.restype = _i
    lib.moss_photometric_loss_roi.argtypes = [_i, _i, _i, _p, _p, _p, _p, _p, _p, _f, _f, _f, _p, _p, _p, _p, C.c_size_t, _p]
    lib.moss_adamw_flat.restype = _i
    lib.moss_adamw_flat.argtypes = [C.c_longlong, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _d, _d, _f, _f, _i, _p]
    lib.moss_adamw_flat_devstep.restype = _i
    lib.moss_adamw_flat_devstep.argtypes = [C.c_longlong, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _d, _d, _f, _f, _p, _p]
    lib.moss_adamw_flat_range.restype = _i
    lib.moss_adamw_flat_range.argtypes = [C.c_longlong, C.c_longlong, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _d, _d, _f, _f, _i, _p, _p]
    lib.moss_adamw_flat_guarded.restype = _i
    lib.moss_adamw_flat_guarded.argtypes = [C.c_longlong, C.c_longlong, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _d, _d, _f, _f, _p, _p, C.c_uint32, _p]
    lib.moss_adamw_flat_ex.restype = _i
    lib.moss_adamw_flat_ex.argtypes = [_p, _p]
    lib.moss_gaussian_activate_forward.restype = _i
    lib.moss_gaussian_activate_forward.argtypes = [_i, _i] + [_p] * 12
    lib.moss_gaussian_activate_backward.restype = _i
    lib.moss_gaussian_activate_backward.argtypes = [_i, _i] + [_p] * 15
    lib.moss_raster_profile_enable.restype = None
    lib.moss_raster_profile_enable.argtypes = [C.c_uint32]
    lib.moss_raster_profile_read.restype = _i
    lib.moss_raster_profile_read.argtypes = [_p, _p]
    lib.moss_raster_export_geometry.restype = _i
    lib.moss_raster_export_geometry.argtypes = [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p]
    lib.moss_raster_export_binning.restype = _i
    lib.moss_raster_export_binning.argtypes = [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]


class FusedAdamWStruct(C.Structure):
    """``moss_fused_adamw`` of include/moss_raster.h (host struct handed to ``moss_raster_backward_raw_adamw``)."""
    _fields_ = [("tensors", C.c_uint32), ("exp_avg", C.c_void_p * 5), ("exp_avg_sq", C.c_void_p * 5), ("lr", C.c_float * 5),
                ("lr_sh_rest", C.c_float), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("step_state", C.c_void_p), ("lr_segment", C.c_int32 * 5), ("sh_active_degree", C.c_int32), ("sh_inactive_zero", C.c_int32)]


class AdamWFlatArgs(C.Structure):
    """``moss_adamw_flat_args`` of include/moss_raster.h (``moss_adamw_flat_ex``: every form of the flat update + the degree-aware SH update)."""
    _fields_ = [("first", C.c_longlong), ("count", C.c_longlong), ("params", C.c_void_p), ("grads", C.c_void_p), ("exp_avg", C.c_void_p),
                ("exp_avg_sq", C.c_void_p), ("num_segments", C.c_int), ("segment_end", C.c_void_p), ("segment_lr", C.c_void_p),
                ("segment_period", C.c_void_p), ("segment_split", C.c_void_p), ("segment_lr2", C.c_void_p), ("segment_active", C.c_void_p),
                ("inactive_zero", C.c_int), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("step", C.c_int), ("step_state", C.c_void_p), ("skip_word", C.c_void_p), ("skip_mask", C.c_uint32),
                ("num_grads_extra", C.c_int), ("grads_extra", C.c_void_p * 3), ("grad_scale", C.c_float)]


class AdamWMultiArgs(C.Structure):
    """``moss_adamw_multi_args`` of include/moss_raster.h (``moss_adamw_multi``: up to eight tensors with buffers of their own, one launch)."""
    _fields_ = [("num_tensors", C.c_int32), ("numel", C.c_longlong * 8), ("params", C.c_void_p * 8), ("grads", C.c_void_p * 8),
                ("exp_avg", C.c_void_p * 8), ("exp_avg_sq", C.c_void_p * 8), ("lr", C.c_float * 8), ("step", C.c_int32 * 8),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_float), ("weight_decay", C.c_float)]


OPT_BITS = {"means3D": 1, "sh": 2, "opacity": 4, "scales": 8, "rotations": 16}      # MOSS_OPT_*; position = index in the struct's arrays


def lib() -> C.CDLL:
    """Load (once) and return the HIP library.  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        f"{LIB_PATH} is missing: the MI355X HIP library has not been built "
                        "(run `python -m moss_amd.build`); there is no CPU/PyTorch fallback for this op")
                handle = C.CDLL(LIB_PATH)
                handle.moss_abi_version.restype = _i
                if handle.moss_abi_version() != ABI_VERSION:
                    raise ImportError(f"{LIB_PATH} implements ABI version {handle.moss_abi_version()}, this binding needs {ABI_VERSION}: "
                                      "rebuild it (python -m moss_amd.build --force)")
                _declare(handle)
                _lib = handle
    return _lib


_ext = None


def ext():
    """Load (once) and return the compiled PyTorch-ROCm extension module ``_moss_C`` (rasterize_gaussians,
    rasterize_gaussians_backward, mark_visible: the reference's ``_C``).  Raises ImportError if it has not been built."""
    global _ext
    if _ext is None:
        with _lock:
            if _ext is None:
                if not os.path.exists(EXT_PATH):
                    raise ImportError(
                        f"{EXT_PATH} is missing: the PyTorch extension of the MI355X rasterizer has not been built "
                        "(run `python -m moss_amd.build`); there is no CPU/PyTorch fallback for this op")
                lib()                                        # libmoss_raster.so first: _moss_C.so links against it
                import importlib.machinery
                import importlib.util
                import torch  # noqa: F401  (libtorch must be loaded before the extension)
                loader = importlib.machinery.ExtensionFileLoader("_moss_C", EXT_PATH)
                spec = importlib.util.spec_from_loader("_moss_C", loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                # abi_version() is the extension's COMPILE-TIME MOSS_ABI_VERSION: a stale _moss_C.so next to a rebuilt library is caught
                if mod.abi_version() != lib().moss_abi_version():
                    raise ImportError(f"{EXT_PATH} was compiled against ABI version {mod.abi_version()}, libmoss_raster.so implements "
                                      f"{lib().moss_abi_version()}: rebuild both (python -m moss_amd.build --force)")
                _ext = mod
    return _ext


STAGES = ["preprocess_fwd", "scan", "scatter", "chunk_sort", "blend_fwd", "blend_bwd", "preprocess_bwd", "merge_gather"]    # (every stage is ONE kernel)


def profile_enable(stages=None):
    """Enable HIP-event timing for the named stages (None = all, [] = off)."""
    names = STAGES if stages is None else stages
    mask = 0
    for n in names:
        mask |= 1 << STAGES.index(n)
    lib().moss_raster_profile_enable(mask)


def profile_read():
    """{stage: (total_ms, count)} since the last read (synchronises the recorded events)."""
    ms = (C.c_float * 8)()
    cnt = (C.c_uint32 * 8)()
    check(lib().moss_raster_profile_read(ms, cnt), "profile_read")
    return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(STAGES)}


def check(rc: int, what: str) -> int:
    if rc < 0:
        msg = lib().moss_last_error().decode(errors="replace")
        raise RuntimeError(f"{what}: {ERR_NAMES.get(rc, rc)}: {msg}")
    return rc

"""Python op surface of the MI355X Gaussian rasterizer -- a drop-in for the reference package
``diff_gaussian_rasterization`` (submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py).

Same public names, signatures, argument meaning, outputs and error behaviour:

=============================  ==========================================================  =================
name                           role                                                        reference lines
=============================  ==========================================================  =================
GaussianRasterizationSettings  per-view constants (NamedTuple, 12 fields)                  __init__.py:160-172
GaussianRasterizer             nn.Module; ``forward`` validates inputs, ``markVisible``     __init__.py:174-223
rasterize_gaussians            functional entry                                            __init__.py:21-42
_RasterizeGaussians            autograd.Function: 4 outputs (color, radii, depth, alpha),  __init__.py:44-158
                               3 incoming grads, 9 returned grads
=============================  ==========================================================  =================

MOSS imports it as ``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
(gaussian_renderer/__init__.py:16); the repo-root package ``diff_gaussian_rasterization`` re-exports this module
under that name.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C
from ._C import CapacityOverflow, RasterContext, set_async, check_async_status, set_grad_sink  # noqa: F401  (opt-in: asynchronous forward / graph capture, gradient sinks; per-rasterizer state)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def cpu_deep_copy_tuple(input_tuple):
    """Host copies of every tensor argument, taken BEFORE a debug-mode native call so that a failing call can be
    replayed from ``snapshot_fw.dump`` / ``snapshot_bw.dump`` (reference __init__.py:17-19, :83-90, :135-142)."""
    return tuple(a.cpu().clone() if isinstance(a, torch.Tensor) else a for a in input_tuple)


def _call_native(fn, args, debug, dump_name, what):
    # (`debug` is the reference's bool -- or this library's MOSS_DEBUG_* bit set, whose bit 0 is that bool: only it asks for snapshots)
    if not (int(debug) & 1):
        return fn(*args)
    snapshot = cpu_deep_copy_tuple(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(snapshot, dump_name)
        print(f"\nAn error occured in {what}. Please forward {dump_name} for debugging.")
        raise


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                transforms=None, raw_flags=0, context=None, translation=None):
        # (the thirteenth input, with raw_flags & RAW_POSE: means3D are the canonical positions and the op poses them itself, T x + translation)
        # (the tenth input is an addition: per-Gaussian 3x3 transforms applied to the covariance inside the op, SURVEY 8f row n2;
        #  the eleventh too: which of opacities / scales / rotations are RAW parameters whose getter runs inside the op;
        #  the twelfth: the RasterContext -- asynchronous-forward policy and gradient sinks -- of the rasterizer that calls)
        rs = raster_settings
        raw_flags = int(raw_flags)
        native_args = (
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width,
            sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, transforms, raw_flags, context, translation)
        ctx.context = context
        (num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer) = _call_native(
            _C.rasterize_gaussians, native_args, rs.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        # an output that takes no part in the loss reaches backward as None (NULL at the C ABI = zeros) instead of as a
        # zero-filled image; the values computed are the same as with the reference's materialised zeros
        ctx.set_materialize_grads(False)
        ctx.has_transforms = transforms is not None
        ctx.has_translation = translation is not None
        ctx.raw_flags = raw_flags
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              geomBuffer, binningBuffer, imgBuffer, alpha, *(() if transforms is None else (transforms,)),
                              *(() if translation is None else (translation,)), *((opacities,) if raw_flags else ()))
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        saved = ctx.saved_tensors
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
         geomBuffer, binningBuffer, imgBuffer, alpha) = saved[:11]
        transforms = saved[11] if ctx.has_transforms else None
        translation = saved[12] if ctx.has_translation else None
        raw_opacities = saved[-1] if ctx.raw_flags else None
        if grad_out_color is None and grad_depth is None and grad_alpha is None:
            return (None,) * 13
        native_args = (
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, grad_alpha,
            sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, alpha, rs.debug,
            transforms, ctx.raw_flags, raw_opacities, ctx.context, translation, False)
        grads = _call_native(_C.rasterize_gaussians_backward, native_args, rs.debug, "snapshot_bw.dump", "backward")
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
         grad_scales, grad_rotations) = grads[:8]
        grad_transforms = grads[8] if transforms is not None else None
        grad_translation = grads[9] if translation is not None else None
        # one gradient per forward() input, in forward()'s order; raster_settings gets None
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, None if transforms is not None else grad_cov3Ds_precomp, None, grad_transforms, None, None, grad_translation)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        transforms=None, raw_flags=0, context=None, translation=None):
    # An EVALUATION render -- grad mode off (render_ZJU.py:56-72 renders under torch.no_grad()) or no input that could receive a
    # gradient -- is told so at the C ABI (MOSS_FORWARD_ONLY): same images bit for bit, none of the state only a backward reads, a
    # binning buffer of 62 B per instance instead of ~370.  No autograd node: the outputs are plain tensors, as torch itself returns
    # them from a Function applied without grad.
    if (context or _C.DEFAULT).forward_only_renders and not (
            torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in
                                            (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, transforms, translation))):
        rs = raster_settings
        res = _call_native(_C.rasterize_gaussians, (
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered,
            int(rs.debug) | _C.FORWARD_ONLY, transforms, int(raw_flags), context, translation), rs.debug, "snapshot_fw.dump", "forward")
        return res[1], res[4], res[2], res[3]                # color, radii, depth, alpha
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, transforms, raw_flags, context, translation)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings, context=None):
        """``context`` (an addition): a :class:`RasterContext` -- the asynchronous-forward policy and gradient sinks of THIS
        rasterizer; None = the process-wide default one (``set_async`` / ``set_grad_sink`` without a context configure that)."""
        super().__init__()
        self.raster_settings = raster_settings
        self.context = context

    def markVisible(self, positions):
        """Boolean (P,) mask of the points that pass the near-plane test of this camera."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, transforms=None, raw_flags=0, translation=None):
        """``transforms`` (an addition): (P,3,3) per-Gaussian matrices; with scales and rotations the op then builds
        T (R S S^T R^T) T^T itself -- MOSS's Python get_covariance -- and returns a gradient for the transforms too.
        ``raw_flags`` (an addition): ``_C.RAW_OPACITY | _C.RAW_SCALE | _C.RAW_ROTATION`` -- those inputs are GaussianModel's raw
        parameters (``_opacity``, ``_scaling``, ``_rotation``); sigmoid / exp / normalize run inside the op and the gradients
        come back w.r.t. the raw parameters (no activation kernels either side).  With ``_C.RAW_POSE`` OR-ed in (needs ``transforms``),
        ``means3D`` are the CANONICAL positions and the op poses them itself, T x + ``translation`` ((P,3), optional): the reference's
        caller does that with torch ops (gaussian_renderer/__init__.py:74-77); gradients come back for the canonical positions, the
        transforms and the translation."""
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        have_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (have_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        # an absent optional is handed to the native side as an empty tensor (null data pointer)
        absent = torch.Tensor([])
        shs = absent if shs is None else shs
        colors_precomp = absent if colors_precomp is None else colors_precomp
        scales = absent if scales is None else scales
        rotations = absent if rotations is None else rotations
        cov3D_precomp = absent if cov3D_precomp is None else cov3D_precomp
        if transforms is not None and (scales.numel() == 0 or cov3D_precomp.numel() != 0):
            raise Exception('transforms need the scale/rotation pair (and no precomputed 3D covariance)!')
        if raw_flags and (scales.numel() == 0 or cov3D_precomp.numel() != 0):
            raise Exception('raw_flags need the scale/rotation pair (and no precomputed 3D covariance)!')
        if (translation is not None or (int(raw_flags) & _C.RAW_POSE)) and (transforms is None or not (int(raw_flags) & _C.RAW_POSE)):
            raise Exception('a translation / RAW_POSE needs the transforms and the RAW_POSE flag!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs, transforms,
                                   raw_flags, self.context, translation)

"""Counterpart of the reference's render binding, gaussian_renderer/__init__.py:21-136 -- the one Python file of
MOSS that changes when the rasterizer is swapped (north_star).  Same call signature, same settings construction
(:36-52), same zero ``means2D`` gradient sink (:29-33), same input-mode selection (:85-109), same output keys (:124-136).

Out of scope here: the LBS / pose-refinement branch (:57-72).  If ``pc`` offers
``coarse_deform_c2source`` it is called exactly as the reference does; otherwise the Gaussians render where they are
(optionally moved by explicit ``transforms`` / ``translation``, the cheap branch :73-77).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

PipelineDefaults = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=True, debug=False)  # arguments/__init__.py:57-61


def eval_sh_rgb(deg, sh, dirs):
    """SH -> RGB in torch for ``pipe.convert_SHs_python`` (the reference imports utils/sh_utils.eval_sh :57-112)."""
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435]
    res = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                   + C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10]
                       + C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14]
                       + C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


_ZERO_POINTS = {}


def _zero_points(like: torch.Tensor) -> torch.Tensor:
    key = (like.device, tuple(like.shape), like.dtype)
    z = _ZERO_POINTS.get(key)
    if z is None:
        if len(_ZERO_POINTS) > 8:
            _ZERO_POINTS.clear()
        z = _ZERO_POINTS[key] = torch.zeros(like.shape, dtype=like.dtype, device=like.device)
    return z


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, return_smpl_rot=False,
           transforms=None, translation=None):
    """Render one view.  ``bg_color`` must be on the GPU.

    ``pipe.fused_activations`` (an addition, default off): read the activated parameters from ``pc.activate(pipe.grad_bucket)``
    -- one HIP kernel instead of the five torch getters (moss_amd/activations.py); values and gradients are the same."""
    # ``pipe.raw_parameters_in_op`` (an addition, default off): hand the rasterizer the RAW ``_opacity / _scaling / _rotation`` and let
    # it run sigmoid / exp / normalize inside its preprocess kernels (C ABI moss_raster_forward_raw): no activation kernel either way.
    raw_flags = 0
    if (getattr(pipe, "raw_parameters_in_op", False) and not pipe.compute_cov3D_python
            and all(hasattr(pc, a) for a in ("_opacity", "_scaling", "_rotation"))):
        raw_flags = 7                                        # _C.RAW_OPACITY | _C.RAW_SCALE | _C.RAW_ROTATION
        if getattr(pc, "spatially_ordered", False):          # (GaussianSet.reorder_spatially: index neighbours are spatial neighbours)
            raw_flags |= 8                                   # _C.HINT_SPATIAL_ORDER
    elif getattr(pipe, "fused_activations", False):
        pc = pc.activate(getattr(pipe, "grad_bucket", None))
    xyz = pc.get_xyz
    # the zero "means2D" whose .grad receives the screen-space gradient (reference :29-33 builds it as zeros + 0 and retains its
    # grad; a leaf needs neither the add kernel nor retain_grad and exposes the same .grad)
    # The VALUES are never read by anything (the rasterizer takes means2D only to hang the gradient on it), so every call shares
    # one cached block of zeros per (device, shape) -- a fresh leaf over the same storage, with its own .grad -- instead of
    # launching a fill kernel per render.
    screenspace_points = _zero_points(xyz).detach().requires_grad_(True)

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings, context=getattr(pipe, "raster_context", None))

    means3D = xyz
    bweights = correct_Rs = pose_out = None
    # ``pipe.pose_in_op`` (an addition; with ``pipe.transforms_in_op``): the op poses the canonical positions itself, T x + translation
    # (C ABI MOSS_RAW_POSE) -- the reference does it here with torch ops (:74-77) -- and returns the gradients of x, T and the translation
    pose_in_op = (transforms is not None and getattr(pipe, "pose_in_op", False) and getattr(pipe, "transforms_in_op", False)
                  and not pipe.compute_cov3D_python and not pipe.convert_SHs_python)
    op_translation = None
    if pose_in_op:
        raw_flags |= 16                                      # _C.RAW_POSE (composes with activated OR raw opacity / scales / rotations)
        op_translation = None if translation is None else translation.squeeze()
    elif transforms is not None:
        # = torch.matmul(transforms, means3D[..., None]).squeeze(-1) (reference :74-75), written as an elementwise product + row sum:
        # a batched GEMM over 100k 3x3 matrices goes through hipBLASLt on MI355X and costs ~0.9 ms per call (and twice more backward)
        means3D = (transforms * means3D[..., None, :]).sum(-1) + (0 if translation is None else translation)
    elif hasattr(pc, "coarse_deform_c2source"):
        _, means3D, bweights, transforms, translation = pc.coarse_deform_c2source(
            means3D[None], viewpoint_camera.smpl_param, viewpoint_camera.big_pose_smpl_param,
            viewpoint_camera.big_pose_world_vertex[None])
    if means3D.dim() != 2:
        means3D = means3D.squeeze()
    means2D = screenspace_points
    # which tensors go in is decided by the three activation bits alone: RAW_POSE / HINT_SPATIAL_ORDER say nothing about them
    # (round 4 tested `if raw_flags`: pose_in_op without raw_parameters_in_op handed the op logits as opacities)
    raw_params = bool(raw_flags & 7)
    opacity = pc._opacity if raw_params else pc.get_opacity

    scales = rotations = cov3D_precomp = op_transforms = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier, None if transforms is None else transforms.squeeze())
    else:
        scales, rotations = (pc._scaling, pc._rotation) if raw_params else (pc.get_scaling, pc.get_rotation)
        # an addition (pipe.transforms_in_op): the per-Gaussian LBS transform of the covariance is applied INSIDE the op instead of
        # by the torch ops of get_covariance (the reference ignores `transforms` in this branch, gaussian_renderer/__init__.py:92-93)
        if transforms is not None and getattr(pipe, "transforms_in_op", False):
            op_transforms = transforms.squeeze()

    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = means3D - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh_rgb(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii, depth, alpha = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
        **({} if op_transforms is None else {"transforms": op_transforms}), **({"raw_flags": raw_flags} if raw_flags else {}),
        **({} if op_translation is None else {"translation": op_translation}))

    return RenderOutput({"render": rendered_image, "render_depth": depth, "render_alpha": alpha,
                         "viewspace_points": screenspace_points, "radii": radii,
                         "transforms": transforms, "translation": translation, "correct_Rs": correct_Rs, "pose_out": pose_out,
                         "lbs_weights": bweights, "means3D": means3D})


class RenderOutput(dict):
    """The reference's result dict (gaussian_renderer/__init__.py:125-136).  ``visibility_filter`` (= ``radii > 0``) is computed
    the first time it is asked for (a minimal kernel costs 4-5 us of a 0.3 ms step).  MOSS itself asks for it in EVERY iteration
    (train_ZJU.py:101, and uses it at :172-174 while it densifies), so its training loop pays that launch every step -- bench.py's
    `callers.dropin_*` variants do, the headline step (which has no use for the mask) does not.  ``out["visibility_filter"]``,
    ``out.get(..)`` and ``"visibility_filter" in out`` all work; ``keys()`` / ``items()`` list it once it exists."""
    _LAZY = "visibility_filter"

    def __missing__(self, key):
        if key == self._LAZY:
            value = dict.__getitem__(self, "radii") > 0
            self[key] = value
            return value
        raise KeyError(key)

    def __contains__(self, key):
        return key == self._LAZY or dict.__contains__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default


def camera_view(cam, device):
    """Adapt a ``scenes.make_camera`` namespace to the attribute names render() reads from MOSS's Camera
    (scene/cameras.py:17-72)."""
    return SimpleNamespace(
        FoVx=cam.FoVx, FoVy=cam.FoVy, image_height=cam.H, image_width=cam.W,
        world_view_transform=cam.viewmatrix.to(device), full_proj_transform=cam.projmatrix.to(device),
        camera_center=cam.campos.to(device))

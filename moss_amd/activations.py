"""The Gaussian parameter activations of MOSS's ``GaussianModel`` getters as ONE HIP kernel each way
(C ABI ``moss_gaussian_activate_forward/backward``, csrc/activations.hip).

Reference semantics, scene/gaussian_model.py:46-53 and :134-166::

    get_xyz = _xyz                  get_features = cat((_features_dc, _features_rest), dim=1)
    get_opacity = sigmoid(_opacity) get_scaling = exp(_scaling)      get_rotation = F.normalize(_rotation)

The backward writes the gradients of the RAW parameters directly into their final destination (``sink``: the slices of a
``moss_amd.dist.GradBucket``) and hands those views to autograd, which adopts them as ``.grad`` without a copy.
"""
from __future__ import annotations

import torch

from ._lib import check, lib


class _ActivateGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, features_dc, features_rest, opacity, scaling, rotation, sink):
        with_features = features_dc is not None
        ins = (xyz, features_dc, features_rest, opacity, scaling, rotation)
        for t in ins:
            if t is not None and (not t.is_cuda or t.dtype != torch.float32):
                raise RuntimeError("activate_gaussians needs float32 GPU tensors; this op has no CPU path")
        ins = tuple(None if t is None else t.contiguous() for t in ins)
        xyz, features_dc, features_rest, opacity, scaling, rotation = ins
        P = int(xyz.shape[0])
        K = (1 + int(features_rest.shape[1])) if with_features else 0
        if (with_features and (features_dc.shape != (P, 1, 3) or features_rest.shape != (P, K - 1, 3))) or rotation.shape != (P, 4) or \
                scaling.shape != (P, 3) or opacity.numel() != P or xyz.shape != (P, 3):
            raise RuntimeError("activate_gaussians: inconsistent parameter shapes")
        dev = xyz.device
        o_xyz = torch.empty_like(xyz)
        o_feat = torch.empty((P, K, 3), dtype=torch.float32, device=dev) if with_features else None
        o_opa = torch.empty_like(opacity)
        o_scl = torch.empty_like(scaling)
        o_rot = torch.empty_like(rotation)
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            rc = lib().moss_gaussian_activate_forward(
                P, K, xyz.data_ptr(), ptr(features_dc), ptr(features_rest) if K > 1 else None, opacity.data_ptr(),
                scaling.data_ptr(), rotation.data_ptr(), o_xyz.data_ptr(), ptr(o_feat), o_opa.data_ptr(), o_scl.data_ptr(),
                o_rot.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        check(rc, "gaussian_activate_forward")
        ctx.save_for_backward(rotation, o_opa, o_scl)
        ctx.meta = (P, K, [None if t is None else t.shape for t in ins])
        ctx.sink = sink
        ctx.set_materialize_grads(False)
        if not with_features:
            o_feat = torch.empty(0, device=dev)          # placeholder output (autograd wants a tensor); not differentiable
            ctx.mark_non_differentiable(o_feat)
        return o_xyz, o_feat, o_opa, o_scl, o_rot

    @staticmethod
    def backward(ctx, g_xyz, g_feat, g_opa, g_scl, g_rot):
        rotation, o_opa, o_scl = ctx.saved_tensors
        P, K, shapes = ctx.meta
        dev = rotation.device
        gs = [None if g is None else g.contiguous() for g in (g_xyz, g_feat if K > 0 else None, g_opa, g_scl, g_rot)]
        dests = []
        for i, shape in enumerate(shapes):
            if shape is None:
                dests.append(None)
                continue
            d = ctx.sink[i]() if ctx.sink is not None and ctx.sink[i] is not None else None
            dests.append(d if d is not None else torch.empty(shape, dtype=torch.float32, device=dev))
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            rc = lib().moss_gaussian_activate_backward(
                P, K, rotation.data_ptr(), o_opa.data_ptr(), o_scl.data_ptr(),
                ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gs[4]),
                dests[0].data_ptr(), ptr(dests[1]), ptr(dests[2]) if K > 1 else None, dests[3].data_ptr(),
                dests[4].data_ptr(), dests[5].data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        check(rc, "gaussian_activate_backward")
        out = tuple(d if (need and d is not None) else None for d, need in zip(dests, ctx.needs_input_grad[:6]))
        del dests
        return out + (None,)


def activate_gaussians(xyz, features_dc, features_rest, opacity, scaling, rotation, bucket=None):
    """Returns (xyz, features (P,K,3), opacity, scaling, rotation) -- the rasterizer-facing values of the six raw parameters.
    With ``bucket`` (a GradBucket holding these parameters) the backward writes their gradients straight into the bucket.
    ``features_dc = features_rest = None``: the SH features take no part (callers that store them as one (P,K,3) tensor pass that
    tensor to the rasterizer as it is); the second return value is then an empty placeholder."""
    sink = None
    if bucket is not None:
        params = (xyz, features_dc, features_rest, opacity, scaling, rotation)
        sink = tuple(None if p is None else (lambda p=p: bucket.sink_for(p)) for p in params)
    return _ActivateGaussians.apply(xyz, features_dc, features_rest, opacity, scaling, rotation, sink)

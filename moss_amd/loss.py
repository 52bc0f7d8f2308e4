"""Photometric losses of the measured training step (host side, torch ops).

Semantics follow the reference's utils/loss_utils.py: ``l1_loss`` (:41-42), ``l2_loss`` (:44-45), ``ssim`` with an
11x11 Gaussian window, sigma 1.5, zero padding, depthwise (:47-87); the step combines them as
``L1 + 0.2 * (1 - SSIM)`` (+ 0.5 * mask L2 on the alpha image), the rasterizer-facing terms of train_ZJU.py:111-131.
Pinned by tests/golden/loss_*.npz (generated from the reference functions); MOSS's own composition -- bound_mask selection for L1 / mask L2,
boundingRect crop for SSIM, train_ZJU.py:108-119 -- is ``training_loss_moss`` (torch) / ``training_loss_moss_fused`` (HIP), pinned by
tests/golden/loss_moss.npz.
"""
from __future__ import annotations

from math import exp

import torch
import torch.nn.functional as F

_WINDOWS = {}


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def _window(window_size: int, channel: int, like: torch.Tensor) -> torch.Tensor:
    key = (window_size, channel, like.device, like.dtype)
    w = _WINDOWS.get(key)
    if w is None:
        g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
        g = (g / g.sum()).unsqueeze(1)
        w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
        w = w2.expand(channel, 1, window_size, window_size).contiguous().to(device=like.device, dtype=like.dtype)
        _WINDOWS[key] = w
    return w


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = _window(window_size, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)


def bounding_rect(bound_mask):
    """cv2.boundingRect(bound_mask) (train_ZJU.py:115) of a (1,H,W) / (H,W) 0/1 mask as (x, y, w, h) Python ints (a host read: do it once
    per view when the view is loaded, like the mask's pixel count)."""
    # (the script casts the mask to uint8 first -- a truncation -- and boundingRect takes the non-zero bytes)
    m = bound_mask.reshape(bound_mask.shape[-2], bound_mask.shape[-1]).to(torch.uint8) != 0
    ys, xs = m.any(1).nonzero().flatten(), m.any(0).nonzero().flatten()
    if ys.numel() == 0:
        return 0, 0, 0, 0
    return int(xs[0]), int(ys[0]), int(xs[-1] - xs[0] + 1), int(ys[-1] - ys[0] + 1)


def training_loss_moss(image, alpha, gt_image, bkgd_mask, bound_mask, rect=None, lambda_dssim=0.2, lambda_mask=0.5, ssim_fn=None):
    """MOSS's own expression, torch ops, line by line (train_ZJU.py:108-119,131): L1 and mask L2 over the pixels of ``bound_mask``
    (1,H,W), SSIM on the crop ``rect`` = boundingRect(bound_mask) of both images.  The reference form :func:`training_loss_moss_fused` is
    tested against."""
    sel = bound_mask.reshape(bound_mask.shape[-2], bound_mask.shape[-1]) == 1          # (`bound_mask[0]==1`, :111)
    ll1 = l1_loss(image.permute(1, 2, 0)[sel], gt_image.permute(1, 2, 0)[sel])
    mask_loss = l2_loss(alpha.reshape(sel.shape)[sel], bkgd_mask.reshape(sel.shape)[sel])
    x, y, w, h = rect if rect is not None else bounding_rect(bound_mask)
    s = (ssim_fn or ssim)(image[:, y:y + h, x:x + w].unsqueeze(0), gt_image[:, y:y + h, x:x + w].unsqueeze(0))
    return ll1 + lambda_mask * mask_loss + lambda_dssim * (1.0 - s)


def training_loss(image, alpha, gt_image, gt_mask, lambda_dssim=0.2, lambda_mask=0.5, ssim_fn=None):
    """L1 + lambda_mask * L2(alpha, mask) + lambda_dssim * (1 - SSIM)  (train_ZJU.py:111-112,119,131).  ``ssim_fn``: the SSIM used
    (default: the torch restatement of the reference's; ``ssim_fused`` = the same value from the HIP kernels)."""
    ll1 = l1_loss(image, gt_image)
    mask_loss = l2_loss(alpha, gt_mask)
    s = (ssim_fn or ssim)(image.unsqueeze(0), gt_image.unsqueeze(0))
    return ll1 + lambda_mask * mask_loss + lambda_dssim * (1.0 - s)


class _FusedSSIM(torch.autograd.Function):
    """mean SSIM(img1, img2) and its gradient w.r.t. img1 from the two loss kernels (C ABI moss_photometric_loss_weighted with
    lambda_l1 = 0, lambda_dssim = 1, no mask term: total = 1 - SSIM, so d SSIM / d img1 = -dL_dimage)."""

    @staticmethod
    def forward(ctx, img1, img2):
        from ._lib import check, lib
        L = lib()
        C, H, W = img1.shape
        a, b = img1.contiguous(), img2.contiguous()
        out = torch.empty(4, dtype=torch.float32, device=a.device)
        d_img = torch.empty((C, H, W), dtype=torch.float32, device=a.device)
        nbytes = int(L.moss_loss_workspace_bytes(C, H, W))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        with torch.cuda.device(a.device):
            rc = L.moss_photometric_loss_weighted(C, H, W, a.data_ptr(), b.data_ptr(), None, None, 0.0, 1.0, 0.0, out.data_ptr(),
                                                  d_img.data_ptr(), None, ws.data_ptr(), nbytes,
                                                  torch.cuda.current_stream(a.device).cuda_stream)
        check(rc, "photometric_loss_weighted")
        ctx.save_for_backward(d_img)
        return out[2]

    @staticmethod
    def backward(ctx, grad_out):
        (d_img,) = ctx.saved_tensors
        return -grad_out * d_img, None


def ssim_fused(img1, img2, window_size=11, size_average=True):
    """Drop-in for the reference's ``utils.loss_utils.ssim`` (:47-87) as MOSS calls it (train_ZJU.py:119: two (1,3,h,w) crops, the
    defaults): the same value, the gradient w.r.t. ``img1``, from two HIP kernels instead of five depthwise 11x11 convolutions and
    ~20 elementwise kernels forward plus their autograd mirror (on MI355X through MIOpen: 8 x 177 us per step).  In MOSS:
    ``from moss_amd.loss import ssim_fused as ssim`` (patches/train_ZJU.diff).  float32 GPU tensors, (C,H,W) or (1,C,H,W); the second
    image gets no gradient (MOSS's is the ground truth)."""
    if window_size != 11 or not size_average:
        return ssim(img1, img2, window_size, size_average)       # (other windows / per-image means: the torch expressions)
    if img1.dim() == 4:
        if img1.shape[0] != 1:
            return ssim(img1, img2, window_size, size_average)
        img1, img2 = img1[0], img2[0]
    if not img1.is_cuda or img1.dtype != torch.float32 or img2.dtype != torch.float32:
        raise RuntimeError("ssim_fused needs float32 GPU tensors (the product path has no CPU fallback; moss_amd.loss.ssim is the torch form)")
    return _FusedSSIM.apply(img1, img2.detach())


class _FusedPhotometricLoss(torch.autograd.Function):
    """HIP implementation of :func:`training_loss` (include/moss_raster.h: moss_photometric_loss): the loss and its
    gradient w.r.t. image and alpha come out of two fused kernels; backward only scales them by the incoming gradient."""

    @staticmethod
    def forward(ctx, image, alpha, gt_image, gt_mask, lambda_dssim, lambda_mask, terms_out=None):
        from ._lib import check, lib
        L = lib()
        if not image.is_cuda:
            raise RuntimeError("fused loss needs GPU tensors; use training_loss() for the torch reference on CPU")
        C, H, W = image.shape
        image_c, gt_c = image.contiguous(), gt_image.contiguous()
        alpha_c, mask_c = alpha.contiguous(), gt_mask.contiguous()
        if terms_out is not None:
            if terms_out.shape != (4,) or terms_out.dtype != torch.float32 or terms_out.device != image.device or not terms_out.is_contiguous():
                raise RuntimeError("fused loss: terms_out must be 4 contiguous float32 values on the image's device")
            out = terms_out
        else:
            out = torch.empty(4, dtype=torch.float32, device=image.device)
        # both gradient images in one buffer: backward scales them by the incoming gradient with ONE kernel
        d_both = torch.empty((C + 1, H, W), dtype=torch.float32, device=image.device)
        d_img, d_alpha = d_both[:C], d_both[C:]
        nbytes = int(L.moss_loss_workspace_bytes(C, H, W))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=image.device)
        with torch.cuda.device(image.device):
            rc = L.moss_photometric_loss(C, H, W, image_c.data_ptr(), gt_c.data_ptr(), alpha_c.data_ptr(), mask_c.data_ptr(),
                                         float(lambda_dssim), float(lambda_mask), out.data_ptr(), d_img.data_ptr(),
                                         d_alpha.data_ptr(), ws.data_ptr(), nbytes,
                                         torch.cuda.current_stream(image.device).cuda_stream)
        check(rc, "photometric_loss")
        ctx.save_for_backward(d_both)
        ctx.C, ctx.alpha_shape = C, alpha.shape
        ctx.terms = out
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        (d_both,) = ctx.saved_tensors
        # the loss is normally the root of the graph: its incoming gradient is then the constant 1 handed to backward() by
        # `backward_from_loss` below, recognised by identity -- multiplying by exactly 1.0 would be a no-op kernel
        unit = _UNIT.get(d_both.device)
        scaled = d_both if (unit is not None and grad_out.data_ptr() == unit.data_ptr()) else grad_out * d_both
        return scaled[:ctx.C], scaled[ctx.C:].reshape(ctx.alpha_shape), None, None, None, None, None


_UNIT = {}


def backward_from_loss(loss):
    """``loss.backward()`` without the two minimal kernels autograd would launch for a root loss (ones_like fill + scaling of the
    loss gradients by 1.0): the unit gradient is a cached constant that the fused loss recognises."""
    one = _UNIT.get(loss.device)
    if one is None:
        one = _UNIT[loss.device] = torch.ones((), dtype=loss.dtype, device=loss.device)
    torch.autograd.backward(loss, grad_tensors=one)


def training_loss_fused(image, alpha, gt_image, gt_mask, lambda_dssim=0.2, lambda_mask=0.5, terms_out=None):
    """Same value and gradients as :func:`training_loss`, computed by the fused HIP kernels.  ``terms_out`` (optional, 4 floats):
    where the kernels write [loss, L1, SSIM, mask L2] -- e.g. ``GradBucket.loss_terms``, so that the loss travels with the
    gradients in the one all-reduce without a copy; the returned loss is then ``terms_out[0]``."""
    return _FusedPhotometricLoss.apply(image, alpha, gt_image, gt_mask, lambda_dssim, lambda_mask, terms_out)


class ViewRegion:
    """What MOSS's loss needs of a view besides the two target images, prepared ONCE when the view is loaded: ``bound`` (H,W) uint8 on
    the device, ``rect`` = 5 int32 on the device (x, y, w, h of cv2.boundingRect(bound_mask), the mask's pixel count).  ``copy_(other)``
    rewrites both in place -- how a step captured in a hipGraph changes view."""

    def __init__(self, bound_mask, rect=None):
        m = (bound_mask.reshape(bound_mask.shape[-2], bound_mask.shape[-1]) == 1)      # the script's selection: `bound_mask[0]==1`
        self.bound = m.to(torch.uint8).contiguous()
        x, y, w, h = rect if rect is not None else bounding_rect(bound_mask)
        self.xywh = (x, y, w, h)
        inside = m[y:y + h, x:x + w]                          # (pixels of the mask outside a caller's rectangle count for nothing)
        self.rect = torch.tensor([x, y, w, h, int(inside.sum())], dtype=torch.int32, device=m.device)

    def copy_(self, other):
        self.bound.copy_(other.bound)
        self.rect.copy_(other.rect)
        self.xywh = other.xywh
        return self


class _FusedMossLoss(torch.autograd.Function):
    """C ABI moss_photometric_loss_roi: value and gradients of :func:`training_loss_moss` from the two loss kernels."""

    @staticmethod
    def forward(ctx, image, alpha, gt_image, bkgd_mask, region, lambda_dssim, lambda_mask, terms_out=None):
        from ._lib import check, lib
        L = lib()
        if not image.is_cuda or image.dtype != torch.float32:
            raise RuntimeError("fused loss needs float32 GPU tensors; training_loss_moss() is the torch form")
        C, H, W = image.shape
        if tuple(region.bound.shape) != (H, W) or region.bound.device != image.device:
            raise RuntimeError("fused loss: the view's region does not belong to this image")
        image_c, gt_c = image.contiguous(), gt_image.contiguous()
        alpha_c, mask_c = alpha.contiguous(), bkgd_mask.to(torch.float32).contiguous()
        if terms_out is not None:
            if terms_out.shape != (4,) or terms_out.dtype != torch.float32 or terms_out.device != image.device or not terms_out.is_contiguous():
                raise RuntimeError("fused loss: terms_out must be 4 contiguous float32 values on the image's device")
            out = terms_out
        else:
            out = torch.empty(4, dtype=torch.float32, device=image.device)
        d_both = torch.empty((C + 1, H, W), dtype=torch.float32, device=image.device)
        d_img, d_alpha = d_both[:C], d_both[C:]
        nbytes = int(L.moss_loss_workspace_bytes(C, H, W))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=image.device)
        with torch.cuda.device(image.device):
            rc = L.moss_photometric_loss_roi(C, H, W, image_c.data_ptr(), gt_c.data_ptr(), alpha_c.data_ptr(), mask_c.data_ptr(),
                                             region.bound.data_ptr(), region.rect.data_ptr(), 1.0, float(lambda_dssim), float(lambda_mask),
                                             out.data_ptr(), d_img.data_ptr(), d_alpha.data_ptr(), ws.data_ptr(), nbytes,
                                             torch.cuda.current_stream(image.device).cuda_stream)
        check(rc, "photometric_loss_roi")
        ctx.save_for_backward(d_both)
        ctx.C, ctx.alpha_shape = C, alpha.shape
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        (d_both,) = ctx.saved_tensors
        unit = _UNIT.get(d_both.device)
        scaled = d_both if (unit is not None and grad_out.data_ptr() == unit.data_ptr()) else grad_out * d_both
        return scaled[:ctx.C], scaled[ctx.C:].reshape(ctx.alpha_shape), None, None, None, None, None, None


def training_loss_moss_fused(image, alpha, gt_image, bkgd_mask, region, lambda_dssim=0.2, lambda_mask=0.5, terms_out=None):
    """``Ll1 + lambda_mask * mask_loss + lambda_dssim * (1 - ssim_loss)`` with the three terms EXACTLY as MOSS forms them
    (train_ZJU.py:108-119,131: bound_mask selection, boundingRect crop) -- :func:`training_loss_moss` -- from two HIP kernels.  ``region``:
    the view's :class:`ViewRegion` (made once per view: the rectangle and the pixel count are host reads).  In MOSS this replaces lines
    :111-119 (patches/train_ZJU.diff keeps them and swaps only ``ssim``; this is the one-call form).  The remaining terms of :131 (lpips,
    s3im, nll) are other subsystems' and are added to the returned loss by the caller."""
    return _FusedMossLoss.apply(image, alpha, gt_image, bkgd_mask, region, lambda_dssim, lambda_mask, terms_out)

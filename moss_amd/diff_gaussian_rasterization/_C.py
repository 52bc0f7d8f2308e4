"""Drop-in for the reference's compiled ``diff_gaussian_rasterization._C`` module (pybind11 exports at
submodules/diff-gaussian-rasterization/ext.cpp:15-18; torch glue in rasterize_points.cu / rasterize_points.h).

Same three callables, same positional arguments, same return tuples.  Each one unwraps the tensors to raw device
pointers and calls the C ABI of ``include/moss_raster.h`` through ctypes on the CURRENT torch HIP stream
(the reference launches on the legacy default stream -- a wart, not a contract).  Tensors are only used for what the
reference's glue uses them for: allocating outputs and the three opaque scratch buffers.
"""
from __future__ import annotations

import threading

import torch

from .._lib import ALLOC_FN, check, lib

NUM_CHANNELS = 3   # submodules/diff-gaussian-rasterization/cuda_rasterizer/config.h:14

_tls = threading.local()
last_num_rendered = 0   # num_rendered of the most recent forward call (read by bench.py for its byte accounting)


class _AsyncState:
    """Opt-in asynchronous forward (C ABI moss_raster_forward_async): no host read-back of num_rendered.

    The reference blocks in every forward to size its binning buffer (rasterizer_impl.cu:283).  In a training loop R
    drifts slowly, so here the buffer is sized for ``margin x`` the last value seen; the true R stays on the device.  A
    frame that needs more than the capacity renders nothing and sets a flag, which is read back (without blocking) and
    raised by a later call or by :func:`check_async_status`.  With no synchronisation left in the step, a whole
    training iteration can be captured in a hipGraph (``torch.cuda.graph``)."""
    enabled = False
    capacity = 0
    margin = 2.0
    pending = None          # (pinned status tensor, torch.cuda.Event)
    last_needed = 0
    status_buf = None       # one pinned 32-byte landing buffer, reused (only one status copy is pending at a time)
    last_img_buffer = None  # image buffer (holds the status header) of the most recent asynchronous forward


ASYNC = _AsyncState()

# Optional destination for dL_dsh: a callable returning a fresh (P, M, 3) float32 tensor (e.g. a view into a flat gradient
# bucket) that the backward kernel fills instead of a newly allocated one; autograd then adopts it as .grad without a copy.
# The same for dL_dmeans3D / dL_dopacity / dL_dscales / dL_drotations -- meaningful when those inputs ARE the parameters (raw mode,
# no transform on the means): then no activation kernel and no copy stands between the backward kernel and the bucket.
GRAD_SINK = {"sh": None, "means3D": None, "opacity": None, "scales": None, "rotations": None}

RAW_OPACITY, RAW_SCALE, RAW_ROTATION = 1, 2, 4       # include/moss_raster.h MOSS_RAW_*


def set_grad_sink(sh=None, means3D=None, opacity=None, scales=None, rotations=None):
    GRAD_SINK.update(sh=sh, means3D=means3D, opacity=opacity, scales=scales, rotations=rotations)


def _sink(name, shape, dev):
    fn = GRAD_SINK.get(name)
    if fn is None or shape[0] == 0:
        return None
    cand = fn()
    if cand is not None and tuple(cand.shape) == tuple(shape) and cand.dtype == torch.float32 and cand.device == dev and cand.is_contiguous():
        return cand
    return None


def set_async(enabled: bool, capacity: int = 0, margin: float = 2.0):
    ASYNC.enabled, ASYNC.capacity, ASYNC.margin, ASYNC.pending = bool(enabled), int(capacity), float(margin), None


def _status_buffer():
    if ASYNC.status_buf is None:
        ASYNC.status_buf = torch.zeros(8, dtype=torch.int32).pin_memory()
    return ASYNC.status_buf


def _consume_status(block: bool):
    """Look at the status words of the previous asynchronous forward, if they have arrived (or wait when block=True)."""
    if ASYNC.pending is None:
        return
    status, ev = ASYNC.pending
    if block:
        ev.synchronize()
    elif not ev.query():
        return
    ASYNC.pending = None
    needed, flags = int(status[6]), int(status[2])
    ASYNC.last_needed = needed
    if needed * 1.25 > ASYNC.capacity:                       # drifting towards the limit: grow ahead of time
        ASYNC.capacity = int(needed * ASYNC.margin) + 1024
    if flags & 2:
        raise RuntimeError(f"rasterize_gaussians (async): a frame needed {needed} (Gaussian, tile) instances but the binning "
                           f"buffer was sized for fewer; that frame rendered nothing. Capacity is now {ASYNC.capacity}.")
    if flags & 1:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")


def check_async_status(img_buffer=None):
    """Synchronously verify the most recent asynchronous forward (call outside graph capture, e.g. every N steps)."""
    if img_buffer is None:
        img_buffer = ASYNC.last_img_buffer
    if img_buffer is not None:
        status = _status_buffer()
        dev = img_buffer.device
        with torch.cuda.device(dev):
            check(lib().moss_raster_read_status(img_buffer.data_ptr(), status.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                  "read_status")
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev))
        ASYNC.pending = (status, ev)
    _consume_status(block=True)

lib()   # fail at import time if the HIP library is missing: there is no fallback


@ALLOC_FN
def _grow(user, nbytes):
    """Replaces resizeFunctional (rasterize_points.cu:27-33): grow scratch tensor number `user` and hand back its pointer."""
    t = _tls.buffers[int(user or 0)]
    t.resize_(int(nbytes))
    return t.data_ptr()


def _ptr(t: torch.Tensor, name: str, dtype=torch.float32):
    """Raw device pointer of an optional tensor; an EMPTY tensor means "absent" and maps to NULL
    (diff_gaussian_rasterization/__init__.py:200-210, rasterize_points.cu:96-108)."""
    if t is None or t.numel() == 0:
        return None, t
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); this op has no CPU path")
    t = t.contiguous()
    return t.data_ptr(), t


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, transforms=None, raw_flags=0):
    """RasterizeGaussiansCUDA, rasterize_points.cu:35-119.
    ``raw_flags`` (an addition): RAW_OPACITY | RAW_SCALE | RAW_ROTATION -- those inputs are MOSS's raw parameters and the getters
    (sigmoid / exp / normalize) run inside the op (C ABI moss_raster_forward_raw); needs scales and rotations, no cov3D_precomp.
    ``transforms`` (an addition): (P,3,3) per-Gaussian matrices applied to the scale/rotation covariance
    inside the op (Sigma' = T Sigma T^T, what MOSS's Python get_covariance builds); needs scales and rotations, no cov3D_precomp.
    Returns (num_rendered, out_color (3,H,W), out_depth (1,H,W), out_alpha (1,H,W), radii (P,), geomBuffer,
    binningBuffer, imgBuffer)."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")          # rasterize_points.cu:57-59
    if not means3D.is_cuda:
        raise RuntimeError("means3D must live on the GPU; this op has no CPU path")
    L = lib()
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    fopts = dict(dtype=torch.float32, device=dev)
    # every element is written by the kernels (or memset by the library when P == 0): no zero-fill pass needed
    out_color = torch.empty((NUM_CHANNELS, H, W), **fopts)
    out_depth = torch.empty((1, H, W), **fopts)
    out_alpha = torch.empty((1, H, W), **fopts)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    geom = torch.empty((0,), dtype=torch.uint8, device=dev)
    binning = torch.empty((0,), dtype=torch.uint8, device=dev)
    img = torch.empty((0,), dtype=torch.uint8, device=dev)

    M = int(sh.size(1)) if sh.numel() != 0 else 0                                   # rasterize_points.cu:85-89
    keep = []
    def p(t, name):
        ptr, c = _ptr(t, name)
        keep.append(c)
        return ptr
    if transforms is not None:
        if transforms.numel() != 9 * P or scales.numel() == 0 or rotations.numel() == 0 or cov3D_precomp.numel() != 0:
            raise RuntimeError("transforms must be (P,3,3) and comes with scales and rotations (no cov3D_precomp)")
        if debug:
            raise RuntimeError("debug mode is not available together with transforms")
    raw_flags = int(raw_flags)
    if raw_flags:
        if scales.numel() == 0 or rotations.numel() == 0 or cov3D_precomp.numel() != 0 or debug:
            raise RuntimeError("raw_flags comes with scales and rotations (no cov3D_precomp, no debug mode)")
    use_async = ASYNC.enabled and not debug and ASYNC.capacity > 0 and P > 0
    capturing = torch.cuda.is_current_stream_capturing()
    if use_async and not capturing:
        _consume_status(block=False)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _tls.buffers = (geom, binning, img)
        try:
            fwd = L.moss_raster_forward_async if use_async else L.moss_raster_forward
            if transforms is not None:
                fwd = L.moss_raster_forward_tf                # cov3D_precomp slot carries the transforms, last int the capacity (-1 = sync)
            if raw_flags:
                fwd = L.moss_raster_forward_raw               # as _tf (transforms may be NULL), raw_flags before the capacity
            in_op = transforms is not None or raw_flags
            rc = fwd(
                _grow, 0, _grow, 1, _grow, 2,
                P, int(degree), M,
                p(background, "background"), W, H,
                p(means3D, "means3D"), p(sh, "sh"), p(colors, "colors_precomp"), p(opacity, "opacity"),
                p(scales, "scales"), float(scale_modifier), p(rotations, "rotations"),
                p(cov3D_precomp, "cov3D_precomp") if not in_op else (None if transforms is None else p(transforms, "transforms")),
                p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), p(campos, "campos"),
                float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                out_color.data_ptr(), out_depth.data_ptr(), out_alpha.data_ptr(), radii.data_ptr() if P else None,
                *((raw_flags,) if raw_flags else ()),
                (int(ASYNC.capacity) if use_async else (-1 if in_op else int(bool(debug)))), stream)
        finally:
            _tls.buffers = None
        rendered = check(rc, "rasterize_gaussians")
        if use_async:
            ASYNC.last_img_buffer = img
        if use_async and not capturing and ASYNC.pending is None:
            status = _status_buffer()
            check(L.moss_raster_read_status(img.data_ptr(), status.data_ptr(), stream), "read_status")
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev))
            ASYNC.pending = (status, ev)
    global last_num_rendered
    if ASYNC.enabled and not use_async and P > 0:          # first (synchronous) call of an async session: learn the size
        ASYNC.capacity = max(ASYNC.capacity, int(rendered * ASYNC.margin) + 1024)
        ASYNC.last_needed = rendered
    last_num_rendered = ASYNC.last_needed if use_async else rendered
    return rendered, out_color, out_depth, out_alpha, radii, geom, binning, img


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, dL_dout_alpha,
                                 sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, debug, transforms=None,
                                 raw_flags=0, opacities=None):
    """RasterizeGaussiansBackwardCUDA, rasterize_points.cu:121-206.
    ``raw_flags`` / ``opacities``: backward of the raw-parameter forward (gradients w.r.t. the raw parameters).
    Returns (dL_dmeans2D (P,3), dL_dcolors (P,3), dL_dopacity (P,1), dL_dmeans3D (P,3), dL_dcov3D (P,6),
    dL_dsh (P,M,3), dL_dscales (P,3), dL_drotations (P,4)) -- plus dL_dtransforms (P,3,3) when ``transforms`` was given."""
    L = lib()
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(alphas.size(-2)), int(alphas.size(-1))      # incoming gradients may be None (= zeros): sizes come from alphas
    M = int(sh.size(1)) if sh.numel() != 0 else 0
    fopts = dict(dtype=torch.float32, device=dev)
    # The reference zero-fills nine tensors here (300 B per Gaussian, rasterize_points.cu:158-166); the HIP backward
    # writes every element exactly once, so plain allocations suffice.  P == 0 keeps the reference's zeros.
    alloc = torch.zeros if P == 0 else torch.empty
    def out(name, shape):
        t = _sink(name, shape, dev)
        return t if t is not None else alloc(shape, **fopts)
    dL_dmeans3D = out("means3D", (P, 3))
    dL_dmeans2D = alloc((P, 3), **fopts)
    dL_dcolors = alloc((P, NUM_CHANNELS), **fopts)
    dL_dconic = alloc((P, 2, 2), **fopts)
    dL_dopacity = out("opacity", (P, 1))
    dL_dcov3D = alloc((P, 6), **fopts)
    dL_dsh = out("sh", (P, M, 3)) if M != 0 else alloc((P, M, 3), **fopts)
    dL_dscales = out("scales", (P, 3))
    dL_drotations = out("rotations", (P, 4))
    dL_dtransforms = alloc((P, 3, 3), **fopts) if transforms is not None else None
    raw_flags = int(raw_flags)
    if P != 0 and raw_flags:
        if opacities is None:
            raise RuntimeError("the raw-parameter backward needs the raw opacities")
        keep = []
        def p(t, name, dtype=torch.float32):
            ptr, c = _ptr(t, name, dtype)
            keep.append(c)
            return ptr
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = L.moss_raster_backward_raw(
                P, int(degree), M, int(R),
                p(background, "background"), W, H,
                p(means3D, "means3D"), p(sh, "sh"), p(colors, "colors_precomp"), p(opacities, "opacity"),
                p(scales, "scales"), float(scale_modifier), p(rotations, "rotations"),
                None if transforms is None else p(transforms, "transforms"),
                p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), p(campos, "campos"),
                float(tan_fovx), float(tan_fovy),
                p(geomBuffer, "geomBuffer", torch.uint8), p(binningBuffer, "binningBuffer", torch.uint8),
                p(imageBuffer, "imageBuffer", torch.uint8),
                p(dL_dout_color, "dL_dout_color"), p(dL_dout_depth, "dL_dout_depth"), p(dL_dout_alpha, "dL_dout_alpha"),
                dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(), dL_dcolors.data_ptr(),
                dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(), dL_dsh.data_ptr() if M else None,
                dL_dscales.data_ptr(), dL_drotations.data_ptr(), None if transforms is None else dL_dtransforms.data_ptr(),
                raw_flags, stream)
        check(rc, "rasterize_gaussians_backward")
        res = (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
        return res + (dL_dtransforms,) if transforms is not None else res
    if P != 0 and transforms is not None:
        keep = []
        def p(t, name, dtype=torch.float32):
            ptr, c = _ptr(t, name, dtype)
            keep.append(c)
            return ptr
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = L.moss_raster_backward_tf(
                P, int(degree), M, int(R),
                p(background, "background"), W, H,
                p(means3D, "means3D"), p(sh, "sh"), p(colors, "colors_precomp"),
                p(scales, "scales"), float(scale_modifier), p(rotations, "rotations"), p(transforms, "transforms"),
                p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), p(campos, "campos"),
                float(tan_fovx), float(tan_fovy),
                p(geomBuffer, "geomBuffer", torch.uint8), p(binningBuffer, "binningBuffer", torch.uint8),
                p(imageBuffer, "imageBuffer", torch.uint8),
                p(dL_dout_color, "dL_dout_color"), p(dL_dout_depth, "dL_dout_depth"), p(dL_dout_alpha, "dL_dout_alpha"),
                dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(), dL_dcolors.data_ptr(),
                dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(), dL_dsh.data_ptr() if M else None,
                dL_dscales.data_ptr(), dL_drotations.data_ptr(), dL_dtransforms.data_ptr(), stream)
        check(rc, "rasterize_gaussians_backward")
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dtransforms
    if P != 0:
        keep = []
        def p(t, name, dtype=torch.float32):
            ptr, c = _ptr(t, name, dtype)
            keep.append(c)
            return ptr
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            rc = L.moss_raster_backward(
                P, int(degree), M, int(R),
                p(background, "background"), W, H,
                p(means3D, "means3D"), p(sh, "sh"), p(colors, "colors_precomp"), p(alphas, "alphas"),
                p(scales, "scales"), float(scale_modifier), p(rotations, "rotations"), p(cov3D_precomp, "cov3D_precomp"),
                p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), p(campos, "campos"),
                float(tan_fovx), float(tan_fovy),
                p(radii, "radii", torch.int32), p(geomBuffer, "geomBuffer", torch.uint8),
                p(binningBuffer, "binningBuffer", torch.uint8), p(imageBuffer, "imageBuffer", torch.uint8),
                p(dL_dout_color, "dL_dout_color"), p(dL_dout_depth, "dL_dout_depth"), p(dL_dout_alpha, "dL_dout_alpha"),
                dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(), dL_dcolors.data_ptr(),
                dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(), dL_dsh.data_ptr() if M else None,
                dL_dscales.data_ptr(), dL_drotations.data_ptr(), int(bool(debug)), stream)
        check(rc, "rasterize_gaussians_backward")
    if transforms is not None:
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dtransforms
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible, rasterize_points.cu:208-227: bool (P,), True where z_view > 0.2."""
    if not means3D.is_cuda:
        raise RuntimeError("means3D must live on the GPU; this op has no CPU path")
    L = lib()
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m, _m = _ptr(means3D, "means3D")
        v, _v = _ptr(viewmatrix, "viewmatrix")
        pr, _pr = _ptr(projmatrix, "projmatrix")
        with torch.cuda.device(dev):
            rc = L.moss_raster_mark_visible(P, m, v, pr, present.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        check(rc, "mark_visible")
    return present

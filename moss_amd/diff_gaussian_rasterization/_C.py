"""Drop-in for the reference's compiled ``diff_gaussian_rasterization._C`` module (pybind11 exports at
submodules/diff-gaussian-rasterization/ext.cpp:15-18; torch glue in rasterize_points.cu / rasterize_points.h).

Same three callables, same positional arguments, same return tuples.  The glue itself is COMPILED: ``moss_amd/lib/_moss_C.so``
(``moss_amd/csrc/torch_binding.cpp``, a PyTorch-ROCm C++ extension built by ``moss_amd/build.py``) allocates the outputs and the
three opaque scratch tensors, unwraps every tensor to a raw device pointer and calls the C ABI of ``include/moss_raster.h`` on the
CURRENT torch HIP stream (the reference launches on the legacy default stream -- a wart, not a contract).  What stays in Python is
policy only, and it is PER RASTERIZER, not per process (:class:`RasterContext`): the capacity bookkeeping of the opt-in asynchronous
forward and the optional gradient sinks.  Two rasterizer users in one process (a training view and an evaluation view at another
resolution) give each its own context; everything that names none shares :data:`DEFAULT`.
"""
from __future__ import annotations

import torch

from .._lib import check, ext, lib

NUM_CHANNELS = 3   # submodules/diff-gaussian-rasterization/cuda_rasterizer/config.h:14

RAW_OPACITY, RAW_SCALE, RAW_ROTATION = 1, 2, 4       # include/moss_raster.h MOSS_RAW_*
HINT_SPATIAL_ORDER = 8                               # MOSS_HINT_SPATIAL_ORDER: OR-ed into raw_flags, changes no result
RAW_POSE = 16                                        # MOSS_RAW_POSE: means3D are canonical positions, posed inside the op (T x + translation)
SH_GRAD_ACTIVE_ONLY = 32                             # MOSS_SH_GRAD_ACTIVE_ONLY (backward): dL_dsh is written for the active degree's coefficients only
# bits of the `debug` argument (GaussianRasterizationSettings.debug may be the reference's bool or an OR of these; include/moss_raster.h)
DEBUG_SYNC, DEBUG_NO_BLOCK_CULL, DEBUG_EXACT_MATH, DEBUG_TRACE = 1, 2, 4, 8
FORWARD_ONLY = 16                                    # MOSS_FORWARD_ONLY: a bit of the same argument -- no backward will follow (evaluation render)

last_num_rendered = 0   # num_rendered of the most recent forward call of ANY context (kept for callers that predate contexts)

_SINK_NAMES = ("sh", "means3D", "opacity", "scales", "rotations")
FRAME_STATE_DROPPED_WORD = 4                         # include/moss_raster.h MOSS_FRAME_STATE_DROPPED_WORD


class CapacityOverflow(RuntimeError):
    """An asynchronous forward needed more (Gaussian, tile) instances than its binning capacity -- or more gradient-record cells / key
    bucket slots than a buffer of that capacity holds: that frame rendered nothing (outputs = background, gradients = 0).
    ``needed`` / ``capacity``: the capacity that would have held the frame and the capacity the context has now."""

    def __init__(self, needed, capacity):
        super().__init__(f"rasterize_gaussians (async): a frame needed {needed} (Gaussian, tile) instances but the binning "
                         f"buffer was sized for fewer; that frame rendered nothing. Capacity is now {capacity}.")
        self.needed, self.capacity = needed, capacity


class RasterContext:
    """State of ONE user of the rasterizer.

    * Opt-in asynchronous forward (C ABI ``moss_raster_forward_async``): no host read-back of num_rendered.  The reference blocks in
      every forward to size its binning buffer (rasterizer_impl.cu:283).  In a training loop R drifts slowly, so here the buffer is
      sized for ``margin x`` the last value seen; the true R stays on the device.  A frame that needs more than the capacity renders
      nothing and sets a flag, which is read back (without blocking) and raised by a later call or by :meth:`check_status`.  With no
      synchronisation left in the step, a whole training iteration can be captured in a hipGraph (``torch.cuda.graph``).
    * Gradient sinks: per input a callable returning a tensor (e.g. a view into a flat gradient bucket) that the backward kernel
      fills instead of a newly allocated one; autograd then adopts it as ``.grad`` without a copy.  The kernels OVERWRITE a sink, so a
      sink must hand out each destination at most once per backward pass (``GradBucket.sink_for`` does: a second request in the same
      step gets None and the op falls back to a fresh tensor, which autograd accumulates normally)."""

    def __init__(self):
        self.enabled = False
        self.capacity = 0
        self.margin = 2.0
        self.pending = None          # (pinned status tensor, torch.cuda.Event)
        self.last_needed = 0
        self.status_buf = None       # one pinned 32-byte landing buffer, reused (only one status copy is pending at a time)
        self.last_img_buffer = None  # image buffer (holds the status header) of the most recent asynchronous forward
        self.last_num_rendered = 0
        self.sinks = dict.fromkeys(_SINK_NAMES)
        self.fused_adamw = None                              # FlatAdamW.fuse_into_backward: the backward kernel takes the optimizer step
        # opt-in (MOSS_SH_GRAD_ACTIVE_ONLY): the `sh` gradient sink's destination holds zeros above the active SH degree and nobody
        # reads them (a zero-initialised GradBucket consumed by the degree-aware FlatAdamW and the active-degree exchange): the raw
        # backward then writes only the active coefficients of dL_dsh -- at degree 0 a twelfth of the kernel's largest output
        self.sh_grad_active_only = False
        # Renders without grad (evaluation, render_ZJU.py:56-72) are told MOSS_FORWARD_ONLY: same images, a 62 B / instance binning buffer
        # instead of ~400.  The price on the CAPACITY-BOUNDED path: its keys take the scan -> scatter chain (six launches) where the
        # training forward buckets them (four) -- ~15 us per render at 100k Gaussians.  False: no-grad renders run the training forward.
        self.forward_only_renders = True
        self._capacity_floor = 0                             # relearn_capacity(): the capacity that was in force before
        self.frame_state = None      # device block the asynchronous forward keeps its per-frame counters in (all-zero between calls)
        self._retired_frame_states = []   # outgrown blocks: a captured hipGraph may still hold their address (see _frame_state)
        self._raised_overflows = 0        # overflows of this context already raised as CapacityOverflow (not "dropped by a replay")

    # ---- asynchronous forward -------------------------------------------------------------------------------------------------
    def set_async(self, enabled: bool, capacity: int = 0, margin: float = 2.0):
        self.enabled, self.capacity, self.margin, self.pending = bool(enabled), int(capacity), float(margin), None

    def relearn_capacity(self):
        """Forget the learned binning capacity: the NEXT forward of this context runs synchronously (like the first one of an
        asynchronous session) and sizes the capacity from what it finds -- for the moments the set of Gaussians changes from outside
        (densify / prune).  Call it outside graph capture; a hipGraph captured with the old capacity must be captured again."""
        self._consume_quiet()
        # (the capacity in force is remembered: if it still holds the new set with 25 % to spare it is KEPT -- the scratch buffers of a
        # re-captured step then have the sizes of the ones they replace and come out of the allocator's cache instead of hipMalloc)
        self._capacity_floor = self.capacity
        self.capacity = 0
        self.pending = None

    def _consume_quiet(self):
        """Pick up a pending status report (it belongs to a frame of the OLD set) without losing an overflow it carries."""
        try:
            self._consume_status(block=True)
        except CapacityOverflow:
            pass                                             # (counted in _raised_overflows: the frame rendered nothing, the caller re-learns anyway)

    def _status_buffer(self):
        if self.status_buf is None:
            self.status_buf = torch.zeros(8, dtype=torch.int32).pin_memory()
        return self.status_buf

    def _consume_status(self, block: bool):
        """Look at the status words of the previous asynchronous forward, if they have arrived (or wait when block=True)."""
        if self.pending is None:
            return
        status, ev = self.pending
        if block:
            ev.synchronize()
        elif not ev.query():
            return
        self.pending = None
        needed, flags = int(status[6]), int(status[2])
        want = max(needed, int(status[3]))                      # the capacity that holds the frame: its instances, and the record pool / key
        self.last_needed = needed                               # buckets that are sized from the capacity (include/moss_raster.h, status words)
        if want * 1.25 > self.capacity:                         # drifting towards the limit: grow ahead of time
            self.capacity = int(want * self.margin) + 1024
        if flags & 2:
            self._raised_overflows += 1                         # the caller hears about this frame here: it is not a silently dropped one
            raise CapacityOverflow(want, self.capacity)
        if flags & 1:
            raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")

    def _request_status(self, img_buffer, dev, stream=None):
        status = self._status_buffer()
        with torch.cuda.device(dev):
            s = torch.cuda.current_stream(dev)
            check(lib().moss_raster_read_status(img_buffer.data_ptr(), status.data_ptr(), s.cuda_stream), "read_status")
            ev = torch.cuda.Event(); ev.record(s)
        self.pending = (status, ev)

    def _clear_captured_status(self):
        """After a graph CAPTURE: ``last_img_buffer`` is the captured forward's image buffer -- memory of the graph's pool that nothing has
        written yet (a capture executes no kernel).  A ``check_status()`` before the first replay would read whatever the block held
        before (seen: the bits of 1.0f in word [3] -> a "needed capacity" of 10^9 and an 800 GB allocation at the next capture).  Its
        status words are zeroed here, eagerly: "nothing rendered, nothing needed" until a replay says otherwise."""
        img = self.last_img_buffer
        if img is not None and img.is_cuda and img.numel() * img.element_size() >= 32:
            img.view(torch.uint8).reshape(-1)[:32].zero_()

    def check_status(self, img_buffer=None):
        """Synchronously verify the most recent asynchronous forward (call outside graph capture, e.g. every N steps)."""
        if img_buffer is None:
            img_buffer = self.last_img_buffer
        if img_buffer is not None:
            self._request_status(img_buffer, img_buffer.device)
        self._consume_status(block=True)

    def _frame_state(self, dev, width, height):
        """The zero-initialised ``frame_state`` block of the asynchronous forwards (C ABI ``moss_raster_forward_async``) for this
        context (one per context: its forwards are ordered on one stream); a new one when the device changes or a larger image comes
        along.  With it no clear kernel runs per forward.

        A block that has been handed out is NEVER freed while the context lives: its address is a kernel argument of any hipGraph
        captured with this context (``GraphedStep``), and a graph replayed after, say, an eager full-resolution evaluation render on
        the same context would otherwise add its tile histograms into freed -- possibly reused -- memory.  Outgrown
        blocks (a few KB each) are parked in ``_retired_frame_states``; each stays all-zero between the forwards that use it."""
        n = int(lib().moss_raster_frame_state_bytes(int(width), int(height)))
        fs = self.frame_state
        if fs is None or fs.device != dev or fs.numel() < n:
            if fs is not None:
                self._retired_frame_states.append(fs)       # (the active block is never in the list: see below)
            for i, old in enumerate(self._retired_frame_states):   # an older block of this device that is large enough serves again
                if old.device == dev and old.numel() >= n:
                    self.frame_state = self._retired_frame_states.pop(i)
                    return self.frame_state
            fs = self.frame_state = torch.zeros(n, dtype=torch.uint8, device=dev)
        return fs

    def _dropped_words(self):
        return [fs.view(torch.int32)[FRAME_STATE_DROPPED_WORD:FRAME_STATE_DROPPED_WORD + 1]
                for fs in [self.frame_state] + self._retired_frame_states if fs is not None and fs.is_cuda]

    def read_dropped_frames(self, reset: bool = True) -> int:
        """How many asynchronous forwards of this context overflowed their capacity (each rendered NOTHING: zero gradients; a guarded
        optimizer step -- ``FlatAdamW.step(skip_word=frame_status_word(img))`` -- skips itself on such a frame) since the last reset:
        the sticky counter the library keeps in the frame state.  Overflows that were already RAISED to the caller as
        :class:`CapacityOverflow` are not counted again (``_consume_status`` takes them off).  One device sum, one host read;
        synchronises the device: call it outside graph capture (``GraphedStep.check`` does)."""
        words = self._dropped_words()
        if not words:
            return 0
        n = int(torch.stack([w.to(words[0].device) for w in words]).sum().item()) - self._raised_overflows
        if reset:
            for w in words:
                w.zero_()
            self._raised_overflows = 0
        return max(n, 0)

    # ---- gradient sinks ---------------------------------------------------------------------------------------------------------
    def set_grad_sink(self, sh=None, means3D=None, opacity=None, scales=None, rotations=None):
        self.sinks.update(sh=sh, means3D=means3D, opacity=opacity, scales=scales, rotations=rotations)

    def _sink(self, name):
        fn = self.sinks.get(name)
        return None if fn is None else fn()


DEFAULT = RasterContext()
ASYNC = DEFAULT            # the names round 1 exposed; both are the default context
GRAD_SINK = DEFAULT.sinks


def set_grad_sink(sh=None, means3D=None, opacity=None, scales=None, rotations=None, context=None):
    (context or DEFAULT).set_grad_sink(sh=sh, means3D=means3D, opacity=opacity, scales=scales, rotations=rotations)


def set_async(enabled: bool, capacity: int = 0, margin: float = 2.0, context=None):
    (context or DEFAULT).set_async(enabled, capacity, margin)


def check_async_status(img_buffer=None, context=None):
    (context or DEFAULT).check_status(img_buffer)


def frame_status_word(img_buffer):
    """The 32-bit status word of a forward's image buffer as a one-element int32 view (bit 0: a prefiltered point was culled, bit 1:
    the frame overflowed its capacity and rendered nothing): what ``FlatAdamW.step(skip_word=...)`` tests ON THE DEVICE."""
    return img_buffer.view(torch.int32)[2:3]


lib()   # fail at import time if the HIP library is missing: there is no fallback
ext()   # ... or if the compiled torch extension is


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, transforms=None, raw_flags=0, context=None, translation=None):
    """RasterizeGaussiansCUDA, rasterize_points.cu:35-119.
    ``raw_flags`` (an addition): RAW_OPACITY | RAW_SCALE | RAW_ROTATION -- those inputs are MOSS's raw parameters and the getters
    (sigmoid / exp / normalize) run inside the op (C ABI moss_raster_forward_raw); needs scales and rotations, no cov3D_precomp.
    ``transforms`` (an addition): (P,3,3) per-Gaussian matrices applied to the scale/rotation covariance
    inside the op (Sigma' = T Sigma T^T, what MOSS's Python get_covariance builds); needs scales and rotations, no cov3D_precomp.
    ``context`` (an addition): the :class:`RasterContext` whose asynchronous-forward policy applies (default: the shared one).
    ``raw_flags & RAW_POSE`` / ``translation`` (additions): means3D are the canonical positions; the op poses them with the transforms
    (and the optional (P,3) translation) itself -- gaussian_renderer/__init__.py:74-77 without the torch ops.
    Returns (num_rendered, out_color (3,H,W), out_depth (1,H,W), out_alpha (1,H,W), radii (P,), geomBuffer,
    binningBuffer, imgBuffer)."""
    cx = context or DEFAULT
    P = int(means3D.size(0)) if means3D.ndimension() >= 1 else 0
    use_async = cx.enabled and not (int(debug) & 1) and cx.capacity > 0 and P > 0 and means3D.is_cuda   # (bit 0 = the reference's debug: synchronous)
    capturing = use_async and torch.cuda.is_current_stream_capturing()
    if use_async and not capturing:
        cx._consume_status(block=False)
    res = ext().rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, float(scale_modifier), cov3D_precomp,
                                    viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy), int(image_height), int(image_width), sh,
                                    int(degree), campos, bool(prefiltered), int(debug), transforms, int(raw_flags),
                                    int(cx.capacity) if use_async else -1,
                                    cx._frame_state(means3D.device, image_width, image_height) if use_async else None, translation)
    rendered, img = res[0], res[7]
    if use_async:
        cx.last_img_buffer = img
        if not capturing and cx.pending is None:
            cx._request_status(img, means3D.device)
    elif cx.enabled and P > 0:                               # first (synchronous) call of an async session: learn the size
        learning = cx.capacity == 0                          # (debug bit 0 / CPU tensors also come here, with a capacity already learned)
        new_cap = int(rendered * cx.margin) + 1024
        if learning and cx._capacity_floor >= 1.25 * rendered:
            new_cap = cx._capacity_floor                     # re-learning (relearn_capacity): the old capacity still holds the new set
        if learning:
            cx._capacity_floor = 0
        cx.capacity = max(cx.capacity, new_cap)
        cx.last_needed = rendered
        if learning and means3D.is_cuda and rendered > 0:    # ... including what the frame asks of the record pool (status word [3])
            cx._consume_status(block=True)                   # (a report still pending from an earlier asynchronous frame is not lost)
            cx._request_status(img, means3D.device)
            cx._consume_status(block=True)                   # (grows the capacity if word [3] x 1.25 exceeds it)
    global last_num_rendered
    cx.last_num_rendered = last_num_rendered = cx.last_needed if use_async else rendered
    return res


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, dL_dout_alpha,
                                 sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, debug, transforms=None,
                                 raw_flags=0, opacities=None, context=None, translation=None, all_outputs=True):
    """RasterizeGaussiansBackwardCUDA, rasterize_points.cu:121-206.
    ``raw_flags`` / ``opacities``: backward of the raw-parameter forward (gradients w.r.t. the raw parameters).
    Returns (dL_dmeans2D (P,3), dL_dcolors (P,3), dL_dopacity (P,1), dL_dmeans3D (P,3), dL_dcov3D (P,6),
    dL_dsh (P,M,3), dL_dscales (P,3), dL_drotations (P,4)) -- plus dL_dtransforms (P,3,3) when ``transforms`` was given, plus
    dL_dtranslation (P,3) when ``translation`` was (RAW_POSE: dL_dmeans3D is then w.r.t. the canonical positions).
    ``all_outputs=False`` (what the autograd function passes): gradients nobody can receive -- dL_dcolors without colors_precomp,
    dL_dcov3D without cov3D_precomp, and the intermediate dL_dconic the reference fills and returns to nobody (rasterize_points.cu:160)
    -- are not computed into memory and come back as None; so do the gradients of tensors whose AdamW update the kernel applied
    itself (``context.fused_adamw``)."""
    cx = context or DEFAULT
    if int(raw_flags) and opacities is None and means3D.size(0) != 0:
        raise RuntimeError("the raw-parameter backward needs the raw opacities")
    fused = 0
    if cx.fused_adamw is not None and means3D.size(0) != 0:
        if not int(raw_flags):
            raise RuntimeError("this context's optimizer step is fused into the backward (FlatAdamW.fuse_into_backward): the op must be "
                               "given the raw parameters (raw_flags)")
        cx.fused_adamw.check_inputs(means3D=means3D, sh=sh, opacity=opacities, scales=scales, rotations=rotations)
        fused = cx.fused_adamw.address
    sinks = [None if (fused and n in cx.fused_adamw.param_ptrs) else cx._sink(n) for n in ("means3D", "opacity", "sh", "scales", "rotations")]
    if cx.sh_grad_active_only and int(raw_flags) and sinks[2] is not None:
        raw_flags = int(raw_flags) | SH_GRAD_ACTIVE_ONLY
    return tuple(ext().rasterize_gaussians_backward(
        background, means3D, radii, colors, scales, rotations, float(scale_modifier), cov3D_precomp, viewmatrix, projmatrix,
        float(tan_fovx), float(tan_fovy), dL_dout_color, dL_dout_depth, dL_dout_alpha, sh, int(degree), campos, geomBuffer, int(R),
        binningBuffer, imageBuffer, alphas, int(debug), transforms, int(raw_flags), opacities,
        *sinks, translation, fused, bool(all_outputs)))


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible, rasterize_points.cu:208-227: bool (P,), True where z_view > 0.2."""
    return ext().mark_visible(means3D, viewmatrix, projmatrix)

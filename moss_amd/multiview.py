"""B views per optimizer step on one GPU: the data-parallel step of SURVEY.md section 8(e) with the "ranks" on ONE device.

Per-view training is embarrassingly parallel (north_star); MOSS itself takes one view per iteration (train_ZJU.py:92-100).  With N GPUs
the frame-parallel step averages the N views' gradients and takes ONE AdamW step (``moss_amd.dist``).  A single 512 x 512 view cannot
fill an MI355X -- each of the step's kernels is a latency chain that leaves most issue slots empty -- so the same
semantics are offered on one device: B cameras in, B images out, the B gradient sets averaged IN A FIXED ORDER (view 0 + view 1 + ...,
then x 1/B: bitwise reproducible, and bit-identical to accumulating the same views one after the other), one AdamW step.

Every view runs the ordinary single-view chain -- ``render()`` -> fused loss -> backward -- on a ``RasterContext`` and a gradient buffer
of its own; with ``parallel_streams`` the B chains are issued on B HIP streams and joined in front of the update, inside ONE captured
hipGraph (parallel branches), so the kernels of different views overlap.  No kernel knows about views.

After a densification event (``FlatAdamW.append_rows`` / ``prune_rows`` on ``MultiViewStep.opt``) build a NEW ``MultiViewStep`` on the model:
the per-view gradient buffers, alias leaves and capacities all follow the row count (``moss_amd.surgery.densification_event`` re-captures
one context and one ``GraphedStep``: the single-view step).
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import dist as mdist
from . import loss as mloss
from .diff_gaussian_rasterization import RasterContext
from .gaussian_renderer import render
from .graphs import GraphedStep
from .optim import FlatAdamW

__all__ = ["MultiViewStep", "MultiViewRender"]

_NAMES = ("_xyz", "_features", "_opacity", "_scaling", "_rotation")


class _ViewAlias:
    """What ``render()`` reads of a GaussianModel, with leaves of its own that ALIAS the model's parameter storage: view b's backward
    hangs its gradients on these, so no two views ever touch the same ``.grad`` (autograd would add the second into the first)."""

    def __init__(self, pc):
        self.max_sh_degree, self.active_sh_degree = pc.max_sh_degree, pc.active_sh_degree
        self.unified_features = True
        self.spatially_ordered = getattr(pc, "spatially_ordered", False)
        for n in _NAMES:
            setattr(self, n, getattr(pc, n).detach().requires_grad_(True))       # same storage, a separate autograd leaf

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return self._features


class MultiViewStep:
    def __init__(self, pc, B, cameras, targets, bg, transforms=None, translation=None, parallel_streams=True, eps=1e-15, fused_sum=True):
        """``pc``: a ``GaussianSet`` with unified SH; ``cameras`` / ``targets``: B camera views and B (image, mask) pairs; ``transforms``
        (P,3,3) / ``translation`` (P,3): the frame's LBS table (one for all views here, or a list of B)."""
        if not getattr(pc, "unified_features", False):
            raise ValueError("MultiViewStep works on a GaussianSet(unified_features=True)")
        self.pc, self.B, self.bg = pc, int(B), bg
        self.cameras, self.targets = list(cameras), list(targets)
        per_view = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * self.B
        self.transforms, self.translation = per_view(transforms), per_view(translation)
        if not 1 <= self.B <= 4:
            raise ValueError("1 to 4 views per step")
        self.parallel_streams = bool(parallel_streams) and self.B > 1
        self.fused_sum = bool(fused_sum) and self.B > 1        # False: the average by torch ops in front of the update (the reference form)
        params = [getattr(pc, n) for n in _NAMES]
        self.bucket = bucket = mdist.GradBucket(params)
        self.opt = FlatAdamW(pc.param_groups(), bucket, eps=eps, capturable=True)
        dev = params[0].device
        self.dev = dev
        # view 0 writes its gradients straight into the optimizer's bucket; the others into buffers of the same layout
        self.flats = [bucket.flat] + [torch.zeros_like(bucket.flat) for _ in range(self.B - 1)]
        self.views = []
        for b in range(self.B):
            alias = _ViewAlias(pc)
            cx = RasterContext()
            cx.set_async(True)
            flat = self.flats[b]
            # (a NEW view object per request: autograd adopts a fresh, exclusively owned tensor as .grad without copying it)
            where = {n: (off, p.numel(), tuple(p.shape)) for n, p, off in zip(_NAMES, params, bucket.offsets)}
            sink = lambda n, flat=flat, where=where: flat[where[n][0]:where[n][0] + where[n][1]].view(where[n][2])
            cx.set_grad_sink(means3D=lambda k=sink: k("_xyz"), sh=lambda k=sink: k("_features"), opacity=lambda k=sink: k("_opacity"),
                             scales=lambda k=sink: k("_scaling"), rotations=lambda k=sink: k("_rotation"))
            pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=False,
                                   transforms_in_op=True, pose_in_op=True, raw_parameters_in_op=True, raster_context=cx)
            self.views.append(SimpleNamespace(alias=alias, ctx=cx, pipe=pipe, terms=flat[bucket.tail:bucket.tail + 4],
                                              stream=None if b == 0 else torch.cuda.Stream(dev)))
        self.images = [None] * self.B
        self.graphed = None

    def _view(self, b):
        v = self.views[b]
        for n in _NAMES:
            getattr(v.alias, n).grad = None
        out = render(self.cameras[b], v.alias, v.pipe, self.bg, transforms=self.transforms[b], translation=self.translation[b])
        gt, mask = self.targets[b]
        loss = mloss.training_loss_fused(out["render"], out["render_alpha"], gt, mask, terms_out=v.terms)
        mloss.backward_from_loss(loss)
        self.images[b] = out["render"].detach()

    def compute(self):
        """The B chains (in parallel branches when asked for), the fixed-order average, ONE AdamW step."""
        cur = torch.cuda.current_stream(self.dev)
        if self.parallel_streams:
            for b in range(1, self.B):
                self.views[b].stream.wait_stream(cur)
            self._view(0)
            for b in range(1, self.B):
                with torch.cuda.stream(self.views[b].stream):
                    self._view(b)
            for b in range(1, self.B):
                cur.wait_stream(self.views[b].stream)
        else:
            for b in range(self.B):
                self._view(b)
        if self.fused_sum:
            # ((g0 + g1) + g2 ...) x 1/B inside the update kernel (C ABI moss_adamw_flat_ex: grads_extra): no pass of its own over the buffers
            self.opt.step(extra_grads=self.flats[1:], grad_scale=1.0 / self.B)
        else:
            n = self.bucket.n_params
            acc = self.flats[0][:n]
            for b in range(1, self.B):                         # fixed order: ((g0 + g1) + g2) + ...
                acc.add_(self.flats[b][:n])
            if self.B > 1:
                acc.mul_(1.0 / self.B)
            self.opt.step()
        return {"images": list(self.images)}

    eager_step = compute

    def capture(self, warmup=2):
        self.graphed = GraphedStep(self.compute, warmup=warmup, device=self.dev, context=self.views[0].ctx,
                                   extra_contexts=[v.ctx for v in self.views[1:]])
        return self.graphed

    def step(self):
        return (self.graphed or self.compute)()

    def check(self):
        """Every view's last frame fitted its capacity (raises ``CapacityOverflow`` otherwise; re-capture after it)."""
        for v in self.views:
            v.ctx.check_status()


class MultiViewRender:
    """B evaluation renders at a time (render_ZJU.py:56-72 renders its test views one after the other): the B forward-only chains --
    ``render()`` under ``torch.no_grad()``, each on a ``RasterContext`` of its own -- on B HIP streams inside one captured hipGraph.
    A 512 x 512 render is a chain of latency-bound kernels that leaves most of the device idle; B of them overlap.  ``cameras`` may be
    replaced between replays only by cameras whose tensors are updated IN PLACE (the graph holds their addresses)."""

    def __init__(self, pc, cameras, bg, transforms=None, translation=None, parallel_streams=True, pipe_flags=None):
        self.pc, self.bg = pc, bg
        self.cameras = list(cameras)
        self.B = len(self.cameras)
        per_view = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * self.B
        self.transforms, self.translation = per_view(transforms), per_view(translation)
        self.parallel_streams = bool(parallel_streams) and self.B > 1
        dev = pc._xyz.device
        self.dev = dev
        self.views = []
        for b in range(self.B):
            cx = RasterContext()
            cx.set_async(True)
            flags = dict(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=False,
                         transforms_in_op=transforms is not None, pose_in_op=transforms is not None,
                         raw_parameters_in_op=all(hasattr(pc, a) for a in ("_opacity", "_scaling", "_rotation")), raster_context=cx)
            flags.update(pipe_flags or {})
            self.views.append(SimpleNamespace(ctx=cx, pipe=SimpleNamespace(**flags), stream=None if b == 0 else torch.cuda.Stream(dev)))
        self.graphed = None

    def _view(self, b):
        v = self.views[b]
        with torch.no_grad():
            out = render(self.cameras[b], self.pc, v.pipe, self.bg, transforms=self.transforms[b], translation=self.translation[b])
        return out["render"], out["render_depth"], out["render_alpha"]

    def compute(self):
        cur = torch.cuda.current_stream(self.dev)
        outs = [None] * self.B
        if self.parallel_streams:
            for b in range(1, self.B):
                self.views[b].stream.wait_stream(cur)
            outs[0] = self._view(0)
            for b in range(1, self.B):
                with torch.cuda.stream(self.views[b].stream):
                    outs[b] = self._view(b)
            for b in range(1, self.B):
                cur.wait_stream(self.views[b].stream)
        else:
            for b in range(self.B):
                outs[b] = self._view(b)
        return outs

    def capture(self, warmup=2):
        self.compute()                                       # (the first, synchronous renders size the capacities)
        torch.cuda.synchronize(self.dev)
        self.graphed = GraphedStep(self.compute, warmup=warmup, device=self.dev, context=self.views[0].ctx,
                                   extra_contexts=[v.ctx for v in self.views[1:]])
        return self.graphed

    def __call__(self):
        """[(image (3,H,W), depth (1,H,W), alpha (1,H,W))] of the B views (the captured graph's static outputs)."""
        return (self.graphed or self.compute)()

    def check(self):
        for v in self.views:
            v.ctx.check_status()

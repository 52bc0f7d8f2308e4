#!/usr/bin/env python3
"""bench.py -- train iters/sec of the rasterizer hot path (BASELINE.json metric) on N MI355X of one node.

One STEP = one training iteration on one view of the `configs[2]` workload (100k Gaussians, 512x512, SH degree 3):
  render() -> HIP rasterizer forward -> L1 + 0.2*(1-SSIM) + 0.5*maskL2 -> backward (HIP rasterizer backward, down to the raw
  Gaussian parameters) -> [N>1: one RCCL all-reduce of the flat gradient bucket] -> AdamW step.
Inputs are resident in HBM before the timed region.  N>1 is frame-parallel (one camera per rank, replicas of the
Gaussians, weak scaling): value = N * steps / max-over-ranks time.

Launch: python bench.py [--gpus N --steps K --warmup W]
  N > 1 without a torchrun environment: this process starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
  --master-addr 127.0.0.1 --master-port <free> bench.py ...` itself (BEFORE it touches a GPU), relays rank 0's JSON line and exits
  non-zero if any rank failed.  Under torchrun (RANK / WORLD_SIZE set, as the driver launches it) it is one rank of the job.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s nominal
HBM_ACHIEVABLE_GBS = 5924.0           # measured on this part: scripts/hbm_bandwidth.py (triad, 1 GiB buffers), profiles/r01_hbm_bandwidth.json
MIN_TIMED_MS = 50.0                   # a timed region shorter than this is re-measured over LONG_STEPS steps as well (reported beside)
LONG_STEPS = 200


def algorithmic_bytes(P, Pv, R, N, tiles, K, scale_rot_mode, transforms=False, fused_adamw=False):
    """bytes that MUST cross HBM per fwd+bwd step, from the measured P, Pv, R, N.  transforms: the in-op
    LBS covariance (row n2) reads 36 B per Gaussian more in both directions and writes dL_dtransforms (36 B per Gaussian).
    fused_adamw: the per-Gaussian backward also takes the optimizer step (moss_raster_backward_raw_adamw): per parameter (11 + 3K per
    Gaussian, ALL P of them) both moments read, parameter and both moments written = 20 B, the parameters of the Gaussians that were
    not rendered read as well, and the gradients of the updated tensors (position, SH, opacity, scale, rotation) no longer written."""
    c = (28 if scale_rot_mode else 24) + (36 if transforms else 0)
    fwd = {
        "preprocess_fwd": P * (12 + c + 4) + Pv * 12 * K + P * 8 + Pv * 43,
        "scan": 8 * P,
        "scatter": Pv * 24 + 12 * R,
        "chunk_sort": 16 * R + 8 * tiles,                    # keys read + written in place
        "merge_gather": 8 * R + 8 * R,                        # keys read, sorted ids + per-tile ranges written (24R + 8R + 8 tiles for the sort)
        "blend_fwd": 44 * R + 28 * N,
    }
    bwd = {
        "blend_bwd": 44 * R + 28 * N + 36 * Pv,
        # (dL_dcov3D, 24 B, is written only when the covariance was an input: cov3D_precomp)
        "preprocess_bwd": 36 * Pv + Pv * (12 + 4 + 24 + 12 * K + 3) + Pv * (12 + (0 if scale_rot_mode else 24) + 12 * K + 4) + (28 * Pv if scale_rot_mode else 0) + P * 12
                          + (Pv * 36 + P * 36 if transforms else 0),
    }
    if fused_adamw:
        n_par = 11 + 3 * K
        bwd["preprocess_bwd"] += P * n_par * 20 + (P - Pv) * n_par * 4 - Pv * (12 + 12 * K + 4 + 28)
    return fwd, bwd


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg5", "moss7k", "moss45k", "moss45k_512"],
                    help="cfg2 / cfg3 / cfg5 = BASELINE configs[1] / [2] / [4]; the others are not BASELINE configurations (analysis aids): "
                         "MOSS's own floor and ceiling -- 6 890 (scene/dataset_readers.py:720) and 45 695 Gaussians "
                         "(scene/gaussian_model.py:496).  moss45k_512 = the ceiling at MOSS's ZJU-MoCap resolution, 512 x 512 (ZJU frames are "
                         "loaded at image_scaling 0.5, scene/dataset_readers.py:540; cfg2 is the floor at that resolution); moss7k / moss45k = "
                         "floor / ceiling at 1024 x 1024, the MonoCap resolution (image_scaling 1.0, :299)")
    ap.add_argument("--mode", default="lbs", choices=["precomp", "scale_rot", "lbs", "lbs_python"],
                    help="lbs (default: MOSS's data flow, gaussian_renderer/__init__.py:88-93 -- scales, rotations and a per-Gaussian "
                         "3x3 LBS transform that changes every frame -- with the covariance product inside the op); "
                         "scale_rot = cov3D computed inside the rasterizer from scales+rotations (the reference's "
                         "compute_cov3D_python=False path, gaussian_renderer/__init__.py:92-93); precomp = cov3D computed by "
                         "torch ops and passed in (MOSS's shipped default, arguments/__init__.py:60); lbs = per-Gaussian 3x3 LBS-like "
                         "transforms applied to the covariance INSIDE the op (extension row n2) vs lbs_python = the same transforms "
                         "through MOSS's Python get_covariance (scene/gaussian_model.py:37-44)")
    ap.add_argument("--torch-adamw", action="store_true", help="use torch.optim.AdamW instead of the flat fused HIP AdamW")
    ap.add_argument("--forward", default="async", choices=["sync", "async"],
                    help="sync = the reference's behaviour (the host reads num_rendered back in every forward); async = "
                         "capacity-bounded forward with no host read-back (moss_raster_forward_async)")
    ap.add_argument("--graph", type=int, default=1, choices=[0, 1],
                    help="1 = capture the per-rank compute of a step (render, loss, backward[, AdamW]) in one hipGraph and replay "
                         "it (needs --forward async); 0 = launch every kernel eagerly")
    ap.add_argument("--torch-activations", action="store_true",
                    help="compute the parameter activations (exp / sigmoid / normalize / cat) with torch ops and let autograd "
                         "accumulate into the bucket, instead of the fused HIP activation kernels writing into it")
    ap.add_argument("--activations", default="in_op", choices=["in_op", "fused"],
                    help="in_op = the rasterizer takes the raw _opacity / _scaling / _rotation and runs sigmoid / exp / normalize inside "
                         "its preprocess kernels (moss_raster_forward_raw: no activation launches); fused = one activation kernel "
                         "each way (moss_gaussian_activate_*).  --torch-activations overrides both")
    ap.add_argument("--target", default="body", choices=["body", "smooth"],
                    help="ground truth of the photometric loss: body = a render of a DIFFERENT random Gaussian body (other points, "
                         "other colours) through the same camera, mask = its alpha > 0.5 -- a masked person on black, like MOSS's "
                         "ZJU-MoCap frames; smooth = a full-frame smooth colour field (drives a few dozen Gaussians to cover the "
                         "whole image within ~250 steps, a regime real captures do not have)")
    ap.add_argument("--order", default="as_generated", choices=["as_generated", "morton"],
                    help="index order of the synthetic Gaussians: as generated (uncorrelated with position -- the default, and the "
                         "least favourable) or re-indexed along a Morton curve (moss_amd.densify.spatial_order); a side experiment, "
                         "named in config.workload when used")
    ap.add_argument("--fused-optimizer", type=int, default=1, choices=[0, 1],
                    help="1 (default): the per-Gaussian backward kernel takes the AdamW step of the parameters it differentiates "
                         "(FlatAdamW.fuse_into_backward) wherever the step is local to the rank (N = 1, loss_only); 0: gradients into the "
                         "bucket, then the flat AdamW kernel (always so for the gradient exchanges).  Same bits either way")
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "sharded", "loss_only"],
                    help="N > 1 only: which form of the step `value` reports (ALL THREE are measured by every N > 1 run and reported at top "
                         "level as value_allreduce / value_sharded / value_loss_only with their ms_per_step_* and replicas_identical_*). "
                         "allreduce (the default: SURVEY 8e, the data-parallel form that exercises RCCL; rounds 1-3 and 5) = ONE model, "
                         "frames sharded over the ranks, one all-reduce (mean) of the flat gradient bucket, then the full AdamW on every "
                         "rank; sharded = reduce-scatter, AdamW on the rank's 1/N of the parameters (moments memory and update time / N), "
                         "all-gather of the updated parameters (moss_amd.dist.ShardedStep); loss_only (round 4's headline) = BASELINE "
                         "configs[3] as written -- \"frames of six subjects sharded across 8 GPUs, RCCL loss all-reduce\": every rank "
                         "trains its OWN model on its own frames, exactly the N = 1 step, and RCCL all-reduces the 4-float loss block only")
    ap.add_argument("--debug-bits", type=int, default=0,
                    help="OR-ed into the op's `debug` argument (include/moss_raster.h MOSS_DEBUG_*): 8 = MOSS_DEBUG_TRACE, roctx ranges around "
                         "every stage launcher (use with --graph 0 under `rocprofv3 --kernel-trace --marker-trace`); 4 = "
                         "MOSS_DEBUG_EXACT_MATH (checking mode, slower blend kernels: not a valid `value`)")
    ap.add_argument("--loss", default="full", choices=["full", "moss"],
                    help="full (default, the metric's form since round 1: L1 / SSIM / mask L2 over the whole frame) or moss: MOSS's OWN expression "
                         "(train_ZJU.py:108-119: L1 and mask L2 over the view's bound_mask, SSIM on its bounding rectangle; "
                         "moss_photometric_loss_roi) -- an analysis aid: `config.loss` says which, the driver's default command is `full`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-callers", action="store_true", help="skip the drop-in / lbs-in-op caller variants reported beside the headline")
    ap.add_argument("--callers-only", default="", help="comma-separated names: measure only these caller variants (profiling aid, e.g. "
                                                         "rocprofv3 --kernel-trace --stats -- python3 bench.py --callers-only patched_moss_pattern)")
    ap.add_argument("--cpu-iters", type=int, default=20,
                    help="iterations of the CPU oracle baseline (about 0.57 s each on one core: 20 = the 10-30 s sample the contract asks for)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="TEST HOOK (tests/test_host_cpu.py): no GPU work at all -- every rank stands in for its step with a host-side "
                         "gradient bucket, so that the launcher, the rendezvous, the all-reduce, the max-over-ranks timing and the JSON "
                         "schema of the N > 1 path can be exercised on a machine without GPUs (backend gloo)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 started as a plain `python bench.py --gpus N`: become the launcher.  Nothing here may touch the GPU -- on this pool a
# process that has initialised HIP must not exec or be replaced, so the children are started first and this process only waits.

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(argv):
    args = parse_args(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"[bench] {args.gpus}-rank job failed (exit code {proc.returncode}, JSON line {'missing' if line is None else 'present'})", file=sys.stderr)
        return proc.returncode or 1
    res = json.loads(line)
    if res.get("n_gpus") != args.gpus:
        print(f"[bench] asked for {args.gpus} ranks, the job reported {res.get('n_gpus')}", file=sys.stderr)
        return 1
    print(line)
    return 0


def _synthetic_bound_mask(gt_mask, grow=12):
    """(1,H,W) uint8: the silhouette's bounding box grown by ``grow`` pixels with its corners cut (an octagon) -- the shape MOSS's
    projected 3-D body box has (a convex region a little larger than the person)."""
    import torch
    m = gt_mask.reshape(gt_mask.shape[-2], gt_mask.shape[-1]) > 0
    H, W = m.shape
    ys, xs = m.any(1).nonzero().flatten(), m.any(0).nonzero().flatten()
    if ys.numel() == 0:
        return torch.ones(1, H, W, dtype=torch.uint8, device=m.device)
    x0, x1 = max(int(xs[0]) - grow, 0), min(int(xs[-1]) + grow + 1, W)
    y0, y1 = max(int(ys[0]) - grow, 0), min(int(ys[-1]) + grow + 1, H)
    yy, xx = torch.meshgrid(torch.arange(H, device=m.device), torch.arange(W, device=m.device), indexing="ij")
    cut = (min(x1 - x0, y1 - y0)) // 4
    dx, dy = torch.minimum(xx - x0, x1 - 1 - xx), torch.minimum(yy - y0, y1 - 1 - yy)
    box = (xx >= x0) & (xx < x1) & (yy >= y0) & (yy < y1) & (dx + dy >= cut)
    return box.to(torch.uint8)[None]


class Harness:
    """One configuration of the training step on this rank: model, optimizer, gradient bucket, rasterizer context, step functions."""

    n_exchange = 0

    def __init__(self, args, dev, rank, world, scene, cam, gt, gt_mask, bg, *, mode, activations, torch_activations, torch_adamw,
                 forward, graph, fused_loss=True, caller_side=None, lbs_T=None, exchange="allreduce", fused_optimizer=True,
                 raw_in_op=False, sh_degree=3):
        import torch
        from types import SimpleNamespace
        from moss_amd import dist as mdist
        from moss_amd import diff_gaussian_rasterization as dgr
        from moss_amd.gaussian_model import GaussianSet
        from moss_amd.gaussian_renderer import render
        from moss_amd import loss as mloss
        self.torch, self.dev, self.world, self.args = torch, dev, world, args
        self.mode, self.forward = mode, forward
        self._cam, self._bg = cam, bg
        self.ctx = dgr.RasterContext()                       # this harness's own asynchronous-forward state and gradient sinks
        unified = not torch_adamw and not torch_activations
        # (below degree 3: MOSS's own state -- the coefficients above the active degree are still the zeros they were created as,
        # scene/gaussian_model.py:179-181, and have never received a gradient)
        self.pc = pc = GaussianSet(scene, sh_degree=sh_degree, device=dev, unified_features=unified, zero_inactive_sh=sh_degree < 3)
        self.pipe = pipe = SimpleNamespace(
            convert_SHs_python=False, compute_cov3D_python=(mode in ("precomp", "lbs_python")), debug=int(getattr(args, "debug_bits", 0)),
            fused_activations=not torch_activations, transforms_in_op=(mode == "lbs"),
            pose_in_op=(mode == "lbs" and not torch_activations),    # the op poses the canonical positions itself (MOSS_RAW_POSE)
            # (raw_in_op: MOSS's own parameter tensors -- separate f_dc / f_rest, torch.optim -- with only the three getters moved into
            # the op: what `pipe.raw_parameters_in_op = True` in patches/train_ZJU.diff does)
            raw_parameters_in_op=((not torch_activations and activations == "in_op" and unified and mode in ("scale_rot", "lbs"))
                                  or (raw_in_op and mode in ("scale_rot", "lbs"))),
            raster_context=self.ctx)
        self.lbs_T = lbs_T
        self.exchange_kind = exchange if (world > 1 and not torch_adamw) else "allreduce"
        # loss_only: independent models -- the optimizer step is local to the rank (inside the captured step), like at N = 1
        self.local_opt = world == 1 or self.exchange_kind == "loss_only"
        self.bucket = bucket = mdist.GradBucket(list(pc.parameters()), world=world if self.exchange_kind == "sharded" else 1)
        pipe.grad_bucket = bucket
        self.sharded = None
        if torch_adamw == "moss_amd":
            # MOSS's optimizer construction with the class swapped (patches/gaussian_model.diff): same groups, same per-group lr, same
            # state keys -- one kernel for the six single-tensor groups (moss_adamw_multi) instead of torch's nine multi-tensor launches per group
            from moss_amd.optim import AdamW as MossAdamW
            self.opt = MossAdamW(pc.param_groups(), lr=0.0, eps=1e-15)
        elif torch_adamw:
            self.opt = torch.optim.AdamW(pc.param_groups(), lr=0.0, eps=1e-15)          # scene/gaussian_model.py:226
        else:
            from moss_amd.optim import FlatAdamW
            # same rule, one kernel over the bucket (sharded: over this rank's 1/N of it, between a reduce-scatter and an all-gather)
            self.opt = FlatAdamW(pc.param_groups(), bucket, eps=1e-15, capturable=True,
                                 shard=(rank, world) if self.exchange_kind == "sharded" else None)
            if self.exchange_kind == "sharded":
                self.sharded = mdist.ShardedStep(bucket, self.opt, rank, world)
            if unified:
                # the degree-aware SH update (MOSS trains below degree 3 for 2999 of 3000 iterations): moments of never-active coefficients
                # are not touched, their known-zero parameters neither; dL_dsh is written for the active coefficients only
                self.opt.set_active_sh_degree(sh_degree)
                self.ctx.sh_grad_active_only = True
        self.use_graph = bool(graph) and forward == "async" and not torch_adamw and caller_side is None
        self.ctx.set_async(forward == "async")
        if unified:
            sinks = {"sh": lambda: bucket.sink_for(pc._features)}             # dL_dsh is written straight into the gradient bucket
            if pipe.raw_parameters_in_op:                                     # ... and so are the raw-parameter gradients
                sinks.update(opacity=lambda: bucket.sink_for(pc._opacity), scales=lambda: bucket.sink_for(pc._scaling),
                             rotations=lambda: bucket.sink_for(pc._rotation))
                if lbs_T is None or pipe.pose_in_op:                          # (posed OUTSIDE the op, the means are not the parameter)
                    sinks["means3D"] = lambda: bucket.sink_for(pc._xyz)
            self.ctx.set_grad_sink(**sinks)
        # The optimizer step INSIDE the backward kernel (FlatAdamW.fuse_into_backward, C ABI moss_raster_backward_raw_adamw): whenever the
        # step is local to the rank and the op is given the parameters themselves (raw, unified SH; positions posed inside the op)
        self.fused_opt = bool(fused_optimizer and self.local_opt and not torch_adamw and pipe.raw_parameters_in_op and caller_side is None
                              and (lbs_T is None or pipe.pose_in_op))
        if self.fused_opt:
            # (N > 1: only ever with the loss-only exchange -- every rank trains a model of its own -- which is what local_only asserts)
            self.opt.fuse_into_backward(self.ctx, means3D=pc._xyz, sh=pc._features, opacity=pc._opacity, scales=pc._scaling,
                                        rotations=pc._rotation, local_only=(world > 1 and self.exchange_kind == "loss_only"))
        # fused_loss: True = the whole loss as the two HIP kernels; "ssim" = MOSS's torch loss expression with ONLY its ssim() call
        # replaced (moss_amd.loss.ssim_fused: what patches/train_ZJU.diff does); False = the reference's torch functions throughout
        import functools
        training_loss = (mloss.training_loss_fused if fused_loss is True else
                         functools.partial(mloss.training_loss, ssim_fn=mloss.ssim_fused) if fused_loss == "ssim" else mloss.training_loss)
        if fused_loss == "moss":
            # MOSS's OWN expression (train_ZJU.py:108-119): L1 / mask L2 over the view's bound_mask, SSIM on its bounding rectangle, from
            # the same two kernels (C ABI moss_photometric_loss_roi).  The bound_mask of a synthetic view: MOSS projects the 3-D box of
            # the body (a convex region around the person); here the target silhouette's bounding box grown by 12 px, corners cut.
            self.region = region = mloss.ViewRegion(_synthetic_bound_mask(gt_mask))
            training_loss = lambda img, a, g_, m_, terms_out=None: mloss.training_loss_moss_fused(img, a, g_, m_, region, terms_out=terms_out)
        self.caller_side = caller_side
        stats = None
        if caller_side == "torch":
            # the three statistics MOSS keeps on GaussianModel (scene/gaussian_model.py:198,204-205), updated with its own torch expressions
            P = scene.means3D.shape[0]
            stats = SimpleNamespace(xyz_gradient_accum=torch.zeros((P, 1), device=dev), denom=torch.zeros((P, 1), device=dev),
                                    max_radii2D=torch.zeros((P,), device=dev))
        elif caller_side == "fused":
            from moss_amd.densify import DensifyStats
            stats = DensifyStats(scene.means3D.shape[0], device=dev)
        self.stats = stats

        def compute():                      # everything of a step that is local to this rank
            if torch_adamw:
                self.opt.zero_grad(set_to_none=True)
            elif pipe.fused_activations:
                bucket.detach_grads()       # gradients are WRITTEN into the bucket by the backward kernels
            else:
                bucket.attach()             # zero the bucket; autograd accumulates into it
            out = render(cam, pc, pipe, bg, transforms=self.lbs_T)      # (self.lbs_T: re-indexed by a densification event)
            if fused_loss is True or fused_loss == "moss":
                # the loss kernels write [loss, L1, SSIM, mask] into the bucket's tail: it travels with the gradients, no copy
                loss = training_loss(out["render"], out["render_alpha"], gt, gt_mask, terms_out=bucket.loss_terms)
                mloss.backward_from_loss(loss)
            else:
                loss = training_loss(out["render"], out["render_alpha"], gt, gt_mask)
                loss.backward()
            if caller_side == "torch":
                # train_ZJU.py:101,171-174 + GaussianModel.add_densification_stats (scene/gaussian_model.py:815-817), as written there
                vis, radii, vsp = out["visibility_filter"], out["radii"], out["viewspace_points"]
                stats.max_radii2D[vis] = torch.max(stats.max_radii2D[vis], radii[vis])
                stats.xyz_gradient_accum[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
                stats.denom[vis] += 1
            elif caller_side == "fused":
                stats.add(out["radii"], out["viewspace_points"].grad)
                _ = out["visibility_filter"]
            if not torch_adamw and pipe.fused_activations and not self.fused_opt:
                bucket.collect()
            if self.fused_opt:
                pass                        # the backward kernel took the step (and skipped it for a frame that overflowed)
            elif self.local_opt:
                # a frame that overflowed its capacity rendered nothing: the update kernel reads the frame's status word and skips
                # itself (inside a captured step nobody else can; moss_adamw_flat_guarded)
                img = None if torch_adamw else self.ctx.last_img_buffer
                if img is not None and forward == "async":
                    self.opt.step(skip_word=dgr._C.frame_status_word(img))
                else:
                    self.opt.step()
            # detached: holding an output with a grad_fn would keep this step's autograd graph (and its AccumulateGrad nodes,
            # bound to the stream they were created on) alive into the next step / into graph capture
            return {"radii": out["radii"]}

        self.compute = compute
        self.step = self.eager_step
        self.graphed = None
        self.graph_note = "eager launches"
        self.t_phase = [0.0, 0.0, 0.0]       # allreduce: [all-reduce, AdamW, -]; sharded: [reduce-scatter, AdamW on the shard, all-gather]
        self._ev = None
        self._ev_pending = False

    def exchange(self):
        """N > 1, after the rank-local part of the step.  allreduce: ONE RCCL all-reduce (mean) of the flat gradient bucket (+ loss
        block), then the full AdamW.  sharded: reduce-scatter, AdamW on this rank's shard, all-gather of the parameters
        (moss_amd.dist.ShardedStep, spelled out here so that each phase sits between two events)."""
        torch, mdist = self.torch, sys.modules["moss_amd.dist"]
        if self._ev is None:
            self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if self._ev_pending:                                 # the previous step's pairs (long complete: no stall)
            self._ev[3].synchronize()
            for i in range(3):
                self.t_phase[i] += self._ev[i].elapsed_time(self._ev[i + 1])
            self.n_exchange += 1
        e = self._ev
        e[0].record()
        if self.exchange_kind == "loss_only":
            self.bucket.all_reduce_loss_only(self.world)
            e[1].record(); e[2].record(); e[3].record()
        elif self.sharded is None:
            # (only the active SH coefficients travel while the degree is below its maximum: MOSS raises it every 1000 iterations; the
            # bench trains at degree 3, where this is the plain all-reduce)
            self.bucket.all_reduce_mean(None, self.world, sh_param=getattr(self.pc, "_features", None) if self.pc.unified_features else None,
                                        active_sh_degree=self.pc.active_sh_degree)
            e[1].record()
            self.opt.step()
            e[2].record(); e[3].record()
        else:
            sh, per = self.sharded, self.bucket.shard_len
            mdist._reduce_scatter_mean(self.opt.grad_shard, self.bucket.flat, self.world)
            e[1].record()
            self.opt.step()
            if sh.rank == sh.tail_rank:
                sh.loss_terms.copy_(self.opt.grad_shard[sh.tail_off:sh.tail_off + 4])
            e[2].record()
            torch.distributed.all_gather_into_tensor(self.opt.flat_params, self.opt.flat_params[sh.rank * per:(sh.rank + 1) * per])
            e[3].record()
        self._ev_pending = True

    def exchange_report(self):
        """{phase: ms per step} over the steps since the counters were reset."""
        n = max(self.n_exchange, 1)
        t = [round(x / n, 4) for x in self.t_phase]
        if self.exchange_kind == "loss_only":
            return {"loss_allreduce_ms": t[0], "bytes_reduced": 16, "adamw_elements": int(self.opt.count),
                    "adamw": "inside the rank-local step" + (": taken by the per-Gaussian backward kernel" if self.fused_opt else "")}
        if self.sharded is None:
            return {"allreduce_ms": t[0], "adamw_ms": t[1], "bytes_reduced": int(self.bucket.flat.numel() * 4), "adamw_elements": int(self.opt.count)}
        return {"reduce_scatter_ms": t[0], "adamw_ms": t[1], "all_gather_ms": t[2], "bytes_reduced": int(self.bucket.flat.numel() * 4),
                "adamw_elements": int(self.opt.count)}

    def eager_step(self):
        out = self.compute()
        if self.world > 1:
            self.exchange()
        return out

    def capture(self):
        """The first (synchronous) forward sized the binning buffer; nothing in compute() talks to the host any more, so the whole
        per-rank step is captured once and replayed: its launches become one hipGraphLaunch."""
        from moss_amd.graphs import GraphedStep
        self.graphed = graphed = GraphedStep(self.compute, warmup=3, device=self.dev, context=self.ctx)
        replays = [0]

        def graph_step():
            graphed()
            replays[0] += 1
            if replays[0] % 512 == 0:        # long runs: the scene grows while it trains; re-capture before the baked-in
                graphed.check()              # binning capacity overflows (one synchronisation per 512 steps)
            if self.world > 1:
                self.exchange()
            return graphed.outputs           # (re-bound by a re-capture: never cache it)

        for _ in range(5):
            graph_step()
        self.torch.cuda.synchronize(self.dev)
        self.step = graph_step
        self.graph_note = "one hipGraph replay per step" + ((" + eager RCCL all-reduce of the 16-byte loss block" if self.exchange_kind == "loss_only" else
                                                             " + eager RCCL all-reduce and AdamW" if self.sharded is None else
                                                             " + eager RCCL reduce-scatter, AdamW on the rank's shard, all-gather") if self.world > 1 else "")

    def densify_event(self, step, reset_opacity=False):
        """One scripted densification event on this harness (moss_amd.scenes.scripted_densification -> moss_amd.surgery.densification_event):
        clone + split + prune of the CURRENT set, optimizer moments / bucket / statistics / LBS table re-laid-out, capacity re-learned by a
        forward-only probe, the captured step re-captured.  MOSS does this every 100 iterations from 400 to 2000 (train_ZJU.py:171-186)."""
        torch = self.torch
        from moss_amd import scenes
        from moss_amd.gaussian_renderer import render
        from moss_amd.surgery import densification_event
        pc = self.pc
        t = {"xyz": pc._xyz.data, "f_dc": pc._features_dc.data, "f_rest": pc._features_rest.data, "opacity": pc._opacity.data,
             "scaling": pc._scaling.data, "rotation": pc._rotation.data}
        ev = scenes.scripted_densification(t, step, self.dev, reset_opacity=reset_opacity)

        def probe():                                         # forward only (no gradient, no optimizer step): sizes the capacity of the new set
            with torch.no_grad():
                render(self._cam, pc, self.pipe, self._bg, transforms=self.lbs_T)
        pg = {} if self.lbs_T is None else {"T": self.lbs_T}
        return densification_event(pc, self.opt, append=ev["append"], prune=ev["prune"], reset_opacity=ev["reset_opacity"],
                                   stats=self.stats if hasattr(self.stats, "prune") else None, context=self.ctx, graphed=self.graphed, probe=probe,
                                   per_gaussian=pg, after_surgery=lambda d: setattr(self, "lbs_T", d.get("T", self.lbs_T)))

    def time_steps(self, n, barrier=None):
        torch = self.torch
        if barrier:
            barrier()
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        out = None
        for _ in range(n):
            out = self.step()
        torch.cuda.synchronize(self.dev)
        if barrier:
            barrier()
        return time.perf_counter() - t0, out


def cpu_quota():
    """CPUs this process may use (cgroup quota / affinity): moss_amd.host.cpu_quota."""
    from moss_amd.host import cpu_quota as q
    return q()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(argv))
    if args.dry_run_cpu:
        return dry_run_cpu(args)

    import torch
    from moss_amd import dist as mdist
    from moss_amd import scenes, _lib
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(_lib.EXT_PATH)):   # normally built by __graft_entry__.build(); hipcc is on the GPU box too
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            from moss_amd import build as hip_build
            hip_build.build()                               # (links under a temporary name and renames: never a half-written file)
        else:                                               # one rank builds, the others wait for the files
            for _ in range(2400):
                if os.path.exists(_lib.LIB_PATH) and os.path.exists(_lib.EXT_PATH):
                    break
                time.sleep(0.5)
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import render, camera_view
    from types import SimpleNamespace
    # Python's cyclic collector: a full (generation 2) pass over a process that has torch loaded walks ~10^6 module-level objects -- 45-55 ms
    # of HOST time (measured), at whatever allocation happens to trip it: inside a timed region of 20 x 0.19 ms, in the middle of a
    # densification event, or -- before moss_amd.graphs.capturing held it off -- inside a graph capture, where freeing a dropped
    # GraphedStep aborts the process.  What exists now (the modules) is moved to the permanent generation: later passes look only at what
    # this program creates (a few ms at worst).  INTEGRATION.md recommends the same two lines to a training script.
    # CPU threads: torch sizes its OpenMP team from the machine (128 of 256 hardware threads on the GPU boxes), the container's CPU QUOTA
    # is 16.  A parallel CPU op -- anything over >= 32k elements: building a scene, a harness, a scripted event -- then burns the quota of
    # the 100 ms scheduler period in a few ms and the kernel suspends the whole process, HIP runtime threads included, until the period
    # ends: 10-90 ms stalls that land in whatever is being timed a moment later (scripts/micro/cpu_parallel_stall.py: 40.3 / 60.3 / 50.3 ms
    # with 128 threads, none with 8).  The team is sized to the quota.
    from moss_amd.host import limit_cpu_threads
    limit_cpu_threads()                                      # (the ranks of one node share the quota: LOCAL_WORLD_SIZE)
    import gc
    import moss_amd.graphs, moss_amd.surgery, moss_amd.multiview, moss_amd.optim, moss_amd.loss, moss_amd.densify   # noqa: F401,E401
    gc.collect()
    gc.freeze()

    rank, world, local_rank = mdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: start N ranks (python bench.py --gpus N does it itself)"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if os.environ.get("MOSS_FORCE_DEVICE"):             # testing aid: several ranks on one GPU (with MOSS_DIST_BACKEND=gloo)
        local_rank = int(os.environ["MOSS_FORCE_DEVICE"])
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    rccl_ranks = None
    if world > 1:
        rccl_ranks = torch.distributed.get_world_size()
        assert rccl_ranks == args.gpus, f"the communicator has {rccl_ranks} ranks, --gpus {args.gpus}"

    # ---- workload (identical Gaussians on every rank, one camera per rank) ---------------------------------
    poses = scenes.look_at_ring(max(world, 8))
    maker = {"cfg2": scenes.config2, "cfg3": scenes.config3, "cfg5": scenes.config5,
             "moss7k": lambda seed=scenes.SEED: scenes.body_scene(6_890, 1024, 1024, 1080.0, init_like=True, seed=seed, name="moss7k"),
             "moss45k": lambda seed=scenes.SEED: scenes.body_scene(45_695, 1024, 1024, 1080.0, init_like=False, seed=seed, name="moss45k"),
             "moss45k_512": lambda seed=scenes.SEED: scenes.body_scene(45_695, 512, 512, 540.0, init_like=False, seed=seed, name="moss45k_512")}[args.config]
    scene = maker()
    if args.order == "morton":
        from moss_amd.densify import spatial_order
        perm = spatial_order(scene.means3D)
        for k_, v_ in list(vars(scene).items()):
            if torch.is_tensor(v_) and v_.dim() >= 1 and v_.shape[0] == scene.P:
                setattr(scene, k_, v_[perm].contiguous())
    if world > 1:
        R_, t_ = poses[rank % len(poses)]
        c0 = scene.camera
        scene.camera = scenes.make_camera(c0.W, c0.H, float(c0.K[0, 0]), float(c0.K[1, 1]), float(c0.K[0, 2]), float(c0.K[1, 2]), R_, t_)
    cam = camera_view(scene.camera, dev)
    H, W = scene.camera.H, scene.camera.W

    def lbs_transforms():
        gT = torch.Generator().manual_seed(1234)
        return (torch.eye(3) + 0.05 * torch.randn(scene.means3D.shape[0], 3, 3, generator=gT)).to(dev)

    bg = torch.zeros(3, device=dev)
    if args.target == "smooth":
        gt = scenes.synthetic_target(H, W).to(dev)
        gt_mask = (gt.mean(0, keepdim=True) > 0.5).float()
    else:
        gt_scene = maker(seed=scenes.SEED + 7)
        gt_pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
        with torch.no_grad():
            gt_out = render(cam, GaussianSet(gt_scene, sh_degree=3, device=dev), gt_pipe, bg)
        gt = gt_out["render"].detach().clamp(0, 1).contiguous()
        gt_mask = (gt_out["render_alpha"].detach() > 0.5).float().contiguous()
        del gt_out, gt_scene

    h = Harness(args, dev, rank, world, scene, cam, gt, gt_mask, bg, mode=args.mode, activations=args.activations,
                torch_activations=args.torch_activations, torch_adamw=args.torch_adamw, forward=args.forward, graph=args.graph,
                lbs_T=lbs_transforms() if args.mode in ("lbs", "lbs_python") else None, exchange=args.exchange,
                fused_optimizer=bool(args.fused_optimizer), fused_loss=("moss" if args.loss == "moss" else True))
    opt, bucket, pc = h.opt, h.bucket, h.pc
    pc.spatially_ordered = args.order == "morton"

    t_start = time.perf_counter()

    def note(msg):
        if os.environ.get("MOSS_BENCH_VERBOSE") and rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.2f}s] {msg}", file=sys.stderr, flush=True)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # ---- warmup (also finds the dominant kernel with all stages timed) -------------------------------------
    _lib.profile_enable(None)
    out = None
    n_warm = max(args.warmup, 1)
    for i in range(n_warm):
        if i == n_warm // 2 and i > 0:
            torch.cuda.synchronize(dev)
            _lib.profile_read()              # discard the first half: first launches include code-object loading
        out = h.step()
    torch.cuda.synchronize(dev)
    prof = _lib.profile_read()
    stage_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}
    dominant = max(stage_ms, key=stage_ms.get)
    _lib.profile_enable([dominant])          # two events per step around the dominant kernel only

    note(f"warmup done; stages {stage_ms}")
    use_graph = h.use_graph
    if use_graph:
        _lib.profile_enable([])              # hipEvent pairs cannot be read back from inside a captured graph
        try:
            out = None
            h.capture()
        except Exception as e:                                   # keep measuring: same kernels, launched one by one
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize(dev)
            use_graph = False
            h.step = h.eager_step
            _lib.profile_enable([dominant])

    # The step trains (AdamW moves the Gaussians), so the workload drifts from iteration to iteration.  The measurement passes after
    # the timed region restore this snapshot and REPLAY THE SAME K ITERATIONS (the kernels are deterministic), so the per-kernel
    # durations they report belong to exactly the frames the timed region rendered.
    snap = opt.snapshot() if hasattr(opt, "snapshot") else None
    h.t_phase = [0.0, 0.0, 0.0]
    h.n_exchange = 0
    note("entering timed region")
    # ---- timed region: EXACTLY K steps between barrier+sync pairs -------------------------------------------
    elapsed, out = h.time_steps(args.steps, barrier)
    note(f"timed region done: {elapsed:.3f}s")
    if args.forward == "async":
        # the last frame of the timed region (graph mode: the graph's own buffers) rendered within its binning capacity, i.e. it
        # really did the work; an overflowed frame would have produced a background image and is an error here
        h.ctx.check_status()
        assert h.ctx.last_needed > 0
    exchange_rep = h.exchange_report() if (world > 1 and h.n_exchange) else None
    dom_in_region = None
    replicas_identical = None
    if not use_graph:
        dom_in_region = _lib.profile_read()[dominant]          # hipEvent pairs recorded inside the timed region
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
        # frame-parallel replicas must hold bit-identical parameters after the same sequence of averaged gradients
        if hasattr(opt, "flat_params") and h.exchange_kind != "loss_only":     # (loss_only: independent models, nothing to compare)
            chk = opt.flat_params[:bucket.n_params].double().sum().reshape(1)
            lo, hi = chk.clone(), chk.clone()
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
            replicas_identical = bool((lo == hi).item())
    # a timed region of a few milliseconds says little: measure LONG_STEPS more steps as well and report both
    long_run = None
    if 1e3 * elapsed < MIN_TIMED_MS and args.steps < LONG_STEPS:
        t_long, _ = h.time_steps(LONG_STEPS, barrier)
        if world > 1:
            tt = torch.tensor([t_long], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            t_long = float(tt.item())
        long_run = {"steps": LONG_STEPS, "value": round(world * LONG_STEPS / t_long, 3), "ms_per_step": round(1e3 * t_long / LONG_STEPS, 4),
                    "why": f"the {args.steps}-step timed region lasted {1e3 * elapsed:.1f} ms (< {MIN_TIMED_MS:.0f} ms)"}

    # ---- N > 1: the OTHER exchange variant, measured the same way (every rank takes part), reported beside the headline ----------
    exchange_variants = None
    if world > 1 and not args.torch_adamw:
        exchange_variants = {h.exchange_kind: dict(exchange_rep or {}, value=round(world * args.steps / elapsed, 3),
                                                   ms_per_step=round(1e3 * elapsed / args.steps, 4), steps=args.steps,
                                                   replicas_identical=replicas_identical, headline=True)}
        for other in [k for k in ("allreduce", "sharded", "loss_only") if k != h.exchange_kind]:
            h2 = Harness(args, dev, rank, world, scene, cam, gt, gt_mask, bg, mode=args.mode, activations=args.activations,
                         torch_activations=args.torch_activations, torch_adamw=False, forward=args.forward, graph=args.graph,
                         lbs_T=lbs_transforms() if args.mode in ("lbs", "lbs_python") else None, exchange=other,
                         fused_optimizer=bool(args.fused_optimizer))
            for _ in range(max(args.warmup, 3)):
                h2.step()
            torch.cuda.synchronize(dev)
            if h2.use_graph:
                h2.capture()
            h2.t_phase = [0.0, 0.0, 0.0]; h2.n_exchange = 0
            n2 = max(args.steps, 50)
            t2, _ = h2.time_steps(n2, barrier)
            tt = torch.tensor([t2], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            t2 = float(tt.item())
            same = None
            if other != "loss_only":
                chk = h2.opt.flat_params[:h2.bucket.n_params].double().sum().reshape(1); lo, hi = chk.clone(), chk.clone()
                torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
                torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
                same = bool((lo == hi).item())
            exchange_variants[other] = dict(h2.exchange_report(), value=round(world * n2 / t2, 3), ms_per_step=round(1e3 * t2 / n2, 4), steps=n2,
                                            replicas_identical=same, headline=False)
            del h2
            torch.cuda.empty_cache()

    # ---- per-kernel device times: eager replay of the same K iterations with a hipEvent pair around every kernel of the op -------
    # (graph mode: events inside a replayed graph cannot be read back, so this replay is also where the dominant kernel's launch
    # duration comes from; it is not part of `value`)
    if snap is not None:
        opt.restore(snap)
    _lib.profile_enable(None)
    for _ in range(args.steps):
        out = h.eager_step()
    torch.cuda.synchronize(dev)
    prof = _lib.profile_read()
    _lib.profile_enable([])
    stage_ms = {k: round(v[0] / v[1], 5) if v[1] else 0.0 for k, v in prof.items()}
    dominant = max(stage_ms, key=stage_ms.get) if use_graph else dominant
    dom_ms, dom_n = dom_in_region if dom_in_region is not None else prof[dominant]
    dom_ms = dom_ms / max(dom_n, 1)

    note(f"stage pass done {stage_ms}")
    # ---- the RASTERIZER's own kernels: when the AdamW step rides inside the per-Gaussian backward, that kernel's time is not the
    # rasterizer's.  The same K iterations once more with the step taken by the flat kernel (bit-identical parameters: the frames are the
    # same), every kernel of the op timed: what `rasterizer_roofline` is computed from.
    raster_stage_ms = stage_ms
    was_fused = h.fused_opt
    if h.fused_opt and snap is not None:
        opt.restore(snap)
        opt.unfuse()
        h.fused_opt = False
        _lib.profile_enable(None)
        for _ in range(args.steps):
            out = h.eager_step()
        torch.cuda.synchronize(dev)
        prof2 = _lib.profile_read()
        _lib.profile_enable([])
        raster_stage_ms = {k: round(v[0] / v[1], 5) if v[1] else 0.0 for k, v in prof2.items()}
        note(f"rasterizer-only pass done {raster_stage_ms}")
    if rank != 0:
        return

    if args.forward == "async":
        h.ctx.check_status()                 # also brings num_rendered of the last replayed frame to the host
    # ---- measured problem statistics and the roofline -------------------------------------------------------
    radii = out["radii"]
    P = int(radii.numel()); Pv = int((radii > 0).sum().item())
    R = int(h.ctx.last_needed if args.forward == "async" else h.ctx.last_num_rendered)
    N = H * W
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    fwd_b, bwd_b = algorithmic_bytes(P, Pv, R, N, tiles, 16, args.mode in ("scale_rot", "lbs"), transforms=args.mode == "lbs",
                                     fused_adamw=was_fused)
    all_b = dict(fwd_b); all_b.update(bwd_b)
    # SURVEY 8(d) as written: the rasterizer's forward + backward, NO optimizer bytes
    rf_b, rb_b = algorithmic_bytes(P, Pv, R, N, tiles, 16, args.mode in ("scale_rot", "lbs"), transforms=args.mode == "lbs", fused_adamw=False)
    raster_b = dict(rf_b); raster_b.update(rb_b)
    if stage_ms.get("scan", 0.0) == 0.0:
        # asynchronous forward: the scan rides along with another kernel (no launch of its own) -- its bytes count there: with the
        # scatter kernel (rounds 2-4), or, when no scatter kernel runs either (round 5: the preprocess kernel writes the keys into
        # per-tile buckets, the scan is one block of the sort kernel), scan -> chunk_sort and scatter -> preprocess_fwd
        no_scatter = stage_ms.get("scatter", 0.0) == 0.0
        for b in (all_b, raster_b):
            if no_scatter:
                b["chunk_sort"] += b.pop("scan")
                b["preprocess_fwd"] += b.pop("scatter")
            else:
                b["scatter"] += b.pop("scan")
        drop = ("scan", "scatter") if no_scatter else ("scan",)
        for k in drop:
            stage_ms.pop(k, None)
        raster_stage_ms = {k: v for k, v in raster_stage_ms.items() if k not in drop}
    dom_bytes = all_b[dominant]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    total_bytes = sum(all_b.values())
    raster_bytes = sum(raster_b.values())
    iters_per_s = world * args.steps / elapsed
    raster_ms = sum(stage_ms.values())
    raster_only_ms = sum(raster_stage_ms.values())
    raster_gbs = raster_bytes / (raster_only_ms * 1e-3) / 1e9 if raster_only_ms > 0 else 0.0
    headline = args.config == "cfg3" and args.mode == "lbs"
    pmc = _pmc_traffic() if headline else {}
    # every stage against the roofline, not only the dominant one (durations: kernel-attached events of the eager replay)
    stages = {}
    for k, ms in stage_ms.items():
        gbs = all_b[k] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        stages[k] = {"ms": ms, "algorithmic_bytes": int(all_b[k]), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 5)}
        stages[k].update(_traffic_fields(pmc.get(k)))
        stages[k]["bound"] = _bound_of(gbs / HBM_PEAK_GBS, pmc.get(k))
        if k == "merge_gather":
            # SURVEY 8(d) prices the sort at keys + ids; what this kernel must ALSO move is the per-instance record: a 64-byte gather of the
            # Gaussian's geometry record and the 48-byte record + 2-byte block mask it emits -- counted here so that `traffic` has something
            # to be compared with (rounds 1-5: "5.7-8x algorithmic", against the keys alone)
            full = all_b[k] + R * (64 + 48 + 2)
            stages[k]["algorithmic_bytes_incl_record_gather"] = int(full)
            stages[k]["frac_incl_record_gather"] = round(full / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else 0.0
        if k == "blend_bwd" and stages[k].get("traffic_writes"):
            # SURVEY 8(d) prices this kernel's output at the reference's: 36 B per visible Gaussian, ADDED atomically (backward.cu:566-580).
            # This design adds nothing atomically (bitwise reproducible gradients): the kernel WRITES one 48-byte record per (entry, 4x4
            # block) pair that blended, and the per-Gaussian backward sums a Gaussian's records in a fixed order.  Its writes are those
            # records -- `traffic` above the algorithmic bytes here is that choice, not a re-read: the READS are below the algorithmic ones.
            stages[k]["gradient_records_written"] = int(stages[k]["traffic_writes"] // 48)
            stages[k]["gradient_records_per_instance"] = round(stages[k]["traffic_writes"] / 48.0 / max(R, 1), 2)
            stages[k]["algorithmic_bytes_with_the_records"] = int(44 * R + 28 * N + stages[k]["traffic_writes"])
            stages[k]["traffic_note"] = ("writes = one 48 B gradient record per (entry, block) pair that blended, where the reference adds 36 B per visible "
                                         "Gaussian atomically; reads (traffic_reads) are below the algorithmic 44 R + 28 N")

    result = {
        "metric": "train iters/sec (fwd+bwd, 512x512, ~100k Gaussians)" if args.config == "cfg3" else f"train iters/sec ({args.config})",
        "value": round(iters_per_s, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: {P} Gaussians on a synthetic capsule body, {W}x{H}, SH degree 3, "
                               f"step = render + L1 + 0.2(1-SSIM) + 0.5 maskL2 + backward + AdamW; one view per GPU per step"
                               + ("; MOSS's data flow (scales + rotations + a per-Gaussian LBS transform per frame, gaussian_renderer/__init__.py:88-93) "
                                  "with the covariance product, the getters, the loss and AdamW as this library's kernels and the step replayed as one "
                                  "hipGraph: what a maintainer gets by applying patches/*.diff; `value_dropin` in this line = the same workload "
                                  "through the UNCHANGED MOSS call pattern (only the three packages swapped)" if headline else "")
                   if args.config == "cfg3" else args.config,
                   "target": args.target, "input_mode": args.mode, "index_order": args.order,
                   "activations": "torch" if args.torch_activations else ("in_op" if h.pipe.raw_parameters_in_op else "fused"), "P": P, "visible": Pv, "num_rendered": R, "pixels": N,
                   "parallelism": ("single GPU" if world == 1 else
                                   (f"`value` = loss_only: frames / subjects sharded over {world} GPUs, one model per GPU, RCCL all-reduce of the "
                                    f"16-byte loss block (configs[3] as written)" if h.exchange_kind == "loss_only" else
                                    f"`value` = {h.exchange_kind}: dp{world}, ONE model, frames sharded over {world} GPUs, replicas kept identical by "
                                    + ("one RCCL all-reduce (mean) of the flat gradient bucket + the full AdamW per rank" if h.exchange_kind == "allreduce"
                                       else "reduce-scatter + AdamW on the rank's shard + all-gather") + " (SURVEY 8e)")
                                   + "; value_allreduce / value_sharded / value_loss_only in this line: the same step under each of the three exchanges"),
                   "forward": args.forward, "launch": h.graph_note if use_graph else "eager launches",
                   # (was_fused: what the TIMED region ran -- the rasterizer-only pass above takes the step out of the backward afterwards)
                   "optimizer": ("AdamW step taken by the per-Gaussian backward kernel (moss_raster_backward_raw_adamw); bit-identical to the flat "
                                 "kernel, reported beside as callers.unfused_optimizer") if was_fused else "flat AdamW kernel over the gradient bucket",
                   "loss": ("full frame: L1 + 0.2 (1 - SSIM) + 0.5 mask L2 (moss_photometric_loss)" if args.loss == "full" else
                            "MOSS's own expression: L1 and mask L2 over the view's bound_mask, SSIM on its bounding rectangle (moss_photometric_loss_roi)"),
                   "glue": "compiled PyTorch-ROCm extension moss_amd/lib/_moss_C.so over the C ABI of libmoss_raster.so",
                   **({"debug_bits": int(args.debug_bits)} if args.debug_bits else {})},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5),
                     # the same kernel by SURVEY 8(d)'s bytes ALONE (no optimizer traffic), timed as the plain instantiation in the
                     # rasterizer-only pass: what `frac` would be if the AdamW step were not riding in it
                     **({"frac_8d_bytes_only": round(raster_b[dominant] / (raster_stage_ms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "avg_launch_ms_8d_only": raster_stage_ms[dominant], "algorithmic_bytes_8d_only": int(raster_b[dominant])}
                        if (dominant in raster_b and raster_stage_ms.get(dominant, 0) > 0 and was_fused) else {}),
                     "bound_by_the_counters": _bound_of(achieved / HBM_PEAK_GBS, pmc.get(dominant)),
                     # the best a plain streaming kernel reaches on this part (profiles/r01_hbm_bandwidth.json: triad over 1 GiB buffers)
                     "peak_achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 5),
                     # PMC bytes were collected on the headline workload with exactly these kernel sources (profiles/pmc_latest.json is
                     # stamped with the sha256 of moss_amd/csrc/): null for any other workload or any other source state
                     **_traffic_fields(pmc.get(dominant)), "traffic_source": PMC_SOURCE,
                     "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 5),
                     "timing": ("hipEvents attached to the kernel (hipExtLaunchKernelGGL start/stop) on its launch stream, over an eager "
                                "replay of the SAME K iterations (parameters and optimizer state restored to the start of the "
                                "graph-replay timed region)") if use_graph else
                               "hipEvents attached to the kernel (hipExtLaunchKernelGGL start/stop) on its launch stream, inside the timed region"},
        # north_star: "rasterizer forward+backward ... as % of HBM roofline".  bytes = SURVEY 8(d)'s formula at the measured P, Pv, R, N
        # with NO optimizer bytes; time = the SUM of the op's seven kernels (kernel-attached events, eager replay of the timed region's
        # frames, the AdamW step taken by the flat kernel so that the per-Gaussian backward is the rasterizer's own kernel)
        "rasterizer_roofline": {"bound": "hbm", "algorithmic_bytes": int(raster_bytes), "kernels_ms": round(raster_only_ms, 5),
                                "achieved": round(raster_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(raster_gbs / HBM_PEAK_GBS, 5),
                                "floor_ms_at_peak": round(raster_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, 5),
                                "frac_of_achievable": round(raster_gbs / HBM_ACHIEVABLE_GBS, 5),
                                "stages_ms": raster_stage_ms,
                                "what": "SURVEY 8(d) bytes of forward + backward (no optimizer, no loss) / summed kernel time of the op's "
                                        "kernels; the AdamW step outside the per-Gaussian backward for this measurement"},
        "stages_ms": stage_ms,
        "stages": stages,
        "rasterizer_ms_per_step": round(raster_ms, 4),
        "step_algorithmic_bytes": int(raster_bytes),
        # ONE definition, the one of rounds 1-3 (comparable across rounds): SURVEY 8(d)'s rasterizer bytes (no optimizer, no loss) moved
        # per WHOLE step time (loss kernels, optimizer and launch gaps included in the time) / 8 TB/s.  Round 4 counted the fused
        # optimizer's 118 MB in the numerator: that figure is `step_hbm_frac_incl_optimizer`.
        "step_hbm_frac": round(raster_bytes * (iters_per_s / world) / (HBM_PEAK_GBS * 1e9), 5),
        "step_hbm_frac_incl_optimizer": round(total_bytes * (iters_per_s / world) / (HBM_PEAK_GBS * 1e9), 5),
        "step_algorithmic_bytes_incl_optimizer": int(total_bytes),
    }
    if long_run is not None:
        result["long_run"] = long_run
    if h.graphed is not None:
        result["graph_recaptures"], result["dropped_frames"] = h.graphed.recaptures, h.graphed.dropped_frames
    if world > 1:
        result["rccl_ranks"] = rccl_ranks
        result["backend"] = torch.distributed.get_backend()
        result["replicas_identical"] = replicas_identical
        result["exchange"] = h.exchange_kind
        if exchange_rep is not None:
            result.update({k: v for k, v in exchange_rep.items() if k.endswith("_ms")})
            result["allreduce_bytes"] = exchange_rep["bytes_reduced"]
        result["exchange_variants"] = exchange_variants
        for kind, rec in (exchange_variants or {}).items():      # every form of the step at top level, whichever one `value` is
            result[f"value_{kind}"] = rec["value"]
            result[f"ms_per_step_{kind}"] = rec["ms_per_step"]
            result[f"replicas_identical_{kind}"] = rec["replicas_identical"]      # (loss_only: null -- the models are different by design)
    if world == 1:
        result["densify_side_ms"] = densify_side(pc, out)
    if world == 1 and not args.no_callers and headline:
        del h, opt, bucket, pc
        torch.cuda.empty_cache()
        result["callers"] = caller_variants(args, dev, scene, cam, gt, gt_mask, bg, lbs_transforms())
        result["value_dropin"] = result["callers"].get("dropin_unchanged", {}).get("value")
        result["value_patched_moss"] = result["callers"].get("patched_moss_pattern", {}).get("value")
        result["value_patched_moss_one_call_loss"] = result["callers"].get("patched_moss_pattern_one_call_loss", {}).get("value")
        if "spatial_order" in result["callers"]:
            result["value_spatial_order"] = result["callers"]["spatial_order"].get("value")
        result["value_no_transforms"] = result["callers"].get("no_transforms", {}).get("value")
        if "unfused_optimizer" in result["callers"]:
            result["value_unfused_optimizer"] = result["callers"]["unfused_optimizer"].get("value")
        result["value_precomp"] = result["callers"].get("precomp_graph", {}).get("value")
        if "small_P_cfg2" in result["callers"]:
            result["value_small_P_cfg2"] = result["callers"]["small_P_cfg2"].get("value")
        if "densify_schedule" in result["callers"]:
            result["value_densify_schedule"] = result["callers"]["densify_schedule"].get("value")
        for b_ in (2, 4):
            if f"multi_view_b{b_}" in result["callers"]:
                result[f"value_views_per_s_b{b_}"] = result["callers"][f"multi_view_b{b_}"].get("value")
        for d_ in (0, 1, 2):
            if f"sh_degree_{d_}" in result["callers"]:
                result[f"value_sh_degree_{d_}"] = result["callers"][f"sh_degree_{d_}"].get("value")
    if world == 1 and not args.no_callers and headline and (not args.callers_only or "eval" in args.callers_only.split(",")):
        # the evaluation path (render_ZJU.py:56-72): forward-only renders of the same Gaussians, on configs[2] and configs[4]
        torch.cuda.empty_cache()
        result["eval"] = {"configs[2]": eval_block(dev, scene, cam, bg, lbs_transforms())}
        try:
            sc5 = scenes.config5()
            gT5 = torch.Generator().manual_seed(1234)
            T5 = (torch.eye(3) + 0.05 * torch.randn(sc5.means3D.shape[0], 3, 3, generator=gT5)).to(dev)
            result["eval"]["configs[4]"] = eval_block(dev, sc5, camera_view(sc5.camera, dev), bg, T5)
            del sc5, T5
        except Exception as e:                               # a side measurement must not take the headline down with it
            result["eval"]["configs[4]"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        result["value_eval_fps"] = result["eval"]["configs[2]"].get("fps_async_graph")
        torch.cuda.empty_cache()
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(scene, args, gt, gt_mask)
        result["cpu_baseline_autograd"] = cpu_baseline_autograd()
    print(json.dumps(result))


def caller_variants(args, dev, scene, cam, gt, gt_mask, bg, lbs_T, steps=100, warmup=10):
    """The same workload the way other callers drive the op, reported BESIDE the headline (never as `value`):
      dropin_unchanged   MOSS's call pattern with nothing but the three packages swapped: compute_cov3D_python=True with per-Gaussian
                         transforms (gaussian_renderer/__init__.py:88-91, arguments/__init__.py:60), the five torch getters, the
                         reference's torch loss (utils/loss_utils.py), torch.optim.AdamW over the six groups, synchronous forward,
                         eager launches, visibility_filter + the densification statistics every step with MOSS's own torch
                         expressions (train_ZJU.py:101,171-174)
      dropin_fused_sides the same call pattern with the caller-side rows of SURVEY 8(f) switched to this repository's kernels: fused
                         L1+SSIM loss, DensifyStats, flat AdamW, fused activation kernels (still cov3D_precomp from Python, still
                         synchronous and eager)
      no_transforms      scales + rotations only (no LBS transform), otherwise as the headline: rounds 1-3's headline configuration
      precomp_graph      cov3D_precomp built in Python from the fused-activated parameters (MOSS's shipped compute_cov3D_python=True,
                         without LBS transforms), asynchronous forward, one hipGraph per step: `value_precomp`
      spatial_order      the HEADLINE configuration on the same Gaussians re-indexed along a Morton curve (densify.spatial_order, what
                         a caller would do whenever it rebuilds its tensors: densify / prune): the synthetic scene's own index order is
                         uncorrelated with position -- the least favourable case for the per-tile counters and the gathers by
                         Gaussian index; MOSS's order (SMPL mesh order + appended clones) lies between the two"""
    import copy
    import torch
    res = {}
    specs = {
        "dropin_unchanged": dict(mode="lbs_python", activations="fused", torch_activations=True, torch_adamw=True, forward="sync", graph=0,
                                 fused_loss=False, caller_side="torch"),
        "dropin_fused_sides": dict(mode="lbs_python", activations="fused", torch_activations=False, torch_adamw=False, forward="sync", graph=0,
                                   fused_loss=True, caller_side="fused"),
        # exactly what patches/gaussian_renderer.diff + patches/train_ZJU.diff turn MOSS's call pattern into: transforms in the op,
        # asynchronous forward, the statistics kernel, ssim() from the HIP kernels inside MOSS's own torch loss expression, the
        # opacity / scaling / rotation getters inside the op, the optimizer class swapped for moss_amd.optim.AdamW (same groups,
        # state keys and per-group lr: MOSS's densification surgery keeps working) -- MOSS's parameter tensors and the rest of its
        # torch loss kept, eager launches
        "patched_moss_pattern": dict(mode="lbs", activations="fused", torch_activations=True, torch_adamw="moss_amd", forward="async", graph=0,
                                     fused_loss="ssim", caller_side="fused", raw_in_op=True),
        # the same with train_ZJU.py:108-119 -- MOSS's bound_mask selection, crop and three loss terms: ~40 torch launches forward and
        # backward -- replaced by the ONE call moss_amd.loss.training_loss_moss_fused (C ABI moss_photometric_loss_roi), INTEGRATION section 3
        "patched_moss_pattern_one_call_loss": dict(mode="lbs", activations="fused", torch_activations=True, torch_adamw="moss_amd", forward="async",
                                                   graph=0, fused_loss="moss", caller_side="fused", raw_in_op=True),
        # the op without per-Gaussian transforms (the reference's compute_cov3D_python=False path): rounds 1-3's headline
        "no_transforms": dict(mode="scale_rot", activations="in_op", torch_activations=False, torch_adamw=False, forward="async", graph=1),
        # MOSS's shipped input mode (compute_cov3D_python=True, arguments/__init__.py:60: the covariance built by torch ops from the
        # activated scales / rotations and handed over as cov3D_precomp) through the same graph path as the headline
        "precomp_graph": dict(mode="precomp", activations="fused", torch_activations=False, torch_adamw=False, forward="async", graph=1),
    }
    if args.mode in ("scale_rot", "lbs") and args.forward == "async" and args.graph and not args.torch_adamw and args.fused_optimizer:
        # the headline configuration with the optimizer as a kernel of its own (rounds 1-4's step): gradients into the bucket, flat AdamW
        specs["unfused_optimizer"] = dict(mode=args.mode, activations=args.activations, torch_activations=args.torch_activations,
                                          torch_adamw=False, forward="async", graph=1, fused_optimizer=False)
    if args.order == "as_generated" and args.mode in ("scale_rot", "lbs") and args.forward == "async" and args.graph:
        # (the pair is measured the same way -- same harness, same number of replays -- so that their RATIO is the effect of the order)
        same = dict(mode=args.mode, activations=args.activations, torch_activations=args.torch_activations,
                    torch_adamw=args.torch_adamw, forward="async", graph=1, fused_optimizer=bool(args.fused_optimizer))
        specs["as_generated_order"] = dict(same)
        specs["spatial_order"] = dict(same)
    if args.mode == "lbs" and args.forward == "async" and args.graph and not args.torch_adamw:
        # the regime MOSS actually trains in (6 890 SMPL-vertex Gaussians at the start, scene/dataset_readers.py:720; at most 45 695,
        # scene/gaussian_model.py:496): BASELINE configs[1] through the headline's harness
        specs["small_P_cfg2"] = dict(mode="lbs", activations=args.activations, torch_activations=args.torch_activations,
                                     torch_adamw=False, forward="async", graph=1, fused_optimizer=bool(args.fused_optimizer))
    if args.mode == "lbs" and args.forward == "async" and args.graph and not args.torch_adamw:
        # the headline at the SH degrees MOSS actually trains at: 0 / 1 / 2 for iterations 1-2999, 3 for the last one (train_ZJU.py:85-86,
        # scene/gaussian_model.py:171-173).  Coefficients above the active degree start as zeros (:179-181) and are never touched
        for d_ in (0, 1, 2):
            specs[f"sh_degree_{d_}"] = dict(mode="lbs", activations=args.activations, torch_activations=args.torch_activations,
                                            torch_adamw=False, forward="async", graph=1, fused_optimizer=bool(args.fused_optimizer), sh_degree=d_)
        # (degree 0 with the optimizer as a kernel of its own -- the form every N > 1 gradient exchange runs: what the degree-aware FLAT
        # update and the active-only dL_dsh are worth there; compare with unfused_optimizer, the same at degree 3)
        specs["sh_degree_0_unfused_optimizer"] = dict(specs["sh_degree_0"], fused_optimizer=False)
    if args.mode == "lbs" and args.forward == "async" and args.graph and not args.torch_adamw:
        # the headline's step with MOSS's OWN loss expression for the three rasterizer-facing terms (train_ZJU.py:108-119: L1 and mask L2
        # over the view's bound_mask, SSIM on its bounding rectangle) instead of the full-frame form the headline carries since round 1
        specs["moss_loss_expression"] = dict(mode="lbs", activations=args.activations, torch_activations=args.torch_activations,
                                             torch_adamw=False, forward="async", graph=1, fused_optimizer=bool(args.fused_optimizer), fused_loss="moss")
    if args.mode == "lbs" and args.forward == "async" and args.graph and not args.torch_adamw and args.fused_optimizer:
        # the headline's step THROUGH MOSS's densification schedule (train_ZJU.py:171-186: an event every 100 iterations): 400 steps with a
        # scripted clone / split / prune event after every 100th (an opacity reset with the second), the optimizer's rows, the bucket, the
        # capacity and the captured graph rebuilt at each (moss_amd.surgery.densification_event) -- on the headline workload and on
        # configs[1], MOSS's own starting point
        same = dict(mode="lbs", activations=args.activations, torch_activations=args.torch_activations, torch_adamw=False,
                    forward="async", graph=1, fused_optimizer=True)
        specs["densify_schedule"] = dict(same)
        specs["densify_schedule_cfg2"] = dict(same)
    if args.mode == "lbs" and args.forward == "async" and args.graph and not args.torch_adamw:
        # B views per optimizer step on THIS GPU (moss_amd/multiview.py): the data-parallel step of SURVEY 8(e) -- B cameras, the B
        # gradient sets averaged in a fixed order, one AdamW step -- with the B single-view chains on B HIP streams inside one hipGraph.
        # Reported as VIEWS per second beside `value` (which stays one view per step), never as it.
        for b_ in (1, 2, 4):
            specs[f"multi_view_b{b_}"] = dict(views=b_)
    only = [x for x in getattr(args, "callers_only", "").split(",") if x]
    for name, kw in specs.items():
        if only and name not in only:
            continue
        if name.startswith("multi_view_b"):
            try:
                res[name] = _multi_view(dev, scene, bg, lbs_T, kw["views"], steps)
            except Exception as e:
                res[name] = {"value": None, "error": f"{type(e).__name__}: {str(e)[:200]}"}
            torch.cuda.empty_cache()
            continue
        try:
            sc, T_ = scene, lbs_T
            cam_, gt_, mask_ = cam, gt, gt_mask
            if name in ("small_P_cfg2", "densify_schedule_cfg2"):
                from moss_amd import scenes as _scenes
                from moss_amd.gaussian_model import GaussianSet as _GS
                from moss_amd.gaussian_renderer import render as _render, camera_view as _camera_view
                from types import SimpleNamespace as _NS
                sc = _scenes.config2()
                cam_ = _camera_view(sc.camera, dev)
                gT = torch.Generator().manual_seed(1234)
                T_ = (torch.eye(3) + 0.05 * torch.randn(sc.means3D.shape[0], 3, 3, generator=gT)).to(dev)
                with torch.no_grad():
                    o_ = _render(cam_, _GS(_scenes.config2(seed=_scenes.SEED + 7), sh_degree=3, device=dev),
                                 _NS(convert_SHs_python=False, compute_cov3D_python=False, debug=False), bg)
                gt_ = o_["render"].detach().clamp(0, 1).contiguous()
                mask_ = (o_["render_alpha"].detach() > 0.5).float().contiguous()
                del o_
            if name == "spatial_order":
                from moss_amd.densify import spatial_order
                perm = spatial_order(scene.means3D)
                sc = copy.copy(scene)
                for k_, v_ in list(vars(sc).items()):
                    if torch.is_tensor(v_) and v_.dim() >= 1 and v_.shape[0] == scene.P:
                        setattr(sc, k_, v_[perm].contiguous())
                T_ = lbs_T[perm.to(lbs_T.device)].contiguous()
            h = Harness(args, dev, 0, 1, sc, cam_, gt_, mask_, bg, lbs_T=T_ if kw["mode"] in ("lbs", "lbs_python") else None, **kw)
            if name == "spatial_order":
                h.pc.spatially_ordered = True                # what GaussianSet.reorder_spatially() leaves behind: render() hints the op
            seg3 = name in ("as_generated_order", "spatial_order", "small_P_cfg2", "moss_loss_expression") or name.startswith("sh_degree_")
            n_steps = 3 * steps if seg3 else steps
            for _ in range(warmup):
                h.step()
            torch.cuda.synchronize(dev)
            if h.use_graph:
                h.capture()
                h.time_steps(warmup)                         # (replays: clocks and caches as in the timed region)
            if name.startswith("densify_schedule"):
                res[name] = _densify_schedule(h, dev, sc)
                del h
                torch.cuda.empty_cache()
                continue
            if seg3:
                # three segments, the median one reported (with the segments beside it): late in a process that has built and dropped
                # four harnesses a segment now and then takes a one-off host stall of tens of milliseconds, which is not what the
                # pair is there to compare
                segs = sorted(h.time_steps(steps)[0] for _ in range(3))
                dt, n_steps = segs[1], steps
            else:
                segs = None
                dt, _ = h.time_steps(n_steps)
            if kw["forward"] == "async":
                h.ctx.check_status()
            res[name] = {"value": round(n_steps / dt, 2), "unit": "iters/s", "ms_per_step": round(1e3 * dt / n_steps, 4), "steps": n_steps,
                         "launch": h.graph_note if h.use_graph else "eager launches", "forward": kw["forward"]}
            if name == "moss_loss_expression":
                x_, y_, w_, h_ = h.region.xywh
                res[name]["workload"] = (f"the headline's workload and step with MOSS's own loss expression: bound_mask of {int(h.region.rect[4])} pixels, "
                                         f"SSIM on its bounding rectangle {w_}x{h_} at ({x_},{y_}) of the {gt_.shape[-1]}x{gt_.shape[-2]} frame")
            if name == "small_P_cfg2" or name.startswith("sh_degree_") or name == "moss_loss_expression":
                res[name].setdefault("workload", "")
                if name != "moss_loss_expression":
                    res[name]["workload"] = (f"BASELINE configs[1]: {sc.means3D.shape[0]} Gaussians, 512x512, the headline's step (lbs, fused optimizer, one hipGraph)"
                                         if name == "small_P_cfg2" else
                                         f"the headline's workload and step at ACTIVE SH degree {kw['sh_degree']} (coefficients above it zero, as MOSS creates them)")
                # its kernels, one by one (eager replay with the library's kernel-attached events)
                _lib_ = sys.modules["moss_amd._lib"]
                _lib_.profile_enable(None)
                for _ in range(20):
                    h.eager_step()
                torch.cuda.synchronize(dev)
                res[name]["stages_us"] = {k: round(1e3 * v[0] / v[1], 1) for k, v in _lib_.profile_read().items() if v[1]}
                _lib_.profile_enable([])
            if segs is not None:
                res[name]["segments_ms_per_step"] = [round(1e3 * x / steps, 4) for x in segs]
            del h
        except Exception as e:                               # a side measurement must not take the headline down with it
            res[name] = {"value": None, "error": f"{type(e).__name__}: {str(e)[:200]}"}
        torch.cuda.empty_cache()
    return res


def _multi_view(dev, scene, bg, lbs_T, B, steps):
    """B views of the headline's workload per optimizer step (cameras on a ring around the body, one target per camera)."""
    import torch
    from types import SimpleNamespace
    from moss_amd import scenes
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import camera_view, render
    from moss_amd.multiview import MultiViewStep
    c0 = scene.camera
    cams = [camera_view(scenes.make_camera(c0.W, c0.H, float(c0.K[0, 0]), float(c0.K[1, 1]), float(c0.K[0, 2]), float(c0.K[1, 2]), R_, t_), dev)
            for R_, t_ in scenes.look_at_ring(8)[:B]]
    gt_scene = scenes.config3(seed=scenes.SEED + 7) if scene.name == "cfg3" else scene
    gts = []
    with torch.no_grad():
        gpc = GaussianSet(gt_scene, sh_degree=3, device=dev)
        for cam_ in cams:
            o = render(cam_, gpc, SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False), bg)
            gts.append((o["render"].detach().clamp(0, 1).contiguous(), (o["render_alpha"].detach() > 0.5).float().contiguous()))
        del gpc
    pc = GaussianSet(scene, sh_degree=3, device=dev, unified_features=True)
    mv = MultiViewStep(pc, B, cams, gts, bg, lbs_T)
    for _ in range(3):
        mv.eager_step()
    torch.cuda.synchronize(dev)
    mv.capture()
    for _ in range(10):
        mv.step()
    torch.cuda.synchronize(dev)
    segs = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            mv.step()
        torch.cuda.synchronize(dev)
        segs.append((time.perf_counter() - t0) / steps)
    mv.check()
    dt = sorted(segs)[1]
    return {"value": round(B / dt, 2), "unit": "views/s", "views_per_step": B, "optimizer_steps_per_s": round(1.0 / dt, 2), "ms_per_step": round(1e3 * dt, 4),
            "ms_per_view": round(1e3 * dt / B, 4), "steps": steps, "segments_ms_per_step": [round(1e3 * x, 4) for x in segs],
            "launch": f"one hipGraph replay per step: {B} single-view chains (render, loss, backward into the view's gradient buffer) on {B} HIP "
                      f"streams, joined in front of ONE flat AdamW kernel that forms the fixed-order average of the {B} buffers itself",
            "workload": f"{scene.name}: {B} cameras on a ring per optimizer step; gradients = the mean over the views (the N-GPU data-parallel step of SURVEY 8e on one device)"}


def _warm_surgery(dev, sizes=(64,)):
    """Throw-away densification events on models of the given sizes: torch loads the code object of every kernel it has not used yet at
    its first launch (`hipLaunchKernel` calls of 20-120 ms in a HIP API trace of this function's absence), and an event is the first user
    of boolean indexing, cat, repeat ... in this process -- at EVERY size class for which torch picks another kernel variant (the bench
    frame's third event crosses 131 072 rows).  That one-off cost per process is not what `event_ms` is there to report, so the sizes the
    schedule will pass through are visited here first."""
    import torch
    from moss_amd import dist as mdist, scenes
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.optim import FlatAdamW
    from moss_amd.densify import DensifyStats
    from moss_amd.surgery import densification_event
    for n in sizes:
        pc = GaussianSet(scenes.config1(P=int(n)), sh_degree=3, device=dev, unified_features=True)
        opt = FlatAdamW(pc.param_groups(), mdist.GradBucket(list(pc.parameters())), eps=1e-15, capturable=True)
        stats = DensifyStats(int(n), device=dev)
        t = {"xyz": pc._xyz.data, "f_dc": pc._features_dc.data, "f_rest": pc._features_rest.data, "opacity": pc._opacity.data,
             "scaling": pc._scaling.data, "rotation": pc._rotation.data}
        T = torch.eye(3, device=dev).repeat(int(n), 1, 1)
        for step in (1, 2):
            ev = scenes.scripted_densification(t, step, dev, reset_opacity=True)
            rep = densification_event(pc, opt, append=ev["append"], prune=ev["prune"], reset_opacity=True, stats=stats, per_gaussian={"T": T})
            T = rep["per_gaussian"]["T"]
            t = {"xyz": pc._xyz.data, "f_dc": pc._features_dc.data, "f_rest": pc._features_rest.data, "opacity": pc._opacity.data,
                 "scaling": pc._scaling.data, "rotation": pc._rotation.data}
        del pc, opt, stats, T, t
    torch.cuda.synchronize(dev)


def _densify_schedule(h, dev, sc, steps=400, every=100):
    """`steps` replays of the captured headline step with a scripted densification event after every `every`-th: wall time of the whole
    schedule (events included) and of the steps alone, the cost of each event, the row counts."""
    import torch
    P0 = int(h.pc._xyz.shape[0])
    _warm_surgery(dev, sizes=(64, P0, int(1.2 * P0), int(1.45 * P0)))
    from moss_amd.surgery import reserve_workspace
    reserve_workspace(3072 * 2 * int(h.pc._xyz.shape[0]), dev)      # (the set grows by <= 50 % over the schedule; see the function)
    h.graphed.reserve_pool(max(2 * 512 * int(h.ctx.capacity), 64 << 20))      # (the capacity grows with the set: scratch of twice today's)
    torch.cuda.synchronize(dev)
    t_steps, events, rows, phases, first, select = 0.0, [], [int(h.pc._xyz.shape[0])], [], [], []
    done = 0
    while done < steps:
        # (the first replay of a freshly captured graph pays its upload: timed on its own and reported as part of what an event costs)
        dt1, _ = h.time_steps(1)
        dt, _ = h.time_steps(every - 1)
        t_steps += dt + dt1
        if done:
            first.append(round(1e3 * dt1, 3))
        done += every
        torch.cuda.synchronize(dev)
        te = time.perf_counter()
        rep = h.densify_event(done, reset_opacity=(done == 2 * every))
        torch.cuda.synchronize(dev)
        # (what the bench's stand-in for MOSS's own selection logic -- seeded clone / split / prune choices, scenes.scripted_densification --
        # costs on top of the event proper: reported, not part of `value`)
        select.append(round(1e3 * (time.perf_counter() - te) - rep["event_ms"], 3))
        events.append(rep["event_ms"]); rows.append(rep["rows_after"])
        phases.append({k: rep[k] for k in ("surgery_ms", "probe_ms", "capture_ms", "device_mallocs", "device_mallocs_by_phase", "device_frees", "gc_full_collections")})
    torch.cuda.synchronize(dev)
    t_all = t_steps + 1e-3 * sum(events)                     # the steps and the events (each timed with the device synchronised on both sides)
    h.ctx.check_status()
    return {"value": round(steps / t_all, 2), "unit": "iters/s", "ms_per_step": round(1e3 * t_all / steps, 4), "steps": steps,
            "events": len(events), "event_ms": events, "event_ms_mean": round(sum(events) / len(events), 3),
            "event_ms_median": sorted(events)[len(events) // 2], "event_phases_ms": phases,
            "first_replay_after_event_ms": first, "scripted_selection_ms": select,
            "value_between_events": round(steps / t_steps, 2), "ms_per_step_between_events": round(1e3 * t_steps / steps, 4),
            "rows": rows, "graph_recaptures": h.graphed.recaptures, "dropped_frames": h.graphed.dropped_frames,
            "workload": f"{sc.name}: the headline's step (lbs, fused optimizer, one hipGraph) with a scripted clone / split / prune event every "
                        f"{every} steps (MOSS: train_ZJU.py:171-186); an event = optimizer rows + moments, bucket, statistics, LBS table "
                        f"re-laid-out, capacity re-learned by a forward-only probe, the step re-captured",
            "launch": h.graph_note}


def eval_block(dev, scene, cam, bg, lbs_T, n=200):
    """Novel-view rendering as MOSS's render_ZJU.py:56-72 does it -- `render(view, gaussians, pipeline, background)` under
    torch.no_grad(), here with the per-Gaussian LBS transforms and translation handed to the op (`transforms=`, `translation=`) -- at SH
    degree 3.  The glue tells the C ABI MOSS_FORWARD_ONLY: same images bit for bit, no backward state, 62 B of binning buffer per instance.
      fps_sync         the reference's behaviour: every render sizes its binning buffer from a host read-back (one synchronisation)
      fps_async_graph  capacity-bounded forward (no read-back), the render captured once in a hipGraph and replayed
      fps_training_forward_async_graph   the TRAINING forward (grad mode on, no backward run) the same way: what the flag saves"""
    import torch
    from types import SimpleNamespace
    from moss_amd import _lib
    from moss_amd import diff_gaussian_rasterization as dgr
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import render
    from moss_amd.graphs import GraphedStep
    pc = GaussianSet(scene, sh_degree=3, device=dev, unified_features=True)
    tl = torch.zeros(scene.means3D.shape[0], 3, device=dev)
    res = {"workload": f"{scene.name}: {scene.means3D.shape[0]} Gaussians, {scene.camera.W}x{scene.camera.H}, SH degree 3, "
                       "render(view, pc, pipe, bg, transforms=, translation=) under torch.no_grad()"}

    def make(async_):
        cx = dgr.RasterContext()
        cx.set_async(async_)
        pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=False,
                               transforms_in_op=True, pose_in_op=True, raw_parameters_in_op=True, raster_context=cx)
        return cx, pipe

    def timed(fn, k):
        fn(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / k

    # --- synchronous, eager (the reference's flow)
    cx, pipe = make(False)

    def eval_render():
        with torch.no_grad():
            return render(cam, pc, pipe, bg, transforms=lbs_T, translation=tl)["render"]
    for _ in range(5):
        eval_render()
    dt = timed(eval_render, n)
    R = int(cx.last_num_rendered)
    res.update(fps_sync=round(1.0 / dt, 1), ms_sync=round(1e3 * dt, 4), num_rendered=R,
               binning_bytes_per_instance=round(_lib.lib().moss_raster_binning_bytes_forward_only(R) / max(R, 1), 1),
               binning_bytes_per_instance_training=round(_lib.lib().moss_raster_binning_bytes(R) / max(R, 1), 1))
    # --- asynchronous + one hipGraph per render
    cx, pipe = make(True)
    for _ in range(3):
        eval_render()
    torch.cuda.synchronize(dev)
    g = GraphedStep(lambda: eval_render(), warmup=2, device=dev, context=cx)
    dt = timed(g, n)
    cx.check_status()
    res.update(fps_async_graph=round(1.0 / dt, 1), ms_async_graph=round(1e3 * dt, 4))
    # its kernels, one by one (eager, kernel-attached events)
    _lib.profile_enable(None)
    for _ in range(20):
        eval_render()
    torch.cuda.synchronize(dev)
    res["stages_us"] = {k: round(1e3 * v[0] / v[1], 1) for k, v in _lib.profile_read().items() if v[1]}
    _lib.profile_enable([])
    del g
    # --- B renders at a time: B forward-only chains on B HIP streams inside one hipGraph (moss_amd.multiview.MultiViewRender)
    try:
        from moss_amd import scenes as _sc
        from moss_amd.gaussian_renderer import camera_view as _cv
        from moss_amd.multiview import MultiViewRender
        c0 = scene.camera
        for B in (2, 4):
            cams = [_cv(_sc.make_camera(c0.W, c0.H, float(c0.K[0, 0]), float(c0.K[1, 1]), float(c0.K[0, 2]), float(c0.K[1, 2]), R_, t_), dev)
                    for R_, t_ in _sc.look_at_ring(8)[:B]]
            mr = MultiViewRender(pc, cams, bg, transforms=lbs_T, translation=tl)
            mr.capture()
            dt = timed(mr, max(n // B, 20))
            mr.check()
            res[f"fps_async_graph_{B}_streams"] = round(B / dt, 1)
            del mr
    except Exception as e:
        res["fps_async_graph_streams_error"] = f"{type(e).__name__}: {str(e)[:200]}"
    # --- the training forward the same way (grad mode on; the autograd node is built and dropped, no backward)
    cx, pipe = make(True)

    def train_forward():
        return render(cam, pc, pipe, bg, transforms=lbs_T, translation=tl)["render"].detach()
    for _ in range(3):
        train_forward()
    torch.cuda.synchronize(dev)
    g = GraphedStep(train_forward, warmup=2, device=dev, context=cx)
    dt = timed(g, n)
    res.update(fps_training_forward_async_graph=round(1.0 / dt, 1), ms_training_forward_async_graph=round(1e3 * dt, 4))
    _lib.profile_enable(None)
    for _ in range(20):
        train_forward()
    torch.cuda.synchronize(dev)
    res["stages_us_training_forward"] = {k: round(1e3 * v[0] / v[1], 1) for k, v in _lib.profile_read().items() if v[1]}
    _lib.profile_enable([])
    return res


def densify_side(pc, out):
    """the KL-guided densify test (k = 2 self query + kl_div, every 100 steps in MOSS) and the per-step densification
    statistics are reported BESIDE the step metric, never inside it."""
    import torch
    from moss_amd.densify import DensifyStats, cal_kl
    with torch.no_grad():
        xyz, rot, scl = pc.get_xyz.detach(), pc._rotation.detach(), pc.get_scaling.detach()
        radii = out["radii"]
        stats = DensifyStats(xyz.shape[0], device=xyz.device)
        grad = torch.randn(xyz.shape[0], 3, device=xyz.device)
        res = {}
        for name, fn in (("cal_kl_every_100_steps", lambda: cal_kl(xyz, rot, scl)), ("statistics_per_step", lambda: stats.add(radii, grad))):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn()
            b.record(); torch.cuda.synchronize()
            res[name] = round(a.elapsed_time(b) / 10, 4)
    return res


def csrc_sha256():
    """sha256 over the kernel sources (moss_amd/csrc/*, sorted by name): ties a committed counter summary to the code it measured."""
    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "moss_amd", "csrc")
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name), "rb") as f:
            hsh.update(name.encode()); hsh.update(f.read())
    return hsh.hexdigest()


def _pmc_traffic():
    """{stage or kernel: HBM bytes per launch} from the committed rocprofv3 --pmc summary (profiles/pmc_latest.json) IF it was
    measured on the kernel sources of this checkout (its `csrc_sha256` stamp matches), else {} (traffic: null)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            data = json.load(f)
        if data.get("csrc_sha256") != csrc_sha256():
            return {}
        return {k: v for k, v in data.items() if isinstance(v, dict)}
    except Exception:
        return {}


PMC_SOURCE = ("profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collected by the builder (scripts/gpu_profile_round.sh, "
              "separate counter-only passes over the eager form of this step), committed and stamped with the sha256 of moss_amd/csrc/; "
              "NOT measured in this run -- null when the stamp does not match the checkout")


def _traffic_fields(rec):
    """traffic (+ how to read it) of one stage from its PMC record: a number for kernels whose reads are wide coalesced streams (the
    guide's calibrated 2 x FETCH_SIZE + WRITE_SIZE), the UPPER end of [raw, 2 x raw] for gather kernels (interval beside it)."""
    if not rec or rec.get("hbm_bytes_per_launch") is None:
        return {"traffic": None}
    out = {"traffic": rec["hbm_bytes_per_launch"]}
    if rec.get("traffic_bound") == "upper" and rec.get("hbm_bytes_interval"):
        out["traffic_interval"] = rec["hbm_bytes_interval"]
        out["traffic_is"] = "upper bound (gather reads: FETCH_SIZE doubling uncalibrated)"
    if rec.get("hbm_read_bytes") is not None:
        out["traffic_reads"], out["traffic_writes"] = rec["hbm_read_bytes"], rec["hbm_write_bytes"]
    for k in ("valu_issue_frac", "wave_cycles_waiting_frac", "insts_valu_per_launch", "insts_salu_per_launch", "kernel_cycles"):
        if k in rec:
            out[k] = rec[k]
    return out


HBM_BOUND_FRAC, VALU_BOUND_FRAC = 0.35, 0.25


def _bound_of(frac_hbm, rec):
    """What a stage is bound by, from its numbers: `hbm` -- it moves its algorithmic bytes at >= 35 % of the nominal 8 TB/s (>= half of
    what a streaming kernel reaches on this part, 5.9 TB/s); `valu_issue` -- not that, and >= 25 % of ALL the device's vector issue
    slots held an instruction (committed counters: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles) -- an AVERAGE over the
    SIMDs: in the blend kernels the SIMDs that drew the long items are full while others idle); `latency` --
    neither: dependent round trips and tails at one or two waves per SIMD.  null without counters of this source state."""
    if frac_hbm >= HBM_BOUND_FRAC:
        return "hbm"
    v = (rec or {}).get("valu_issue_frac")
    if v is None:
        return None
    return "valu_issue" if v >= VALU_BOUND_FRAC else "latency"


def cpu_baseline(scene, args, gt, gt_mask):
    """The CPU oracle (single-threaded C port of the reference algorithm) + the same loss in torch on ONE thread, timed on
    this box's host cores on a bounded sample of the same workload.  A reported baseline, never the product path."""
    import torch
    from moss_amd.loss import training_loss
    from tests import helpers as hp
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    d = hp.inputs_of(scene, "precomp" if args.mode == "lbs_python" else ("scale_rot" if args.mode == "lbs" else args.mode))
    gt_c, mask_c = gt.cpu(), gt_mask.cpu()
    n = max(1, args.cpu_iters)
    t0 = time.perf_counter()
    for _ in range(n):
        fw = hp.oracle_forward(d)
        img = torch.from_numpy(fw.color).requires_grad_(True)
        alpha = torch.from_numpy(fw.alpha).requires_grad_(True)
        loss = training_loss(img, alpha, gt_c, mask_c)
        loss.backward()
        hp.oracle_backward(d, fw, img.grad, torch.zeros(1, d.H, d.W), alpha.grad)
    dt = time.perf_counter() - t0
    torch.set_num_threads(nthreads)
    return {"value": round(n / dt, 4), "unit": "iters/s", "cores": 1, "kind": "port",
            "sample": f"{n} full iterations of the same workload (C oracle fwd+bwd + torch loss, 1 thread) in {dt:.1f} s",
            "host_cores_available": os.cpu_count()}


def cpu_baseline_autograd(budget_s=8.0):
    """The baseline north_star names: the naive PyTorch-autograd per-pixel rasterizer (oracle/autograd_rasterizer.py, float64) on the
    host cores, fwd+bwd, on BASELINE configs[0] (256 Gaussians, 128x128 -- the case it is meant for; it takes its tile lists from
    the C oracle's binning, which is inside the timed iteration).  A reported, non-target baseline."""
    import torch
    from moss_amd import scenes
    from oracle import autograd_rasterizer as ag
    from tests import helpers as hp
    d = hp.inputs_of(scenes.config1(), "scale_rot"); c = d.cam
    f64 = lambda t: t.double()

    def it():
        fw = hp.oracle_forward(d)
        leaf = lambda t: t.double().clone().requires_grad_(True)
        means, opa, shs, scl, rot = leaf(d.means3D), leaf(d.opacities), leaf(d.shs), leaf(d.scales), leaf(d.rotations)
        col, dep, alp = ag.render(fw, means, opa, f64(c.viewmatrix), f64(c.projmatrix), f64(c.campos), c.tanfovx, c.tanfovy, f64(d.bg),
                                  d.degree, shs=shs, scales=scl, rotations=rot, cov3D_precomp=None)
        (col.sum() + alp.sum()).backward()
    it()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 200:
        it(); n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "iters/s", "cores": torch.get_num_threads(), "host_cores_available": os.cpu_count(), "cpu_quota": cpu_quota(), "kind": "port",
            "workload": "BASELINE configs[0]: 256 Gaussians, 128x128, fwd+bwd", "sample": f"{n} iterations in {dt:.1f} s",
            "what": "naive PyTorch-autograd rasterizer (oracle/autograd_rasterizer.py, float64, torch intra-op threads as stated)"}


def dry_run_cpu(args):
    """TEST HOOK: the N-rank control flow of this file without any GPU (see --dry-run-cpu)."""
    import torch
    import torch.distributed as dist
    from moss_amd import dist as mdist
    os.environ.setdefault("MOSS_DIST_BACKEND", "gloo")
    rank, world, _ = mdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    ranks = dist.get_world_size() if world > 1 else 1
    variants = {}

    class _SgdShard:
        """stand-in optimizer of the sharded path: p -= 0.1 g on this rank's shard (what FlatAdamW(shard=...) is to ShardedStep)"""
        def __init__(self, bucket, rank):
            per = bucket.shard_len
            self.flat_params = torch.zeros(bucket.flat.numel())
            self.grad_shard = torch.zeros(per)
            self.first = min(rank * per, bucket.n_params)
            self.count = min(self.first + per, bucket.n_params) - self.first

        def step(self):
            self.flat_params[self.first:self.first + self.count].add_(self.grad_shard[:self.count], alpha=-0.1)

    def run(kind):
        params = [torch.nn.Parameter(torch.zeros(1000, 3)), torch.nn.Parameter(torch.zeros(1000, 16, 3))]
        bucket = mdist.GradBucket(params, world=world if kind == "sharded" else 1)
        n = bucket.n_params
        loss_only = kind == "loss_only"                      # independent models: only the loss block is exchanged
        if kind == "sharded" and world > 1:
            opt = _SgdShard(bucket, rank)
            sharded = mdist.ShardedStep(bucket, opt, rank, world)
            flat_params = opt.flat_params
        else:
            sharded, flat_params = None, torch.zeros(n)
        t_ar = [0.0]

        def step():
            bucket.flat[:n].fill_(float(rank + 1))           # "the backward": rank-dependent gradients
            bucket.loss_terms.fill_(float(rank + 1))
            t0 = time.perf_counter()
            if sharded is not None:
                sharded.step()
            elif loss_only:
                bucket.all_reduce_loss_only(world)
            else:
                bucket.all_reduce_mean(None, world)
            t_ar[0] += time.perf_counter() - t0
            if sharded is None:
                flat_params[:n].add_(bucket.flat[:n], alpha=-0.1)      # "the optimizer"
        for _ in range(max(args.warmup, 1)):
            step()
        t_ar[0] = 0.0
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        identical = True
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            chk = flat_params[:n].double().sum().reshape(1); lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            identical = None if loss_only else bool((lo == hi).item())          # (loss_only: the ranks' models are their own)
            mean = (world + 1) / 2.0                                            # the mean of 1..world
            loss_seen = float((sharded.loss_terms if sharded is not None else bucket.loss_terms)[0])
            assert abs(loss_seen - mean) < 1e-6, (kind, loss_seen)
            want = -0.1 * (float(rank + 1) if loss_only else mean) * (max(args.warmup, 1) + args.steps)
            assert not loss_only or abs(float(bucket.flat[0]) - float(rank + 1)) < 1e-6        # its gradients never travelled
            assert abs(float(flat_params[0]) - want) < 1e-4 * abs(want) and abs(float(flat_params[n - 1]) - want) < 1e-4 * abs(want), kind
        return {"value": round(world * args.steps / elapsed, 3), "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                "exchange_ms": round(1e3 * t_ar[0] / args.steps, 4), "replicas_identical": identical,
                "checksum": float(flat_params[:n].double().sum())}

    variants[args.exchange] = run(args.exchange)
    if world > 1:
        for other in [k for k in ("allreduce", "sharded", "loss_only") if k != args.exchange]:
            variants[other] = run(other)
        assert variants["allreduce"]["checksum"] == variants["sharded"]["checksum"], "the two gradient-exchange paths left different parameters"
    head = variants[args.exchange]
    if rank == 0:
        per_kind = {}
        for kind, rec in variants.items():
            per_kind[f"value_{kind}"] = rec["value"]; per_kind[f"ms_per_step_{kind}"] = rec["ms_per_step"]
            per_kind[f"replicas_identical_{kind}"] = rec["replicas_identical"]
        print(json.dumps({**per_kind, "metric": "train iters/sec (fwd+bwd, 512x512, ~100k Gaussians)", "value": head["value"],
                          "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "DRY RUN: no GPU work (launcher / collective plumbing test)",
                          "config": {"workload": "dry run", "parallelism": f"`value` = {args.exchange}"}, "rccl_ranks": ranks, "backend": dist.get_backend() if world > 1 else None,
                          "replicas_identical": all(v["replicas_identical"] for v in variants.values() if v["replicas_identical"] is not None), "exchange": args.exchange,
                          "allreduce_ms": head["exchange_ms"], "adamw_ms": 0.0, "exchange_variants": variants}))


if __name__ == "__main__":
    main()

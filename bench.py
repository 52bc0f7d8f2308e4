#!/usr/bin/env python3
"""bench.py -- train iters/sec of the rasterizer hot path (BASELINE.json metric) on N MI355X of one node.

One STEP = one training iteration on one view of the `configs[2]` workload (100k Gaussians, 512x512, SH degree 3):
  render() -> HIP rasterizer forward -> L1 + 0.2*(1-SSIM) + 0.5*maskL2 -> backward (HIP rasterizer backward, down to the raw
  Gaussian parameters) -> [N>1: one RCCL all-reduce of the flat gradient bucket] -> AdamW step.
Inputs are resident in HBM before the timed region.  N>1 is frame-parallel (one camera per rank, replicas of the
Gaussians, weak scaling): value = N * steps / max-over-ranks time.

Launch: python bench.py [--gpus 1 --steps K --warmup W]      or, for N>1,
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 5924.0           # measured: scripts/hbm_bandwidth.py (triad, 1 GiB buffers), profiles/r01_hbm_bandwidth.json      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(P, Pv, R, N, tiles, K, scale_rot_mode):
    """bytes that MUST cross HBM per fwd+bwd step, from the measured P, Pv, R, N."""
    c = 28 if scale_rot_mode else 24
    fwd = {
        "preprocess_fwd": P * (12 + c + 4) + Pv * 12 * K + P * 8 + Pv * 43,
        "scan": 8 * P,
        "scatter": Pv * 24 + 12 * R,
        "tile_sort": 24 * R + 8 * R + 8 * tiles,
        "blend_fwd": 44 * R + 28 * N,
    }
    bwd = {
        "blend_bwd": 44 * R + 28 * N + 36 * Pv,
        "preprocess_bwd": 36 * Pv + Pv * (12 + 4 + 24 + 12 * K + 3) + Pv * (12 + 24 + 12 * K + 4) + (28 * Pv if scale_rot_mode else 0) + P * 12,
    }
    return fwd, bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--mode", default="scale_rot", choices=["precomp", "scale_rot", "lbs", "lbs_python"],
                    help="scale_rot = cov3D computed inside the rasterizer from scales+rotations (the reference's "
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=20,
                    help="iterations of the CPU oracle baseline (about 0.57 s each on one core: 20 = the 10-30 s sample the contract asks for)")
    args = ap.parse_args()

    import torch
    from moss_amd import dist as mdist
    from moss_amd import scenes, _lib
    if not os.path.exists(_lib.LIB_PATH):               # normally built by __graft_entry__.build(); hipcc is on the GPU box too
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            from moss_amd import build as hip_build
            hip_build.build()
        else:                                           # one rank builds, the others wait for the file
            for _ in range(1200):
                if os.path.exists(_lib.LIB_PATH):
                    break
                time.sleep(0.5)
            time.sleep(2.0)
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import render, camera_view
    from moss_amd.loss import training_loss_fused as training_loss, backward_from_loss     # HIP-fused L1 + SSIM + mask loss
    from types import SimpleNamespace

    rank, world, local_rank = mdist.init_from_env()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if os.environ.get("MOSS_FORCE_DEVICE"):             # testing aid: several ranks on one GPU (with MOSS_DIST_BACKEND=gloo)
        local_rank = int(os.environ["MOSS_FORCE_DEVICE"])
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    # ---- workload (identical Gaussians on every rank, one camera per rank) ---------------------------------
    poses = scenes.look_at_ring(max(world, 8))
    maker = {"cfg2": scenes.config2, "cfg3": scenes.config3, "cfg5": scenes.config5}[args.config]
    scene = maker()
    if world > 1:
        R_, t_ = poses[rank % len(poses)]
        c0 = scene.camera
        scene.camera = scenes.make_camera(c0.W, c0.H, float(c0.K[0, 0]), float(c0.K[1, 1]), float(c0.K[0, 2]), float(c0.K[1, 2]), R_, t_)
    cam = camera_view(scene.camera, dev)
    H, W = scene.camera.H, scene.camera.W
    # SH coefficients as one (P,16,3) parameter (no per-step concat) whenever the flat optimizer can give dc / rest their two rates
    unified = not args.torch_adamw and not args.torch_activations
    pc = GaussianSet(scene, sh_degree=3, device=dev, unified_features=unified)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=(args.mode in ("precomp", "lbs_python")), debug=False,
                           fused_activations=not args.torch_activations, transforms_in_op=(args.mode == "lbs"),
                           raw_parameters_in_op=(not args.torch_activations and args.activations == "in_op" and unified
                                                 and args.mode in ("scale_rot", "lbs")))
    lbs_T = None
    if args.mode in ("lbs", "lbs_python"):
        gT = torch.Generator().manual_seed(1234)
        lbs_T = (torch.eye(3) + 0.05 * torch.randn(scene.means3D.shape[0], 3, 3, generator=gT)).to(dev)
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
    bucket = mdist.GradBucket(list(pc.parameters()))
    pipe.grad_bucket = bucket
    if args.torch_adamw:
        opt = torch.optim.AdamW(pc.param_groups(), lr=0.0, eps=1e-15, fused=True)      # gaussian_model.py:226
    else:
        from moss_amd.optim import FlatAdamW
        opt = FlatAdamW(pc.param_groups(), bucket, eps=1e-15, capturable=True)         # same rule, one kernel over the bucket
    from moss_amd import diff_gaussian_rasterization as dgr
    use_graph = bool(args.graph) and args.forward == "async" and not args.torch_adamw
    dgr.set_async(args.forward == "async")
    if unified:
        sinks = {"sh": lambda: bucket.sink_for(pc._features)}             # dL_dsh is written straight into the gradient bucket
        if pipe.raw_parameters_in_op:                                     # ... and so are the raw-parameter gradients
            sinks.update(opacity=lambda: bucket.sink_for(pc._opacity), scales=lambda: bucket.sink_for(pc._scaling),
                         rotations=lambda: bucket.sink_for(pc._rotation))
            if lbs_T is None:                                             # (with a transform the means are not the parameter)
                sinks["means3D"] = lambda: bucket.sink_for(pc._xyz)
        dgr.set_grad_sink(**sinks)

    def compute():                      # everything of a step that is local to this rank
        if pipe.fused_activations:
            bucket.detach_grads()       # gradients are WRITTEN into the bucket by the activation backward kernel
        else:
            bucket.attach()             # zero the bucket; autograd accumulates into it
        out = render(cam, pc, pipe, bg, transforms=lbs_T)
        # the loss kernels write [loss, L1, SSIM, mask] into the bucket's tail: it travels with the gradients, no copy
        loss = training_loss(out["render"], out["render_alpha"], gt, gt_mask, terms_out=bucket.loss_terms)
        backward_from_loss(loss)
        if pipe.fused_activations:
            bucket.collect()
        if world == 1:
            opt.step()
        # detached: holding an output with a grad_fn would keep this step's autograd graph (and its AccumulateGrad nodes,
        # bound to the stream they were created on) alive into the next step / into graph capture
        return {"radii": out["radii"]}

    def eager_step():
        out = compute()
        if world > 1:
            bucket.all_reduce_mean(None, world)          # ONE RCCL all-reduce of the flat gradient bucket (+ loss slot)
            opt.step()
        return out

    step = eager_step

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
        out = step()
    torch.cuda.synchronize(dev)
    prof = _lib.profile_read()
    stage_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}
    dominant = max(stage_ms, key=stage_ms.get)
    _lib.profile_enable([dominant])          # two events per step around the dominant kernel only

    note(f"warmup done; stages {stage_ms}")
    graph_note = "eager launches"
    if use_graph:
        # The first (synchronous) forward above sized the binning buffer; nothing in compute() talks to the host any more, so
        # the whole per-rank step is captured once and replayed: ~50 launches become one hipGraphLaunch.
        _lib.profile_enable([])              # hipEvent pairs cannot be read back from inside a captured graph
        try:
            out = None
            from moss_amd.graphs import GraphedStep
            graphed = GraphedStep(compute, warmup=3, device=dev)

            replays = [0]

            def graph_step():
                graphed()
                replays[0] += 1
                if replays[0] % 512 == 0:        # long runs: the scene grows while it trains; re-capture before the baked-in
                    graphed.check()              # binning capacity overflows (one synchronisation per 512 steps)
                if world > 1:
                    bucket.all_reduce_mean(None, world)
                    opt.step()
                return graphed.outputs                       # (re-bound by a re-capture: never cache it)

            for _ in range(5):
                graph_step()
            torch.cuda.synchronize(dev)
            step = graph_step
            graph_note = "one hipGraph replay per step" + (" + eager RCCL all-reduce and AdamW" if world > 1 else "")
        except Exception as e:                                   # keep measuring: same kernels, launched one by one
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize(dev)
            use_graph = False
            _lib.profile_enable([dominant])

    # The step trains (AdamW moves the Gaussians), so the workload drifts from iteration to iteration.  The measurement passes after
    # the timed region restore this snapshot and REPLAY THE SAME K ITERATIONS (the kernels are deterministic), so the per-kernel
    # durations they report belong to exactly the frames the timed region rendered.
    snap = opt.snapshot() if hasattr(opt, "snapshot") else None
    note("entering timed region")
    # ---- timed region: EXACTLY K steps between barrier+sync pairs -------------------------------------------
    barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev); barrier()
    elapsed = time.perf_counter() - t0
    note(f"timed region done: {elapsed:.3f}s")
    if args.forward == "async":
        # the last frame of the timed region (graph mode: the graph's own buffers) rendered within its binning capacity, i.e. it
        # really did the work; an overflowed frame would have produced a background image and is an error here
        dgr.check_async_status()
        assert dgr._C.ASYNC.last_needed > 0
    dom_in_region = None
    replicas_identical = None
    if not use_graph:
        dom_in_region = _lib.profile_read()[dominant]          # hipEvent pairs recorded inside the timed region
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
        # frame-parallel replicas must hold bit-identical parameters after the same sequence of averaged gradients
        if hasattr(opt, "flat_params"):
            chk = opt.flat_params.double().sum().reshape(1)
            lo, hi = chk.clone(), chk.clone()
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
            replicas_identical = bool((lo == hi).item())

    # ---- per-kernel device times: eager replay of the same K iterations with a hipEvent pair around every kernel of the op -------
    # (graph mode: events inside a replayed graph cannot be read back, so this replay is also where the dominant kernel's launch
    # duration comes from; it is not part of `value`)
    if snap is not None:
        opt.restore(snap)
    _lib.profile_enable(None)
    for _ in range(args.steps):
        out = eager_step()
    torch.cuda.synchronize(dev)
    prof = _lib.profile_read()
    _lib.profile_enable([])
    stage_ms = {k: round(v[0] / v[1], 5) if v[1] else 0.0 for k, v in prof.items()}
    dominant = max(stage_ms, key=stage_ms.get) if use_graph else dominant
    dom_ms, dom_n = dom_in_region if dom_in_region is not None else prof[dominant]
    dom_ms = dom_ms / max(dom_n, 1)

    note(f"stage pass done {stage_ms}")
    if rank != 0:
        return

    if args.forward == "async":
        dgr.check_async_status()                 # also brings num_rendered of the last replayed frame to the host
    # ---- measured problem statistics and the roofline -------------------------------------------------------
    radii = out["radii"]
    P = int(radii.numel()); Pv = int((radii > 0).sum().item())
    from moss_amd.diff_gaussian_rasterization import _C
    R = int(_C.last_num_rendered)
    N = H * W
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    fwd_b, bwd_b = algorithmic_bytes(P, Pv, R, N, tiles, 16, args.mode in ("scale_rot", "lbs"))
    all_b = dict(fwd_b); all_b.update(bwd_b)
    dom_bytes = all_b[dominant]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    total_bytes = sum(all_b.values())
    iters_per_s = world * args.steps / elapsed
    raster_ms = sum(stage_ms.values())

    result = {
        "metric": "train iters/sec (fwd+bwd, 512x512, ~100k Gaussians)" if args.config == "cfg3" else f"train iters/sec ({args.config})",
        "value": round(iters_per_s, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: {P} Gaussians on a synthetic capsule body, {W}x{H}, SH degree 3, "
                               f"step = render + L1 + 0.2(1-SSIM) + 0.5 maskL2 + backward + AdamW; one view per GPU per step"
                   if args.config == "cfg3" else args.config,
                   "target": args.target, "input_mode": args.mode,
                   "activations": "torch" if args.torch_activations else ("in_op" if pipe.raw_parameters_in_op else "fused"), "P": P, "visible": Pv, "num_rendered": R, "pixels": N,
                   "parallelism": f"frame-parallel x{world}" if world > 1 else "single GPU",
                   "forward": args.forward, "launch": graph_note},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5),
                     # the best a plain streaming kernel reaches on this part (profiles/r01_hbm_bandwidth.json: triad over 1 GiB buffers)
                     "peak_achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 5),
                     # PMC bytes were collected on the headline workload (profiles/pmc_latest.json): null for any other
                     "traffic": _pmc_traffic(dominant) if (args.config == "cfg3" and args.mode == "scale_rot") else None,
                     "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 5),
                     "timing": ("hipEvents attached to the kernel (hipExtLaunchKernelGGL start/stop) on its launch stream, over an eager "
                                "replay of the SAME K iterations (parameters and optimizer state restored to the start of the "
                                "graph-replay timed region)") if use_graph else
                               "hipEvents attached to the kernel (hipExtLaunchKernelGGL start/stop) on its launch stream, inside the timed region"},
        "stages_ms": stage_ms,
        "rasterizer_ms_per_step": round(raster_ms, 4),
        "step_algorithmic_bytes": int(total_bytes),
        "step_hbm_frac": round(total_bytes * (iters_per_s / world) / (HBM_PEAK_GBS * 1e9), 5),
    }

    if replicas_identical is not None:
        result["replicas_identical"] = replicas_identical
    if world == 1:
        result["densify_side_ms"] = densify_side(pc, out)
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(scene, args, gt, gt_mask)
    print(json.dumps(result))


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


def _pmc_traffic(kernel):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary (profiles/pmc_latest.json), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(scene, args, gt, gt_mask):
    """The CPU oracle (single-threaded C port of the reference algorithm) + the same loss in torch on ONE thread, timed on
    this box's host cores on a bounded sample of the same workload.  A reported baseline, never the product path."""
    import numpy as np
    import torch
    from moss_amd.loss import training_loss
    from tests import helpers as hp
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
    return {"value": round(n / dt, 4), "unit": "iters/s", "cores": 1, "kind": "port",
            "sample": f"{n} full iterations of the same workload (C oracle fwd+bwd + torch loss, 1 thread) in {dt:.1f} s",
            "host_cores_available": os.cpu_count()}


if __name__ == "__main__":
    main()

"""Randomised check of the optimizer step inside the backward kernel (FlatAdamW.fuse_into_backward, C ABI
moss_raster_backward_raw_adamw) against backward -> bucket -> flat AdamW: random scenes (1-4000 Gaussians -- most P are no multiple of
4 or 64 --, ragged images, SH degrees 0-3, with and without per-Gaussian transforms / in-op posing, the spatial-order hint, weight
decay on and off), three steps each; parameters, both moments and the step count must agree BIT FOR BIT.
Usage: python scripts/fuzz_fused.py [n_cases] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from types import SimpleNamespace
from moss_amd.dist import GradBucket
from moss_amd.gaussian_model import GaussianSet
from moss_amd.gaussian_renderer import render, camera_view
from moss_amd.optim import FlatAdamW
from moss_amd.diff_gaussian_rasterization import _C
from fuzz_scenes import random_scene

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
nothing = 0
for seed in range(first, first + n_cases):
    if (seed - first) % 100 == 0:
        print(f"[progress] seed {seed} ({seed - first} of {n_cases} done, {bad} flagged)", flush=True)
    s, _, degree, _ = random_scene(seed)
    g = torch.Generator().manual_seed(seed + 77)
    lbs = bool(seed & 1)
    hint = bool(seed & 2)
    wd = 0.01 if seed & 4 else 0.0
    cam = camera_view(s.camera, dev)
    bg = s.bg.to(dev)
    w = torch.rand(3, s.camera.H, s.camera.W, generator=g).to(dev)
    T = s.transforms.to(dev).contiguous() if lbs else None
    tl = (0.01 * torch.randn(s.P, 3, generator=g)).to(dev) if lbs and (seed & 8) else None

    def make(fused):
        pc = GaussianSet(s, sh_degree=3, device=dev, unified_features=True)
        pc.active_sh_degree = degree
        pc.spatially_ordered = hint
        cx = _C.RasterContext()
        cx.set_async(True, capacity=2_000_000)
        pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, raster_context=cx,
                               raw_parameters_in_op=True, transforms_in_op=lbs, pose_in_op=lbs)
        bucket = GradBucket(list(pc.parameters()))
        opt = FlatAdamW(pc.param_groups(), bucket, eps=1e-15, weight_decay=wd, capturable=True)
        if fused:
            opt.fuse_into_backward(cx, means3D=pc._xyz, sh=pc._features, opacity=pc._opacity, scales=pc._scaling, rotations=pc._rotation)

        def step():
            if not fused:
                bucket.attach()
            kw = {} if T is None else ({"transforms": T} if tl is None else {"transforms": T, "translation": tl})
            out = render(cam, pc, pipe, bg, **kw)
            ((out["render"] * w).sum() + out["render_alpha"].sum()).backward()
            opt.step(skip_word=None if fused else _C.frame_status_word(cx.last_img_buffer))
            return out
        return SimpleNamespace(opt=opt, step=step, cx=cx)

    try:
        a, b = make(False), make(True)
        for it in range(3):
            oa, ob = a.step(), b.step()
        torch.cuda.synchronize()
        if int((oa["radii"] > 0).sum()) == 0:
            nothing += 1
        assert torch.equal(oa["render"], ob["render"]), "images differ"
        for name in ("flat_params", "exp_avg", "exp_avg_sq"):
            x, y = getattr(a.opt, name), getattr(b.opt, name)
            assert torch.equal(x, y), f"{name}: {int((x != y).sum())} elements differ, max {float((x - y).abs().max()):.3e}"
        assert a.opt.step_count() == b.opt.step_count() == 3
        assert bool(torch.isfinite(b.opt.flat_params).all())
    except Exception as ex:
        bad += 1
        print(f"seed {seed}: P={s.P} {s.camera.W}x{s.camera.H} lbs={lbs} translation={tl is not None} hint={hint} wd={wd} deg={degree}: {type(ex).__name__}: {str(ex)[:300]}")
print(f"{n_cases - bad} / {n_cases} random cases: fused == flat bit for bit ({nothing} of them rendered nothing: zero gradients, the step still taken)")
sys.exit(1 if bad else 0)

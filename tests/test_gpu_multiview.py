"""B views per optimizer step on one GPU (moss_amd/multiview.py; VERDICT r5 "next round" 8): the data-parallel step of SURVEY 8(e) with
the ranks on one device.  The captured form -- B chains on B HIP streams inside one hipGraph, the gradient average formed inside the
update kernel (C ABI moss_adamw_flat_ex: grads_extra) -- must equal, BIT FOR BIT, accumulating the same B views one after the other on
one stream with torch adds, scaling by 1/B and taking the same AdamW step."""
from types import SimpleNamespace

import pytest
import torch

from moss_amd import scenes

pytestmark = pytest.mark.gpu


def _setup(gpu, B):
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import camera_view, render
    scene = scenes.config2()
    c0 = scene.camera
    cams = [camera_view(scenes.make_camera(c0.W, c0.H, float(c0.K[0, 0]), float(c0.K[1, 1]), float(c0.K[0, 2]), float(c0.K[1, 2]), R, t), gpu)
            for R, t in scenes.look_at_ring(8)[:B]]
    bg = torch.zeros(3, device=gpu)
    T = (torch.eye(3) + 0.05 * torch.randn(scene.P, 3, 3, generator=torch.Generator().manual_seed(1234))).to(gpu)
    gt_scene = scenes.config2(seed=scenes.SEED + 7)
    gts = []
    with torch.no_grad():
        for cam in cams:
            o = render(cam, GaussianSet(gt_scene, sh_degree=3, device=gpu), SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False), bg)
            gts.append((o["render"].detach().clamp(0, 1).contiguous(), (o["render_alpha"].detach() > 0.5).float().contiguous()))
    return scene, cams, gts, bg, T


@pytest.mark.parametrize("B", [2, 4])
def test_parallel_captured_views_equal_sequential_accumulation(gpu, hip_lib, B):
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.multiview import MultiViewStep
    scene, cams, gts, bg, T = _setup(gpu, B)
    res = {}
    for key, kw in (("sequential", dict(parallel_streams=False, fused_sum=False)), ("parallel_graph", dict(parallel_streams=True, fused_sum=True))):
        pc = GaussianSet(scene, sh_degree=3, device=gpu, unified_features=True)
        mv = MultiViewStep(pc, B, cams, gts, bg, T, **kw)
        snap = mv.opt.snapshot()
        mv.eager_step()                                      # capacities (a step: undone below)
        torch.cuda.synchronize(gpu)
        if key == "parallel_graph":
            mv.capture(warmup=1)
        mv.opt.restore(snap)
        for _ in range(6):
            out = mv.step()
        torch.cuda.synchronize(gpu)
        mv.check()
        assert mv.opt.step_count() == 6
        res[key] = (mv.opt.flat_params.clone(), mv.opt.exp_avg.clone(), mv.opt.exp_avg_sq.clone(), [im.clone() for im in out["images"]])
    a, b = res["sequential"], res["parallel_graph"]
    assert float(a[1].abs().max()) > 0
    for x, y, what in zip(a[:3], b[:3], ("parameters", "exp_avg", "exp_avg_sq")):
        assert torch.equal(x, y), what
    for i, (x, y) in enumerate(zip(a[3], b[3])):
        assert torch.equal(x, y), f"image of view {i}"
    assert not torch.equal(a[3][0], a[3][1])                 # (the views are different cameras)


def test_one_view_is_the_ordinary_unfused_step(gpu, hip_lib):
    """B = 1: nothing but render -> loss -> backward into the bucket -> flat AdamW."""
    from moss_amd import dist as mdist
    from moss_amd import loss as mloss
    from moss_amd.diff_gaussian_rasterization import RasterContext
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import render
    from moss_amd.multiview import MultiViewStep
    from moss_amd.optim import FlatAdamW
    scene, cams, gts, bg, T = _setup(gpu, 1)
    pc = GaussianSet(scene, sh_degree=3, device=gpu, unified_features=True)
    mv = MultiViewStep(pc, 1, cams, gts, bg, T)
    for _ in range(3):
        mv.eager_step()
    pc2 = GaussianSet(scene, sh_degree=3, device=gpu, unified_features=True)
    bucket = mdist.GradBucket(list(pc2.parameters()))
    opt = FlatAdamW(pc2.param_groups(), bucket, eps=1e-15, capturable=True)
    cx = RasterContext(); cx.set_async(True)
    cx.set_grad_sink(sh=lambda: bucket.sink_for(pc2._features), opacity=lambda: bucket.sink_for(pc2._opacity), scales=lambda: bucket.sink_for(pc2._scaling),
                     rotations=lambda: bucket.sink_for(pc2._rotation), means3D=lambda: bucket.sink_for(pc2._xyz))
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=True, transforms_in_op=True, pose_in_op=True,
                           raw_parameters_in_op=True, raster_context=cx, grad_bucket=bucket)
    for _ in range(3):
        bucket.detach_grads()
        out = render(cams[0], pc2, pipe, bg, transforms=T)
        mloss.backward_from_loss(mloss.training_loss_fused(out["render"], out["render_alpha"], gts[0][0], gts[0][1], terms_out=bucket.loss_terms))
        bucket.collect()
        opt.step()
    torch.cuda.synchronize(gpu)
    # (the two buckets list the parameters in different orders: compared tensor by tensor)
    for name in ("_xyz", "_features", "_opacity", "_scaling", "_rotation"):
        a, b = getattr(pc, name), getattr(pc2, name)
        assert torch.equal(a.data, b.data), name
        ia = {id(p): i for i, p in enumerate(mv.opt.bucket.params)}[id(a)]
        ib = {id(p): i for i, p in enumerate(opt.bucket.params)}[id(b)]
        for x, y in zip(mv.opt._moments_of(ia), opt._moments_of(ib)):
            assert torch.equal(x, y), name
    assert float(mv.opt.exp_avg.abs().max()) > 0


def test_parallel_evaluation_renders_equal_single_renders(gpu, hip_lib):
    """MultiViewRender: four forward-only renders on four streams in one hipGraph == the four renders one after the other, bit for bit."""
    from moss_amd.diff_gaussian_rasterization import RasterContext
    from moss_amd.gaussian_model import GaussianSet
    from moss_amd.gaussian_renderer import render
    from moss_amd.multiview import MultiViewRender
    scene, cams, gts, bg, T = _setup(gpu, 4)
    pc = GaussianSet(scene, sh_degree=3, device=gpu, unified_features=True)
    tl = (0.01 * torch.randn(scene.P, 3, generator=torch.Generator().manual_seed(3))).to(gpu)
    mr = MultiViewRender(pc, cams, bg, transforms=T, translation=tl)
    mr.capture()
    outs = mr()
    torch.cuda.synchronize(gpu)
    mr.check()
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, fused_activations=False, transforms_in_op=True,
                           pose_in_op=True, raw_parameters_in_op=True, raster_context=RasterContext())
    for b, cam in enumerate(cams):
        with torch.no_grad():
            ref = render(cam, pc, pipe, bg, transforms=T, translation=tl)
        for got, want, name in zip(outs[b], (ref["render"], ref["render_depth"], ref["render_alpha"]), ("render", "depth", "alpha")):
            assert torch.equal(got, want), (b, name)
    assert not torch.equal(outs[0][0], outs[1][0])

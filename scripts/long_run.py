"""Stability check: the bench's training step (graph replay) for a few thousand iterations on the body target; prints the loss
trajectory, checks every 500 steps that nothing overflowed and everything is finite.  Run TWICE from the same start -- with the
flat AdamW kernel after the backward, then with the step taken inside the backward kernel (FlatAdamW.fuse_into_backward) -- and the
parameters and moments after the last step must be equal bit for bit.
Usage: python scripts/long_run.py [steps] [order, one digit per run: 0 = flat, 1 = fused; default 01]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from moss_amd import scenes, dist as mdist
from moss_amd.gaussian_model import GaussianSet
from moss_amd.gaussian_renderer import render, camera_view
from moss_amd.loss import training_loss_fused, backward_from_loss
from moss_amd.optim import FlatAdamW
from moss_amd.graphs import GraphedStep
from moss_amd import diff_gaussian_rasterization as dgr

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda:0")
scene = scenes.config3()
cam = camera_view(scene.camera, dev)
def run(fused):
    pc = GaussianSet(scene, sh_degree=3, device=dev, unified_features=True)
    bg = torch.zeros(3, device=dev)
    pipe0 = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    with torch.no_grad():
        o = render(cam, GaussianSet(scenes.config3(seed=scenes.SEED + 7), sh_degree=3, device=dev), pipe0, bg)
    gt = o["render"].detach().clamp(0, 1).contiguous(); gt_mask = (o["render_alpha"].detach() > 0.5).float().contiguous()
    bucket = mdist.GradBucket(list(pc.parameters()))
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, raw_parameters_in_op=True, grad_bucket=bucket)
    opt = FlatAdamW(pc.param_groups(), bucket, eps=1e-15, capturable=True)
    cx = dgr.RasterContext()
    cx.set_async(True, **({"capacity": int(os.environ["LR_CAPACITY"])} if os.environ.get("LR_CAPACITY") else {}))
    pipe.raster_context = cx
    if fused:
        opt.fuse_into_backward(cx, means3D=pc._xyz, sh=pc._features, opacity=pc._opacity, scales=pc._scaling, rotations=pc._rotation)
    else:
        cx.set_grad_sink(sh=lambda: bucket.sink_for(pc._features), means3D=lambda: bucket.sink_for(pc._xyz), opacity=lambda: bucket.sink_for(pc._opacity),
                         scales=lambda: bucket.sink_for(pc._scaling), rotations=lambda: bucket.sink_for(pc._rotation))

    def compute():
        bucket.detach_grads()
        out = render(cam, pc, pipe, bg)
        loss = training_loss_fused(out["render"], out["render_alpha"], gt, gt_mask, terms_out=bucket.loss_terms)
        backward_from_loss(loss)
        if not fused:
            bucket.collect()
            img = cx.last_img_buffer                         # (None after the first, synchronous forward that sizes the capacity)
            opt.step(skip_word=None if (img is None or os.environ.get("LR_NOGUARD")) else dgr._C.frame_status_word(img))
        return {"radii": out["radii"]}

    for _ in range(3):
        compute()
    torch.cuda.synchronize()
    step = GraphedStep(compute, device=dev, context=cx)
    t0 = time.time(); last = 0
    for it in range(1, steps + 1):
        out = step()
        if it % 500 == 0 or it == steps:
            torch.cuda.synchronize()
            recaptured = step.check()
            terms = bucket.loss_terms.cpu().tolist()
            finite = bool(torch.isfinite(opt.flat_params).all()) and bool(torch.isfinite(bucket.loss_terms).all())
            dt = time.time() - t0
            print(f"step {it:5d}: loss {terms[0]:.5f} (L1 {terms[1]:.5f}, SSIM {terms[2]:.4f}, mask {terms[3]:.5f})  visible {int((out['radii'] > 0).sum())}  "
                  f"instances {cx.last_needed}  finite {finite}  {1e3 * dt / (it - last):.3f} ms/step" + ("  [re-captured, capacity %d]" % cx.capacity if recaptured else ""))
            assert finite
            t0 = time.time(); last = it
    torch.cuda.synchronize()
    return opt.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count(), step.dropped_frames


import itertools
order = [bool(int(c)) for c in (sys.argv[2] if len(sys.argv) > 2 else "01")]
res = []
for f in order:
    print("--- " + ("AdamW step inside the per-Gaussian backward kernel" if f else "flat AdamW kernel after the backward"))
    res.append(run(f))
ok = True
for (i, x), (j, y) in itertools.combinations(enumerate(res), 2):
    same = [bool(torch.equal(u, v)) for u, v in zip(x[:3], y[:3])]
    print(f"run {i} (fused={order[i]}) vs run {j} (fused={order[j]}): steps {x[3]} / {y[3]}, dropped {x[4]} / {y[4]}; parameters, exp_avg, exp_avg_sq bit-identical: {same}")
    ok = ok and all(same) and x[3] == y[3]
assert ok
print("ok")

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from moss_amd import scenes, _lib
from moss_amd import dist as mdist
from moss_amd.optim import FlatAdamW
from moss_amd.gaussian_model import GaussianSet
from moss_amd.gaussian_renderer import render, camera_view
from moss_amd.loss import training_loss_fused as training_loss
from moss_amd.diff_gaussian_rasterization import _C
gpu = torch.device("cuda:0")
s = scenes.config3()
pc = GaussianSet(s, sh_degree=3, device=gpu)
cam = camera_view(s.camera, gpu)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
bg = torch.zeros(3, device=gpu)
H, W = s.camera.H, s.camera.W
gt = scenes.synthetic_target(H, W).to(gpu)
gt_mask = (gt.mean(0, keepdim=True) > 0.5).float()
bucket = mdist.GradBucket(list(pc.parameters()))
opt = FlatAdamW(pc.param_groups(), bucket, eps=1e-15)
print([ (g.get("name"), g["lr"]) for g in pc.param_groups()])
def step():
    bucket.attach()
    out = render(cam, pc, pipe, bg)
    loss = training_loss(out["render"], out["render_alpha"], gt, gt_mask)
    loss.backward()
    opt.step()
    return out, loss
for it in range(1001):
    out, loss = step()
    if it % 100 == 0:
        r = out["radii"]
        tt = ((2 * r + 16) // 16).float() ** 2
        _lib.profile_enable(None); _lib.profile_read()
        for _ in range(10): step()
        torch.cuda.synchronize()
        pr = _lib.profile_read(); _lib.profile_enable([])
        print(it, "loss %.4f" % float(loss), "R", _C.last_num_rendered, "radius max", int(r.max()), "n(r>32)", int((r > 32).sum()), "n(r>128)", int((r > 128).sum()),
              {k[:14]: round(v[0] / max(v[1], 1), 3) for k, v in pr.items()}, flush=True)

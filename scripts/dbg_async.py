import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from moss_amd import scenes, _lib
from moss_amd.gaussian_model import GaussianSet
from moss_amd.gaussian_renderer import render, camera_view
from moss_amd.loss import training_loss_fused as training_loss
import moss_amd.diff_gaussian_rasterization as dgr
gpu = torch.device("cuda:0")
s = scenes.config3()
pc = GaussianSet(s, sh_degree=3, device=gpu)
cam = camera_view(s.camera, gpu)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
bg = torch.zeros(3, device=gpu)
H, W = s.camera.H, s.camera.W
gt = scenes.synthetic_target(H, W).to(gpu)
gt_mask = (gt.mean(0, keepdim=True) > 0.5).float()
def compute():
    pc.zero_grad()
    out = render(cam, pc, pipe, bg)
    loss = training_loss(out["render"], out["render_alpha"], gt, gt_mask)
    loss.backward()
def run(tag, n=50):
    for _ in range(5): compute()
    torch.cuda.synchronize()
    _lib.profile_enable(None); _lib.profile_read()
    na = torch.cuda.memory_stats()["num_device_alloc"]
    t0 = time.perf_counter()
    for _ in range(n): compute()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    pr = _lib.profile_read(); _lib.profile_enable([])
    print(tag, "host ms/step %.3f total ms/step %.3f" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3),
          "device allocs", torch.cuda.memory_stats()["num_device_alloc"] - na,
          {k: round(v[0] / max(v[1], 1), 4) for k, v in pr.items()}, flush=True)
run("sync ")
dgr.set_async(True)
run("async")
print("capacity", dgr._C.ASYNC.capacity, "needed", dgr._C.ASYNC.last_needed)
run("async")
dgr.set_async(True, capacity=300000)
run("async cap=300000")
dgr.set_async(False)
run("sync ")

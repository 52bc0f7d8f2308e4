import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from moss_amd import scenes
from moss_amd.gaussian_model import GaussianSet
from moss_amd.gaussian_renderer import render, camera_view
import moss_amd.diff_gaussian_rasterization as dgr
gpu = torch.device("cuda:0")
s = scenes.config2()
pc = GaussianSet(s, device=gpu)
cam = camera_view(s.camera, gpu)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
bg = torch.zeros(3, device=gpu)
w = torch.rand(3, s.camera.H, s.camera.W, device=gpu)
params = list(pc.parameters())
grads = [torch.zeros_like(p) for p in params]
def compute():
    for p, g in zip(params, grads):
        g.zero_(); p.grad = g
    out = render(cam, pc, pipe, bg)
    ((out["render"] * w).sum() + out["render_alpha"].sum()).backward()
    return out["render"].detach(), out["radii"]
dgr.set_async(True)
compute()
print("capacity", dgr._C.ASYNC.capacity)
side = torch.cuda.Stream(gpu)
side.wait_stream(torch.cuda.current_stream(gpu))
with torch.cuda.stream(side):
    compute()
torch.cuda.current_stream(gpu).wait_stream(side)
torch.cuda.synchronize(gpu)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    g_img, g_radii = compute()
for trial in range(3):
    if trial:
        with torch.no_grad():
            pc._xyz.add_(0.01 * trial)
    graph.replay()
    torch.cuda.synchronize(gpu)
    img_g = g_img.clone(); rad_g = g_radii.clone(); grads_g = [g.clone() for g in grads]
    img_e, rad_e = compute()
    torch.cuda.synchronize(gpu)
    d = (img_g - img_e).abs()
    print("trial", trial, "img maxdiff", float(d.max()), "ndiff", int((d > 0).sum()), "radii diff", int((rad_g != rad_e).sum()),
          "img_g sum", float(img_g.sum()), "img_e sum", float(img_e.sum()))
    for a, b in zip(grads_g, grads):
        print("   grad maxdiff", float((a - b).abs().max()), float(a.abs().max()), float(b.abs().max()))
dgr.check_async_status()

"""Per-stage device times of the raw op (no loss / optimizer) on one config, via the library's HIP-event hooks."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moss_amd import scenes, _lib
from tests import helpers as hp

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg3")
ap.add_argument("--mode", default="precomp")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--order", default="as_is", help="as_is | morton (Gaussians re-indexed along a 3-D Morton curve) | random")
args = ap.parse_args()
dev = torch.device("cuda:0")
# (moss45k: MOSS's own ceiling -- 45 695 Gaussians, scene/gaussian_model.py:496 -- at its ZJU-MoCap resolution, 1024 x 1024)
scene = {"cfg1": scenes.config1, "cfg2": scenes.config2, "cfg3": scenes.config3, "cfg5": scenes.config5,
         "moss45k": lambda: scenes.body_scene(45_695, 1024, 1024, 1080.0, init_like=False, name="moss45k"),
         "moss7k": lambda: scenes.body_scene(6_890, 1024, 1024, 1080.0, init_like=True, name="moss7k")}[args.config]()
if args.order != "as_is":
    from moss_amd.densify import spatial_order
    perm = spatial_order(scene.means3D) if args.order == "morton" else torch.randperm(scene.P, generator=torch.Generator().manual_seed(1))
    for k, v in list(vars(scene).items()):
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == scene.P:
            setattr(scene, k, v[perm].contiguous())
d = hp.inputs_of(scene, args.mode)
dc, dd, da = hp.image_grads(d.H, d.W)
dc, dd, da = dc.to(dev), dd.to(dev), da.to(dev)
for _ in range(5):
    t = hp.hip_forward(d, dev); hp.hip_backward(d, t, dc, dd, da, dev)
torch.cuda.synchronize()
_lib.profile_enable(None)
t0 = time.perf_counter()
for _ in range(args.iters):
    t = hp.hip_forward(d, dev); hp.hip_backward(d, t, dc, dd, da, dev)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.iters * 1e3
prof = _lib.profile_read()
out = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}
out["sum"] = round(sum(out.values()), 4); out["wall_ms_per_iter"] = round(wall, 4); out["R"] = t.R
print(args.config, os.environ.get("MOSS_BLEND_CULL", "1"), json.dumps(out))

"""Phase breakdown of the per-Gaussian backward kernel WITH the AdamW step inside (FlatAdamW.fuse_into_backward), beside the plain one:
thread 0 of every block stamps s_memtime at the phase boundaries (diagnostic build: MOSS_AMD_LIB_DIR=lib_diag)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
from moss_amd import scenes, _lib
from moss_amd.dist import GradBucket
from moss_amd.gaussian_model import GaussianSet
from moss_amd.gaussian_renderer import render, camera_view
from moss_amd.optim import FlatAdamW
from moss_amd.diff_gaussian_rasterization import _C
dev = torch.device("cuda:0")
L = _lib.lib()
assert L.moss_build_has_diagnostics(), "run with MOSS_AMD_LIB_DIR=lib_diag (python -m moss_amd.build --diag)"
L.moss_raster_debug_set_stamps.argtypes = [ctypes.c_void_p]
cfg = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "cfg3"
s = {"cfg2": scenes.config2, "cfg3": scenes.config3, "cfg5": scenes.config5,
     "moss45k": lambda: scenes.body_scene(45_695, 1024, 1024, 1080.0, init_like=False, name="moss45k")}[cfg]()
cam = camera_view(s.camera, dev)
bg = torch.zeros(3, device=dev)
w = torch.rand(3, s.camera.H, s.camera.W, device=dev)
for fused in (False, True):
    pc = GaussianSet(s, sh_degree=3, device=dev, unified_features=True)
    cx = _C.RasterContext()
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False, raster_context=cx, raw_parameters_in_op=True)
    bucket = GradBucket(list(pc.parameters()))
    opt = FlatAdamW(pc.param_groups(), bucket, eps=1e-15, capturable=True)
    if fused:
        opt.fuse_into_backward(cx, means3D=pc._xyz, sh=pc._features, opacity=pc._opacity, scales=pc._scaling, rotations=pc._rotation)
    else:
        cx.set_grad_sink(sh=lambda: bucket.sink_for(pc._features), opacity=lambda: bucket.sink_for(pc._opacity),
                         scales=lambda: bucket.sink_for(pc._scaling), rotations=lambda: bucket.sink_for(pc._rotation),
                         means3D=lambda: bucket.sink_for(pc._xyz))
    buf = torch.zeros(16 * 2048 * 16, dtype=torch.int64, device=dev)
    for it in range(4):
        bucket.detach_grads()
        out = render(cam, pc, pipe, bg)
        loss = (out["render"] * w).sum() + out["render_alpha"].sum()
        torch.cuda.synchronize()
        if it == 3:
            L.moss_raster_debug_set_stamps(buf.data_ptr())
        loss.backward()
        torch.cuda.synchronize()
        L.moss_raster_debug_set_stamps(None)
    st = buf.cpu().numpy().reshape(-1, 16); st = st[st[:, 0] > 0]
    ph = np.diff(st[:, :7], axis=1)
    print(f"--- fused = {fused}: blocks {len(st)}")
    print("mean cycles per phase [gather+SHload, barrier, pre-SH math+writes, SH, scale/rot, tail writes (+ the 11 scalars' update), copy-out / SH update]:", ph.mean(0).astype(int))
    sub = np.diff(st[:, 8:13], axis=1)
    print("inside phase 0 [.., masks requested, masks in + counted, records gathered, coop gather] mean:", sub.mean(0).astype(int), "p90:", np.percentile(sub, 90, axis=0).astype(int))
    print("p90:", np.percentile(ph, 90, axis=0).astype(int), " block total mean", int((st[:, 6] - st[:, 0]).mean()), "max", int((st[:, 6] - st[:, 0]).max()))
    tot = st[:, 6] - st[:, 0]
    worst = np.argsort(-tot)[:8]
    print("slowest blocks: total cycles", tot[worst].astype(int).tolist(), "| [masks requested, masks in, records gathered, coop gather]:", sub[worst].astype(int).tolist())
    r0 = st[:, 13].min()
    print("realtime (us): block starts median %.2f p90 %.2f last %.2f | block ends median %.2f p90 %.2f last %.2f" % (
        np.median(st[:, 13] - r0) / 100, np.percentile(st[:, 13] - r0, 90) / 100, (st[:, 13].max() - r0) / 100,
        np.median(st[:, 14] - r0) / 100, np.percentile(st[:, 14] - r0, 90) / 100, (st[:, 14].max() - r0) / 100))

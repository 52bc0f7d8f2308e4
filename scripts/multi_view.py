"""B views per optimizer step on ONE GPU (VERDICT r5 "next round" 8): B cameras in, B images out, the B views' gradients averaged in a
fixed order, ONE AdamW step -- the semantics of the N-GPU data-parallel step of SURVEY 8(e) with N = B.

A 512 x 512 view cannot fill 256 CUs: every kernel of the step is a latency chain.  Here the B views' chains --
forward, loss, backward, each on its own RasterContext and gradient bucket -- are issued on B HIP streams INSIDE ONE captured hipGraph
(parallel branches; joined in front of the update), so that one view's kernels fill the issue slots and CUs the other's leave idle.

usage: python scripts/multi_view.py [B ...]          (default: 1 2 4)
"""
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from moss_amd import dist as mdist                       # noqa: E402
from moss_amd import loss as mloss                       # noqa: E402
from moss_amd import scenes                              # noqa: E402
from moss_amd.multiview import MultiViewStep             # noqa: E402
from moss_amd.gaussian_model import GaussianSet          # noqa: E402
from moss_amd.gaussian_renderer import camera_view, render   # noqa: E402


def main():
    Bs = [int(x) for x in sys.argv[1:]] or [1, 2, 4]
    dev = torch.device("cuda", 0)
    scene = scenes.config3()
    poses = scenes.look_at_ring(8)
    c0 = scene.camera
    cams = [camera_view(scenes.make_camera(c0.W, c0.H, float(c0.K[0, 0]), float(c0.K[1, 1]), float(c0.K[0, 2]), float(c0.K[1, 2]), R, t), dev)
            for R, t in poses]
    bg = torch.zeros(3, device=dev)
    gT = torch.Generator().manual_seed(1234)
    T = (torch.eye(3) + 0.05 * torch.randn(scene.P, 3, 3, generator=gT)).to(dev)
    gt_scene = scenes.config3(seed=scenes.SEED + 7)
    gts = []
    with torch.no_grad():
        for cam in cams:
            o = render(cam, GaussianSet(gt_scene, sh_degree=3, device=dev), SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False), bg)
            gts.append((o["render"].detach().clamp(0, 1).contiguous(), (o["render_alpha"].detach() > 0.5).float().contiguous()))
    for B in Bs:
        for streams in ((False, True) if B > 1 else (False,)):
            pc = GaussianSet(scene, sh_degree=3, device=dev, unified_features=True)
            mv = MultiViewStep(pc, B, cams[:B], gts[:B], bg, T, parallel_streams=streams)
            for _ in range(3):
                mv.eager_step()
            torch.cuda.synchronize(dev)
            mv.capture()
            for _ in range(10):
                mv.step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            n = 200
            for _ in range(n):
                mv.step()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / n
            mv.check()
            print(f"B={B} streams={streams}: {1e3 * dt:.4f} ms per step, {B / dt:.1f} views/s, {1 / dt:.1f} optimizer steps/s", flush=True)


if __name__ == "__main__":
    main()

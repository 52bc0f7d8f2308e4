"""The two loss kernels in their three forms on a masked frame (a rendered body on black against another body): full frame
(moss_photometric_loss) and MOSS's own expression on the body's bound_mask / bounding rectangle (moss_photometric_loss_roi), 512 x 512
and 1024 x 1024.  hipEvents around graphs of 20 calls.  usage: python scripts/loss_times_roi.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moss_amd import _lib
from moss_amd.graphs import capturing
from moss_amd.loss import ViewRegion
dev = torch.device("cuda:0")
L = _lib.lib()
C = 3
for H in (512, 1024):
    W = H
    g = torch.Generator().manual_seed(3)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    body = lambda cx, cy, rx, ry: (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 < 1.0).float()
    k_ = H / 512.0
    m1, m2 = body(250 * k_, 260 * k_, 70 * k_, 200 * k_), body(262 * k_, 256 * k_, 74 * k_, 196 * k_)
    img = (torch.rand(C, H, W, generator=g) * m1).to(dev).contiguous(); gt = (torch.rand(C, H, W, generator=g) * m2).to(dev).contiguous()
    alpha = (m1[None] * 0.97).to(dev).contiguous(); mask = m2[None].to(dev).contiguous()
    # bound_mask: the two bodies' bounding box grown by 12 px (MOSS: the projected 3-D box of the body)
    u = (m1 + m2) > 0
    ys, xs = u.any(1).nonzero().flatten(), u.any(0).nonzero().flatten()
    bm = torch.zeros(1, H, W, dtype=torch.uint8)
    bm[0, max(int(ys[0]) - 12, 0):int(ys[-1]) + 13, max(int(xs[0]) - 12, 0):int(xs[-1]) + 13] = 1
    region = ViewRegion(bm.to(dev))
    ws = torch.empty(int(L.moss_loss_workspace_bytes(C, H, W)), dtype=torch.uint8, device=dev)
    out = torch.zeros(4, device=dev); dimg = torch.empty_like(img); dalpha = torch.empty_like(alpha)

    def full():
        s = torch.cuda.current_stream().cuda_stream
        assert L.moss_photometric_loss(C, H, W, img.data_ptr(), gt.data_ptr(), alpha.data_ptr(), mask.data_ptr(), 0.2, 0.5, out.data_ptr(),
                                       dimg.data_ptr(), dalpha.data_ptr(), ws.data_ptr(), ws.numel(), s) == 0

    def roi():
        s = torch.cuda.current_stream().cuda_stream
        assert L.moss_photometric_loss_roi(C, H, W, img.data_ptr(), gt.data_ptr(), alpha.data_ptr(), mask.data_ptr(), region.bound.data_ptr(),
                                           region.rect.data_ptr(), 1.0, 0.2, 0.5, out.data_ptr(), dimg.data_ptr(), dalpha.data_ptr(),
                                           ws.data_ptr(), ws.numel(), s) == 0

    for name, call in (("full frame", full), ("bound_mask + rectangle", roi)):
        for _ in range(20): call()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with capturing(gr):
            for _ in range(20): call()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): gr.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"{H}x{W} {name:24s}: both kernels {e0.elapsed_time(e1) / 200 * 1e3:6.2f} us per call   (rectangle {region.xywh}, {int(region.rect[4])} mask pixels; loss {out[0].item():.5f})", flush=True)

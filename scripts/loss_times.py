"""Time of the two fused-loss kernels (ssim_pass1 / ssim_pass2) on a masked frame like the bench's: a rendered body on black against
another body (about a quarter of the 32 x 32 tiles hold anything), 512 x 512 x 3.  hipEvents around batches of calls; the per-kernel
split comes from running this under `rocprofv3 --kernel-trace --stats`.  usage: python scripts/loss_times.py [full]   (full: no empty tile)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moss_amd import _lib
from moss_amd.graphs import capturing
dev = torch.device("cuda:0")
L = _lib.lib()
C, H, W = 3, (1024 if "1024" in sys.argv else 512), (1024 if "1024" in sys.argv else 512)
g = torch.Generator().manual_seed(3)
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
def body(cx, cy, rx, ry):
    m = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 < 1.0).float()
    return m
if "full" in sys.argv:
    img = torch.rand(C, H, W, generator=g); gt = torch.rand(C, H, W, generator=g); alpha = torch.rand(1, H, W, generator=g); mask = (torch.rand(1, H, W, generator=g) > 0.5).float()
else:
    k_ = H / 512.0
    m1, m2 = body(250 * k_, 260 * k_, 70 * k_, 200 * k_), body(262 * k_, 256 * k_, 74 * k_, 196 * k_)
    img = torch.rand(C, H, W, generator=g) * m1; gt = torch.rand(C, H, W, generator=g) * m2
    alpha = m1[None] * 0.97; mask = m2[None]
img, gt, alpha, mask = (t.to(dev).contiguous() for t in (img, gt, alpha, mask))
tiles = sum(1 for by in range(H // 32) for bx in range(W // 32)
            if float(img[:, max(0, 32 * by - 5):32 * by + 37, max(0, 32 * bx - 5):32 * bx + 37].abs().sum() + gt[:, max(0, 32 * by - 5):32 * by + 37, max(0, 32 * bx - 5):32 * bx + 37].abs().sum()) > 0)
ws = torch.empty(int(L.moss_loss_workspace_bytes(C, H, W)), dtype=torch.uint8, device=dev)
out = torch.zeros(4, device=dev); dimg = torch.empty_like(img); dalpha = torch.empty_like(alpha)
def call():
    s = torch.cuda.current_stream().cuda_stream          # (inside torch.cuda.graph: the capture stream)
    rc = L.moss_photometric_loss(C, H, W, img.data_ptr(), gt.data_ptr(), alpha.data_ptr(), mask.data_ptr(), 0.2, 0.5, out.data_ptr(),
                                 dimg.data_ptr(), dalpha.data_ptr(), ws.data_ptr(), ws.numel(), s)
    assert rc == 0
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
print(f"{H}x{W}: non-empty tiles (halo included): {tiles} of {(H // 32) * (W // 32)}; loss {out.tolist()}; both kernels: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per call (graph of 20 calls)")
import hashlib
print("checksums", float(dimg.double().abs().sum()), float(dalpha.double().abs().sum()),
      "sha256 of (loss terms, dL_dimage, dL_dalpha):", hashlib.sha256(out.cpu().numpy().tobytes() + dimg.cpu().numpy().tobytes() + dalpha.cpu().numpy().tobytes()).hexdigest()[:16])

"""Phase stamps of the two loss kernels (thread 0 of every workgroup, s_memrealtime = 100 MHz device-wide; diagnostic build:
MOSS_AMD_LIB_DIR=lib_diag MOSS_LOSS_STAMPS=1): start, loads consumed, tile in LDS, horizontal pass done, vertical pass + stores issued, end."""
import ctypes, os, sys
os.environ["MOSS_LOSS_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
assert L.moss_build_has_diagnostics(), "run with MOSS_AMD_LIB_DIR=lib_diag"
L.moss_raster_debug_set_stamps.argtypes = [ctypes.c_void_p]
C, H, W = 3, (1024 if "1024" in sys.argv else 512), (1024 if "1024" in sys.argv else 512)
g = torch.Generator().manual_seed(3)
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
body = lambda cx, cy, rx, ry: (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 < 1.0).float()
k_ = H / 512.0
m1, m2 = body(250 * k_, 260 * k_, 70 * k_, 200 * k_), body(262 * k_, 256 * k_, 74 * k_, 196 * k_)
img = (torch.rand(C, H, W, generator=g) * m1).to(dev); gt = (torch.rand(C, H, W, generator=g) * m2).to(dev)
alpha = (m1[None] * 0.97).to(dev).contiguous(); mask = m2[None].to(dev).contiguous()
ws = torch.empty(int(L.moss_loss_workspace_bytes(C, H, W)), dtype=torch.uint8, device=dev)
out = torch.zeros(4, device=dev); dimg = torch.empty_like(img); dalpha = torch.empty_like(alpha)
buf = torch.zeros(2 * 8 * 4096, dtype=torch.int64, device=dev)
def call():
    assert L.moss_photometric_loss(C, H, W, img.data_ptr(), gt.data_ptr(), alpha.data_ptr(), mask.data_ptr(), 0.2, 0.5, out.data_ptr(), dimg.data_ptr(),
                                   dalpha.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream) == 0
for _ in range(5): call()
torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(buf.data_ptr()); call(); torch.cuda.synchronize(); L.moss_raster_debug_set_stamps(None)
s = buf.cpu().numpy().astype(np.float64)
for name, off, last in (("ssim_pass1 [start, loads consumed, tile in LDS, horizontal done, vertical + stores issued, reduced]", 0, 5),
                        ("ssim_pass2 [start, loads consumed, maps in LDS, horizontal done, vertical + stores issued]", 8 * 4096, 4)):
    w = s[off: off + 8 * 4096].reshape(-1, 8)
    w = w[w[:, 0] > 0]
    t0 = w[:, 0].min()
    print(name, "workgroups", len(w))
    d = np.diff(w[:, :last + 1], axis=1) / 100
    print("   per-workgroup phase durations (us) median", np.round(np.median(d, axis=0), 2), "p90", np.round(np.percentile(d, 90, axis=0), 2),
          "| workgroup lifetime median %.2f p90 %.2f" % (np.median(w[:, last] - w[:, 0]) / 100, np.percentile(w[:, last] - w[:, 0], 90) / 100))
    for i in range(last + 1):
        print("   stamp %d: median %6.2f  p10 %6.2f  p90 %6.2f  max %6.2f us" % (i, np.median(w[:, i] - t0) / 100, np.percentile(w[:, i] - t0, 10) / 100,
                                                                                np.percentile(w[:, i] - t0, 90) / 100, (w[:, i].max() - t0) / 100))

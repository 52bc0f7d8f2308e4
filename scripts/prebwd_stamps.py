"""Phase breakdown of the per-Gaussian backward kernel (thread 0 of every block stamps s_memtime at phase boundaries)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import scenes, _lib
from tests import helpers as hp
dev = torch.device("cuda:0")
d = hp.inputs_of(scenes.config3(), "scale_rot")
dc, dd, da = hp.image_grads(d.H, d.W)
dc, dd, da = dc.to(dev), dd.to(dev), da.to(dev)
L = _lib.lib()
L.moss_raster_debug_set_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16 * 2048 * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    t = hp.hip_forward(d, dev); hp.hip_backward(d, t, dc, dd, da, dev)
t = hp.hip_forward(d, dev)
torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(buf.data_ptr())
hp.hip_backward(d, t, dc, dd, da, dev); torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(None)
s = buf.cpu().numpy().reshape(-1, 16)
nb = (100000 + 63) // 64
s = s[:nb]
t0 = s[:, 0].min()
ph = np.diff(s[:, :7], axis=1)
print("blocks", nb, " kernel span (cycles):", int(s[:, 6].max() - t0), " block start spread:", int(s[:, 0].max() - t0))
print("mean cycles per phase [gather+SHload, barrier, pre-SH math+writes, SH, scale/rot, tail writes, copy-out]:", ph.mean(0).astype(int))
f = s[:, [0, 8, 9, 10, 11, 12, 1]]
print("fine phase 1 [first-level loads, pos loads, mask loads, record batches, coop loop, SH->LDS]:", np.diff(f, axis=1).mean(0).astype(int))
f2 = s[:, [10, 7, 15, 11]]
print("balanced gather [count + file descriptors, request + wait for the records, scans + hand-over]:", np.diff(f2, axis=1).mean(0).astype(int))
print("p90:", np.percentile(ph, 90, axis=0).astype(int), " block total mean", int((s[:, 6] - s[:, 0]).mean()), "max", int((s[:, 6] - s[:, 0]).max()))

r0 = s[:, 13].min()
print("realtime (us): block starts median %.2f p90 %.2f last %.2f | block ends median %.2f p90 %.2f last %.2f" % (
    np.median(s[:, 13] - r0) / 100, np.percentile(s[:, 13] - r0, 90) / 100, (s[:, 13].max() - r0) / 100,
    np.median(s[:, 14] - r0) / 100, np.percentile(s[:, 14] - r0, 90) / 100, (s[:, 14].max() - r0) / 100))

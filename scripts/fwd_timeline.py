"""Timeline of the forward blend kernel's heavy items on the device-wide 100 MHz clock (s_memrealtime): when each block item's
blender (half 0) started and ended relative to the earliest start, and the kernel's host-measured duration beside it."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import scenes, _lib
from tests import helpers as hp
dev = torch.device("cuda:0")
cfg = getattr(scenes, sys.argv[1] if len(sys.argv) > 1 else "config3")()
d = hp.inputs_of(cfg, "precomp")
L = _lib.lib()
L.moss_raster_debug_set_stamps.argtypes = [ctypes.c_void_p]
T_pad = 4096
buf = torch.zeros(16 * T_pad * 8, dtype=torch.int64, device=dev)
for _ in range(3): hp.hip_forward(d, dev)
torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(buf.data_ptr())
hp.hip_forward(d, dev); torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8).astype(np.uint64)
idx = np.nonzero(s[:, 4] > 0)[0]
w = s[idx]
start = w[:, 4].astype(np.int64); end = (w[:, 6] >> np.uint64(16)).astype(np.int64); rounds = (w[:, 6] & np.uint64(0xffff)).astype(np.int64)
t0 = start.min()
st_us = (start - t0) / 100.0; en_us = (end - t0) / 100.0
print("items stamped:", len(w), " with trips:", int((rounds > 0).sum()))
print("item START us after the first: percentiles 50/90/99/100:", np.percentile(st_us, [50, 90, 99, 100]).round(1))
print("item END   us after the first start: percentiles 10/50/90/99/100:", np.percentile(en_us, [10, 50, 90, 99, 100]).round(1))
late = np.argsort(-en_us)[:14]
print("last to end:  rank blk   start   end   entries  trips  cycles   per-trip")
for i in late:
    print(f"   {idx[i] // 16:5d} {idx[i] % 16:3d} {st_us[i]:7.1f} {en_us[i]:6.1f} {int(w[i,1]):8d} {int(w[i,7]):6d} {int(w[i,0]):7d} {int(w[i,5]) / max(int(w[i,7]),1):8.0f}")
hist, edges = np.histogram(en_us, bins=np.arange(0, en_us.max() + 2.5, 2.5))
print("ends per 2.5 us bin:", list(zip(edges[:-1].astype(int).tolist(), hist.tolist())))
hist, edges = np.histogram(st_us, bins=np.arange(0, st_us.max() + 2.5, 2.5))
print("starts per 2.5 us bin:", list(zip(edges[:-1].astype(int).tolist(), hist.tolist())))

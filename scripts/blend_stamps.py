"""Phase breakdown of the forward blend kernel's slowest workgroups (s_memtime stamps, diagnostics build path)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import scenes, _lib
from tests import helpers as hp
dev = torch.device("cuda:0")
d = hp.inputs_of(scenes.config3(), "precomp")
L = _lib.lib()
L.moss_raster_debug_set_stamps.argtypes = [ctypes.c_void_p]
T_pad = 1024
buf = torch.zeros(4 * T_pad * 8, dtype=torch.int64, device=dev)
for _ in range(3): hp.hip_forward(d, dev)
torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(buf.data_ptr())
hp.hip_forward(d, dev); torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8)
order = np.argsort(-s[:, 0])[:12]
print("wg      total   bar1  stage   bar2   cull  trips | batches trips   (cycles of the 100 MHz-independent shader clock)")
for i in order:
    print(f"{i:5d} {s[i,0]:8d} {s[i,1]:6d} {s[i,2]:6d} {s[i,3]:6d} {s[i,4]:6d} {s[i,5]:7d} | {s[i,6]:4d} {s[i,7]:6d}   per-trip {s[i,5]/max(s[i,7],1):.0f} cyc")
tot = s[s[:, 6] > 0]
print("all WGs with work:", len(tot), "sum total cycles", tot[:, 0].sum(), "max", tot[:, 0].max(), "mean", int(tot[:, 0].mean()))
print("phase shares of the top-12:", (s[order, 1:6].sum(0) / s[order, 0].sum()).round(3))
# load-balance view: every queued item (tile, quadrant) with work, in queue (LPT) order
items = s[: 4 * T_pad]
work = items[items[:, 6] > 0]
tot_cycles = work[:, 0].astype(np.float64)
print("items with work:", len(work), " sum of item cycles:", int(tot_cycles.sum()), " longest item:", int(tot_cycles.max()),
      " sum/768 workgroups:", int(tot_cycles.sum() / 768), " (span lower bounds: longest item vs. perfectly balanced)")
print("item cycles percentiles 50/90/99/100:", np.percentile(tot_cycles, [50, 90, 99, 100]).astype(int))
print("trips per item percentiles 50/90/99/100:", np.percentile(work[:, 7], [50, 90, 99, 100]).astype(int), " total trips", int(work[:, 7].sum()),
      " batches total", int(work[:, 6].sum()))
print("phase shares over all items:", (work[:, 1:6].sum(0) / work[:, 0].sum()).round(3), "(top, stage, barrier, cull, trips)")

"""Per-item breakdown of the forward blend kernel as its BLENDER waves see it (s_memtime stamps): cycles in trips, cycles starved
(waiting for the pair's scanner), cycles until the first trip."""
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
buf = torch.zeros(16 * T_pad * 8, dtype=torch.int64, device=dev)
for _ in range(3): hp.hip_forward(d, dev)
torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(buf.data_ptr())
hp.hip_forward(d, dev); torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8)
s[:, 6] &= 0xffff                                           # (the upper bits carry the item's end on the realtime clock: fwd_timeline.py)
work = s[s[:, 6] > 0]
order = np.argsort(-work[:, 0])[:12]
print("  blender cycles  n entries  rounds  trips(scheduled)  trip cycles  starved  first trip after")
for i in order:
    w = work[i]
    print(f"{w[0]:10d} {w[1]:10d} {w[6]:6d} {w[7]:6d} {w[5]:10d} {w[2]:8d} {w[3]:8d}   per-trip {w[5]/max(w[7],1):.0f}")
tc = work[:, 0].astype(np.float64)
print("items with work:", len(work), "sum cycles", int(tc.sum()), "longest", int(tc.max()), "sum/3072 waves", int(tc.sum() / 3072))
print("percentiles 50/90/99/100:", np.percentile(tc, [50, 90, 99, 100]).astype(int))
print("total rounds", int(work[:, 6].sum()), "scheduled trips", int(work[:, 7].sum()), "trip share", round(work[:, 5].sum() / tc.sum(), 3), "starved share", round(work[:, 2].sum() / tc.sum(), 3))

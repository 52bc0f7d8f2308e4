"""Per-item picture of the forward blend kernel (diagnostic build: MOSS_AMD_LIB_DIR=lib_diag): trips, cycles per trip, start / end on
the device-wide clock, by the dispatch class of the item's workgroup -- what a cost-aware dealing of the items would have to beat."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import scenes, _lib
from tests import helpers as hp
dev = torch.device("cuda:0")
cfg = getattr(scenes, sys.argv[1] if len(sys.argv) > 1 else "config3")()
d = hp.inputs_of(cfg, "scale_rot")
L = _lib.lib()
assert L.moss_build_has_diagnostics(), "run with MOSS_AMD_LIB_DIR=lib_diag"
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
start = w[:, 4].astype(np.int64); end = (w[:, 6] >> np.uint64(16)).astype(np.int64)
t0 = start.min()
st, en = (start - t0) / 100.0, (end - t0) / 100.0
trips = w[:, 7].astype(np.int64); cyc = w[:, 0].astype(np.int64); tripcyc = w[:, 5].astype(np.int64); n = w[:, 1].astype(np.int64)
rank = idx // 16; blk = idx % 16
print("items", len(w), "total trips", int(trips.sum()), "max trips", int(trips.max()), "kernel ends at", en.max())
print("trips percentiles 50/75/90/99/100:", np.percentile(trips, [50, 75, 90, 99, 100]))
order = np.argsort(-trips)
print("TOP 24 by trips:  rank blk entries trips  start  end  cyc/trip")
for i in order[:24]:
    print(f"   {rank[i]:5d} {blk[i]:3d} {n[i]:7d} {trips[i]:5d} {st[i]:6.1f} {en[i]:6.1f} {tripcyc[i] / max(trips[i], 1):7.0f}")
order = np.argsort(-en)
starve = w[:, 2].astype(np.int64); tfirst = w[:, 3].astype(np.int64); rounds = (w[:, 6] & np.uint64(0xffff)).astype(np.int64)
print("LAST 24 to end:   tile blk entries trips  start  end  cyc/trip | item kcyc = trips + starve + rest ; first trip at kcyc ; rounds")
for i in order[:24]:
    print(f"   {rank[i]:5d} {blk[i]:3d} {n[i]:7d} {trips[i]:5d} {st[i]:6.1f} {en[i]:6.1f} {tripcyc[i] / max(trips[i], 1):7.0f} | "
          f"{cyc[i]/1e3:6.1f} = {tripcyc[i]/1e3:5.1f} + {starve[i]/1e3:5.1f} + {(cyc[i]-tripcyc[i]-starve[i])/1e3:5.1f} ; {tfirst[i]/1e3:5.1f} ; {rounds[i]}")
big = trips >= 60
print("items with >= 60 trips:", int(big.sum()), " mean share of the item's cycles: trips %.2f starve %.2f" % (float((tripcyc[big] / cyc[big]).mean()), float((starve[big] / cyc[big]).mean())))
print("all items: total kcyc %.0f = trips %.0f + starve %.0f" % (cyc.sum() / 1e3, tripcyc.sum() / 1e3, starve.sum() / 1e3))
# trips within a tile: how unequal are the 16 blocks?
by_rank = {}
for r, t in zip(rank, trips): by_rank.setdefault(int(r), []).append(int(t))
top = sorted(by_rank, key=lambda r: -max(by_rank[r]))[:10]
for r in top: print("tile rank", r, "block trips", sorted(by_rank[r], reverse=True))
# first items (started within 3 us) vs later ones
first = st < 3.0
print("first-wave items:", int(first.sum()), " their trips sum", int(trips[first].sum()), " later items", int((~first).sum()), "trips", int(trips[~first].sum()))
print("cyc/trip of first-wave items by start order quartile:", [round(float(np.mean(tripcyc[first][q] / np.maximum(trips[first][q], 1))), 0) for q in np.array_split(np.argsort(idx[first]), 4)])
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fwd_items.npy"),
        np.stack([rank, blk, n, trips, (st * 100).astype(np.int64), (en * 100).astype(np.int64), tripcyc, cyc, starve]))

"""Per-wave breakdown of the backward blend kernel (s_memtime stamps): segment phase vs block items, pop waits, imbalance."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import scenes, _lib
from tests import helpers as hp
dev = torch.device("cuda:0")
d = hp.inputs_of(scenes.config3(), "scale_rot")
L = _lib.lib()
L.moss_raster_debug_set_bwd_stamps.argtypes = [ctypes.c_void_p]
dc, dd, da = hp.image_grads(d.H, d.W)
nw = 256 * 4 * 6
buf = torch.zeros(nw * 16, dtype=torch.int64, device=dev)
t = hp.hip_forward(d, dev)
for _ in range(3): hp.hip_backward(d, t, dc, dd, da, dev)
torch.cuda.synchronize()
L.moss_raster_debug_set_bwd_stamps(buf.data_ptr())
hp.hip_backward(d, t, dc, dd, da, dev); torch.cuda.synchronize()
L.moss_raster_debug_set_bwd_stamps(None)
s = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()      # w[0..2] are s_memrealtime ticks (100 MHz, one clock for the whole device); the item costs are s_memtime cycles
us = lambda v: np.round((v - t0) / 100.0, 2)
print("waves", len(s), "| kernel span", us(s[:, 2].max()), "us | wave starts: first 0, median", us(np.median(s[:, 0])), "last", us(s[:, 0].max()))
print("segment phase ends: min/median/p90/max", us(s[:, 1].min()), us(np.median(s[:, 1])), us(np.percentile(s[:, 1], 90)), us(s[:, 1].max()))
print("wave ends: min/median/p90/max", us(s[:, 2].min()), us(np.median(s[:, 2])), us(np.percentile(s[:, 2], 90)), us(s[:, 2].max()))
hist, edges = np.histogram((s[:, 2] - t0) / 100.0, bins=12)
print("wave end histogram (us):", [(round(float(e), 1), int(h)) for e, h in zip(edges[:-1], hist)])
print("segments per wave: mean %.2f max %d total %d; invalid slots popped total %d" % (s[:, 3].mean(), s[:, 3].max(), s[:, 3].sum(), s[:, 6].sum()))
print("cycles per segment item: mean %.0f; pop wait per pop: %.0f; share of phase 1 spent in items %.2f, in pops %.2f" % (
    s[:, 4].sum() / max(s[:, 3].sum(), 1), s[:, 5].sum() / max((s[:, 3] + s[:, 6] + 8).sum(), 1),
    s[:, 4].sum() / (s[:, 1] - s[:, 0]).sum(), s[:, 5].sum() / (s[:, 1] - s[:, 0]).sum()))
print("tail items per wave: mean %.2f max %d; cycles per tail item %.0f; pop wait per tail pop %.0f" % (
    s[:, 7].mean(), s[:, 7].max(), s[:, 8].sum() / max(s[:, 7].sum(), 1), s[:, 9].sum() / max((s[:, 7] + 1).sum(), 1)))
for q in range(8):
    m = s[:, 10] == q
    if not m.any(): continue
    x = s[m]
    print(" xcd", q, "waves", int(m.sum()), "start median", us(np.median(x[:, 0])), "segments", int(x[:, 3].sum()), "tails", int(x[:, 7].sum()), "| segment phase end median/max", us(np.median(x[:, 1])), us(x[:, 1].max()),
          "| end median/max", us(np.median(x[:, 2])), us(x[:, 2].max()))
# implied clock of s_memtime against the 100 MHz real-time counter: waves with exactly one segment, phase 1 = pops + the item
one = s[s[:, 3] == 1]
if len(one):
    dur_us = (one[:, 1] - one[:, 0]) / 100.0
    cyc = one[:, 4] + one[:, 5]
    print("phase 1 of one-segment waves: median %.2f us, item+pop cycles median %.0f -> %.2f cycles per ns; item alone %.0f cycles, pop+table %.0f" % (
        np.median(dur_us), np.median(cyc), np.median(cyc / dur_us) / 1000.0, np.median(one[:, 4]), np.median(one[:, 5])))
tl = s[s[:, 7] > 0]
if len(tl):
    dur_us = (tl[:, 2] - tl[:, 1]) / 100.0
    cyc = tl[:, 8] + tl[:, 9]
    print("phase 2 of waves with tail items: median %.2f us for %.2f items, item+pop cycles median %.0f -> %.2f cycles per ns" % (
        np.median(dur_us), np.median(tl[:, 7]), np.median(cyc), np.median(cyc / dur_us) / 1000.0))

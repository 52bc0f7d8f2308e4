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
nw = 256 * 4 * 4
buf = torch.zeros(nw * 16, dtype=torch.int64, device=dev)
t = hp.hip_forward(d, dev)
for _ in range(3): hp.hip_backward(d, t, dc, dd, da, dev)
torch.cuda.synchronize()
L.moss_raster_debug_set_bwd_stamps(buf.data_ptr())
hp.hip_backward(d, t, dc, dd, da, dev); torch.cuda.synchronize()
L.moss_raster_debug_set_bwd_stamps(None)
s = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
print("waves", len(s), "kernel span (cycles of s_memtime)", s[:, 2].max() - t0, "start spread", s[:, 0].max() - t0)
print("phase1 end: min/mean/max", (s[:, 1] - t0).min(), (s[:, 1] - t0).mean(), (s[:, 1] - t0).max())
print("end: min/mean/max", (s[:, 2] - t0).min(), (s[:, 2] - t0).mean(), (s[:, 2] - t0).max())
print("segments per wave: mean %.2f max %d total %d; invalid slots popped total %d" % (s[:, 3].mean(), s[:, 3].max(), s[:, 3].sum(), s[:, 6].sum()))
print("cycles per segment item: mean %.0f; pop wait per pop: %.0f; share of phase 1 spent in items %.2f, in pops %.2f" % (
    s[:, 4].sum() / max(s[:, 3].sum(), 1), s[:, 5].sum() / max((s[:, 3] + s[:, 6] + 8).sum(), 1),
    s[:, 4].sum() / (s[:, 1] - s[:, 0]).sum(), s[:, 5].sum() / (s[:, 1] - s[:, 0]).sum()))
print("tail items per wave: mean %.2f max %d; cycles per tail item %.0f; pop wait per tail pop %.0f" % (
    s[:, 7].mean(), s[:, 7].max(), s[:, 8].sum() / max(s[:, 7].sum(), 1), s[:, 9].sum() / max((s[:, 7] + 1).sum(), 1)))
for q in range(8):
    m = s[:, 10] == q
    print(" xcd", q, "waves", int(m.sum()), "segments", int(s[m, 3].sum()), "phase1 end mean", int((s[m, 1] - t0).mean()), "end mean", int((s[m, 2] - t0).mean()))

"""Per-item picture of the forward blend's quads (diagnostic build: MOSS_AMD_LIB_DIR=lib_diag): when every heavy item started and ended
on the device clock, how its pieces were walked (owner exact / owner speculative / re-walked), what the owner waited for."""
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
cyc = w[:, 0].astype(np.int64); n = w[:, 1].astype(np.int64); idle = w[:, 2].astype(np.int64); rewalk = (w[:, 3] & np.uint64(0xffff)).astype(np.int64)
d_res = (w[:, 3] >> np.uint64(16)).astype(np.int64)
walk = w[:, 5].astype(np.int64); n_exact = np.zeros(len(w), np.int64); n_spec = (w[:, 6] & np.uint64(0xffff)).astype(np.int64)
pieces = w[:, 7].astype(np.int64) // 16
dur = en - st
print("items", len(w), "pieces combined", int(pieces.sum()), "walked by the owner", int(n_spec.sum()), "pixels resolved", int(rewalk.sum()), "resolve kcyc", int(d_res.sum() // 1000))
print("kernel: first start 0, last end %.1f us; starts pct 50/90/99/100: %s" % (en.max(), np.percentile(st, [50, 90, 99, 100]).round(1)))
print("item duration us pct 10/50/90/99/100:", np.percentile(dur, [10, 50, 90, 99, 100]).round(1))
print("pieces per item pct 50/90/99/100:", np.percentile(pieces, [50, 90, 99, 100]))
print("owner cycles: total %.0f k = walk %.0f k + idle %.0f k + rest %.0f k" % (cyc.sum() / 1e3, walk.sum() / 1e3, idle.sum() / 1e3, (cyc - walk - idle).sum() / 1e3))
order = np.argsort(-en)
print("LAST 20 to end: rank blk entries pieces exact spec rewalk start end | kcyc = walk + idle + rest")
for i in order[:20]:
    print(f"   {idx[i] // 16:5d} {idx[i] % 16:3d} {n[i]:7d} {pieces[i]:4d} {n_exact[i]:3d} {n_spec[i]:3d} {rewalk[i]:3d} {st[i]:6.1f} {en[i]:6.1f} | "
          f"{cyc[i] / 1e3:6.1f} = {walk[i] / 1e3:5.1f} + {idle[i] / 1e3:5.1f} + {(cyc[i] - walk[i] - idle[i]) / 1e3:5.1f}")
order = np.argsort(-dur)
print("LONGEST 12: rank blk entries pieces exact spec rewalk start end")
for i in order[:12]:
    print(f"   {idx[i] // 16:5d} {idx[i] % 16:3d} {n[i]:7d} {pieces[i]:4d} {n_exact[i]:3d} {n_spec[i]:3d} {rewalk[i]:3d} {st[i]:6.1f} {en[i]:6.1f}")
# how busy is the device over time: items in flight per microsecond
for tq in range(0, int(en.max()) + 1, 4):
    print(f"  t={tq:3d} us: items in flight {int(((st <= tq) & (en > tq)).sum())}")

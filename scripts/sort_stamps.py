"""Phase stamps of the two tile-sort kernels (thread 0 of every workgroup, s_memtime): lookup, key load, network / searches, gather, stores."""
import ctypes, os, sys
os.environ["MOSS_SORT_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import scenes, _lib
from tests import helpers as hp
dev = torch.device("cuda:0")
d = hp.inputs_of(scenes.config3(), "scale_rot")
L = _lib.lib()
L.moss_raster_debug_set_stamps.argtypes = [ctypes.c_void_p]
if "--async" in sys.argv:            # the capacity-bounded forward: keys bucketed by the preprocess kernel, the scan inside the sort kernel
    from moss_amd.diff_gaussian_rasterization import _C
    _C.set_async(True)
buf = torch.zeros(131072 + 32768 + 8 * 4096 + 64, dtype=torch.int64, device=dev)
for _ in range(3): hp.hip_forward(d, dev)
torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(buf.data_ptr())
hp.hip_forward(d, dev); torch.cuda.synchronize()
L.moss_raster_debug_set_stamps(None)
s = buf.cpu().numpy()[131072:]
for name, off in (("chunk_sort [lookup, key load, network, store]", 0), ("merge_gather [lookup, keys+searches, gather, emit]", 8 * 1024)):
    w = s[off: off + 8 * 1024].reshape(-1, 8).astype(np.float64)
    scan_row = w[0].copy() if (off == 0 and w[0, 0] == 0 and w[0, 5] > 0) else None     # (--async: workgroup 0 of the sort kernel is the scan block)
    w = w[(w[:, 0] > 0) & (w[:, 4] > 0)]
    # the clocks of the 8 XCDs are not synchronised: spans per XCD (workgroup index % 8)
    ph = np.diff(w[:, :5], axis=1)
    print(name, "workgroups", len(w), "mean cycles per phase", ph.mean(0).astype(int), "p95", np.percentile(ph, 95, axis=0).astype(int), "total mean", int((w[:, 4] - w[:, 0]).mean()), "max", int((w[:, 4] - w[:, 0]).max()))
    t0 = w[:, 5].min()
    print("   realtime (us): workgroup starts median %.2f last %.2f | ends median %.2f p90 %.2f last %.2f" % (
        np.median(w[:, 5] - t0) / 100, (w[:, 5].max() - t0) / 100, np.median(w[:, 7] - t0) / 100, np.percentile(w[:, 7] - t0, 90) / 100, (w[:, 7].max() - t0) / 100))
    if scan_row is not None:
        print("   the scan block (workgroup 0): start %.2f end %.2f" % ((scan_row[5] - t0) / 100, (scan_row[7] - t0) / 100))
    big = w[:, 6] >= (1024 if off == 0 else 3)
    if big.any():
        print("   full chunks / tiles of >= 3 chunks:", int(big.sum()), "mean phases", np.diff(w[big, :5], axis=1).mean(0).astype(int))

w = s[32768: 32768 + 8 * 4096].reshape(-1, 8).astype(np.float64)
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
print("preprocess_forward blocks", len(w), "(realtime us, relative to the first block's start)")
for i, name in ((0, "start"), (1, "loads issued + SH staged"), (2, "geometry done + stored"), (4, "histogram + slot runs"), (5, "flushed, end")):
    print("   %-28s median %6.2f  p90 %6.2f  max %6.2f" % (name, np.median(w[:, i] - t0) / 100, np.percentile(w[:, i] - t0, 90) / 100, (w[:, i].max() - t0) / 100))

w = s[16384: 16384 + 8 * 512].reshape(-1, 8).astype(np.float64)
w = w[w[:, 0] > 0]
t0 = w[:, 0].min() if len(w) else 0.0
scan = w[w[:, 5] > 0]; blk = w[w[:, 4] > 0]
print("scatter blocks", len(blk), "(realtime us, relative to the first block's start)" if len(blk) else "(none: the preprocess kernel wrote the keys)")
for i, name in [] if not len(blk) else ((0, "start"), (1, "own instances counted"), (2, "tile starts known"), (3, "runs reserved"), (4, "keys written, end")):
    print("   %-28s median %6.2f  p90 %6.2f  max %6.2f" % (name, np.median(blk[:, i] - t0) / 100, np.percentile(blk[:, i] - t0, 90) / 100, (blk[:, i].max() - t0) / 100))
if len(scan):
    print("   the scan block: start %.2f end %.2f" % ((scan[0, 0] - t0) / 100, (scan[0, 5] - t0) / 100))

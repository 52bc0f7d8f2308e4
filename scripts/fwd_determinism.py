"""Bitwise reproducibility of the forward blend: the same frame rendered N times (the quads' waves take pieces in whatever order their
timing gives -- the images must not depend on it).  python scripts/fwd_determinism.py [config] [mode] [runs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import scenes
from tests import helpers as hp
dev = torch.device("cuda:0")
cfg = getattr(scenes, sys.argv[1] if len(sys.argv) > 1 else "config3")()
mode = sys.argv[2] if len(sys.argv) > 2 else "scale_rot"
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
d = hp.inputs_of(cfg, mode)
ref = None
bad = 0
for i in range(runs):
    t = hp.hip_forward(d, dev)
    e = hp.hip_export(d, t, dev)
    cur = dict(color=e.color, depth=e.depth, alpha=e.alpha, final_T=e.final_T.reshape(d.H, d.W), n_contrib=e.n_contrib.reshape(d.H, d.W))
    if ref is None:
        ref = cur
        continue
    for k in ref:
        a, b = ref[k], cur[k]
        if not np.array_equal(a, b):
            bad += 1
            diff = (a != b)
            if diff.ndim == 3: diff = diff.any(0)
            ys, xs = np.nonzero(diff)
            print(f"run {i}: {k} differs on {len(ys)} pixels; max |d| {np.abs(a.astype(np.float64) - b.astype(np.float64)).max():.3e}; first at (y={ys[0]}, x={xs[0]}) "
                  f"tile {ys[0] // 16 * ((d.W + 15) // 16) + xs[0] // 16} block {(ys[0] % 16) // 4 * 4 + (xs[0] % 16) // 4}; blocks touched "
                  f"{len(set(zip((ys // 4).tolist(), (xs // 4).tolist())))}")
print("runs", runs, "mismatching tensors", bad)
sys.exit(1 if bad else 0)

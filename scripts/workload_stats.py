"""Tile-list statistics of a configuration's initial scene: list lengths, sort chunks, longest per-pixel lists."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moss_amd import scenes
from tests import helpers as hp
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
scene = {"cfg2": scenes.config2, "cfg3": scenes.config3, "cfg5": scenes.config5}[name]()
d = hp.inputs_of(scene, "scale_rot")
t = hp.hip_forward(d, dev)
e = hp.hip_export(d, t, dev)
n = (e.ranges[:, 1] - e.ranges[:, 0]).astype(np.int64)
print(name, "P", d.P, "R", t.R, "tiles", len(n), "non-empty", int((n > 0).sum()), "heavy(>=32)", int((n >= 32).sum()), "of them >=128:", int((n >= 128).sum()))
print("list length percentiles (non-empty) 10/50/90/99/100:", np.percentile(n[n > 0], [10, 50, 90, 99, 100]).astype(int))
for c in (1024, 2048, 4096, 8192):
    nch = (n + c - 1) // c
    print(" chunk", c, ": chunks", int(nch.sum()), "single-chunk tiles hold", round(float(n[nch == 1].sum()) / n.sum(), 3), "of instances; max chunks/tile", int(nch.max()))
nc = e.n_contrib.reshape(d.H, d.W).astype(np.int64)
print("n_contrib (last contributor position) max", nc.max(), " mean over covered pixels", round(float(nc[nc > 0].mean()), 1))
tt = e.tiles_touched.astype(np.int64)
print("tiles per visible Gaussian: mean", round(float(tt[tt > 0].mean()), 2), "max", tt.max(), " visible", int((tt > 0).sum()))

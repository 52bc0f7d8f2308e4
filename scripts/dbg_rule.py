"""Worst element of one tensor under the single rule, for one fuzz seed: python scripts/dbg_rule.py <seed> [tensor]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from tests import helpers as hp
from fuzz_scenes import random_scene
dev = torch.device("cuda:0")
seed = int(sys.argv[1]); n = sys.argv[2] if len(sys.argv) > 2 else "dL_dcolors"
s, mode, degree, colors = random_scene(seed)
d = hp.inputs_of(s, mode, degree=degree, colors=colors, bg=s.bg.tolist())
fw = hp.oracle_forward(d); fw64 = hp.oracle_forward64(d, fw)
m = hp.stable_mask(d, fw, fw64, thr=1e-4)
t = hp.hip_forward(d, dev)
dc, dd, da = hp.image_grads(d.H, d.W, zero_depth=bool(seed & 1)); dc, dd, da = dc * m, dd * m, da * m
g = hp.hip_backward(d, t, dc, dd, da, dev)
ref32 = hp.oracle_backward(d, fw, dc, dd, da)
mass = hp.oracle_gradient_scales(d, fw, dc, dd, da)
spread, ref64 = hp.reference_noise_floor(d, fw, fw64, dc, dd, da, seed=seed)
a = getattr(g, n).cpu().numpy().astype(np.float64); b = np.asarray(getattr(ref64, n), np.float64); o = np.asarray(getattr(ref32, n), np.float64)
den = spread[n] + hp.RULE_EPS * np.asarray(mass[n]).reshape(a.shape)
r = np.abs(a - b) / np.maximum(den, 1e-300)
k = np.unravel_index(np.argmax(r), r.shape); i = k[0]
print("scene P", s.P, mode, d.W, d.H, "deg", degree, "colors", colors)
print("worst", n, k, "ratio", r[k], "hip", a[k], "f64", b[k], "f32", o[k], "err", abs(a[k] - b[k]), "spread", spread[n][k], "eps*mass", hp.RULE_EPS * np.asarray(mass[n]).reshape(a.shape)[k])
print("gaussian", i, "opacity", float(d.opacities[i]), "radius", int(fw.radii[i]), "tiles", int(fw.tiles_touched[i]), "conic_opacity", fw.conic_opacity[i])
# where does it sit in its tiles' lists, and the final_T of the pixels around its centre
pl = fw.point_list; rg = fw.ranges
pos = np.nonzero(pl == i)[0]
for p_ in pos[:6]:
    tile = int(np.searchsorted(rg[:, 1], p_, side="right"))
    print("  tile", tile, "list length", int(rg[tile, 1] - rg[tile, 0]), "position", int(p_ - rg[tile, 0]))
cx, cy = fw.means2D[i]
print("centre", cx, cy, "final_T there", fw.final_T.reshape(d.H, d.W)[int(np.clip(round(cy), 0, d.H - 1)), int(np.clip(round(cx), 0, d.W - 1))])
# --- is it the forward state?  float64 backward on the HIP forward's (final_T, n_contrib)
e = hp.hip_export(d, t, dev)
import copy
f2 = copy.copy(fw64); f2.final_T = e.final_T.astype(np.float64); f2.n_contrib = e.n_contrib.copy()
ref64_hipstate = hp.oracle_backward(d, f2, dc, dd, da)
c = np.asarray(getattr(ref64_hipstate, n), np.float64)
print("f64 backward on the HIP forward state:", c[k], " |hip - that|", abs(a[k] - c[k]), " |f64 - that|", abs(b[k] - c[k]))
H, W = d.H, d.W
fT_h = e.final_T.reshape(H, W); fT_64 = np.asarray(fw64.final_T).reshape(H, W); nc_h = e.n_contrib.reshape(H, W); nc_64 = np.asarray(fw64.n_contrib).reshape(H, W)
x0, x1 = max(int(cx) - 14, 0), min(int(cx) + 15, W); y0, y1 = max(int(cy) - 14, 0), min(int(cy) + 15, H)
rel = np.abs(fT_h[y0:y1, x0:x1] - fT_64[y0:y1, x0:x1]) / np.maximum(fT_64[y0:y1, x0:x1], 1e-30)
mm = m.numpy().reshape(H, W)[y0:y1, x0:x1] > 0
print("final_T rel diff hip vs f64 around the Gaussian (stable pixels): max", rel[mm].max() if mm.any() else None, "median", np.median(rel[mm]) if mm.any() else None,
      " n_contrib equal on stable:", bool((nc_h[y0:y1, x0:x1][mm] == nc_64[y0:y1, x0:x1][mm]).all()), " stable share", mm.mean())
f32T = np.asarray(fw.final_T).reshape(H, W)
rel32 = np.abs(f32T[y0:y1, x0:x1] - fT_64[y0:y1, x0:x1]) / np.maximum(fT_64[y0:y1, x0:x1], 1e-30)
print("final_T rel diff f32 oracle vs f64: max", rel32[mm].max() if mm.any() else None)

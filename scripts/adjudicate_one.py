"""One fuzz seed against the float64 adjudicator, every gradient tensor: python scripts/adjudicate_one.py SEED [SEED ...]
(scaled = worst element in units of the Gaussian's contribution mass; relmax = worst element / largest value of the tensor)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
from tests import helpers as hp
from fuzz_scenes import random_scene

dev = torch.device("cuda:0")
names = ["dL_dmeans2D", "dL_dconic", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dtransforms"]
for seed in [int(a) for a in sys.argv[1:]]:
    s, mode, degree, colors = random_scene(seed)
    d = hp.inputs_of(s, mode, degree=degree, colors=colors, bg=s.bg.tolist())
    fw = hp.oracle_forward(d); fw64 = hp.oracle_forward64(d, fw)
    m = hp.stable_mask(d, fw, fw64, thr=1e-4)
    t = hp.hip_forward(d, dev)
    dc, dd, da = hp.image_grads(d.H, d.W, zero_depth=bool(seed & 1))
    dc, dd, da = dc * m, dd * m, da * m
    g = hp.hip_backward(d, t, dc, dd, da, dev)
    ref = hp.oracle_backward(d, fw, dc, dd, da); ref64 = hp.oracle_backward(d, fw64, dc, dd, da)
    sc = hp.oracle_gradient_scales(d, fw, dc, dd, da)
    print("seed", seed, "P", s.P, mode, "deg", degree, "%dx%d" % (d.W, d.H), "R", fw.num_rendered)
    for n in names:
        if getattr(g, n, None) is None or not hasattr(ref, n) or not getattr(ref, n).size:
            continue
        a = getattr(g, n).cpu().numpy().reshape(getattr(ref, n).shape); b32 = getattr(ref, n); b64 = getattr(ref64, n)
        line = "  %-14s relmax hip-64 %.2e orc-64 %.2e hip-orc %.2e" % (n, hp.rel_err(a, b64), hp.rel_err(b32, b64), hp.rel_err(a, b32))
        if n in sc:
            e = np.abs(a.astype(np.float64) - b64).reshape(a.shape[0], -1).max(1) if a.ndim > 1 else np.abs(a.astype(np.float64) - b64)
            worst = int(np.argmax(e / np.maximum(np.asarray(sc[n]).reshape(-1)[:len(e)] if np.asarray(sc[n]).size >= len(e) else 1.0, 1e-30))) if np.asarray(sc[n]).size >= len(e) else -1
            line += " | scaled hip-64 %.2e orc-64 %.2e (worst Gaussian %d)" % (hp.scaled_err(a, b64, sc[n])[0], hp.scaled_err(b32, b64, sc[n])[0], worst)
        print(line)

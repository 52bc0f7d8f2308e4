"""For the fuzz seeds flagged by scripts/fuzz_parity.py (tests/golden/fuzz_outlier_seeds.json): HIP vs float32 oracle vs float64
adjudicator, per gradient tensor, end to end on the stable pixels.  Writes gpurun_out/adjudication.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
from tests import helpers as hp
from fuzz_scenes import random_scene

dev = torch.device("cuda:0")
seeds = json.load(open(os.path.join(ROOT, "tests", "golden", "fuzz_outlier_seeds.json")))["seeds"]
names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dtransforms"]
out = {}
for seed in seeds:
    s, mode, degree, colors = random_scene(seed)
    d = hp.inputs_of(s, mode, degree=degree, colors=colors, bg=s.bg.tolist())
    fw = hp.oracle_forward(d)
    if fw.num_rendered == 0:
        continue
    fw64 = hp.oracle_forward64(d, fw)
    m = hp.stable_mask(d, fw, fw64, thr=1e-4)
    t = hp.hip_forward(d, dev)
    e = hp.hip_export(d, t, dev)
    dc, dd, da = hp.image_grads(d.H, d.W, zero_depth=bool(seed & 1))
    dc, dd, da = dc * m, dd * m, da * m
    g = hp.hip_backward(d, t, dc, dd, da, dev)
    ref = hp.oracle_backward(d, fw, dc, dd, da); ref64 = hp.oracle_backward(d, fw64, dc, dd, da)
    sc = hp.oracle_gradient_scales(d, fw, dc, dd, da)
    ok = m.numpy().astype(bool)
    rec = {"P": s.P, "mode": mode, "fragile": float(1 - m.mean()),
           "color_vs64 (hip, orc)": (float(np.abs(e.color - fw64.color)[:, ok].max()), float(np.abs(fw.color - fw64.color)[:, ok].max())),
           "ncontrib_mismatch_stable": int((e.n_contrib[ok.reshape(-1)] != fw.n_contrib[ok.reshape(-1)]).sum())}
    for n in names:
        if getattr(g, n, None) is None or not getattr(ref, n).size:
            continue
        a = getattr(g, n).cpu().numpy(); b32 = getattr(ref, n); b64 = getattr(ref64, n)
        rec[n] = {"scaled (hip-orc, hip-64, orc-64)": (hp.scaled_err(a, b32, sc[n])[0], hp.scaled_err(a, b64, sc[n])[0], hp.scaled_err(b32, b64, sc[n])[0]),
                  "relmax (hip-orc, hip-64, orc-64)": (hp.rel_err(a, b32), hp.rel_err(a, b64), hp.rel_err(b32, b64))}
    out[seed] = rec
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "adjudication.json"), "w"), indent=1)
print(len(out), "seeds adjudicated")

"""The single parity rule (tests/helpers.py: RULE_K, RULE_EPS, reference_noise_floor) over random scenes: for every gradient element
|hip - f64| <= RULE_K * (spread + RULE_EPS * mass).  Prints, per seed, the worst ratio |hip - f64| / (spread + RULE_EPS mass) and the
same ratio for the unperturbed float32 oracle (one of the samples `spread` is the maximum of: <= 1 by construction); at the end the
distribution and the pass rate.
Usage: python scripts/fuzz_rule.py [n_cases] [first_seed] | python scripts/fuzz_rule.py seeds <json list file>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from tests import helpers as hp
from fuzz_scenes import random_scene

dev = torch.device("cuda:0")
if len(sys.argv) > 2 and sys.argv[1] == "seeds":
    seeds = json.load(open(sys.argv[2]))
    seeds = seeds["seeds"] if isinstance(seeds, dict) else seeds
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    seeds = list(range(first, first + n))
NAMES = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dtransforms"]
rows, worst_hip, worst_orc = [], 0.0, 0.0
for i, seed in enumerate(seeds):
    if i % 100 == 0:
        print(f"[progress] {i} of {len(seeds)}; worst so far hip {worst_hip:.2f} oracle32 {worst_orc:.2f}", flush=True)
    s, mode, degree, colors = random_scene(seed)
    d = hp.inputs_of(s, mode, degree=degree, colors=colors, bg=s.bg.tolist())
    fw = hp.oracle_forward(d)
    if fw.num_rendered == 0:
        continue
    fw64 = hp.oracle_forward64(d, fw)
    m = hp.stable_mask(d, fw, fw64, thr=1e-4)
    t = hp.hip_forward(d, dev)
    dc, dd, da = hp.image_grads(d.H, d.W, zero_depth=bool(seed & 1))
    dc, dd, da = dc * m, dd * m, da * m
    g = hp.hip_backward(d, t, dc, dd, da, dev)
    ref32 = hp.oracle_backward(d, fw, dc, dd, da)
    mass = hp.oracle_gradient_scales(d, fw, dc, dd, da)
    cond, ref64 = hp.reference_noise_floor(d, fw, fw64, dc, dd, da, seed=seed)
    row = {"seed": seed}
    for n in NAMES:
        v = getattr(g, n, None)
        if v is None or not torch.is_tensor(v) or v.numel() == 0 or n not in mass:
            continue
        rh, kh = hp.single_rule_ratio(v.cpu().numpy(), getattr(ref64, n), mass[n], cond[n])
        ro, _ = hp.single_rule_ratio(getattr(ref32, n), getattr(ref64, n), mass[n], cond[n])
        row[n] = (round(rh, 3), round(ro, 3))
        worst_hip = max(worst_hip, rh); worst_orc = max(worst_orc, ro)
    rows.append(row)
    mx = max((v[0] for k, v in row.items() if k != "seed"), default=0.0)
    if mx > hp.RULE_K:
        print("OVER", row, flush=True)
allr = np.array([max((v[0] for k, v in r.items() if k != "seed"), default=0.0) for r in rows])
allo = np.array([max((v[1] for k, v in r.items() if k != "seed"), default=0.0) for r in rows])
print("scenes", len(rows), "rule K", hp.RULE_K, "eps", hp.RULE_EPS)
print("hip worst ratio per scene: pct 50/90/99/100", np.percentile(allr, [50, 90, 99, 100]).round(3), " pass", int((allr <= hp.RULE_K).sum()), "of", len(allr))
print("float32 oracle, same ratio: pct 50/90/99/100", np.percentile(allo, [50, 90, 99, 100]).round(3), " pass", int((allo <= hp.RULE_K).sum()), "of", len(allo))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"fuzz_rule_{seeds[0]}_{len(seeds)}.json")
json.dump({"K": hp.RULE_K, "eps": hp.RULE_EPS, "rows": rows}, open(out, "w"))

import json, os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from tests import helpers as hp
from fuzz_scenes import random_scene
dev = torch.device("cuda:0")
names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dtransforms"]
for seed in [int(x) for x in sys.argv[1:]]:
    s, mode, degree, colors = random_scene(seed)
    d = hp.inputs_of(s, mode, degree=degree, colors=colors, bg=s.bg.tolist())
    fw = hp.oracle_forward(d); fw64 = hp.oracle_forward64(d, fw)
    m = hp.stable_mask(d, fw, fw64, thr=1e-4)
    t = hp.hip_forward(d, dev)
    dc, dd, da = hp.image_grads(d.H, d.W, zero_depth=bool(seed & 1)); dc, dd, da = dc*m, dd*m, da*m
    g = hp.hip_backward(d, t, dc, dd, da, dev)
    ref = hp.oracle_backward(d, fw, dc, dd, da); ref64 = hp.oracle_backward(d, fw64, dc, dd, da); refa = hp.oracle_backward(d, fw, dc, dd, da, f32_accumulators=True)
    sc = hp.oracle_gradient_scales(d, fw, dc, dd, da)
    for n in names:
        if getattr(g, n, None) is None or not getattr(ref, n).size: continue
        a = getattr(g, n).cpu().numpy()
        ex, e32, eh = hp.adjudication_excess(a, (getattr(ref, n), getattr(refa, n)), getattr(ref64, n), sc[n], 2.0)
        rh = hp.rel_err(a, getattr(ref64, n)); ro = max(hp.rel_err(getattr(ref, n), getattr(ref64, n)), hp.rel_err(getattr(refa, n), getattr(ref64, n)))
        if ex > 2e-5 or rh > 2*ro + 2e-5 or True:
            # which gaussian
            s_ = np.asarray(sc[n]).reshape(a.shape); err = np.abs(a - getattr(ref64, n)) / np.where(s_ > 0, s_, 1)
            gi = np.unravel_index(np.argmax(err), err.shape)[0]
            print(seed, mode, n, "scaled hip %.1e refs %.1e | rel hip %.1e refs %.1e | worst gaussian %d tiles %d scale %s opa %.3f" % (eh, e32, rh, ro, gi, fw.tiles_touched[gi], np.round(s.scales[gi].numpy(), 4), float(s.opacities[gi])))

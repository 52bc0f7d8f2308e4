"""One fuzz seed's IMAGES against the float32 and the float64 oracle on the stable pixels: python scripts/dbg_image_seed.py SEED [...]
(a seed that misses scripts/fuzz_parity.py's 3e-4 image bar against the float32 oracle: is the kernel further from float64 than the
oracle's own float32 arithmetic is?  The adjudicated image rule of tests/test_gpu_parity_hardened.py for random scenes: <= 4 x + 2e-6)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
from tests import helpers as hp
from tests import test_gpu_parity as tp
from fuzz_scenes import random_scene

dev = torch.device("cuda:0")
for seed in [int(a) for a in sys.argv[1:]]:
    s, mode, degree, colors = random_scene(seed)
    d = hp.inputs_of(s, mode, degree=degree, colors=colors, bg=s.bg.tolist())
    fw = hp.oracle_forward(d); fw64 = hp.oracle_forward64(d, fw)
    t = hp.hip_forward(d, dev); e = hp.hip_export(d, t, dev)
    ok = (np.asarray(fw.margin) > tp.FRAGILE).reshape(d.H, d.W)
    ok64 = hp.stable_mask(d, fw, fw64, thr=1e-4).numpy().astype(bool)
    print("seed", seed, "P", s.P, mode, "deg", degree, "%dx%d" % (d.W, d.H), "R", fw.num_rendered, "stable", float(ok.mean()), "stable (both oracles, 1e-4)", float(ok64.mean()),
          "n_contrib equal on stable:", bool((e.n_contrib.reshape(d.H, d.W)[ok] == np.asarray(fw.n_contrib).reshape(d.H, d.W)[ok]).all()))
    for name, a, b32, b64 in (("color", e.color, fw.color, fw64.color), ("depth", e.depth, fw.depth, fw64.depth), ("alpha", e.alpha, fw.alpha, fw64.alpha)):
        for label, mk in (("margin>2e-5", ok), ("both oracles 1e-4", ok64)):
            sc = np.abs(b32).max() + 1e-30
            print("  %-6s %-18s |hip-orc32| %.2e  |hip-f64| %.2e  |orc32-f64| %.2e   (of the largest value)" % (
                name, label, np.abs(a[:, mk] - b32[:, mk]).max() / sc, np.abs(a[:, mk] - b64[:, mk]).max() / sc, np.abs(b32[:, mk] - b64[:, mk]).max() / sc))
    # the forward-only path gives the same bits
    t2 = hp.hip_forward(d, dev, debug=16)
    print("  forward-only image equal:", bool(torch.equal(t.color, t2.color) and torch.equal(t.alpha, t2.alpha)))

"""Where does run-to-run variation of the gradients come from on the bench frame (config3, segments active)?  (a) ONE forward state,
the backward repeated; (b) forward + backward repeated.  Compares every gradient tensor bitwise with the first run's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moss_amd import scenes
from tests import helpers as hp
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = hp.inputs_of(getattr(scenes, sys.argv[2] if len(sys.argv) > 2 else "config3")(), "scale_rot")
dc, dd, da = hp.image_grads(d.H, d.W)
NAMES = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")


def grads(t):
    g = hp.hip_backward(d, t, dc, dd, da, dev)
    return {k: getattr(g, k).clone() for k in NAMES}


def diff(a, b):
    out = []
    for k in NAMES:
        if not torch.equal(a[k], b[k]):
            m = a[k] != b[k]
            rows = m.reshape(m.shape[0], -1).any(1).nonzero().flatten()
            out.append(f"{k}: {int(m.sum())} elements in {len(rows)} Gaussians (first {rows[:6].tolist()}), max {float((a[k] - b[k]).abs().max()):.3e}")
    return out


t = hp.hip_forward(d, dev)
ref = grads(t)
bad = 0
for i in range(n):
    x = diff(ref, grads(t))
    if x:
        bad += 1
        print(f"(a) backward repeat {i}: " + " | ".join(x))
print(f"(a) same forward state, backward x{n}: {bad} runs differ")
bad = 0
for i in range(n):
    t2 = hp.hip_forward(d, dev)
    same_img = torch.equal(t.color, t2.color) and torch.equal(t.alpha, t2.alpha) and torch.equal(t.depth, t2.depth)
    x = diff(ref, grads(t2))
    if x or not same_img:
        bad += 1
        print(f"(b) forward+backward repeat {i}: images identical {same_img} " + " | ".join(x))
print(f"(b) forward + backward x{n}: {bad} runs differ")

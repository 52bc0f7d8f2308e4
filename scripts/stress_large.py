"""Scale sanity beyond BASELINE's largest config: 1M Gaussians at 2048x2048 (forward + backward twice: finite, deterministic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moss_amd import scenes
from tests import helpers as hp

dev = torch.device("cuda")
P = int(os.environ.get("P", 1_000_000)); S = int(os.environ.get("S", 2048))
sc = scenes.body_scene(P, S, S, 540.0 * S / 512, init_like=False, name="stress")
d = hp.inputs_of(sc, "scale_rot")
outs = []
for it in range(2):
    t0 = time.time()
    t = hp.hip_forward(d, dev)
    dc, dd, da = hp.image_grads(S, S, seed=5)
    g = hp.hip_backward(d, t, dc, dd, da, dev)
    torch.cuda.synchronize()
    print(f"run {it}: R = {t.R}, {time.time() - t0:.2f} s, peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
    outs.append((t.color.clone(), {k: v.clone() for k, v in vars(g).items() if v is not None and v.numel() > 0}))
assert torch.isfinite(outs[0][0]).all()
assert torch.equal(outs[0][0], outs[1][0])
for k in outs[0][1]:
    assert torch.isfinite(outs[0][1][k]).all(), k
    assert torch.equal(outs[0][1][k], outs[1][1][k]), k
print("finite and bit-identical across two runs; radii>0:", int((t.radii > 0).sum()))
# the forward-only path (MOSS_FORWARD_ONLY: 62 B of binning buffer per instance) at this size: the same images, bit for bit
del g, outs[1:]
torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
t2 = hp.hip_forward(d, dev, debug=16)
torch.cuda.synchronize()
assert t2.R == t.R and torch.equal(t2.color, outs[0][0]) and torch.equal(t2.alpha, t.alpha) and torch.equal(t2.depth, t.depth)
print(f"forward only: the same image bit for bit, binning buffer {t2.binning.numel() / max(t2.R, 1):.1f} B per instance (training forward: {t.binning.numel() / max(t.R, 1):.1f}), peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")

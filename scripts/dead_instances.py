"""How many (Gaussian, tile) instances of the reference's binning never reach alpha >= 1/255 at ANY pixel of their tile?
(Measurement for a possible exact cull before the sort; the reference duplicates a Gaussian into every tile of the bounding
square of its 3-sigma radius, auxiliary.h:46-61.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moss_amd import scenes
from tests import helpers

dev = torch.device("cuda")
for name, mk in (("cfg3", scenes.config3), ("cfg2", scenes.config2), ("cfg5", scenes.config5)):
    sc = mk()
    d = helpers.inputs_of(sc, mode="scale_rot")
    t = helpers.hip_forward(d, dev)
    e = helpers.hip_export(d, t, dev)
    R = t.R
    ids = torch.from_numpy(e.point_list.astype(np.int64)).to(dev)
    tile = torch.from_numpy((e.point_list_keys >> np.uint64(32)).astype(np.int64)).to(dev)
    m2 = torch.from_numpy(e.means2D).to(dev)[ids]; co = torch.from_numpy(e.conic_opacity).to(dev)[ids]
    gx = (d.W + 15) // 16
    ox = (tile % gx).float() * 16; oy = (tile // gx).float() * 16
    alive = torch.zeros(R, dtype=torch.bool, device=dev)
    blocks = torch.zeros(R, dtype=torch.int32, device=dev)
    for by in range(4):
        for bx in range(4):
            hit = torch.zeros(R, dtype=torch.bool, device=dev)
            for py in range(4):
                for px in range(4):
                    X = ox + bx * 4 + px; Y = oy + by * 4 + py
                    dx = m2[:, 0] - X; dy = m2[:, 1] - Y
                    power = -0.5 * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) - co[:, 1] * dx * dy
                    a = torch.clamp_max(co[:, 3] * torch.exp(power), 0.99)
                    hit |= (power <= 0) & (a >= 1.0 / 255.0) & (X < d.W) & (Y < d.H)
            alive |= hit; blocks += hit.int()
    print(name, "R", R, "alive", int(alive.sum()), "dead fraction %.3f" % (1 - alive.float().mean().item()),
          "mean live 4x4 blocks per live instance %.2f" % (blocks[alive].float().mean().item()))

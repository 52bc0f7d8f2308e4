"""Random scene generator of the parity sweeps (scripts/fuzz_parity.py, tests/test_gpu_parity_hardened.py): host-side, seeded,
no GPU.  1-4000 Gaussians, ragged image sizes, cameras, 300:1 anisotropy, image-covering Gaussians, opacities 0 / 1 / around
1/255, all three covariance input modes, SH degrees 0-3, colours instead of SHs, random backgrounds."""
import math
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moss_amd import scenes  # noqa: E402


def random_scene(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda lo, hi: lo + (hi - lo) * float(torch.rand(1, generator=g))
    P = int(10 ** r(0, 3.6))
    W, H = int(r(17, 300)), int(r(17, 300))
    s = SimpleNamespace(name=f"fuzz{seed}", P=P, sh_degree=3)
    spread = r(0.2, 1.5)
    s.means3D = (torch.rand(P, 3, generator=g) * 2 - 1) * spread
    s.means3D[:, 2] += r(-0.5, 2.0) * float(torch.rand(1, generator=g) < 0.3)            # some scenes straddle the near plane
    mode_scale = r(-4.5, -1.0)
    s.scales = torch.exp(mode_scale + r(0.1, 1.2) * torch.randn(P, 3, generator=g))
    if float(torch.rand(1, generator=g)) < 0.3:                                         # a few very large Gaussians
        k = max(1, P // 50)
        s.scales[:k] *= 30.0
    q = torch.randn(P, 4, generator=g); s.rotations = q / q.norm(dim=1, keepdim=True)
    s.opacities = torch.sigmoid(r(0.5, 4.0) * torch.randn(P, 1, generator=g) + r(-3, 3))
    ext = torch.rand(P, generator=g)
    s.opacities[ext < 0.03] = 0.0; s.opacities[ext > 0.97] = 1.0
    s.shs = 0.3 * torch.randn(P, 16, 3, generator=g); s.shs[:, 0, :] += (torch.rand(P, 3, generator=g) - 0.5) / 0.28209479177387814
    s.bg = torch.rand(3, generator=g) * float(torch.rand(1, generator=g) < 0.5)
    s.transforms = torch.randn(P, 3, 3, generator=g) * 0.1 + torch.eye(3)
    s.cov3D_precomp = scenes.covariance_precomp(s.scales, s.rotations, 1.0, s.transforms)
    f = r(0.5, 2.0) * max(W, H)
    ang = r(-0.4, 0.4)
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    s.camera = scenes.make_camera(W, H, f, f * r(0.8, 1.25), W / 2 + r(-20, 20), H / 2 + r(-20, 20), R, np.array([r(-0.3, 0.3), r(-0.3, 0.3), r(2.0, 4.0)]))
    mode = ["scale_rot", "precomp", "lbs"][int(torch.randint(0, 3, (1,), generator=g))]
    degree = int(torch.randint(0, 4, (1,), generator=g))
    colors = bool(torch.rand(1, generator=g) < 0.2)
    return s, mode, degree, colors

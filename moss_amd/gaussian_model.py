"""A minimal Gaussian parameter container exposing the property names the render binding reads from MOSS's
``GaussianModel`` (scene/gaussian_model.py:134-173).  It is NOT a rebuild of that class (densification, LBS, ply I/O
are out of scope): it exists so the step harness, the tests and bench.py can drive the
rasterizer exactly the way ``gaussian_renderer.render`` does.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import scenes


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianSet(nn.Module):
    def __init__(self, scene, sh_degree=3, device="cpu"):
        super().__init__()
        self.max_sh_degree = 3
        self.active_sh_degree = sh_degree
        self.motion_offset_flag = True
        dev = torch.device(device)
        self._xyz = nn.Parameter(scene.means3D.clone().to(dev))
        self._features_dc = nn.Parameter(scene.shs[:, :1, :].clone().to(dev))
        self._features_rest = nn.Parameter(scene.shs[:, 1:, :].clone().to(dev))
        self._scaling = nn.Parameter(torch.log(scene.scales).to(dev))
        self._rotation = nn.Parameter(scene.rotations.clone().to(dev))
        self._opacity = nn.Parameter(inverse_sigmoid(scene.opacities.clamp(1e-4, 1 - 1e-4)).to(dev))

    # activations as in scene/gaussian_model.py:46-56,134-166
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def activate(self, bucket=None):
        """All five getters above from ONE fused HIP kernel (moss_amd/activations.py); same values and gradients.  Returns a
        namespace with get_xyz / get_features / get_opacity / get_scaling / get_rotation / get_covariance, i.e. it can stand in
        for ``self`` wherever the render binding reads the activated parameters."""
        from .activations import activate_gaussians
        xyz, feat, opa, scl, rot = activate_gaussians(self._xyz, self._features_dc, self._features_rest, self._opacity,
                                                      self._scaling, self._rotation, bucket)
        raw_rotation = self._rotation
        return SimpleNamespace(
            get_xyz=xyz, get_features=feat, get_opacity=opa, get_scaling=scl, get_rotation=rot,
            get_covariance=lambda scaling_modifier=1, transform=None: scenes.covariance_precomp(scl, raw_rotation, scaling_modifier, transform),
            max_sh_degree=self.max_sh_degree, active_sh_degree=self.active_sh_degree)

    def get_covariance(self, scaling_modifier=1, transform=None):
        return scenes.covariance_precomp(self.get_scaling, self._rotation, scaling_modifier, transform)

    def param_groups(self):
        """AdamW groups with the reference's learning rates (arguments/__init__.py:66-79, gaussian_model.py:215-226)."""
        return [
            {"params": [self._xyz], "lr": 0.00016, "name": "xyz"},
            {"params": [self._features_dc], "lr": 0.0025, "name": "f_dc"},
            {"params": [self._features_rest], "lr": 0.0025 / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": 0.05, "name": "opacity"},
            {"params": [self._scaling], "lr": 0.005, "name": "scaling"},
            {"params": [self._rotation], "lr": 0.001, "name": "rotation"},
        ]

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
    def __init__(self, scene, sh_degree=3, device="cpu", unified_features=False, zero_inactive_sh=False):
        """unified_features: keep the SH coefficients as ONE (P,16,3) parameter ``_features`` -- ``_features_dc`` / ``_features_rest``
        are then views of it and ``get_features`` needs no torch.cat (47 MB of copies per step and direction at 100k Gaussians).
        The reference's two learning rates (f_dc, f_rest = f_dc / 20) become a periodic pattern of one optimizer segment
        (``param_groups()`` -> FlatAdamW); torch.optim cannot express that, so this mode needs moss_amd.optim.FlatAdamW."""
        super().__init__()
        self.unified_features = bool(unified_features)
        self.max_sh_degree = 3
        self.active_sh_degree = sh_degree
        if zero_inactive_sh:
            # MOSS's own state below the maximum degree: features_rest starts as zeros (create_from_pcd, scene/gaussian_model.py:179-181)
            # and a coefficient receives its first gradient when its degree becomes active (oneupSHdegree, :171-173)
            import copy
            scene = copy.copy(scene)
            scene.shs = scene.shs.clone()
            scene.shs[:, (int(sh_degree) + 1) ** 2:, :] = 0.0
        self.motion_offset_flag = True
        dev = torch.device(device)
        self._xyz = nn.Parameter(scene.means3D.clone().to(dev))
        if self.unified_features:
            self._features = nn.Parameter(scene.shs.clone().contiguous().to(dev))
        else:
            self._features_dc = nn.Parameter(scene.shs[:, :1, :].clone().to(dev))
            self._features_rest = nn.Parameter(scene.shs[:, 1:, :].clone().to(dev))
        self._scaling = nn.Parameter(torch.log(scene.scales).to(dev))
        self._rotation = nn.Parameter(scene.rotations.clone().to(dev))
        self._opacity = nn.Parameter(inverse_sigmoid(scene.opacities.clamp(1e-4, 1 - 1e-4)).to(dev))

    spatially_ordered = False

    def oneupSHdegree(self, optimizer=None):
        """scene/gaussian_model.py:171-173 (called every 1000 iterations, train_ZJU.py:85-86); a ``FlatAdamW`` is told the new active
        degree (its degree-aware SH update, ``set_active_sh_degree``).  A captured step must be captured again: the degree is a launch
        argument of the rasterizer and of the update."""
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
        if optimizer is not None and hasattr(optimizer, "set_active_sh_degree"):
            optimizer.set_active_sh_degree(self.active_sh_degree)

    def reorder_spatially(self, optimizer=None):
        """Re-index the Gaussians along a Morton curve of their positions (moss_amd.densify.spatial_order) -- parameters in place and,
        if given, the rows of a ``FlatAdamW``'s moments with them -- and remember that index neighbours are now spatial neighbours
        (``render`` passes the hint to the op).  Results do not depend on the index order; the memory traffic of the binning and of the
        per-Gaussian kernels does (profiles/r02_notes.md, finding 30).  Meant for the moments the set is rebuilt anyway (MOSS:
        densify / prune, scene/gaussian_model.py:densification_postfix); not capturable.  Returns the permutation: per-Gaussian state
        kept OUTSIDE this module and the optimizer (``DensifyStats`` accumulators, an LBS transform table) must be indexed with it too --
        MOSS resets its own accumulators at exactly these moments (densification_postfix).

        ``optimizer``: a ``FlatAdamW`` (its ``permute_rows`` moves parameters and moments together), or a ``torch.optim`` optimizer
        (the ``exp_avg`` / ``exp_avg_sq`` / ``max_exp_avg_sq`` rows of every permuted parameter are permuted with it); anything else
        raises.  ``optimizer=None`` is valid ONLY while no optimizer state exists for these parameters (before the first step, or
        when the caller rebuilds its optimizer afterwards): a Gaussian would otherwise continue with another Gaussian's moments."""
        from .densify import spatial_order
        perm = spatial_order(self._xyz.detach())
        rows = [p for p in self.parameters() if p.dim() >= 1 and p.shape[0] == perm.numel()]
        if optimizer is not None and hasattr(optimizer, "permute_rows"):
            optimizer.permute_rows(perm)                     # (the parameters live in its flat buffer: moved there)
        elif optimizer is None or isinstance(optimizer, torch.optim.Optimizer):
            with torch.no_grad():
                for p in rows:
                    p.copy_(p[perm].clone())
                    st = optimizer.state.get(p, {}) if optimizer is not None else {}
                    for k, v in st.items():                  # torch.optim.Adam(W): exp_avg, exp_avg_sq (, max_exp_avg_sq); `step` is a scalar
                        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == perm.numel():
                            v.copy_(v[perm.to(v.device)].clone())
        else:
            raise TypeError(f"reorder_spatially: cannot permute the state of a {type(optimizer).__name__}; pass a FlatAdamW, a "
                            "torch.optim.Optimizer, or None before any optimizer state exists")
        self.spatially_ordered = True
        return perm

    # ---- the three moments MOSS rebuilds its tensors AND its optimizer state (scene/gaussian_model.py:314-317, :396-411, :436-454), on a
    # FlatAdamW: the Parameter objects stay, their storage and the moments are re-laid-out by the optimizer.  Decision logic (which
    # Gaussians to clone / split / prune: densify_and_clone / _split / _prune, :456-620) is MOSS's and stays there.
    def _flat(self, optimizer):
        if not (hasattr(optimizer, "prune_rows") and hasattr(optimizer, "append_rows") and hasattr(optimizer, "reset_rows")):
            raise TypeError("GaussianSet surgery drives a moss_amd.optim.FlatAdamW; with a torch-state optimizer (torch.optim.AdamW, "
                            "moss_amd.optim.AdamW) MOSS's own cat_tensors_to_optimizer / _prune_optimizer / replace_tensor_to_optimizer apply")
        return optimizer

    def prune_points(self, mask, optimizer, stats=None):
        """``prune_points(mask)`` (scene/gaussian_model.py:396-411): rows where ``mask`` is True are REMOVED from every parameter, from
        both moments of each and from the densification statistics."""
        keep = ~mask.bool()
        self._flat(optimizer).prune_rows(keep)
        if stats is not None:
            stats.prune(keep)

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation, optimizer, stats=None):
        """``densification_postfix`` (scene/gaussian_model.py:436-454): the new Gaussians are appended to every parameter with zero
        moments, and the three statistics start again from zero at the new size."""
        opt = self._flat(optimizer)
        if self.unified_features:
            rows = {id(self._features): torch.cat((new_features_dc, new_features_rest), dim=1)}
        else:
            rows = {id(self._features_dc): new_features_dc, id(self._features_rest): new_features_rest}
        rows.update({id(self._xyz): new_xyz, id(self._opacity): new_opacities, id(self._scaling): new_scaling, id(self._rotation): new_rotation})
        index = {id(p): i for i, p in enumerate(opt.bucket.params)}
        opt.append_rows({index[k]: v for k, v in rows.items()})
        if stats is not None:
            stats.reset(self._xyz.shape[0])

    def reset_opacity(self, optimizer):
        """``reset_opacity`` (scene/gaussian_model.py:314-317): opacity = min(opacity, 0.01) in logits, both moments of it zeroed."""
        with torch.no_grad():
            new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self._opacity) * 0.01))
        self._flat(optimizer).reset_rows(self._opacity, new)

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

    def __getattr__(self, name):
        # unified mode: the reference's two tensors as views of the single parameter
        if name in ("_features_dc", "_features_rest"):
            params = self.__dict__.get("_parameters", {})
            if "_features" in params:
                f = params["_features"]
                return f[:, :1, :] if name == "_features_dc" else f[:, 1:, :]
        return super().__getattr__(name)

    @property
    def get_features(self):
        if self.unified_features:
            return self._features
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def activate(self, bucket=None):
        """All five getters above from ONE fused HIP kernel (moss_amd/activations.py); same values and gradients.  Returns a
        namespace with get_xyz / get_features / get_opacity / get_scaling / get_rotation / get_covariance, i.e. it can stand in
        for ``self`` wherever the render binding reads the activated parameters."""
        from .activations import activate_gaussians
        if self.unified_features:
            xyz, _, opa, scl, rot = activate_gaussians(self._xyz, None, None, self._opacity, self._scaling, self._rotation, bucket)
            feat = self._features                          # handed to the rasterizer as it is
        else:
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
        if self.unified_features:
            k3 = 3 * int(self._features.shape[1])
            feats = [{"params": [self._features], "lr": 0.0025, "lr_pattern": (k3, 3, 0.0025 / 20.0), "name": "features"}]
        else:
            feats = [{"params": [self._features_dc], "lr": 0.0025, "name": "f_dc"},
                     {"params": [self._features_rest], "lr": 0.0025 / 20.0, "name": "f_rest"}]
        return [
            {"params": [self._xyz], "lr": 0.00016, "name": "xyz"},
            *feats,
            {"params": [self._opacity], "lr": 0.05, "name": "opacity"},
            {"params": [self._scaling], "lr": 0.005, "name": "scaling"},
            {"params": [self._rotation], "lr": 0.001, "name": "rotation"},
        ]

"""Densification bookkeeping either side of the rasterizer.

Mirrors the three statistics MOSS keeps on ``GaussianModel`` and what it does with them every step
(train_ZJU.py:171-174, scene/gaussian_model.py:815-817), plus the KL test of its KL-guided densify
(scene/gaussian_model.py:586-598, :758-813).  It does NOT rebuild densify_and_clone / split / prune (out of scope: control
plane of the model).  HIP only (csrc/densify.hip through the C ABI); no CPU path.

Frame-parallel training: every rank accumulates the statistics of ITS views locally; ``sync()`` -- called once,
right before a densification decision -- sums ``xyz_gradient_accum`` and ``denom`` and takes the maximum of ``max_radii2D``
over the ranks, so every replica takes the same decision on the same numbers.
"""
from __future__ import annotations

import torch

from ._lib import check, lib

__all__ = ["DensifyStats", "densify_stats_update", "neighbour_kl", "cal_kl"]


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class DensifyStats:
    """``xyz_gradient_accum (P,1)``, ``denom (P,1)``, ``max_radii2D (P)`` as in GaussianModel.training_setup
    (scene/gaussian_model.py:204-205) / create_from_pcd (:198)."""

    def __init__(self, P: int, device="cuda"):
        self.xyz_gradient_accum = torch.zeros((P, 1), device=device)
        self.denom = torch.zeros((P, 1), device=device)
        self.max_radii2D = torch.zeros((P,), device=device)

    def add(self, radii: torch.Tensor, viewspace_grad: torch.Tensor) -> None:
        """One step's update: ``max_radii2D[vis] = max(.., radii[vis])`` and ``add_densification_stats(viewspace_points, vis)``
        with ``vis = radii > 0``.  ``viewspace_grad`` is ``viewspace_point_tensor.grad`` (P, >= 2 columns)."""
        P = self.denom.shape[0]
        if not radii.is_cuda or not viewspace_grad.is_cuda:
            raise RuntimeError("DensifyStats.add needs GPU tensors; this op has no CPU path")
        if radii.shape != (P,) or radii.dtype != torch.int32 or viewspace_grad.dim() != 2 or viewspace_grad.shape[0] != P \
                or viewspace_grad.shape[1] < 2 or viewspace_grad.dtype != torch.float32:
            raise RuntimeError("DensifyStats.add: expected radii (P) int32 and viewspace_grad (P, >=2) float32")
        if viewspace_grad.stride(1) != 1:
            viewspace_grad = viewspace_grad.contiguous()
        with torch.cuda.device(radii.device):
            check(lib().moss_densify_stats(P, radii.contiguous().data_ptr(), viewspace_grad.data_ptr(), viewspace_grad.stride(0),
                                           self.xyz_gradient_accum.data_ptr(), self.denom.data_ptr(), self.max_radii2D.data_ptr(),
                                           _stream(radii.device)), "densify_stats")

    def sync(self, group=None) -> None:
        """Combine the ranks' locally accumulated statistics (sum, sum, max).  No-op without an initialised process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        sums = torch.cat([self.xyz_gradient_accum.view(-1), self.denom.view(-1)])
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        P = self.denom.shape[0]
        self.xyz_gradient_accum.copy_(sums[:P].view(P, 1))
        self.denom.copy_(sums[P:].view(P, 1))
        dist.all_reduce(self.max_radii2D, op=dist.ReduceOp.MAX, group=group)

    def mean_grads(self) -> torch.Tensor:
        """``grads = xyz_gradient_accum / denom; grads[grads.isnan()] = 0`` (scene/gaussian_model.py:721-722)."""
        g = self.xyz_gradient_accum / self.denom
        g[g.isnan()] = 0.0
        return g

    def reset(self, P: int = None) -> None:
        """All three back to zero -- for ``P`` Gaussians if given: ``densification_postfix`` re-creates them at the new size
        (scene/gaussian_model.py:451-454)."""
        if P is not None and int(P) != self.denom.shape[0]:
            dev = self.denom.device
            self.xyz_gradient_accum = torch.zeros((int(P), 1), device=dev)
            self.denom = torch.zeros((int(P), 1), device=dev)
            self.max_radii2D = torch.zeros((int(P),), device=dev)
        else:
            self.xyz_gradient_accum.zero_(); self.denom.zero_(); self.max_radii2D.zero_()

    def prune(self, keep_mask: torch.Tensor) -> None:
        """``prune_points`` keeps the surviving rows of the three statistics (scene/gaussian_model.py:406-410)."""
        keep = keep_mask.to(self.denom.device).bool()
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep].contiguous()
        self.denom = self.denom[keep].contiguous()
        self.max_radii2D = self.max_radii2D[keep].contiguous()


def densify_stats_update(max_radii2D: torch.Tensor, xyz_gradient_accum: torch.Tensor, denom: torch.Tensor,
                         radii: torch.Tensor, viewspace_grad: torch.Tensor) -> None:
    """The per-step bookkeeping of train_ZJU.py:171-174 on the CALLER's tensors (MOSS's ``gaussians.max_radii2D (P)``,
    ``gaussians.xyz_gradient_accum (P,1)``, ``gaussians.denom (P,1)``), in place, in one launch and without the mask -> index host
    synchronisation: what ``patches/train_ZJU.diff`` puts in the place of those two lines."""
    st = DensifyStats.__new__(DensifyStats)
    st.max_radii2D, st.xyz_gradient_accum, st.denom = max_radii2D, xyz_gradient_accum, denom
    for t in (max_radii2D, xyz_gradient_accum, denom):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("densify_stats_update: the statistics must be contiguous float32 GPU tensors")
    st.add(radii, viewspace_grad)


def neighbour_kl(xyz: torch.Tensor, rotation: torch.Tensor, scaling: torch.Tensor, pair_idx: torch.Tensor) -> torch.Tensor:
    """``kl_div`` (scene/gaussian_model.py:773-813) of Gaussian ``pair_idx[:,0]`` against Gaussian ``pair_idx[:,1]``, gather
    fused in.  ``rotation``: raw quaternions; ``scaling``: activated scales.  Returns (P,) float32."""
    if not xyz.is_cuda:
        raise RuntimeError("neighbour_kl needs GPU tensors; this op has no CPU path")
    N = xyz.shape[0]
    if xyz.shape != (N, 3) or rotation.shape != (N, 4) or scaling.shape != (N, 3) or pair_idx.dim() != 2 or pair_idx.shape[1] != 2 \
            or pair_idx.dtype != torch.int64:
        raise RuntimeError("neighbour_kl: expected xyz (N,3), rotation (N,4), scaling (N,3), pair_idx (P,2) int64")
    P = pair_idx.shape[0]
    out = torch.empty((P,), dtype=torch.float32, device=xyz.device)
    x, r, s = (t.detach().float().contiguous() for t in (xyz, rotation, scaling))
    with torch.cuda.device(xyz.device):
        check(lib().moss_neighbour_kl(P, N, x.data_ptr(), r.data_ptr(), s.data_ptr(), pair_idx.contiguous().data_ptr(),
                                      out.data_ptr(), _stream(xyz.device)), "neighbour_kl")
    return out


def cal_kl(xyz: torch.Tensor, rotation: torch.Tensor, scaling: torch.Tensor, knn_impl: str = None):
    """The KL of every Gaussian against its nearest other Gaussian: the k = 2 self-query (``knn_near_2``) followed by ``kl_div``,
    as in GaussianModel.cal_kl (scene/gaussian_model.py:758-771) and densify_and_clone/split (:586-598).  Returns
    ``(kl (P,), point_ids (P,2))``; the caller compares with its threshold (``> kl_threshold`` in cal_kl, ``<`` at :598)."""
    from .knn_cuda import knn
    _, ids = knn(xyz.detach()[None], xyz.detach()[None], 2, knn_impl)
    return neighbour_kl(xyz, rotation, scaling, ids[0]), ids[0]


def spatial_order(xyz: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """A permutation that lists the Gaussians along a 3-D Morton (Z-order) curve of their positions (`bits` per axis).

    Nothing in the rasterizer depends on the index order of the Gaussians (ties in depth are broken by the index, as in the
    reference's radix sort of (tile | depth) keys, rasterizer_impl.cu:330-337), but its memory traffic does: a block of 256
    consecutive Gaussians adds to the histogram of every tile it touches (preprocess.hip), reserves a run in each of those tiles
    (binning.hip: scatter) and its instances' records are gathered by Gaussian index (merge_gather, per-Gaussian backward).  MOSS
    starts from the SMPL vertices in mesh order and appends clones and splits next to nothing in particular
    (scene/gaussian_model.py: densification_postfix); re-indexing the set along a space-filling curve whenever it is rebuilt anyway
    (densify / prune, every few hundred iterations) keeps index neighbours spatial neighbours.  Apply the returned permutation to
    every per-Gaussian tensor AND to the optimizer state (`FlatAdamW.permute_rows`; `GaussianSet.reorder_spatially` does both)."""
    with torch.no_grad():
        p = xyz.detach().float()
        lo = p.min(dim=0).values
        span = (p.max(dim=0).values - lo).clamp_min(1e-12)
        q = ((p - lo) / span * (2 ** bits - 1)).round().to(torch.int64).clamp_(0, 2 ** bits - 1)
        code = torch.zeros(p.shape[0], dtype=torch.int64, device=p.device)
        for b in range(bits):
            for a in range(3):
                code |= ((q[:, a] >> b) & 1) << (3 * b + a)
        return torch.argsort(code, stable=True)

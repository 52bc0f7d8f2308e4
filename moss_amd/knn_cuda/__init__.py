"""Drop-in for the third-party ``knn_cuda`` package as MOSS uses it.

``from knn_cuda import KNN`` (scene/gaussian_model.py:28); ``KNN(k=1, transpose_mode=True)`` / ``KNN(k=2, transpose_mode=True)``
(:85-86) called as ``dist, idx = knn(ref, query)`` with ``ref (B, Nr, 3)`` and ``query (B, Nq, 3)`` (:586,657,759,827).
Returns ``dist (B, Nq, k)`` float32 Euclidean distances (ascending) and ``idx (B, Nq, k)`` int64 indices into ``ref``.
``transpose_mode=False`` takes / returns the dimension-major layout ``(B, 3, N)`` / ``(B, k, Nq)`` like the original.
Exact, on the GPU, no CPU path (csrc/knn_query.hip): brute force (C ABI ``moss_knn_query``) for small reference sets, a uniform
cell grid over the references (``moss_knn_grid_build`` / ``moss_knn_grid_query``) from ``GRID_MIN_REF`` references up -- both
return identical results, ties included.  ``KnnGrid`` keeps a built grid for reference sets that do not change between calls
(MOSS's template vertices, scene/gaussian_model.py:827).  The original wheel is a binary that is not part of the reference
repository, so this replacement is "parity unpinned": its tests compare with an exhaustive search.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .._lib import check, lib

__all__ = ["KNN", "KnnGrid", "knn"]

GRID_MIN_REF = 2048        # below this the brute-force kernel (one launch) is as fast as building a grid (8 launches)


def _impl(Nr: int, impl) -> str:
    impl = impl or os.environ.get("MOSS_KNN_IMPL", "auto")
    if impl == "auto":
        return "grid" if Nr >= GRID_MIN_REF else "brute"
    if impl not in ("grid", "brute"):
        raise RuntimeError("knn: impl must be 'auto', 'grid' or 'brute'")
    return impl


class KnnGrid:
    """A cell grid built once over ``ref (Nr, 3)``; ``query(points (Nq, 3), k)`` any number of times.  The references are
    snapshotted into the grid: later changes of ``ref`` are not seen."""

    def __init__(self, ref: torch.Tensor):
        if ref.dim() != 2 or ref.shape[1] != 3 or ref.shape[0] < 1:
            raise RuntimeError("KnnGrid: expected ref (Nr, 3) with Nr >= 1")
        if not ref.is_cuda:
            raise RuntimeError("knn needs GPU tensors; this op has no CPU path")
        self.Nr = int(ref.shape[0])
        self.device = ref.device
        ref_c = ref.detach().float().contiguous()
        self._bytes = int(lib().moss_knn_grid_workspace_bytes(self.Nr))
        self._ws = torch.empty(self._bytes, dtype=torch.uint8, device=ref.device)
        with torch.cuda.device(ref.device):
            check(lib().moss_knn_grid_build(self.Nr, ref_c.data_ptr(), self._ws.data_ptr(), self._bytes,
                                            torch.cuda.current_stream(ref.device).cuda_stream), "knn_grid_build")

    def query(self, points: torch.Tensor, k: int, dist: torch.Tensor = None, idx: torch.Tensor = None):
        if points.dim() != 2 or points.shape[1] != 3 or points.device != self.device:
            raise RuntimeError("KnnGrid.query: expected points (Nq, 3) on the grid's device")
        if not 1 <= k <= 4:
            raise RuntimeError("knn: k must be 1..4")
        if self.Nr < k:
            raise RuntimeError("knn: fewer reference points than k")
        Nq = int(points.shape[0])
        pts = points.detach().float().contiguous()
        if dist is None:
            dist = torch.empty((Nq, k), dtype=torch.float32, device=self.device)
        if idx is None:
            idx = torch.empty((Nq, k), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().moss_knn_grid_query(self.Nr, Nq, k, self._ws.data_ptr(), self._bytes, pts.data_ptr(), dist.data_ptr(),
                                            idx.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream), "knn_grid_query")
        return dist, idx


def knn(ref: torch.Tensor, query: torch.Tensor, k: int, impl: str = None):
    """ref (B, Nr, 3), query (B, Nq, 3) -> (dist (B, Nq, k), idx (B, Nq, k) int64)."""
    if ref.dim() != 3 or query.dim() != 3 or ref.shape[0] != query.shape[0] or ref.shape[2] != 3 or query.shape[2] != 3:
        raise RuntimeError("knn: expected ref (B, Nr, 3) and query (B, Nq, 3)")
    if not ref.is_cuda or not query.is_cuda:
        raise RuntimeError("knn needs GPU tensors; this op has no CPU path")
    if not 1 <= k <= 4:
        raise RuntimeError("knn: k must be 1..4")
    B, Nr, Nq = ref.shape[0], ref.shape[1], query.shape[1]
    if Nr < k:
        raise RuntimeError("knn: fewer reference points than k")
    ref_c, query_c = ref.detach().float().contiguous(), query.detach().float().contiguous()
    dist = torch.empty((B, Nq, k), dtype=torch.float32, device=ref.device)
    idx = torch.empty((B, Nq, k), dtype=torch.int64, device=ref.device)
    with torch.cuda.device(ref.device):
        stream = torch.cuda.current_stream(ref.device).cuda_stream
        for b in range(B):
            if _impl(Nr, impl) == "grid":
                KnnGrid(ref_c[b]).query(query_c[b], k, dist[b], idx[b])
                continue
            check(lib().moss_knn_query(Nr, Nq, k, ref_c[b].data_ptr(), query_c[b].data_ptr(), dist[b].data_ptr(), idx[b].data_ptr(),
                                       stream), "knn_query")
    return dist, idx


class KNN(nn.Module):
    def __init__(self, k, transpose_mode=False, impl=None):
        super().__init__()
        self.k = int(k)
        self._t = bool(transpose_mode)
        self._impl = impl                                 # None / "auto": by reference count; "grid"; "brute"

    def forward(self, ref, query):
        if not self._t:                                   # dimension-major in and out
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        d, i = knn(ref, query, self.k, self._impl)
        if not self._t:
            d, i = d.transpose(1, 2).contiguous(), i.transpose(1, 2).contiguous()
        return d, i

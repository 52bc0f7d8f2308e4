"""Drop-in for the third-party ``knn_cuda`` package as MOSS uses it.

``from knn_cuda import KNN`` (scene/gaussian_model.py:28); ``KNN(k=1, transpose_mode=True)`` / ``KNN(k=2, transpose_mode=True)``
(:85-86) called as ``dist, idx = knn(ref, query)`` with ``ref (B, Nr, 3)`` and ``query (B, Nq, 3)`` (:586,657,759,827).
Returns ``dist (B, Nq, k)`` float32 Euclidean distances (ascending) and ``idx (B, Nq, k)`` int64 indices into ``ref``.
``transpose_mode=False`` takes / returns the dimension-major layout ``(B, 3, N)`` / ``(B, k, Nq)`` like the original.
Exact brute force on the GPU (C ABI ``moss_knn_query``, csrc/knn_query.hip); no CPU path.  The original wheel is a binary that
is not part of the reference repository, so this replacement is "parity unpinned": its tests compare with an exhaustive search.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .._lib import check, lib

__all__ = ["KNN", "knn"]


def knn(ref: torch.Tensor, query: torch.Tensor, k: int):
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
            check(lib().moss_knn_query(Nr, Nq, k, ref_c[b].data_ptr(), query_c[b].data_ptr(), dist[b].data_ptr(), idx[b].data_ptr(),
                                       stream), "knn_query")
    return dist, idx


class KNN(nn.Module):
    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = int(k)
        self._t = bool(transpose_mode)

    def forward(self, ref, query):
        if not self._t:                                   # dimension-major in and out
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        d, i = knn(ref, query, self.k)
        if not self._t:
            d, i = d.transpose(1, 2).contiguous(), i.transpose(1, 2).contiguous()
        return d, i

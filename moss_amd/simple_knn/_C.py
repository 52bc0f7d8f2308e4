"""Drop-in for the reference's compiled ``simple_knn._C`` (submodules/simple-knn/ext.cpp:15-16, spatial.cu:16-25)."""
from __future__ import annotations

import torch

from .._lib import check, lib

lib()   # fail at import time if the HIP library is missing: there is no fallback


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """(P,3) float32 device tensor -> (P,) float32: mean squared distance of every point to its 3 nearest other points."""
    if not points.is_cuda:
        raise RuntimeError("points must live on the GPU; this op has no CPU path")
    if points.dtype != torch.float32:
        raise RuntimeError(f"points: expected torch.float32, got {points.dtype}")
    L = lib()
    P = int(points.size(0))
    means = torch.full((P,), 0.0, dtype=torch.float32, device=points.device)      # spatial.cu:21-22
    if P == 0:
        return means
    pts = points.contiguous()
    nbytes = int(L.moss_knn_workspace_bytes(P))
    workspace = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)
    with torch.cuda.device(points.device):
        rc = L.moss_knn_dist2(P, pts.data_ptr(), means.data_ptr(), workspace.data_ptr(), nbytes,
                              torch.cuda.current_stream(points.device).cuda_stream)
    check(rc, "distCUDA2")
    return means

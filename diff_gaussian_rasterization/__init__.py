"""Repo-root alias so MOSS's own import line works unchanged:
``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
(gaussian_renderer/__init__.py:16).  Everything lives in ``moss_amd.diff_gaussian_rasterization``."""
from moss_amd.diff_gaussian_rasterization import (  # noqa: F401
    GaussianRasterizationSettings, GaussianRasterizer, _RasterizeGaussians, rasterize_gaussians, cpu_deep_copy_tuple, _C)

"""Repo-root alias so ``from simple_knn._C import distCUDA2`` (scene/gaussian_model.py:25) works unchanged."""
from moss_amd.simple_knn import _C  # noqa: F401

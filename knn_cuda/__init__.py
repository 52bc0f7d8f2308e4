"""Root alias so that MOSS's ``from knn_cuda import KNN`` (scene/gaussian_model.py:28) resolves to the MI355X implementation."""
from moss_amd.knn_cuda import KNN, KnnGrid, knn  # noqa: F401

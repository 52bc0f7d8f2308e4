"""Drop-in for the reference's ``simple_knn`` package (submodules/simple-knn); MOSS uses ``simple_knn._C.distCUDA2``
(scene/gaussian_model.py:25,185)."""
from . import _C  # noqa: F401

"""setup.py -- builds the two shared objects of the op before the Python packages are collected (see pyproject.toml).

The reference builds its extension with torch.utils.cpp_extension.CUDAExtension (DGR/setup.py:21-29).  Here the kernels are a plain
C-ABI library compiled by hipcc for gfx950 and the torch glue a host-only extension over it; both are produced by
``moss_amd.build.build()`` (in-tree, the same files ``__graft_entry__.build()`` makes) and shipped as package data, so
``pip install [-e] .`` leaves ``import diff_gaussian_rasterization``, ``import simple_knn._C``, ``import knn_cuda`` working without
PYTHONPATH.
"""
import os
import sys

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

ROOT = os.path.dirname(os.path.abspath(__file__))


def _build_native():
    sys.path.insert(0, ROOT)
    try:
        from moss_amd import build as hip_build
        hip_build.build()
        # the installed package carries the C-ABI header next to the kernel sources: it can rebuild itself (moss_amd/build.py INCLUDE_DIR)
        import shutil
        os.makedirs(os.path.join(ROOT, "moss_amd", "include"), exist_ok=True)
        shutil.copy(os.path.join(ROOT, "include", "moss_raster.h"), os.path.join(ROOT, "moss_amd", "include", "moss_raster.h"))
    finally:
        sys.path.pop(0)


class BuildPyWithNative(build_py):
    def run(self):
        _build_native()
        super().run()


class DevelopWithNative(develop):
    def run(self):
        _build_native()
        super().run()


setup(
    name="moss-amd",
    version="0.4.0",
    description="MI355X-native (gfx950) differentiable Gaussian-splatting rasterizer: drop-in for MOSS's diff_gaussian_rasterization, "
                "simple_knn and knn_cuda",
    python_requires=">=3.10",
    packages=find_packages(include=["moss_amd*", "diff_gaussian_rasterization*", "simple_knn*", "knn_cuda*"]),
    package_data={"moss_amd": ["lib/libmoss_raster.so", "lib/_moss_C.so", "csrc/*", "include/*.h"]},
    cmdclass={"build_py": BuildPyWithNative, "develop": DevelopWithNative},
)

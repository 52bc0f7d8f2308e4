"""Builds moss_amd/lib/libmoss_raster.so (the C-ABI library of include/moss_raster.h) with hipcc for gfx950.

In-tree build: the .so is git-ignored but travels to the GPU box with the repo snapshot.
``python -m moss_amd.build [--force] [--diag]``

``--diag`` builds the DIAGNOSTIC variant into ``moss_amd/lib_diag/`` as well: the same sources with ``-DMOSS_DIAG`` plus
``scripts/diag/knobs.cpp`` -- MOSS_* environment knobs that select kernel variants for A/B timing (some give wrong results on
purpose) and the stamp buffers of the timeline scripts.  The product build in ``moss_amd/lib/`` has neither and reads no environment
variable; ``MOSS_AMD_LIB_DIR=lib_diag`` makes ``moss_amd._lib`` load the diagnostic pair instead (scripts/ only).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libmoss_raster.so")
EXT = os.path.join(OUT_DIR, "_moss_C.so")                # PyTorch-ROCm extension module over the C ABI (csrc/torch_binding.cpp)
# The C-ABI header: include/moss_raster.h of the source tree; an INSTALLED copy of the package carries its own (setup.py copies it to
# moss_amd/include/), so that `python -m moss_amd.build --force` -- what the loader's error messages recommend -- works there too.
INCLUDE_DIR = os.path.join(ROOT, "include") if os.path.exists(os.path.join(ROOT, "include", "moss_raster.h")) else os.path.join(HERE, "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", INCLUDE_DIR, "-I", CSRC,
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]
# translation unit -> extra flags.  The per-Gaussian kernels decide integers (radius, tile rectangle, sort key) and must
# round exactly like the CPU oracle: no FMA contraction there.  The blend kernels spell their FMAs explicitly.
SOURCES = {
    "preprocess.hip": ["-ffp-contract=off"],
    "knn.hip": ["-ffp-contract=off"],
    "knn_query.hip": ["-ffp-contract=off"],
    "binning.hip": [],
    "blend.hip": [],
    "loss.hip": [],
    "optim.hip": [],
    "activations.hip": [],
    "densify.hip": [],
    "raster_api.hip": [],
}


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


DIAG_DIR = os.path.join(HERE, "lib_diag")
DIAG_EXTRA_SOURCE = os.path.join(ROOT, "scripts", "diag", "knobs.cpp")


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """Build the product pair (lib/libmoss_raster.so, lib/_moss_C.so); with ``diag`` ALSO the diagnostic pair in lib_diag/."""
    lib = _build_into(OUT_DIR, [], force, verbose)
    if diag:
        _build_into(DIAG_DIR, ["-DMOSS_DIAG"], force, verbose)
    return lib


def _build_into(out_dir: str, defines, force: bool, verbose: bool) -> str:
    os.makedirs(out_dir, exist_ok=True)
    lib_path = os.path.join(out_dir, "libmoss_raster.so")
    # (every header of csrc/: common.h, adamw.h, ... -- a header left out of this list would not rebuild the objects that include it)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(INCLUDE_DIR, "moss_raster.h"), os.path.abspath(__file__)]
    objs = []
    procs = []
    sources = [(os.path.join(CSRC, src), extra) for src, extra in SOURCES.items()]
    if defines:
        if not os.path.exists(DIAG_EXTRA_SOURCE):
            raise RuntimeError("the diagnostic build (--diag) needs scripts/diag/knobs.cpp of the SOURCE tree; an installed package "
                               "carries the product build only")
        sources.append((DIAG_EXTRA_SOURCE, ["-x", "hip"]))
    for s, extra in sources:
        src = os.path.basename(s)
        o = os.path.join(out_dir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            cmd = [HIPCC] + COMMON + list(defines) + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out):
            sys.stderr.write(f"--- hipcc {src} ---\n{out.decode(errors='replace')}\n")
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _newer(lib_path, objs):
        # link under a temporary name and rename: a process that finds the library never sees a half-written file
        tmp = lib_path + f".tmp{os.getpid()}"
        subprocess.check_call([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp] + objs)
        os.replace(tmp, lib_path)
    build_torch_extension(force=force, verbose=verbose, out_dir=out_dir)
    return lib_path


def build_torch_extension(force: bool = False, verbose: bool = False, out_dir: str = OUT_DIR) -> str:
    """<out_dir>/_moss_C.so: host-only C++ (g++), the torch glue of the reference's rasterize_points.cu over the C ABI."""
    src = os.path.join(CSRC, "torch_binding.cpp")
    EXT = os.path.join(out_dir, "_moss_C.so")
    if not (force or _newer(EXT, [src, os.path.join(INCLUDE_DIR, "moss_raster.h"), os.path.abspath(__file__)])):
        return EXT
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include"),
                                INCLUDE_DIR]
    tmp = EXT + f".tmp{os.getpid()}"
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-DTORCH_EXTENSION_NAME=_moss_C",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-Wno-deprecated-declarations"]
    for i in inc:
        cmd += ["-I", i]
    cmd += [src, "-o", tmp, "-L", tlib, "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python", "-lc10_hip", "-ltorch_hip",
            "-L", out_dir, "-lmoss_raster", f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, EXT)
    return EXT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))

"""Builds moss_amd/lib/libmoss_raster.so (the C-ABI library of include/moss_raster.h) with hipcc for gfx950.

In-tree build: the .so is git-ignored but travels to the GPU box with the repo snapshot.
``python -m moss_amd.build [--force]``
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
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
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


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "moss_raster.h"), os.path.abspath(__file__)]
    objs = []
    procs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            cmd = [HIPCC] + COMMON + extra + ["-c", s, "-o", o]
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
    if force or procs or _newer(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Builds the HIP shared libraries for gfx950 with hipcc (cross-compiles without a GPU).

    python -m cu_sdr_collection_amd.build        # or  __graft_entry__.build()

Outputs (git-ignored, but shipped to the GPU box by gpurun):
    cu-sdr-collection_amd/lib/libgnsscorr.so     the product: C-ABI of include/gnsscorr.h
    cu-sdr-collection_amd/lib/libgnsssynth.so    test/bench utility: synthetic IF generator
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")

LIBS = {
    "libgnsscorr.so": ["gnsscorr.hip", "corr_kernel.hip", "corr_fast.hip", "corr_lane.hip", "track.hip", "multi.hip", "stream.hip", "acq.hip", "navsync.hip"],
    "libgnsssynth.so": ["synth.hip"],
}
HEADERS = ["gc_internal.h", "corr_common.h", "devloop.h", os.path.join("..", "..", "include", "gnsscorr.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function", "-fno-slp-vectorize"]
# The correlator's exact paths restate the reference's float64 arithmetic operation by operation
# (fl(a + fl(i*d)), MATLAB's two-sided colon).  HIP's __dadd_rn / __dmul_rn are plain operators defined in a
# header compiled under the default -ffp-contract=fast, so the backend fuses them into v_fma_f64 (one rounding
# instead of two: wrong table index at exact ties).  These translation units therefore forbid contraction;
# wanted FMAs are written as fmaf() / fma().
NO_CONTRACT = {"gnsscorr.hip", "corr_kernel.hip", "corr_fast.hip", "corr_lane.hip", "track.hip"}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build libgnsscorr.so")
    return exe


def _stale(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def build(force: bool = False, verbose: bool = True) -> None:
    os.makedirs(LIBDIR, exist_ok=True)
    for lib, srcs in LIBS.items():
        src_paths = [os.path.join(CSRC, s) for s in srcs]
        missing = [s for s in src_paths if not os.path.exists(s)]
        if missing:
            raise RuntimeError(f"missing sources for {lib}: {missing}")
        target = os.path.join(LIBDIR, lib)
        deps = src_paths + [os.path.join(CSRC, h) for h in HEADERS]
        if not force and not _stale(target, deps):
            continue
        # one object per translation unit, compiled in parallel (the kernels are heavily templated), then linked
        objdir = os.path.join(HERE, "build")
        os.makedirs(objdir, exist_ok=True)
        cflags = [f for f in FLAGS if f != "-shared"]
        jobs = []
        for s in src_paths:
            obj = os.path.join(objdir, os.path.splitext(os.path.basename(s))[0] + ".o")
            if not force and not _stale(obj, [s] + [os.path.join(CSRC, h) for h in HEADERS]):
                jobs.append((obj, None, None))
                continue
            extra = ["-ffp-contract=off"] if os.path.basename(s) in NO_CONTRACT else []
            cmd = [_hipcc(), *cflags, *extra, "-c", s, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            jobs.append((obj, cmd, subprocess.Popen(cmd, cwd=CSRC)))
        for obj, cmd, proc in jobs:
            if proc is not None and proc.wait() != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
        cmd = [_hipcc(), *FLAGS, *[j[0] for j in jobs], "-o", target]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)


if __name__ == "__main__":
    build(force="--force" in sys.argv)

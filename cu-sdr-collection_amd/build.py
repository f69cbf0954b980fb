"""Builds the HIP shared libraries for gfx950 with hipcc (cross-compiles without a GPU).

    python -m cu_sdr_collection_amd.build [--force]      # or  __graft_entry__.build()

Outputs (git-ignored, but shipped to the GPU box by gpurun):
    cu-sdr-collection_amd/lib/libgnsscorr.so     the product: C-ABI of include/gnsscorr.h
    cu-sdr-collection_amd/lib/libgnsssynth.so    test/bench utility: synthetic IF generator
    cu-sdr-collection_amd/lib/BUILD_INFO.json    what was built from what: per translation unit the SHA-256 of everything that
                                                 went into it (source, headers, flags, compiler version) and the compile seconds,
                                                 per library the SHA-256 of the .so

A translation unit is recompiled when that input hash differs from the recorded one (content, not modification times: a fresh
checkout, a touched file or a copied tree cannot pass a stale object off as current); `force=True` / GC_BUILD_FORCE=1 recompiles
everything.  `verify()` re-hashes sources and libraries against BUILD_INFO.json - smoke() calls it on the GPU box, so the binary a
test run loaded is tied to the sources of the same tree.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
INFO = os.path.join(LIBDIR, "BUILD_INFO.json")

_CORR_UNITS = ["gnsscorr.hip", "corr_kernel.hip", "corr_fast.hip", "corr_multi.hip",
               # one source, four translation units (its 148 kernel instantiations took one compiler process 248 s): unit@MACRO=value
               "corr_cboc.hip",   # (round 6: the hybrid for channels with a derived BOC(6,1) arm ships - it beats the lane kernel on config 3's lists)
               "corr_lane.hip@GC_LANE_PART=0", "corr_lane.hip@GC_LANE_PART=1", "corr_lane.hip@GC_LANE_PART=2", "corr_lane.hip@GC_LANE_PART=3",
               "track.hip", "multi.hip", "stream.hip",
               # the acquisition, one translation unit per part of the search (acq_internal.h is what they share)
               "acq_fft.hip", "acq_coarse.hip", "acq_shift.hip", "acq_fine.hip", "acq_cond.hip", "acq_guard.hip", "navsync.hip"]


def _tuned(spec: str) -> str:
    return spec + ("," if "@" in spec else "@") + "GC_TUNING=1"


LIBS = {
    # the product: reads no tuning variable (gc_internal.h GC_TUNE_ENV), carries no experimental kernel
    "libgnsscorr.so": _CORR_UNITS,
    # the same sources with -DGC_TUNING=1: every A/B switch of docs/KNOBS.md live.  Loaded through GC_LIB_PATH by the knob tests, scripts/variants.sh and the profiling scripts.
    "libgnsscorr_tuning.so": [_tuned(u) for u in _CORR_UNITS],
    "libgnsssynth.so": ["synth.hip"],
}
HEADERS = ["gc_internal.h", "corr_common.h", "devloop.h", "acq_guard.h", "acq_internal.h", os.path.join("..", "..", "include", "gnsscorr.h")]
# --offload-compress: the gfx950 code objects inside the fat binary are zstd-compressed (10.1 -> ~1.6 MB; the HIP runtime inflates them
# when the library is loaded: ~20 ms once per process)
FLAGS = ["--offload-arch=gfx950", "--offload-compress", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function", "-fno-slp-vectorize"]
# The correlator's exact paths restate the reference's float64 arithmetic operation by operation
# (fl(a + fl(i*d)), MATLAB's two-sided colon).  HIP's __dadd_rn / __dmul_rn are plain operators defined in a
# header compiled under the default -ffp-contract=fast, so the backend fuses them into v_fma_f64 (one rounding
# instead of two: wrong table index at exact ties).  These translation units therefore forbid contraction;
# wanted FMAs are written as fmaf() / fma().
NO_CONTRACT = {"gnsscorr.hip", "corr_kernel.hip", "corr_fast.hip", "corr_multi.hip", "corr_cboc.hip", "corr_lane.hip", "track.hip"}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build libgnsscorr.so")
    return exe


_VERSION = None


def _compiler_version() -> str:
    global _VERSION
    if _VERSION is None:
        try:
            _VERSION = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, check=True).stdout.strip()
        except (OSError, subprocess.CalledProcessError, RuntimeError):
            _VERSION = "unknown"
    return _VERSION


def _sha_file(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def _unit(spec: str):
    """(source file, extra -D flags, object base name) of a unit spec "file.hip" or "file.hip@MACRO=value[,MACRO=value...]"."""
    src, _, macros = spec.partition("@")
    base = os.path.splitext(src)[0]
    if not macros:
        return src, [], base
    ms = macros.split(",")
    return src, ["-D" + m for m in ms], base + "_" + "_".join("".join(c if c.isalnum() else "_" for c in m) for m in ms)


def _tu_flags(spec: str):
    src, defs, _ = _unit(spec)
    cflags = [f for f in FLAGS if f != "-shared"]
    return cflags + (["-ffp-contract=off"] if src in NO_CONTRACT else []) + defs


def source_hash(src: str, with_compiler: bool = True) -> str:
    """SHA-256 over a translation unit's inputs: its source, every shared header, its flags (and the compiler's version string)."""
    h = hashlib.sha256()
    for p in [os.path.join(CSRC, _unit(src)[0])] + [os.path.join(CSRC, x) for x in HEADERS]:
        h.update(os.path.basename(p).encode() + b"\0" + _sha_file(p).encode() + b"\0")
    h.update(" ".join(_tu_flags(src)).encode())
    if with_compiler:
        h.update(_compiler_version().encode())
    return h.hexdigest()


def _load_info() -> dict:
    try:
        with open(INFO) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def build(force: bool = False, verbose: bool = True) -> dict:
    force = force or os.environ.get("GC_BUILD_FORCE", "") not in ("", "0")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    info = _load_info()
    units = info.setdefault("units", {})
    libs = info.setdefault("libs", {})
    t_all = time.time()
    for lib, srcs in LIBS.items():
        missing = [s for s in srcs if not os.path.exists(os.path.join(CSRC, _unit(s)[0]))]
        if missing:
            raise RuntimeError(f"missing sources for {lib}: {missing}")
        target = os.path.join(LIBDIR, lib)
        # one object per translation unit, compiled in parallel (the kernels are heavily templated), then linked
        jobs, rebuilt = [], False
        for s in srcs:
            obj = os.path.join(objdir, _unit(s)[2] + ".o")
            want = source_hash(s)
            have = units.get(s, {})
            if not force and os.path.exists(obj) and have.get("inputs_sha256") == want and have.get("object_sha256") == _sha_file(obj):
                jobs.append((s, obj, want, None, None, 0.0))
                continue
            cmd = [_hipcc(), *_tu_flags(s), "-c", os.path.join(CSRC, _unit(s)[0]), "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            jobs.append((s, obj, want, cmd, subprocess.Popen(cmd, cwd=CSRC), time.time()))
        pending = [j for j in jobs if j[4] is not None]
        while pending:                        # every unit's own wall time: poll, do not wait for them in list order
            for j in list(pending):
                s, obj, want, cmd, proc, t0 = j
                rc = proc.poll()
                if rc is None:
                    continue
                pending.remove(j)
                if rc != 0:
                    for other in pending:
                        other[4].kill()
                    raise subprocess.CalledProcessError(rc, cmd)
                rebuilt = True
                units[s] = {"inputs_sha256": want, "object_sha256": _sha_file(obj), "compile_seconds": round(time.time() - t0, 1)}
                if verbose:
                    print(f"[build] {s}: {units[s]['compile_seconds']} s (wall; the units compile side by side)", flush=True)
            if pending:
                time.sleep(0.2)
        objs = [j[1] for j in jobs]
        link_key = hashlib.sha256("".join(units[s]["object_sha256"] for s in srcs).encode()).hexdigest()
        have = libs.get(lib, {})
        if rebuilt or force or not os.path.exists(target) or have.get("objects_sha256") != link_key or have.get("sha256") != _sha_file(target):
            cmd = [_hipcc(), *FLAGS, *objs, "-o", target]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=CSRC)
            libs[lib] = {"objects_sha256": link_key, "sha256": _sha_file(target), "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                         "sources": {s: source_hash(s, with_compiler=False) for s in srcs}}
    info["compiler"] = _compiler_version().splitlines()[0] if _compiler_version() else "unknown"
    info["flags"] = FLAGS
    with open(INFO, "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    if verbose:
        for lib in LIBS:
            print(f"[build] {lib} sha256 {libs[lib]['sha256']} ({'forced, ' if force else ''}{time.time() - t_all:.0f} s in all)", flush=True)
    return info


def verify() -> dict:
    """{lib: sha256} of the built libraries after checking them and the sources of this tree against BUILD_INFO.json (works without
    hipcc: the compiler version is not part of this check).  Raises RuntimeError when a library is not the one the recorded build
    produced, or was built from other sources than the ones here."""
    info = _load_info()
    out = {}
    for lib, srcs in LIBS.items():
        rec = info.get("libs", {}).get(lib)
        target = os.path.join(LIBDIR, lib)
        if rec is None or not os.path.exists(target):
            raise RuntimeError(f"{lib}: no recorded build (run cu_sdr_collection_amd.build)")
        sha = _sha_file(target)
        if sha != rec["sha256"]:
            raise RuntimeError(f"{lib}: sha256 {sha[:16]} is not the recorded build's {rec['sha256'][:16]}")
        for s in srcs:
            if rec["sources"].get(s) != source_hash(s, with_compiler=False):
                raise RuntimeError(f"{lib}: built from another {s} (or other headers / flags) than this tree's - rebuild")
        out[lib] = sha
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)

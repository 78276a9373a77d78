"""Build libreevr_amd.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m reevr_amd.build          # or: from reevr_amd.build import build_lib; build_lib()

The shared object lands next to the sources (reevr_amd/csrc/libreevr_amd.so). It is
git-ignored but travels with the working tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libreevr_amd.so")
SOURCES = ["rvc_kernels.hip", "rvc_sweep.hip", "rvc_impulse.hip", "rvc_plan.cpp", "rvc_state.cpp", "rvc_schedule.cpp", "rvc_abi.cpp"]
# per-source extra flags: the kernels are built WITHOUT the SLP vectoriser (rvc_sweep.hip: the comment at its top;
# rvc_kernels.hip: v_pk_* issues at half the rate of the scalar ops on gfx950 and costs a register shuffle per operand pair --
# the 8192-bin inverse transform 614 -> 679 instructions but 239 packed + 168 moves -> 424 scalar + 36 moves, 56 -> 40 registers,
# 115 -> 108 us per 4096 rows; no kernel spills any more)
EXTRA_FLAGS = {"rvc_sweep.hip": ["-fno-slp-vectorize"], "rvc_kernels.hip": ["-fno-slp-vectorize"]}
HEADERS = ["rvc_internal.h", "rvc_set.h", "rvc_fft_lds.hpp", os.path.join("..", "..", "include", "reevr_amd", "rvc.h"),
           os.path.join("..", "..", "include", "reevr_amd", "rvc_debug.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-ffp-contract=fast-honor-pragmas",
                "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def compile_objects(out_dir: str, defines=(), verbose: bool = False):
    """hipcc -c every source (in parallel) into out_dir; returns the object paths."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(out_dir, exist_ok=True)

    def one(src):
        obj = os.path.join(out_dir, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc()] + COMMON_FLAGS + list(defines) + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj
    with ThreadPoolExecutor(len(SOURCES)) as ex:
        return list(ex.map(one, SOURCES))


def link_lib(objs, lib: str, verbose: bool = False):
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    objs = compile_objects(os.path.join(CSRC, "build"), verbose=verbose)
    link_lib(objs, LIB, verbose)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))

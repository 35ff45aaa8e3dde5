"""Builds libvaporetto_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

`python -m vaporetto_amd.build` or `__graft_entry__.build()`.  The .so is git-ignored but travels with the
gpurun snapshot; hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvaporetto_hip.so")
SOURCES = ["model.cpp", "tables.cpp", "capi.cpp", "capi_device.cpp", "capi_host.cpp", "kernels.hip", "kernels_fast.hip", "kernels_tags.hip", "kernels_emit.hip"]
HEADERS = ["model.hpp", "tables.hpp", "layout.h", "kernels.hpp", "device_common.h", "capi_internal.hpp", "patset.hpp", os.path.join("..", "..", "include", "vaporetto_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_hip(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-o", LIB_PATH] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))

"""TEST INFRASTRUCTURE: builds the product's kernel and host sources (vaporetto_amd/csrc) against the CPU emulator
of the HIP execution model (tests/native/hipemu) and loads the result with the C ABI's signatures.

The emulated library exists so that `-m "not gpu"` can run the code the GPU runs (tests/test_kernel_emu.py).  It is
never on the product path: `vaporetto_amd._lib.load()` only ever loads libvaporetto_hip.so and fails without it."""
import ctypes as C
import os
import subprocess

from vaporetto_amd import _lib, build

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "native", "hipemu")
# VPT_EMU_DEFINES="-DVPT_FAST_STAGED=1 ..." builds (and loads) another geometry of the kernels into a library of its own
_DEFINES = os.environ.get("VPT_EMU_DEFINES", "").split()
LIB = os.path.join(HERE, "native", "libvaporetto_emu%s.so" % ("_" + "".join(c for c in "".join(_DEFINES) if c.isalnum()) if _DEFINES else ""))
_EMU_FILES = [os.path.join(EMU, "hipemu.cpp"), os.path.join(EMU, "hip", "hip_runtime.h")]


def build_emulated() -> str:
    import fcntl
    srcs = [os.path.join(build.CSRC, f) for f in build.SOURCES]
    deps = srcs + [os.path.join(build.CSRC, h) for h in build.HEADERS] + _EMU_FILES
    with open(LIB + ".lock", "w") as lock:   # pytest-xdist workers: one of them builds, the others wait for it
        fcntl.flock(lock, fcntl.LOCK_EX)
        if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
            return LIB
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
               "-Wl,-Bsymbolic",   # its hip* definitions bind locally even if a real HIP runtime is loaded in the process
               "-I" + EMU, "-o", LIB + ".tmp"] + _DEFINES
        cmd += [s for s in srcs if s.endswith(".cpp")] + ["-x", "c++"] + [s for s in srcs if s.endswith(".hip")]
        cmd += ["-x", "none", _EMU_FILES[0]]
        subprocess.check_call(cmd)
        os.replace(LIB + ".tmp", LIB)
    return LIB


def load():
    L = C.CDLL(build_emulated())
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L

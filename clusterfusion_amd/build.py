"""Build recipe for libclusterfusion_hip.so (hipcc, gfx950 only, in-tree output).

`python -m clusterfusion_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without
a GPU; the resulting .so travels to the GPU box with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libclusterfusion_hip.so")
SOURCES = ["cf_api.hip", "cf_mla_api.hip"]
# The persistent kernels branch ONCE on the device-side sequence length into straight copies of their whole path (exact wait
# counts per arm, cf_fused_kernel.h).  SimplifyCFG would hoist the arms' identical first instructions -- the weight requests of
# phase 1 -- above that branch and sink their common tails below it; the register allocator then spills the hoisted loads
# (144..461 VGPR spills, each behind an s_waitcnt vmcnt(0)).  Keep the copies apart:
DEVICE_FLAGS = ["-mllvm", "-hoist-common-insts=false", "-mllvm", "-sink-common-insts=false"]


def _headers():
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "clusterfusion_hip.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


FAST = os.path.join(PKG, "_cf_fast.so")
FAST_SRC = os.path.join(CSRC, "cf_torch_binding.cpp")


def build_fast(force: bool = False, verbose: bool = False) -> str:
    """The compiled host binding of the three Llama entries (csrc/cf_torch_binding.cpp -> clusterfusion_amd/_cf_fast.so): plain
    C++ over torch's headers, no device code.  It calls the C-ABI through addresses handed over at import (ops.py), so it links
    against torch only; the HIP runtime it needs one call of is the one torch has already loaded."""
    newest = max(os.path.getmtime(FAST_SRC), os.path.getmtime(os.path.join(ROOT, "include", "clusterfusion_hip.h")))
    if not force and os.path.exists(FAST) and os.path.getmtime(FAST) >= newest:
        return FAST
    import sysconfig
    import torch
    tdir = os.path.dirname(torch.__file__)
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_cf_fast",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I" + os.path.join(tdir, "include"), "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROOT, "include"), FAST_SRC, "-o", FAST + ".tmp",
           "-L" + os.path.join(tdir, "lib"), "-ltorch_python", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
           "-Wl,-rpath," + os.path.join(tdir, "lib")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(FAST + ".tmp", FAST)
    return FAST


def build(force: bool = False, verbose: bool = False) -> str:
    try:
        build_fast(force, verbose)
    except Exception as e:      # noqa: BLE001 -- the binding is an accelerator of the host path, the library below is the product
        print(f"[clusterfusion_amd.build] the compiled host binding did not build ({type(e).__name__}: {e}); the ctypes path serves every "
              "entry (slower per call); tests/test_binding.py will say so", file=sys.stderr)
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + DEVICE_FLAGS
    cmd += os.environ.get("CF_EXTRA_HIPCC_FLAGS", "").split()     # experiments only (-DCF_EXP_...)
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

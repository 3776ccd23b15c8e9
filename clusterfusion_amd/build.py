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


def build(force: bool = False, verbose: bool = False) -> str:
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

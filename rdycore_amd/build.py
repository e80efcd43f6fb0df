"""Builds the native library (HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only
container (the "does it build" check of __graft_entry__.build()).
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "librdyhip.so")
SOURCES = ["rdyhip_api.hip"]
DEPS = SOURCES + ["swe_device.h", "swe_kernels.h", "forcing_kernels.h", "muscl_kernels.h", "halo_exchange.h", "halo_plan.h"]
ARCH = "gfx950"
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def lib_path() -> str:
    # RDYHIP_LIB: load another build of the same ABI (A/B timing of two revisions in one process tree)
    return os.environ.get("RDYHIP_LIB", LIB)


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in DEPS] + [os.path.join(INCLUDE, "rdyhip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> rdycore_amd/csrc/librdyhip.so"""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build librdyhip.so (the HIP extension is required; there is no CPU fallback)")
    extra = os.environ.get("RDYHIP_EXTRA_HIPCC_FLAGS", "").split()   # experiments only
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           f"-I{INCLUDE}", f"-I{CSRC}", f"-I{ROCM}/include"] + extra + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES] + [
               f"-L{ROCM}/lib", "-lrccl"]   # RCCL: the halo exchange behind the ABI (csrc/halo_exchange.h)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))

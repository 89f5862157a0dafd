"""Build libaudioprims_hip.so in-tree with hipcc for gfx950 (no JIT cache: the built
.so travels with the source tree to the GPU box)."""

from __future__ import annotations

import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# AP_LIB_PATH: load another build of the same sources instead (diagnostic builds, tools/diag_clock.py)
LIB_PATH = os.environ.get("AP_LIB_PATH") or os.path.join(PKG_DIR, "libaudioprims_hip.so")
SOURCES = ["audioprims.hip", "host_builders.cpp"]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(PKG_DIR), "include", "audioprims.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library."""
    if os.environ.get("AP_LIB_PATH"):
        return LIB_PATH
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))

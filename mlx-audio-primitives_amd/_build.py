"""Build libaudioprims_hip.so in-tree with hipcc for gfx950 (no JIT cache: the built
.so travels with the source tree to the GPU box).  The translation units are compiled
in parallel into build/obj/*.o (kept out of history and off the GPU box) and linked."""

from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
ROOT = os.path.dirname(PKG_DIR)
OBJ_DIR = os.path.join(ROOT, "build", "obj")
# AP_LIB_PATH: load another build of the same sources instead (diagnostic builds, tools/diag_clock.py)
LIB_PATH = os.environ.get("AP_LIB_PATH") or os.path.join(PKG_DIR, "libaudioprims_hip.so")
SOURCES = ["audioprims.hip", "stft16.hip", "istft16.hip", "host_builders.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _hipcc() -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _deps(src: str) -> list[str]:
    """Files the object was compiled from, out of the -MD depfile of its last compilation."""
    dep = _obj(src)[:-2] + ".d"
    if not os.path.exists(dep):
        return []
    text = open(dep).read().replace("\\\n", " ")
    return [t for t in text.split(":", 1)[1].split() if t]


def _obj_stale(src: str) -> bool:
    obj = _obj(src)
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = _deps(src) or [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(CSRC, src))
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(ROOT, "include", "audioprims.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def _compile(src: str, verbose: bool) -> None:
    obj = _obj(src)
    cmd = [_hipcc()] + FLAGS + ["-MD", "-MF", obj[:-2] + ".d", "-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library."""
    if os.environ.get("AP_LIB_PATH"):
        return LIB_PATH
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [s for s in SOURCES if force or _obj_stale(s)]
    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(lambda s: _compile(s, verbose), todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))

"""Builds libpleas_hip.so in-tree with hipcc for gfx950 (no JIT cache, no cmake)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libpleas_hip.so")


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpleas_hip.so can only be built on a ROCm image")
    return exe


def sources() -> list:
    return sorted(os.path.join(CSRC, s) for s in os.listdir(CSRC) if s.endswith(".hip"))


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(INCLUDE, "pleas_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into one shared library for gfx950 (objects built in parallel)."""
    if not force and not is_stale():
        return LIB
    objs, procs = [], []
    for src in sources():
        obj = os.path.splitext(src)[0] + ".o"
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC, "-c", src,
               "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


SAN_DIR = os.path.join(CSRC, "_san")
SAN_LIB = os.path.join(SAN_DIR, "libpleas_hip_asan.so")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-O1"]


def asan_runtime() -> str:
    """clang's shared AddressSanitizer runtime: must be LD_PRELOADed into a process that dlopens SAN_LIB."""
    out = subprocess.check_output([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    if not os.path.isabs(out) or not os.path.exists(out):
        raise RuntimeError("AddressSanitizer runtime not found next to hipcc's clang (%r)" % out)
    return out


def build_sanitized(force: bool = False) -> str:
    """The same sources with the HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (device code compiled as
    always: GPU sanitizers are not available on the pool).  What it is for: the ~1.5 k lines of host C++ inside the
    .hip files -- plan builders, plan caches, XCD item ordering, lane dealing, the host LAP -- exercised WITHOUT a GPU
    through the `*_ws_bytes` / `*_plan_info` / `pleas_lsap_host` entry points and the argument checks
    (tests/sanitize_driver.py).  Objects and library live under csrc/_san/ and never replace the product library."""
    os.makedirs(SAN_DIR, exist_ok=True)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(INCLUDE, "pleas_hip.h")]
    if not force and os.path.exists(SAN_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(SAN_LIB) for d in deps):
        return SAN_LIB
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(SAN_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        cmd = [_hipcc(), "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC] + SAN_FLAGS + ["-c", src, "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-shared-libsan"] + SAN_FLAGS[:2] + ["-o", SAN_LIB] + objs)
    return SAN_LIB


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        print(build_sanitized(force="--force" in sys.argv))
    else:
        print(build(force="--force" in sys.argv, verbose=True))

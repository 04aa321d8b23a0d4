"""
In-tree build of the HIP extension:  python -m pygho_amd.build [--force]

hipcc cross-compiles gfx950 code objects without a GPU; the resulting
``pygho_amd/_lib/libpygho_hip.so`` travels with the source tree.  HIP runtime symbols are
left to resolve against the ``libamdhip64.so`` that PyTorch-ROCm has already loaded
(same SONAME), so only one HIP runtime lives in the process.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libpygho_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wno-unused-result",
         "-fno-gpu-rdc", "-DNDEBUG"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP extension cannot be built")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [cc, *FLAGS, f"-I{INCLUDE}", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

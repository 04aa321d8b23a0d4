"""
In-tree build of the HIP extension:  python -m pygho_amd.build [--force]

hipcc cross-compiles gfx950 code objects without a GPU; the resulting
``pygho_amd/_lib/libpygho_hip.so`` travels with the source tree.  HIP runtime symbols are
left to resolve against the ``libamdhip64.so`` that PyTorch-ROCm has already loaded
(same SONAME), so only one HIP runtime lives in the process.
"""
from __future__ import annotations

import glob
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libpygho_hip.so")
USAGE = os.path.join(LIBDIR, "resource_usage.json")       # per-kernel registers / scratch / occupancy from the last build
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wno-unused-result",
         "-fno-gpu-rdc", "-DNDEBUG"]


def _resource_usage(remarks: str, src: str) -> dict:
    """parse the compiler's -Rpass-analysis=kernel-resource-usage remarks: {kernel: vgprs, scratch, occupancy, lds} for the kernels
    of namespace pygho (library kernels pulled in from hipCUB are not ours to fix)."""
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1) if m.group(1).startswith("_ZN5pygho") else None
            if cur:
                out[cur] = {"source": src}
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch_bytes_per_lane", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy_waves_per_simd", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds_bytes_per_block", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                out[cur][key] = int(m.group(1))
    return {k: v for k, v in out.items() if "scratch_bytes_per_lane" in v}


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP extension cannot be built")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def up_to_date() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(USAGE):     # the resource-usage record is part of a complete build
        return False
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True, variant: str = "", defines=()) -> str:
    """`variant` + `defines`: an A/B build of the same sources with extra -D flags into _lib/variants/<variant>/ (loaded with
    PYGHO_AMD_LIB=<path>, see _native.py) so that two kernel versions can be timed on the SAME box in one gpurun call."""
    if variant:
        return _build_into(os.path.join(LIBDIR, "variants", variant), list(defines), verbose, force=True)
    if not force and up_to_date():
        return LIB
    return _build_into(LIBDIR, [], verbose, force)


def _build_into(libdir: str, defines, verbose: bool, force: bool = True) -> str:
    """`force` False: objects newer than their source and every header are kept (their resource-usage record is kept beside
    them), so that a one-file change recompiles one file; `__graft_entry__.build()` compiles everything."""
    os.makedirs(libdir, exist_ok=True)
    objdir = os.path.join(libdir, "obj")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    lib, usage_path = os.path.join(libdir, "libpygho_hip.so"), os.path.join(libdir, "resource_usage.json")
    # objects of sources that no longer exist (a shelved experiment's .o stayed beside the live ones for a round) are removed: obj/
    # holds exactly what the library is linked from
    live = {os.path.basename(src) + ".o" for src in sources()}
    live |= {name + ".usage.json" for name in live}
    for name in os.listdir(objdir):
        if name not in live:
            os.remove(os.path.join(objdir, name))

    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        rec = obj + ".usage.json"
        if not force and os.path.exists(obj) and os.path.exists(rec) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in [src, *headers]):
            with open(rec) as f:
                return obj, json.load(f)
        cmd = [cc, *FLAGS, *defines, "-Rpass-analysis=kernel-resource-usage", f"-I{INCLUDE}", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        proc = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if proc.returncode != 0:
            sys.stderr.write(proc.stderr)
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        usage = _resource_usage(proc.stderr, os.path.basename(src))
        with open(rec, "w") as f:
            json.dump(usage, f)
        return obj, usage

    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(compile_one, sources()))
    objs = [r[0] for r in results]
    usage = {k: v for r in results for k, v in r[1].items()}
    with open(usage_path, "w") as f:
        json.dump(usage, f, indent=1, sort_keys=True)
    # a register array demoted to scratch turns into HBM traffic (scratch stores are memory writes): the first 16-byte forms of
    # the masked fill / broadcast kernels wrote 2x their output that way.  Our own kernels must not spill -- fail the build.
    spilled = {k: v["scratch_bytes_per_lane"] for k, v in usage.items() if v["scratch_bytes_per_lane"] > 0}
    if spilled:
        raise RuntimeError(f"kernels with scratch (register arrays demoted to memory): {spilled}")
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *objs, "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    args = sys.argv[1:]
    variant = args[args.index("--variant") + 1] if "--variant" in args else ""
    print(build(force="--force" in args, variant=variant, defines=[a for a in args if a.startswith("-D")]))

#!/usr/bin/env python3
"""
Turn the two rocprofv3 --pmc passes of the bench command (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
profiles/<tag>_traffic.json: HBM bytes per launch of one kernel of the step -- `fast` = seg_gmr_fast_kernel<bf16, SUM, BOTH> (the
by-tuple backward launches; rounds 1-4: also the forward), `fused` = seg_fused_fwd_kernel<bf16, SILU, SUM> (round 5: the forward
with the layer MLP inside) -- corrected
as MI355X_MICROARCH.md prescribes (counters are in KiB; FETCH_SIZE is half-counted on gfx950 for coalesced 16-B reads:
x1.97 from the calibration in profiles/r01_pmc_seg_gmr.md; WRITE_SIZE is exact).

usage: collect_traffic.py <fetch_dir> <write_dir> <kernel_stats.csv> <out.json> [graphs hidden dtype [fast|fused]]
"""
import csv
import glob
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash(files):
    """the same hash bench.py computes: a traffic figure is only reported for the kernel sources it was measured on"""
    h = hashlib.sha256()
    for name in files:
        h.update(open(os.path.join(REPO, "pygho_amd", "csrc", name), "rb").read())
    return h.hexdigest()

# which -> (substring of the rocprof kernel name, the name bench.py reports, source files)
KERNELS = {"fast": ("seg_gmr_fast_kernel<pygho::bf16, 0, 0, false, true, false, false, 0>",     # <T, SUM, BOTH, !SCALED, OFF32, !OUTF32, !THIRD, no act>
                    "seg_gmr_fast_kernel<bf16,SUM,BOTH>", ("common.h", "seg_reduce.hip")),
           "fused": ("seg_fused_fwd_kernel<pygho::bf16, 2, false", "seg_fused_fwd_kernel<bf16,SILU,SUM>", ("common.h", "seg_fused.hip")),
           "dual": ("seg_dual_kernel<pygho::bf16", "seg_dual_kernel<bf16,SUM>", ("common.h", "seg_dual.hip"))}
WHICH = sys.argv[8] if len(sys.argv) > 8 else "fast"
KERNEL, REPORTED, SOURCES = KERNELS[WHICH]
FETCH_CORRECTION = 1.97


def mean_counter(directory, name):
    f = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == name and KERNEL in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000]
    return sum(vals) / len(vals) * 1024.0, len(vals)


def main():
    fetch_dir, write_dir, stats_csv, out = sys.argv[1:5]
    graphs, hidden, dtype = (int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]) if len(sys.argv) > 7 else (8192, 128, "bf16")
    fetch, n = mean_counter(fetch_dir, "FETCH_SIZE")
    write, _ = mean_counter(write_dir, "WRITE_SIZE")
    row = next(r for r in csv.DictReader(open(stats_csv)) if KERNEL in r["Name"])
    json.dump({
        "kernel": REPORTED,
        "config": {"graphs_per_gpu": graphs, "hidden": hidden, "dtype": dtype},
        "kernel_source_sha256": kernel_source_hash(SOURCES),
        "launches_profiled": n,
        "FETCH_SIZE_bytes_raw": fetch, "FETCH_SIZE_correction": FETCH_CORRECTION, "WRITE_SIZE_bytes": write,
        "traffic_bytes_per_launch": fetch * FETCH_CORRECTION + write,
        "rocprof_avg_us": float(row["AverageNs"]) / 1e3, "rocprof_calls": int(row["Calls"]),
        "note": "separate --pmc passes of `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 3 "
                "--warmup 1 --no-cpu-baseline`; launches with grid > 100k threads only (the 8192-graph aggregation; `fast`: the by-tuple backward "
                "launch per layer, `fused`: the forward launch per layer with Linear -> BatchNorm -> act inside; the by-edge backward plan runs on seg_scatter_kernel); "
                "FETCH_SIZE x1.97 per profiles/r01_pmc_seg_gmr.md",
    }, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()

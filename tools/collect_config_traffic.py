#!/usr/bin/env python3
"""
Summarise tools/profile_configs.sh: per kernel entry of bench.py's `configs` object the HBM bytes per launch
(FETCH_SIZE x the gfx950 read correction + WRITE_SIZE, counters in KiB, separate passes) and the rocprofv3 average duration,
tied to the hash of the kernel sources (tools/bench_configs.py reports a figure only for the sources it was measured on).

usage: collect_config_traffic.py <prof_dir> <out.json>
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as BC   # noqa: E402

# entry -> (kernel-name substrings, family, read correction and where it was calibrated)
ENTRIES = {
    "config5_spspmm_fwd": (("seg_gmr_window_kernel", "seg_gmr_tile_kernel", "seg_gmr_fast_kernel"), "seg", 2.05,
                           "x 2.05: calibrated on a launch of known read volume with the same 512-B row gathers (profiles/r02_pmc_i2_window.md)"),
    "config3_mamamm_XA": (("masked_bmm",), "bmm", 1.97, "x 1.97: whole-row 16-B coalesced reads (profiles/r01_pmc_seg_gmr.md)"),
    "config3_mamamm_XY": (("masked_bmm",), "bmm", 1.97, "x 1.97: whole-row 16-B coalesced reads (profiles/r01_pmc_seg_gmr.md)"),
}


def counter_rows(directory):
    rows = []
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    prof, out = sys.argv[1], sys.argv[2]
    # launches in dispatch order: --kernels-only runs config 5's launch 34 times (1 + 3 + 30), then X A 33 times, then X Y 33 times
    per = {}
    for i, names in ((1, ("FETCH_SIZE",)), (2, ("WRITE_SIZE",)), (3, ("TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"))):
        rows = counter_rows(os.path.join(prof, f"pmc{i}"))
        for name in names:
            seq = collections.defaultdict(list)
            for r in rows:
                if r["Counter_Name"] == name:
                    seq[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            per[name] = {k: [v for _, v in sorted(vs)] for k, vs in seq.items()}
    stats = {}
    for f in glob.glob(os.path.join(prof, "stats") + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            stats[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    res = {}
    for key, (subs, family, corr, why) in ENTRIES.items():
        def pick(name):
            # the kernel of this entry: the bmm entries share a name family -- X A launches come first, X Y second
            cands = {k: v for k, v in per.get(name, {}).items() if any(s in k for s in subs) and len(v) >= 20}
            if not cands:
                return None, None
            if family == "bmm":
                k = max(cands, key=lambda q: len(cands[q]))
                v = cands[k]
                if len(cands) >= 2:                      # two different kernels: order of first appearance is not in the CSV; use counts per kernel
                    ks = sorted(cands)
                    k = ks[0] if key.endswith("XA") else ks[-1]
                    v = cands[k]
                else:
                    half = len(v) // 2
                    v = v[:half] if key.endswith("XA") else v[half:]
                return k, v
            k = max(cands, key=lambda q: len(cands[q]))
            return k, cands[k]
        kname, fetch = pick("FETCH_SIZE")
        _, write = pick("WRITE_SIZE")
        if fetch is None or write is None:
            continue
        mean = lambda v: sum(v) / len(v)
        ent = {"kernel": kname, "kernel_source_sha256": BC.source_hash(BC.SOURCES[family]), "launches_profiled": len(fetch),
               "FETCH_SIZE_bytes_raw": mean(fetch) * 1024.0, "FETCH_SIZE_correction": corr, "FETCH_SIZE_correction_source": why,
               "WRITE_SIZE_bytes": mean(write) * 1024.0}
        ent["traffic_bytes_per_launch"] = ent["FETCH_SIZE_bytes_raw"] * corr + ent["WRITE_SIZE_bytes"]
        for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"):
            _, v = pick(c)
            if v:
                ent[c] = mean(v)
        if kname in stats:
            ent["rocprof_avg_us_whole_script"], ent["rocprof_calls_whole_script"] = stats[kname]
        res[key] = ent
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

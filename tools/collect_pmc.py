#!/usr/bin/env python3
"""
Summarise rocprofv3 --pmc passes (one directory per pass, --kernel-trace only) into per-kernel means.

usage: collect_pmc.py <out.json> <name-substring>[,<name-substring>...] <pass_dir> [<pass_dir> ...]
"""
import collections
import csv
import glob
import json
import sys


def main():
    out, subs, dirs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                key = next((s for s in subs if s in name), None)
                if key is None:
                    continue
                short = name.split("(")[0][-90:]
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One steady-state training step out of a rocprofv3 --kernel-trace CSV of bench.py: every kernel in launch order with its duration and
the idle gap in front of it, and a summary by kernel name (launches per step, microseconds per step; which ones are below 50 us).

    python tools/step_trace.py <dir with *_kernel_trace.csv> [--step K] [--list]
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("pygho::", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    return name[:110]


def main():
    d = sys.argv[1]
    which = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -3
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    opt = [i for i, r in enumerate(rows) if "FusedAdam" in r[2] or "fused_adam" in r[2].lower()]
    ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] != i + 1]        # the LAST launch of every optimizer step
    if len(ends) < 4:
        print(f"only {len(ends)} optimizer launches found in {len(rows)} kernels")
        return
    lo, hi = ends[which - 1] + 1, ends[which] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][0], step[-1][1]
    busy = sum(e - s for s, e, _ in step)
    print(f"step {which}: {len(step)} kernels, wall {(t1 - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
    by = collections.OrderedDict()
    prev_end = t0
    for s, e, n in step:
        k = short(n)
        ent = by.setdefault(k, [0, 0.0, 0.0])
        ent[0] += 1
        ent[1] += (e - s) / 1e3
        ent[2] += max(0, s - prev_end) / 1e3
        if "--list" in sys.argv:
            print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} gap {max(0, s - prev_end) / 1e3:6.1f}  {k}")
        prev_end = max(prev_end, e)
    small = {k: v for k, v in by.items() if v[1] / v[0] < 50.0}
    print(f"kernels below 50 us: {sum(v[0] for v in small.values())} launches, {sum(v[1] for v in small.values()):.1f} us, "
          f"gaps in front of them {sum(v[2] for v in small.values()):.1f} us")
    print(f"{'launches':>8} {'us/step':>9} {'avg us':>8} {'gap us':>7}  kernel")
    for k, (n, us, gap) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:8d} {us:9.1f} {us / n:8.1f} {gap:7.1f}  {k}")


if __name__ == "__main__":
    main()

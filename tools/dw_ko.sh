#!/bin/bash
# tools/dw_ko.sh <variant> ...: median duration of the TUPLE-LEVEL launches (> 150 us) of bn_bwd_linear_dw and of the three rowblock
# passes inside the training step, per variant build ("base" = main library), from a rocprofv3 kernel trace
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != base ]; then export PYGHO_AMD_LIB=$R/pygho_amd/_lib/variants/$v/libpygho_hip.so; else unset PYGHO_AMD_LIB; fi
  o=$R/gpurun_out/dwko_$v; rm -rf $o; mkdir -p $o
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $o/t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-regimes --no-configs > $o/line.json 2> $o/err.log
  python3 - "$v" $o <<'PY'
import csv, glob, json, sys, statistics
v, o = sys.argv[1], sys.argv[2]
f = glob.glob(o + "/t/**/*kernel_trace.csv", recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("bn_bwd_linear_dw", "rowblock_linear_kernel<pygho::bf16, 128, 2, 2>", "rowblock_linear_kernel<pygho::bf16, 128, 1, 2>", "rowblock_linear_kernel<pygho::bf16, 128, 0, 0>"):
        if key in n:
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if us > 60: d.setdefault(key[-28:], []).append(us)
line = json.loads(open(o + "/line.json").read().strip().splitlines()[-1])
print(json.dumps({"variant": v, "ms_per_step": round(line["ms_per_step"], 3), "median_us": {k: round(statistics.median(x), 1) for k, x in d.items()}}))
PY
  rm -rf $o/t
done

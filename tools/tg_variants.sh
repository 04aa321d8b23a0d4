#!/bin/bash
# tools/tg_variants.sh <variant> ...: kernel durations (avg / min / max us) of the small-table gradient path per variant build
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != base ]; then export PYGHO_AMD_LIB=$R/pygho_amd/_lib/variants/$v/libpygho_hip.so; else unset PYGHO_AMD_LIB; fi
  rm -rf $R/gpurun_out/tgprof
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tgprof -- python3 $R/tools/table_grad_bench.py > /dev/null 2>&1
  python3 - $v $(find $R/gpurun_out/tgprof -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Name"]
    if "table_grad" in n or "sum_blocks" in n:
        print(sys.argv[1], n[12:60].ljust(50), r["Calls"], "avg %.1f min %.1f max %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf $R/gpurun_out/tgprof
done

#!/bin/bash
# tools/dw_variants.sh <variant> ...: per variant build ("base" = the main library), kernel time and LDS bank conflicts of
# bn_bwd_linear_dw / rowblock_linear inside the training step (rocprofv3 kernel stats, then one PMC pass)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != base ]; then export PYGHO_AMD_LIB=$R/pygho_amd/_lib/variants/$v/libpygho_hip.so; else unset PYGHO_AMD_LIB; fi
  o=$R/gpurun_out/dwv_$v; rm -rf $o; mkdir -p $o
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-regimes --no-configs > $o/line.json 2> $o/err.log
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $o/pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-regimes --no-configs > /dev/null 2>> $o/err.log
  python3 - "$v" $o <<'PY'
import csv, glob, json, sys, collections
v, o = sys.argv[1], sys.argv[2]
st = glob.glob(o + "/stats/**/*kernel_stats.csv", recursive=True)[0]
res = {}
for r in csv.DictReader(open(st)):
    n = r["Name"]
    if "bn_bwd_linear_dw" in n or "rowblock_linear_kernel" in n:
        res[n.split("(")[0][-48:]] = {"avg_us": round(float(r["AverageNs"]) / 1e3, 1), "max_us": round(float(r["MaxNs"]) / 1e3, 1), "calls": int(r["Calls"])}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bn_bwd_linear_dw" in n or "rowblock_linear_kernel" in n:
            acc[n.split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    for c, vals in cs.items():
        res.setdefault(k, {})[c + "_max"] = max(vals)
line = json.loads(open(o + "/line.json").read().strip().splitlines()[-1])
print(json.dumps({"variant": v, "ms_per_step": round(line["ms_per_step"], 3), "kernels": res}))
PY
done
find $R/gpurun_out -path "*dwv_*" -name "*.csv" -size +1M -delete

#!/bin/bash
# tools/step_kernels.sh <tag> [env assignments...]: rocprofv3 kernel stats of the training step, per-step totals of every kernel > 8 us
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
o=$R/gpurun_out/sk_$tag; rm -rf $o; mkdir -p $o
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-regimes --no-configs > $o/line.json 2> $o/err.log
python3 - $o $tag <<'PY'
import csv, glob, json, sys
o, tag = sys.argv[1], sys.argv[2]
st = glob.glob(o + "/stats/**/*kernel_stats.csv", recursive=True)[0]
line = json.loads(open(o + "/line.json").read().strip().splitlines()[-1])
print(f"== {tag}: {line['ms_per_step']:.3f} ms/step")
for r in csv.DictReader(open(st)):
    per_step = float(r["TotalDurationNs"]) / 1e3 / 12
    if per_step > 30: print(f"  {per_step:8.1f} us/step  calls/step {int(r['Calls'])/12:5.1f}  avg {float(r['AverageNs'])/1e3:7.1f}  {r['Name'][:100]}")
PY
find $o -name "*.csv" -size +1M -delete

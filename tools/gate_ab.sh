#!/bin/bash
# the fresh-batch step with the next batch's collate kernel gated behind the readout of the step in flight, against ungated, and the resident batch
cd $GRAFT_REPO_ROOT
for g in none readout none readout; do
  echo "== fresh, --collate-gate $g"
  python3 bench.py --collate-gate $g --steps 40 --warmup 10 --no-cpu-baseline --no-regimes --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], {k: round(v['avg_ms'],4) for k,v in d['kernels'].items() if 'dual' in k or 'fused' in k})"
done
echo "== resident"
python3 bench.py --resident-batch --steps 40 --warmup 10 --no-cpu-baseline --no-regimes --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"

#!/bin/bash
# the training step with the fused backward in its table-gradient form on / off (PYGHO_DUAL_TABLE_GRAD), fresh batches, same box
cd $GRAFT_REPO_ROOT
for f in 0 1 0 1; do
  echo "== fresh, PYGHO_DUAL_TABLE_GRAD=$f"
  PYGHO_DUAL_TABLE_GRAD=$f python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-regimes --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],4), {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done

#!/bin/bash
# the whole training step with and without the fused block forward (PYGHO_FUSED_FWD), resident batch and fresh batches, same box
cd $GRAFT_REPO_ROOT
for f in 0 1 0 1; do
  echo "== resident, PYGHO_FUSED_FWD=$f"
  PYGHO_FUSED_FWD=$f python3 bench.py --resident-batch --steps 30 --warmup 8 --no-cpu-baseline --no-regimes --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['frac'], {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
for f in 0 1; do
  echo "== fresh, PYGHO_FUSED_FWD=$f"
  PYGHO_FUSED_FWD=$f python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-regimes --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['frac'])"
done

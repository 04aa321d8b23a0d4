#!/bin/bash
# the whole training step with and without the fused backward (PYGHO_DUAL_BWD), fresh batches, same box; then kernel variants
cd $GRAFT_REPO_ROOT
for f in 0 1 0 1; do
  echo "== fresh, PYGHO_DUAL_BWD=$f"
  PYGHO_DUAL_BWD=$f python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-regimes --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
echo "== dual kernel alone, default build"
python3 tools/dual_one.py res 20
for v in "$@"; do
  echo "== dual kernel alone, variant $v"
  PYGHO_AMD_LIB=$GRAFT_REPO_ROOT/pygho_amd/_lib/variants/$v/libpygho_hip.so python3 tools/dual_one.py res 20
done

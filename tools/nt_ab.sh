#!/bin/bash
# cache-policy A/B of the fused forward / fused backward row traffic (nt bits on the loads, the stores, both): whole step, fresh
# batches, same box.  usage: nt_ab.sh <variant>...   (variants built with pygho_amd.build.build(variant=..., defines=...))
cd $GRAFT_REPO_ROOT
run() {
  python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-regimes --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
}
for rep in 1 2; do
  echo "== default"; run
  for v in "$@"; do
    echo "== variant $v"
    PYGHO_AMD_LIB=$GRAFT_REPO_ROOT/pygho_amd/_lib/variants/$v/libpygho_hip.so run
  done
done

#!/bin/bash
# what bench.py's own HIP-event timing of the aggregation launches costs the step it measures: events in every step of the timed region
# (rounds 1-5), in every 4th (shipped), in the first one only; fresh batches, same box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for n in 1 4 1000; do
    echo "== --event-steps $n"
    python3 bench.py --steps 40 --warmup 10 --event-steps $n --no-cpu-baseline --no-regimes --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['launches'], round(r['avg_ms'],4), round(r['frac'],4), {k: (v['launches'], round(v['avg_ms'],4)) for k,v in r['other'].items()})"
  done
done

#!/bin/bash
# I2Conv layer (config 5, d = 256) with and without the width-256 streaming kernels, same box
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  echo "PYGHO_ROWBLOCK_256=$v"
  PYGHO_ROWBLOCK_256=$v python3 - <<'PY'
import sys, json, torch
sys.path.insert(0, "tools")
import bench_layers
dev = torch.device("cuda:0")
r = bench_layers.case("I2Conv", 2048, dev)
print(json.dumps({k: r[k] for k in ("ms", "graphs", "tuples", "d")}))
PY
done

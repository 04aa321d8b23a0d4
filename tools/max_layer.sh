#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, json, torch
sys.path.insert(0, "tools")
import bench_layers
r = bench_layers.case("NGNNConv", 8192, torch.device("cuda:0"), aggr="max", kernels=True)
print(json.dumps({"ms": r["ms"], "kernels": {k: {kk: round(vv, 4) if isinstance(vv, float) else vv for kk, vv in v.items()} for k, v in r["kernels"].items() if k.startswith(("seg_ext", "seg_gmr"))}}))
PY

#!/usr/bin/env python3
"""The by-edge gradient launch on the scatter kernel (csrc/seg_scatter.hip) at BASELINE size, N launches (mean by HIP events): the
command of the --pmc passes (tools/pmc_one.sh).  usage: scatter_one.py <zinc|i2> <plain|res> [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth          # noqa: E402
from pygho_amd import segment as S         # noqa: E402

kind, mode = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
d, graphs, key = (256, 2048, "X___X___2___A___0") if kind == "i2" else (128, 8192, "X___X___1___A___0")
hb = synth.make_batch(graphs, kind, seed=1000)
acd = torch.from_numpy(hb.acd[key]).to(dev)
nt, ne = hb.num_tuples, hb.num_edges
plan = _ops.message_plan(acd, nt, nt, ne)
g = torch.randn(nt, d, device=dev).to(torch.bfloat16)
h = torch.randn(nt, d, device=dev).to(torch.bfloat16)
add = torch.randn(ne, d, device=dev).to(torch.bfloat16) if mode == "res" else None
assert S.scatter_plan(plan) is not None
for _ in range(3):
    S.by_edge_product(plan, g, h, None, add)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    S.by_edge_product(plan, g, h, None, add)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nbytes = 2 * d * (2 * nt + ne * (2 if add is not None else 1)) + 8 * plan.m + 4 * (ne + 1)
print(json.dumps({"kind": kind, "mode": mode, "ms": ms, "frac": nbytes / ms / 1e6 / 8000, "alg_GB": nbytes / 1e9}))

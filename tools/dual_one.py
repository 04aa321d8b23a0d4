#!/usr/bin/env python3
"""The fused backward launch (csrc/seg_dual.hip) at BASELINE size, N launches (mean by HIP events): the command of the --pmc passes
(tools/pmc_one.sh) and of the variant A/B.  usage: dual_one.py <plain|res|tg> [reps]   (tg: the table-gradient form the training
step runs -- no per-edge rows at all)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth          # noqa: E402
from pygho_amd import segment as S         # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "res"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
d, graphs, key = 128, 8192, "X___X___1___A___0"
hb = synth.make_batch(graphs, "zinc", seed=1000)
acd = torch.from_numpy(hb.acd[key]).to(dev)
nt, ne = hb.num_tuples, hb.num_edges
plan = _ops.message_plan(acd, nt, nt, ne)
ea = torch.from_numpy(hb.edge_attr).to(dev).long()
g = torch.randn(nt, d, device=dev).to(torch.bfloat16)
h = torch.randn(nt, d, device=dev).to(torch.bfloat16)
table = torch.randn(16, d, device=dev).to(torch.bfloat16)
add = torch.randn(ne, d, device=dev).to(torch.bfloat16) if mode == "res" else None
look_fwd, look_byc = plan.lookup(_ops.flat_index(ea))[:2]
sp = S.scatter_plan(plan)
assert sp is not None and sp.cgap is not None
if mode == "tg":
    run = lambda: S.dual_backward_tg(plan, g, h, table, look_fwd, look_byc)
    nbytes = 2 * d * 3 * nt + 16 * plan.m + 4 * (nt + 1) + 20 * sp.n_chunks
else:
    run = lambda: S.dual_backward(plan, g, h, table, look_byc, add)
    nbytes = 2 * d * (3 * nt + ne * (2 if add is not None else 1)) + 12 * plan.m + 4 * (nt + 1) + 20 * sp.n_chunks
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(json.dumps({"mode": mode, "ms": ms, "frac": nbytes / ms / 1e6 / 8000, "has_to_move_GB": nbytes / 1e9, "chunks": sp.n_chunks}))

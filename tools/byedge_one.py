#!/usr/bin/env python3
"""The by-edge gradient launch of the 2-tuple product at the ZINC shape (8192 graphs, d = 128 bf16) as seg_gmr dispatches it,
N launches (median by HIP events).  usage: byedge_one.py gather <plain|addend> [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pygho_amd import _ops, synth  # noqa: E402
from tile_ab import timed  # noqa: E402

kern, mode = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
d = 128
hb = synth.replicate(synth.make_batch(1024, "zinc", seed=1), 8)
acd = torch.from_numpy(hb.acd["X___X___1___A___0"]).to(dev)
nt, ne = hb.num_tuples, hb.num_edges
G = torch.randn(nt, d, device=dev).to(torch.bfloat16)
H = torch.randn(nt, d, device=dev).to(torch.bfloat16)
R = torch.randn(ne, d, device=dev).to(torch.bfloat16) if mode == "addend" else None
plan = _ops.message_plan(acd, nt, nt, ne)
p, a_g, c_g = plan.by_d()
fn = lambda: _ops.seg_gmr(ne, G, H, p.seg_ptr, a_g, c_g, "sum", None, addend=R)
ms = timed(fn, reps)
nbytes = 2 * d * (2 * nt + ne * (2 if R is not None else 1)) + 8 * plan.m + 4 * (ne + 1)
print(json.dumps({"kernel": kern, "mode": mode, "ms": ms, "frac": nbytes / ms / 1e6 / 8000, "alg_GB": nbytes / 1e9}))

#!/usr/bin/env python3
"""The forward of the tuple initialisation (unit_triple_kernel) at the ZINC shape, 8192 graphs, d = 128 bf16: median launch time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pygho_amd import _ops, synth  # noqa: E402
from tile_ab import timed  # noqa: E402

dev = torch.device("cuda:0")
d = 128
hb = synth.replicate(synth.make_batch(1024, "zinc", seed=1), 8)
row, col = torch.from_numpy(hb.tupleid[0]).to(dev), torch.from_numpy(hb.tupleid[1]).to(dev)
tf = torch.from_numpy(hb.tuplefeat).to(dev).flatten()
left = torch.randn(hb.num_nodes, d, device=dev).to(torch.bfloat16)
right = torch.randn(hb.num_nodes, d, device=dev).to(torch.bfloat16)
table = torch.randn(16, d, device=dev).to(torch.bfloat16)
ref = ((left[row].float() * right[col].float()) * table[tf].float()).to(torch.bfloat16)     # the kernel's order, rounded once
out = _ops.pair_product(left, right, table, row, col, tf)
torch.cuda.synchronize()
ms = timed(lambda: _ops.pair_product(left, right, table, row, col, tf), 30)
nbytes = hb.num_tuples * (d * 2 + 12)
print(json.dumps({"tuples": hb.num_tuples, "ms": ms, "TBps_written_plus_indices": nbytes / ms / 1e9, "equal": bool(torch.equal(out, ref))}))

#!/usr/bin/env python3
"""The eager small-batch regime of bench.py on its own (128 graphs by default): ms per full train step, wall clock over 200 steps,
three repetitions.  usage: eager_small.py [graphs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth  # noqa: E402
from pygho_amd.ngnn import SpModel  # noqa: E402

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
dd = synth.to_datadict(synth.make_batch(graphs, "zinc", seed=7), dev)
torch.manual_seed(0)
model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(dd)
    loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
    loss.backward()
    opt.step()
    return loss


for _ in range(10):
    step()
res = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        loss = step()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 200 * 1e3)
print({"graphs": graphs, "arena_linear": os.environ.get("PYGHO_ARENA_LINEAR", "1"), "defer_counters": os.environ.get("PYGHO_DEFER_COUNTERS", "1"),
       "eager_ms_per_step": [round(r, 3) for r in res], "loss": float(loss)})

#!/usr/bin/env python3
"""Host-side profile (cProfile, tottime) of the eager 128-graph training step: where the Python time of the launch-bound regime goes."""
import cProfile
import os
import pstats
import sys

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


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)

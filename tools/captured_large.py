#!/usr/bin/env python3
"""Does capturing the WHOLE training step into a HIP graph help at the benchmark size (8192 graphs), where the step is GPU-bound?
Eager and captured ms per step on the same resident batch, same model, HIP events over 30 steps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth  # noqa: E402
from pygho_amd.graphs import GraphedStep  # noqa: E402
from pygho_amd.ngnn import SpModel  # noqa: E402

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
dd = synth.to_datadict(synth.make_batch(graphs, "zinc", seed=7), dev)
y = dd["y"].unsqueeze(-1)
for mode in ("eager", "hipgraph", "eager", "hipgraph"):
    torch.manual_seed(0)
    model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
    opt = (torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True) if mode == "eager"
           else torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True))

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(y, pred.float())
        loss.backward()
        opt.step()
        return loss.detach()
    model.prepare(dd)                        # every plan of the batch, the by-edge scatter plans included (the dispatcher never plans)
    if mode == "eager":
        run = step
        for _ in range(16):
            run()
    else:
        for _ in range(4):
            step()
        gs = GraphedStep(step, warmup=2)
        run = gs.replay
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        loss = run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{graphs} graphs {mode:9s} {e0.elapsed_time(e1) / 30:.3f} ms per step, loss {float(loss):.5f}", flush=True)

#!/usr/bin/env python3
"""Which host code issues the small ATen launches of a training step (casts, fills, scalar adds): torch.profiler with stacks over
one eager step of the 128-graph regime; prints, per ATen op that launches a kernel, the innermost pygho_amd / bench frame."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth  # noqa: E402
from pygho_amd.ngnn import SpModel  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dd = synth.to_datadict(synth.make_batch(graphs, "zinc", seed=7), dev)
    torch.manual_seed(0)
    model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)

    def step():                                           # = bench.py's eager small-batch regime
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True,
                                experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
        torch.cuda.synchronize()
    by = collections.Counter()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        if not ev.kernels:
            continue
        frame = next((s for s in ev.stack if "pygho_amd" in s or "small_launches.py" in s), ev.stack[0] if ev.stack else "?")
        by[(ev.name, frame.strip()[-90:])] += 1
    for (name, frame), n in by.most_common(60):
        print(f"{n:4d}  {name:28s} {frame}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Which host-side work a training step does on a FRESH collated batch that it does not do on a resident one: torch.profiler CPU
self-times of both, side by side (128-graph ZINC-shape batches, the launch-bound regime)."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth  # noqa: E402
from pygho_amd.collate import DeviceGraphStore  # noqa: E402
from pygho_amd.ngnn import SpModel  # noqa: E402

KEY = "X___X___1___A___0"
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(1024)], dev)
torch.manual_seed(0)
model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
gen = torch.Generator().manual_seed(0)


def step(dd):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(dd)
    torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float()).backward()
    opt.step()


fresh = [store.collate(torch.randperm(1024, generator=gen)[:graphs]) for _ in range(30)]
for dd in fresh[:10]:
    step(dd)
torch.cuda.synchronize()
res = {}
for name, batches in (("fresh", fresh[10:]), ("resident", [fresh[0]] * 20)):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as pr:
        for dd in batches:
            step(dd)
        torch.cuda.synchronize()
    res[name] = {e.key: (e.count / 20, e.self_cpu_time_total / 20) for e in pr.key_averages()}
keys = sorted(set(res["fresh"]) | set(res["resident"]), key=lambda k: -(res["fresh"].get(k, (0, 0))[1] - res["resident"].get(k, (0, 0))[1]))
print(f"{'op':70s} {'fresh n':>8s} {'us':>8s} {'resid n':>8s} {'us':>8s}")
for k in keys[:40]:
    f, r = res["fresh"].get(k, (0, 0)), res["resident"].get(k, (0, 0))
    print(f"{k[:70]:70s} {f[0]:8.1f} {f[1]:8.1f} {r[0]:8.1f} {r[1]:8.1f}")
tot = {n: sum(v[1] for v in d.values()) for n, d in res.items()}
print("total self CPU us per step:", {n: round(v, 1) for n, v in tot.items()})

#!/usr/bin/env python3
"""Host time of the fresh-batch training loop at small batch sizes, split into its three parts: `DeviceGraphStore.collate`,
`SpModel.prepare` (plans) and the training step, each timed with a device synchronisation after it (so the figures are
host issue time + the GPU tail, not overlapped).  `--cprofile PART` prints the top host functions of one part."""
import argparse
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth  # noqa: E402
from pygho_amd.collate import DeviceGraphStore  # noqa: E402
from pygho_amd.ngnn import SpModel  # noqa: E402

KEY = "X___X___1___A___0"

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=128)
ap.add_argument("--reps", type=int, default=60)
ap.add_argument("--cprofile", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
recs = [synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(1024)]
store = DeviceGraphStore(recs * 4, dev)
torch.manual_seed(0)
model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
gen = torch.Generator().manual_seed(0)


def step(dd):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(dd)
    loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
    loss.backward()
    opt.step()


t = {"collate": 0.0, "prepare": 0.0, "step": 0.0}
prof = cProfile.Profile() if args.cprofile else None
for k in range(args.reps + 10):
    ids = torch.randperm(store.num_graphs, generator=gen)[:args.graphs]
    timed = k >= 10
    for part in ("collate", "prepare", "step"):
        torch.cuda.synchronize()
        if prof is not None and timed and part == args.cprofile:
            prof.enable()
        t0 = time.perf_counter()
        if part == "collate":
            with _ops.deferred_index_checks():
                dd = store.collate(ids)
        elif part == "prepare":
            if os.environ.get("PREPARE", "0") == "1":
                with _ops.deferred_index_checks():
                    model.prepare(dd)
        else:
            step(dd)
        torch.cuda.synchronize()
        if prof is not None and timed and part == args.cprofile:
            prof.disable()
        if timed:
            t[part] += time.perf_counter() - t0
print({k: round(v / args.reps * 1e3, 3) for k, v in t.items()}, "ms per batch,", args.graphs, "graphs")
if prof is not None:
    pstats.Stats(prof).sort_stats("tottime").print_stats(40)

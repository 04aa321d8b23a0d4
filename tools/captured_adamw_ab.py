#!/usr/bin/env python3
"""The captured slot step (fresh batch every replay) with torch's AdamW in its two capturable implementations: foreach (the default
when only `capturable=True` is given: ~20 multi-tensor launches per step) and fused (one launch).  128 and 1024 graphs, 50 replays,
host clock around the loop with a device synchronisation."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth  # noqa: E402
from pygho_amd.collate import DeviceGraphStore  # noqa: E402
from pygho_amd.graphs import SlotStep  # noqa: E402
from pygho_amd.ngnn import SpModel  # noqa: E402

KEY = "X___X___1___A___0"
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
recs = [synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(2048)]
store = DeviceGraphStore(recs * 2, dev)
for graphs in (128, 1024):
    batches = [np.random.default_rng(s).permutation(store.num_graphs)[:graphs] for s in range(60)]
    for kind in ("foreach", "fused", "foreach", "fused"):
        torch.manual_seed(0)
        model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, **({"fused": True} if kind == "fused" else {}))

        def step(dd):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            opt.step()
            return loss.detach()
        ss = SlotStep(store, graphs, step, warmup=3)
        for ids in batches[:10]:
            loss = ss.run(ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ids in batches[10:]:
            loss = ss.run(ids)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 50 * 1e3
        print(f"{graphs} graphs, AdamW {kind}: {ms:.4f} ms per captured fresh-batch step, loss {float(loss):.6f}, replays {ss.replays}, eager fallbacks {ss.eager_steps}", flush=True)

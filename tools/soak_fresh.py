#!/usr/bin/env python3
"""Soak of the fresh-batch training loop: many steps through DeviceGraphStore + BatchPrefetcher at 128 and 8192 graphs; reports
throughput, finite losses, allocator drift (bytes allocated / reserved at the start and the end) and the length of the deferred
range-flag list (must stay bounded).  `--captured`: the same soak through ONE captured step over a fixed-capacity batch slot
(pygho_amd.graphs.SlotStep) at 128 and 1024 graphs, 3000 fresh batches each."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth  # noqa: E402
from pygho_amd.collate import BatchPrefetcher, DeviceGraphStore  # noqa: E402
from pygho_amd.ngnn import SpModel  # noqa: E402
from pygho_amd.plans import _PENDING_ERRORS  # noqa: E402

KEY = "X___X___1___A___0"
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
recs = [synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(2048)]
CAPTURED = "--captured" in sys.argv
if CAPTURED:
    from pygho_amd.graphs import SlotStep  # noqa: E402
    for graphs, steps in ((128, 3000), (1024, 3000)):
        store = DeviceGraphStore(recs * 2, dev)
        torch.manual_seed(0)
        model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True)

        def step(dd):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            opt.step()
            return loss.detach()
        ss = SlotStep(store, graphs, step)
        idr = np.random.default_rng(0)
        losses, mem0, t0 = torch.zeros(steps, device=dev), None, None
        for k in range(steps):
            if k == 50:
                torch.cuda.synchronize()
                mem0, t0 = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved()), time.perf_counter()
            losses[k] = ss.run(idr.permutation(store.num_graphs)[:graphs])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print({"mode": "captured slot step", "graphs": graphs, "steps": steps, "ms_per_step": round(dt / (steps - 50) * 1e3, 3),
               "replays": ss.replays, "eager_fallbacks": ss.eager_steps, "non_finite_losses": int((~torch.isfinite(losses)).sum()),
               "first_loss": round(float(losses[0]), 4), "last_loss": round(float(losses[-20:].mean()), 4),
               "allocated_MB_start_end": [round(mem0[0] / 2 ** 20), round(torch.cuda.memory_allocated() / 2 ** 20)],
               "reserved_MB_start_end": [round(mem0[1] / 2 ** 20), round(torch.cuda.memory_reserved() / 2 ** 20)],
               "pending_flags": len(_PENDING_ERRORS), "planner_fetches_total": _ops.FETCHES[0]}, flush=True)
        del ss, store, model, opt
        _ops.check_deferred_errors()
    sys.exit(0)
for graphs, steps, reps in ((128, 3000, 1), (8192, 150, 8)):
    store = DeviceGraphStore(recs * reps, dev)
    torch.manual_seed(0)
    model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
    gen = torch.Generator().manual_seed(0)
    ids = (torch.randperm(store.num_graphs, generator=gen)[:graphs] for _ in range(steps))
    bad, n, mem0, t0 = 0, 0, None, None
    losses = []
    for k, dd in enumerate(BatchPrefetcher(store, ids)):
        if k == 50:
            torch.cuda.synchronize()
            mem0, t0 = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved()), time.perf_counter()
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        opt.step()
        losses.append(loss.detach())
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ls = torch.stack(losses)
    print({"graphs": graphs, "steps": n, "ms_per_step": round(dt / (n - 50) * 1e3, 3), "non_finite_losses": int((~torch.isfinite(ls)).sum()),
           "first_loss": round(float(ls[0]), 4), "last_loss": round(float(ls[-20:].mean()), 4),
           "allocated_MB_start_end": [round(mem0[0] / 2 ** 20), round(torch.cuda.memory_allocated() / 2 ** 20)],
           "reserved_MB_start_end": [round(mem0[1] / 2 ** 20), round(torch.cuda.memory_reserved() / 2 ** 20)],
           "pending_flags": len(_PENDING_ERRORS), "planner_fetches_total": _ops.FETCHES[0]}, flush=True)
    del store, model, opt
    _ops.check_deferred_errors()

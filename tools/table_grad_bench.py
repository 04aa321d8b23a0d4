#!/usr/bin/env python3
"""The plan-free small-table gradient (csrc/table_grad.hip) against the planned hierarchical segment reduction, at the shapes of the
8192-graph and 128-graph NGNN steps: microseconds per call (HIP events, 50 repetitions, launches of the whole path)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops  # noqa: E402

dev = torch.device("cuda:0")
for m, n_table in ((410_000, 16), (190_000, 32), (52_000, 16), (24_000, 32), (6_400, 16), (3_000, 32)):
    g = torch.randn(m, 128, device=dev).to(torch.bfloat16)
    idx = torch.randint(0, min(n_table, 28), (m,), device=dev)
    plan = _ops.cached_plan(idx, n_table, "scatter")
    for name, fn in (("table_grad", lambda: _ops.table_grad(g, idx, n_table)), ("planned", lambda: _ops.seg_reduce_rows(g, plan, "sum"))):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"m {m:7d} n_table {n_table:3d} {name:10s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us")

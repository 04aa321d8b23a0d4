#!/usr/bin/env python3
"""Kernel-level time table (torch profiler) of one training step of a 6-layer example/zinc.py model, either layout.

    python tools/profile_model.py --conv SSWL --layout dense [--graphs 1024] [--rows 45]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth                                    # noqa: E402
from pygho_amd.honn.SpOperator import parse_precomputekey       # noqa: E402
from pygho_amd.models import MaModel, SpModel                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--conv", default="NGNN")
    ap.add_argument("--layout", default="dense", choices=("dense", "sparse"))
    ap.add_argument("--graphs", type=int, default=1024)
    ap.add_argument("--rows", type=int, default=45)
    ap.add_argument("--f32-act", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    act = None if args.f32_act else torch.bfloat16
    if args.layout == "dense":
        model = MaModel(args.conv, num_layer=6, hiddim=128, act_dtype=act).to(dev)
        hb = synth.make_batch(args.graphs, "zinc", seed=11)
        dd = synth.to_dense_datadict(hb, dev)
    else:
        kind = "i2" if args.conv == "I2GNN" else "zinc"
        model = SpModel(args.conv, num_layer=6, hiddim=128, act_dtype=act).to(dev)
        hb = synth.make_batch(min(args.graphs, 1024), kind, seed=11, keys=tuple(parse_precomputekey(model)))
        if args.graphs > 1024:
            hb = synth.replicate(hb, args.graphs // 1024)
        dd = synth.to_datadict(hb, dev, kind)
    y = dd["y"].unsqueeze(-1)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dict(dd))
        torch.nn.functional.l1_loss(y, pred.float()).backward()
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as pr:
        for _ in range(4):
            step()
        torch.cuda.synchronize()
    print(pr.key_averages().table(sort_by="cuda_time_total", row_limit=args.rows, max_name_column_width=100))


if __name__ == "__main__":
    main()

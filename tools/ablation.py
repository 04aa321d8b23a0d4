#!/usr/bin/env python3
"""What each switchable piece of the benchmarked NGNN step is worth: the step of bench.py (8192 ZINC-shape graphs, 6 layers,
hidden 128, bf16) timed with one module switch of pygho_amd._ops turned off at a time.

    python tools/ablation.py [--graphs 8192] > profiles/rNN_ablation.jsonl
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth                               # noqa: E402
from pygho_amd.ngnn import SpModel                              # noqa: E402
from pygho_amd.parallel import FlatGradSync                     # noqa: E402


def run(dd, dev, fused_adamw=True, steps=12, warmup=4):
    torch.manual_seed(0)
    model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
    sync = FlatGradSync(model.parameters())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=fused_adamw)
    y = dd["y"].unsqueeze(-1)

    def step():
        sync.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        torch.nn.functional.l1_loss(y, pred.float()).backward()
        sync.sync()
        opt.step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def layers(dev):
    """the other layers: one forward + backward of SSWLConv / SUNConv (sparse, 8192 graphs), SUNConv on the padded layout
    (1024 graphs) and I2Conv (2048 3-tuple graphs, hidden 256), each with its own switches off one at a time"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_layers
    import bench_ops
    cases = [
        ("SSWLConv SS", lambda: bench_layers.case("SSWLConv", 8192, dev)["ms"], ("USE_SSWL_BLOCK", "USE_CONCAT_BLOCK")),
        ("SUNConv SS", lambda: bench_layers.case("SUNConv", 8192, dev)["ms"], ("USE_PAIR_COMBINE",)),
        ("SUNConv DD (1024, 37, 37, 128)", lambda: bench_ops.sunconv_case(1024, 37, 128, torch.bfloat16, dev)["ms"],
         ("USE_PAIR_COMBINE", "USE_BMM_LISTS", "USE_BMM_EXTENTS")),
        ("I2Conv SS hidden 256", lambda: bench_layers.case("I2Conv", 2048, dev)["ms"], ("USE_SEG_WINDOW",)),
    ]
    for name, fn, flags in cases:
        base = fn()
        print(json.dumps({"layer": name, "config": "all on", "ms": base}), flush=True)
        for flag in flags:
            setattr(_ops, flag, False)
            try:
                ms = fn()
            finally:
                setattr(_ops, flag, True)
            print(json.dumps({"layer": name, "config": f"{flag} = False", "ms": ms, "delta_ms": ms - base}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=8192)
    ap.add_argument("--layers", action="store_true", help="the per-layer switches instead of the benchmarked step")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.layers:
        layers(dev)
        return
    hb = synth.replicate(synth.make_batch(1024, "zinc", seed=1000), args.graphs // 1024)
    dd = synth.to_datadict(hb, dev)
    base = run(dd, dev)
    print(json.dumps({"config": "all on (bench.py default)", "graphs": hb.num_graphs, "ms": base, "graphs_per_s": hb.num_graphs / base * 1e3}), flush=True)
    ms = run(dd, dev, fused_adamw=False)
    print(json.dumps({"config": "multi-tensor (foreach) AdamW instead of the fused one", "ms": ms, "delta_ms": ms - base}), flush=True)
    for flag in ("USE_ADJ_TABLE", "USE_TABLE_PRODUCT", "USE_FUSED_DW", "USE_BN_BWD_LINEAR", "USE_ROWBLOCK_LINEAR"):
        setattr(_ops, flag, False)
        try:
            ms = run(dd, dev)
        finally:
            setattr(_ops, flag, True)
        print(json.dumps({"config": f"{flag} = False", "ms": ms, "delta_ms": ms - base}), flush=True)
    again = run(dd, dev)
    print(json.dumps({"config": "all on, repeated at the end", "ms": again, "delta_ms": again - base}), flush=True)


if __name__ == "__main__":
    main()

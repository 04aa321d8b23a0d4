#!/usr/bin/env python3
"""Bisect of the multi-capture caveat of pygho_amd/graphs.py WITH THE PACKAGE'S OWN STEP: N captured SpModel training steps (one per
fixed mini-batch) sharing one model and ONE capturable AdamW, an eager kernel between replays, no device synchronisation.  Counts
non-finite losses over `epochs` passes; `--variants` re-runs itself in child processes with pieces replaced (environment switches of
the package) and prints one line per variant.

    python tools/bisect_graph_nan.py [--graphs-captured 4] [--epochs 50] [--sync] [--eager full|pygho|none] [--variants]
"""
import argparse
import json
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args):
    from pygho_amd import _ops, synth
    from pygho_amd.graphs import GraphedStep
    from pygho_amd.ngnn import SpModel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    act = None if args.f32 else torch.bfloat16
    model = SpModel(1, args.layers, args.hidden, act_dtype=act).to(dev)
    if args.stock:                                          # these Linear modules take torch's stock (autocast) path
        for name, m in model.named_modules():
            if isinstance(m, torch.nn.Linear) and any(s in name for s in args.stock.split(",")):
                m.__dict__["_pygho_stock"] = True
                print("stock:", name, file=sys.stderr)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
    dds = [synth.to_datadict(synth.make_batch(args.batch, "zinc", seed=100 + k), dev) for k in range(args.graphs_captured)]

    def make_step(dd):
        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=act is not None):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            opt.step()
            return loss.detach()
        return step

    if args.no_capture:                                     # the same steps, eagerly: is it capture at all?
        fns = [make_step(dd) for dd in dds]
        bad = total = 0
        first_bad = None
        for epoch in range(args.epochs + 3):
            for fn in fns:
                loss = fn()
                total += 1
                ok = bool(torch.isfinite(loss))
                bad += not ok
                if first_bad is None and not ok:
                    first_bad = epoch
        return {"captured": 0, "eager_steps": total, "non_finite": bad, "first_bad_epoch": first_bad,
                "params_non_finite": sum(int(not bool(torch.isfinite(p).all())) for p in model.parameters())}
    steps = [GraphedStep(make_step(dd), warmup=3) for dd in dds]
    bad, total = 0, 0
    scratch = torch.zeros(1024, device=dev)
    first_bad = None
    for epoch in range(args.epochs):
        outs = []
        for k, gs in enumerate(steps):
            out = gs.replay()
            if args.eager == "full":
                _ = torch.full((1,), 7.0, device=dev)                  # an eager kernel + an eager allocation between two replays
            elif args.eager == "add":
                scratch.add_(1.0)                                      # an eager kernel, no allocation
            elif args.eager == "accum":
                outs.append(out + 0.0)                                 # what a training loop does: device-side statistics of the loss
            if args.sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        vals = [bool(torch.isfinite(gs.output)) for gs in steps]
        total += len(vals)
        bad += sum(not v for v in vals)
        if first_bad is None and not all(vals):
            first_bad = epoch
    pbad = sum(int(not bool(torch.isfinite(p).all())) for p in model.parameters())
    return {"captured": args.graphs_captured, "epochs": args.epochs, "eager": args.eager, "sync": args.sync, "non_finite": bad, "of": total,
            "first_bad_epoch": first_bad, "params_non_finite": pbad}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs-captured", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--sync", action="store_true")
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--eager", default="full", choices=["full", "add", "accum", "none"])
    ap.add_argument("--no-capture", action="store_true")
    ap.add_argument("--stock", default="")
    ap.add_argument("--variants", action="store_true")
    args = ap.parse_args()
    if not args.variants:
        print(json.dumps(run(args)))
        sys.exit(0)
    base = [sys.executable, os.path.abspath(__file__), "--epochs", str(args.epochs)]
    variants = [
        ("base 4 graphs, eager full", [], {}),
        ("base 6 graphs", ["--graphs-captured", "6"], {}),
        ("2 graphs", ["--graphs-captured", "2"], {}),
        ("with device sync", ["--sync"], {}),
        ("eager add (no allocation)", ["--eager", "add"], {}),
        ("eager accum", ["--eager", "accum"], {}),
        ("no eager work", ["--eager", "none"], {}),
        ("f32", ["--f32"], {}),
        ("cast arena off", [], {"PYGHO_CAST_ARENA": "0"}),
        ("deferred counters off", [], {"PYGHO_DEFER_COUNTERS": "0"}),
        ("grad chain off", [], {"PYGHO_GRAD_CHAIN": "0"}),
        ("recompute off", [], {"PYGHO_RECOMPUTE_PRE": "0"}),
        ("pair bwd off", [], {"PYGHO_PAIR_BWD": "0"}),
        ("scatter off", [], {"PYGHO_SEG_SCATTER": "0"}),
        ("arena linear off", [], {"PYGHO_ARENA_LINEAR": "0"}),
    ]
    for name, extra, env in variants:
        r = subprocess.run(base + extra, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else f"rc={r.returncode} {r.stderr[-300:]}"
        print(f"{name:32s} {line}", flush=True)

#!/usr/bin/env python3
"""Plain-torch reductions of the multi-capture caveat of pygho_amd/graphs.py (NO pygho_amd code involved): N captured training steps
(one per fixed mini-batch) sharing one model and ONE capturable AdamW, an eager kernel launched between replays, no device
synchronisation.

    python tools/repro_graph_nan.py [--captured 4] [--sync] [--autocast] [--layers 6] [--rows 3000] [--epochs 50]

Round 4 bisect with the package's own step (tools/bisect_graph_nan.py): the default path shows 0 non-finite losses of 500 replays; the
ONE variant that reproduces the NaN (191 of 200) routes the node-level Linears through torch's autocast `F.linear` (library GEMM + the
casts autocast inserts) instead of the package's arena Linear -- i.e. through plain torch ops.  `--autocast` is that path in isolation:
a stack of nn.Linear under bf16 autocast at the node-level height.  Without `--autocast` (f32 Linear) nothing fails (round 3)."""
import argparse

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--captured", type=int, default=4)
ap.add_argument("--sync", action="store_true")
ap.add_argument("--autocast", action="store_true")
ap.add_argument("--layers", type=int, default=6)
ap.add_argument("--rows", type=int, default=3000)
ap.add_argument("--width", type=int, default=128)
ap.add_argument("--epochs", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
mods = []
for _ in range(args.layers):
    mods += [torch.nn.Linear(args.width, args.width), torch.nn.SiLU()]
model = torch.nn.Sequential(*mods, torch.nn.Linear(args.width, 1)).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
batches = [(torch.randn(args.rows, args.width, device=dev), torch.randn(args.rows, 1, device=dev)) for _ in range(args.captured)]


def make_step(x, y):
    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.autocast):
            pred = model(x)
        loss = torch.nn.functional.mse_loss(pred.float(), y)
        loss.backward()
        opt.step()
        return loss.detach()
    return step


graphs = []
for x, y in batches:
    step = make_step(x, y)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    graphs.append((g, out))
bad, first = 0, None
for epoch in range(args.epochs):
    for g, out in graphs:
        g.replay()
        _ = torch.full((1,), 7.0, device=dev)          # any eager kernel between two replays
        if args.sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    n = sum(int(not bool(torch.isfinite(out))) for _, out in graphs)
    bad += n
    if n and first is None:
        first = epoch
print(f"captured steps: {args.captured}, autocast: {args.autocast}, device sync after eager work: {args.sync}, "
      f"non-finite losses: {bad} of {args.epochs * args.captured}, first bad epoch: {first}")

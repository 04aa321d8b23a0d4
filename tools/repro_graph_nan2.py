#!/usr/bin/env python3
"""Closer plain-torch reduction of the captured-step NaN (tools/bisect_graph_nan.py: it appears when `lin_tupleinit1` or `poolmlp.lins.0`
take torch's stock nn.Linear under bf16 autocast inside a captured step, ONE graph, with or without synchronisation).  Shapes of the
128-graph ZINC batch: 3003 node rows, width 128; an embedding feeds two Linears that share their input, their product is reduced,
then Linear -> BatchNorm1d -> SiLU -> Linear; ONE captured training step with a capturable AdamW, replayed.
    python tools/repro_graph_nan2.py [--rows 3003] [--epochs 60] [--no-capture]"""
import argparse

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=3003)
ap.add_argument("--epochs", type=int, default=60)
ap.add_argument("--no-capture", action="store_true")
ap.add_argument("--no-bn", action="store_true")
ap.add_argument("--f32", action="store_true", help="no autocast")
ap.add_argument("--no-cache", action="store_true", help="torch.autocast(cache_enabled=False)")
ap.add_argument("--eval-bn", action="store_true", help="BatchNorm in eval mode (running statistics, no update)")
ap.add_argument("--no-track", action="store_true", help="BatchNorm1d(track_running_stats=False)")
ap.add_argument("--sgd", action="store_true", help="plain SGD instead of capturable AdamW")
ap.add_argument("--affine", action="store_true", help="a hand-written per-channel affine map (two f32 parameters) in place of BatchNorm1d")
ap.add_argument("--single", action="store_true", help="one Linear in front instead of the product of two")
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--no-miopen", action="store_true", help="torch.backends.cudnn.enabled = False: ATen's native batch-norm kernels")
ap.add_argument("--bn-f32-input", action="store_true", help="cast the BatchNorm input to f32 by hand")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
W = 128
if args.no_miopen:
    torch.backends.cudnn.enabled = False


class M(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.emb = torch.nn.Embedding(32, W)
        self.l0, self.l1 = torch.nn.Linear(W, W), torch.nn.Linear(W, W)
        self.p = torch.nn.Linear(W, W)
        self.bn = torch.nn.Identity() if args.no_bn else torch.nn.BatchNorm1d(W, track_running_stats=not args.no_track)
        self.out = torch.nn.Linear(W, 1)
        self.aw, self.ab = torch.nn.Parameter(torch.ones(W)), torch.nn.Parameter(torch.zeros(W))

    def forward(self, idx):
        x = self.emb(idx) if args.f32 else self.emb(idx).to(torch.bfloat16)
        h = self.l0(x) if args.single else self.l0(x) * self.l1(x)
        h = self.p(h)
        if args.affine:
            h = torch.nn.functional.silu(h * self.aw + self.ab)
        else:
            h = torch.nn.functional.silu(self.bn(h.float() if args.bn_f32_input else h))
        return self.out(h)


model = M().to(dev)
if args.eval_bn:
    model.bn.eval()
opt = torch.optim.SGD(model.parameters(), lr=args.lr) if args.sgd else torch.optim.AdamW(model.parameters(), lr=args.lr, capturable=True)
idx = torch.randint(0, 32, (args.rows,), device=dev)
y = torch.randn(args.rows, 1, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.f32, cache_enabled=not args.no_cache):
        pred = model(idx)
    loss = torch.nn.functional.l1_loss(y, pred.float())
    loss.backward()
    opt.step()
    return loss.detach()


bad = 0
if args.no_capture:
    for _ in range(args.epochs):
        bad += int(not bool(torch.isfinite(step())))
else:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    for _ in range(args.epochs):
        g.replay()
        torch.cuda.synchronize()
        bad += int(not bool(torch.isfinite(out)))
print(f"rows {args.rows}, capture {not args.no_capture}, bn {not args.no_bn}, autocast {not args.f32}, miopen {not args.no_miopen}, "
      f"bn input f32 {args.bn_f32_input}, cache {not args.no_cache}, eval-bn {args.eval_bn}, track {not args.no_track}, sgd {args.sgd}, affine {args.affine}, single {args.single}, lr {args.lr}: non-finite losses {bad} of {args.epochs}")

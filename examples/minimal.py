#!/usr/bin/env python3
"""
The flow of the reference's example/minimal.py (NGNN on ZINC, sparse backend) on this backend, with synthetic ZINC-shape graphs
in place of the dataset (no network here): per-graph preprocessing -> a device-resident graph store -> mini-batches collated ON
THE DEVICE -> 6-layer NGNN (bf16 activations, f32 master weights) -> L1 loss -> AdamW.

    python examples/minimal.py [--graphs 8192] [--batch 2048] [--epochs 3]
    python examples/minimal.py --graphs 4096 --batch 128 --epochs 4        # the reference's batch size: captured steps (below)

Small batches are bound by host issue time, not by the GPU (~135 launches per step: 3.6-4.0 ms eager against 1.2 ms of GPU work at
the reference's batch_size = 128, example/minimal.py:119).  Below --capture-below graphs per batch (default 2048) the whole training
step (device collation, forward, backward, AdamW) is therefore captured ONCE into a HIP graph over a fixed-capacity batch slot
(`pygho_amd.graphs.SlotStep`) and replayed for every batch: the dataset is re-shuffled every epoch like the reference's DataLoader
(example/minimal.py:132-140), a batch costs one small upload + one replay (1.2 ms per 128-graph step), and the results equal the eager
loop's bit for bit.  The last, smaller batch of an epoch (and any batch that would not fit the slot's capacities) runs eagerly.

Reference lines: dataset + Sppretransform (example/minimal.py:100-130) -> synth.make_graph (k-hop tuple sampler and the
precomputed "X___X___1___A___0" message triples, as hodata/SpTupleSampler.py:91-126 + SpData.py:115-171 produce them);
SpDataloader (:132-140) -> collate.DeviceGraphStore; model (:37-85) -> pygho_amd.ngnn.SpModel; train loop (:142-160) -> below.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth                                   # noqa: E402
from pygho_amd.collate import BatchPrefetcher, DeviceGraphStore  # noqa: E402
from pygho_amd.graphs import SlotStep                          # noqa: E402
from pygho_amd.honn.SpOperator import parse_precomputekey      # noqa: E402
from pygho_amd.ngnn import SpModel                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--capture-below", type=int, default=2048, help="batches smaller than this train through captured HIP graphs")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = SpModel(1, 6, args.hidden, act_dtype=torch.bfloat16).to(dev)
    keys = tuple(parse_precomputekey(model))                    # the message plans the layers will look up
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    records = [synth.make_graph(rng, "zinc", 3, keys) for _ in range(args.graphs)]
    for r in records:                                           # a learnable synthetic target: mean atom type
        r.y = float(r.x.mean()) / 10.0
    store = DeviceGraphStore(records, dev)
    print(f"{args.graphs} graphs preprocessed and stored on the device in {time.perf_counter() - t0:.1f} s; keys {keys}")
    if args.batch < args.capture_below:
        return train_captured(args, model, store, dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
    for epoch in range(args.epochs):
        perm = torch.randperm(args.graphs, generator=torch.Generator().manual_seed(epoch))
        tot, nb = 0.0, 0
        t0 = time.perf_counter()
        batches = [perm[i:i + args.batch] for i in range(0, args.graphs, args.batch)]
        # block-diagonal batches built by device kernels, one batch ahead on a side stream; a collated batch arrives with every
        # index plan this model's step asks for (other models: pass `model.prepare` as the third argument)
        for dd in BatchPrefetcher(store, batches):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            opt.step()
            tot += float(loss.detach())
            nb += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"epoch {epoch}: mean L1 {tot / nb:.4f}, {args.graphs / dt:,.0f} graphs/s")


def train_captured(args, model, store, dev):
    """ONE captured training step for every mini-batch: fresh shuffled batches each epoch, as the reference's loader draws them"""
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=True)

    def step(dd):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        opt.step()
        return loss.detach()

    t0 = time.perf_counter()
    ss = SlotStep(store, args.batch, step)                       # 3 warm-up steps, then the capture
    torch.cuda.synchronize()
    print(f"captured one {args.batch}-graph step in {time.perf_counter() - t0:.2f} s; capacities {ss.slot.caps}")
    for epoch in range(args.epochs):
        perm = np.random.default_rng(epoch).permutation(args.graphs)
        batches = [perm[i:i + args.batch] for i in range(0, args.graphs, args.batch)]
        t0 = time.perf_counter()
        losses = [ss.run(ids).clone() for ids in batches]        # (the loss tensor of a replay is static: keep a copy)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mean_loss = float(torch.stack([l.reshape(()) for l in losses]).mean())
        print(f"epoch {epoch}: mean L1 {mean_loss:.4f}, {args.graphs / dt:,.0f} graphs/s, {dt / len(batches) * 1e3:.2f} ms per step "
              f"({ss.replays} replays, {ss.eager_steps} eager steps so far)")


if __name__ == "__main__":
    main()

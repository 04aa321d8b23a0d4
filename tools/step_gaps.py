#!/usr/bin/env python3
"""Idle gaps on the GPU timeline of the benchmarked NGNN training step (torch profiler, kernel events sorted by start time):
total idle time per step and the largest gaps with the kernels on either side.

    python tools/step_gaps.py [--graphs 8192] [--optimizer fused|foreach]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth                                    # noqa: E402
from pygho_amd.ngnn import SpModel                              # noqa: E402
from pygho_amd.parallel import FlatGradSync                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=8192)
    ap.add_argument("--optimizer", default="fused")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hb = synth.replicate(synth.make_batch(1024, "zinc", seed=1000), args.graphs // 1024)
    dd = synth.to_datadict(hb, dev)
    torch.manual_seed(0)
    model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
    sync = FlatGradSync(model.parameters())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=args.optimizer == "fused")
    y = dd["y"].unsqueeze(-1)

    def step():
        sync.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        torch.nn.functional.l1_loss(y, pred.float()).backward()
        sync.sync()
        opt.step()
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    steps = 4
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as pr:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    ev = sorted((e for e in pr.events() if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range.end > e.time_range.start),
                key=lambda e: e.time_range.start)
    busy = sum(e.time_range.end - e.time_range.start for e in ev)
    span = ev[-1].time_range.end - ev[0].time_range.start
    gaps = []
    for a, b in zip(ev, ev[1:]):
        g = b.time_range.start - a.time_range.end
        if g > 0:
            gaps.append((g, a.name[:70], b.name[:70]))
    print(f"{len(ev) / steps:.0f} kernels per step, span {span / steps / 1e3:.3f} ms per step, busy {busy / steps / 1e3:.3f} ms, "
          f"idle {(span - busy) / steps / 1e3:.3f} ms")
    small = sum(g for g, _, _ in gaps if g <= 10)
    print(f"gaps <= 10 us: {small / steps / 1e3:.3f} ms per step; larger gaps: {(sum(g for g, _, _ in gaps) - small) / steps / 1e3:.3f} ms per step")
    for g, a, b in sorted(gaps, reverse=True)[:24]:
        print(f"{g:8.1f} us   after {a}\n              before {b}")


if __name__ == "__main__":
    main()

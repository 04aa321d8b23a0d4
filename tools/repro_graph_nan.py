#!/usr/bin/env python3
"""Plain-torch reduction of the caveat in pygho_amd/graphs.py (no pygho_amd code involved): N captured training steps (one per fixed
mini-batch) sharing ONE capturable AdamW, an eager kernel launched between replays.  Result on PyTorch 2.10 / ROCm 7.2, MI355X
(round 3): 0 non-finite losses with 2 / 4 / 6 captured steps, with and without a device synchronisation -- this reduction does
NOT reproduce the NaN seen with the captured SpModel steps.     python tools/repro_graph_nan.py [n_graphs=4] [sync=0|1]"""
import sys
import torch

n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sync = len(sys.argv) > 2 and sys.argv[2] == "1"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.SiLU(), torch.nn.Linear(64, 1)).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
batches = [(torch.randn(256, 64, device=dev), torch.randn(256, 1, device=dev)) for _ in range(n_graphs)]


def make_step(x, y):
    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(model(x), y)
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
bad = 0
for epoch in range(20):
    for g, out in graphs:
        g.replay()
        _ = torch.full((1,), 7.0, device=dev)          # any eager kernel between two replays
        if sync:
            torch.cuda.synchronize()
    bad += sum(int(not bool(torch.isfinite(out))) for _, out in graphs)
print(f"captured steps: {n_graphs}, device sync after eager work: {sync}, non-finite losses: {bad} of {20 * n_graphs}")

"""
Multi-process (gloo, world_size 2) test of the N > 1 path on CPU: graph-sharded data parallelism with ONE flat
gradient all-reduce per step must give the same averaged gradients and the same parameters after an optimizer
step as a single process on the concatenated batch (no BatchNorm, as the reference has no SyncBN).
The compute in the toy model is plain torch: the HIP path itself has no CPU build; what is tested here is the
sharding + gradient exchange (pygho_amd/parallel.py) that bench.py uses over RCCL.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pygho_amd.parallel import FlatGradSync, shard_ranges


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.SiLU(), torch.nn.Linear(16, 1))


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(40, 6, generator=g), torch.randn(40, 1, generator=g)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y = _data()
    lo, hi = shard_ranges(np.ones(40), world)[rank]
    model = _model()
    sync = FlatGradSync(model.parameters())
    sync.broadcast_params(0)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for _ in range(3):
        sync.zero_grad()
        loss = torch.nn.functional.mse_loss(model(x[lo:hi]), y[lo:hi], reduction="sum") / 20.0   # mean over equal shards
        loss.backward()
        sync.sync()
        opt.step()
    if rank == 0:
        ret["flat"] = sync.flat.clone()
        ret["params"] = [p.detach().clone() for p in model.parameters()]
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_allreduce_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    x, y = _data()
    model = _model()
    sync = FlatGradSync(model.parameters())
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for _ in range(3):
        sync.zero_grad()
        torch.nn.functional.mse_loss(model(x), y, reduction="sum").div(40.0).backward()
        sync.sync()
        opt.step()
    torch.testing.assert_close(ret["flat"], sync.flat, rtol=1e-5, atol=1e-6)
    for a, b in zip(ret["params"], model.parameters()):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)


def test_flat_views_alias_parameter_grads():
    model = _model()
    sync = FlatGradSync(model.parameters())
    model(torch.ones(2, 6)).sum().backward()
    expect = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    sync.sync()                                  # no process group: pack only
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(model.parameters(), sync.views))
    assert torch.equal(sync.flat, expect) and float(sync.flat.abs().sum()) > 0
    sync.zero_grad()
    assert all(p.grad is None for p in model.parameters())
    model[0](torch.ones(2, 6)).sum().backward()   # second Linear unused: its slice must read zero after packing
    sync.sync()
    n0 = sum(p.numel() for p in model[0].parameters())
    assert float(sync.flat[:n0].abs().sum()) > 0 and float(sync.flat[n0:].abs().sum()) == 0
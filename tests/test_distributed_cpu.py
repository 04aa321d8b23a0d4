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


def _worker(rank, world, port, ret, overlap=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y = _data()
    lo, hi = shard_ranges(np.ones(40), world)[rank]
    model = _model()
    sync = FlatGradSync(model.parameters(), overlap=overlap, buckets=2)
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
        ret["calls"] = sync.allreduce_calls
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("overlap", [False, True])
def test_two_rank_flat_allreduce_equals_single_process(overlap):
    """overlap=True: the gradient is exchanged in two ranges launched from backward hooks (the later layers' range while backward
    still runs); same averaged gradient, same parameters after the optimizer steps."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret, overlap), nprocs=2, join=True)
    assert ret["calls"] == (6 if overlap else 3)
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

# ---- BASELINE config 4's wording: ONE global batch stream sharded by graph, balanced by message count -----------------------
def _global_records():
    import numpy as np
    from pygho_amd import synth
    rng = np.random.default_rng(11)
    return [synth.make_graph(rng, "zinc", 3, ("X___X___1___A___0",)) for _ in range(12)]


def _ngnn_step_inputs(records):
    import numpy as np
    from pygho_amd import synth
    hb = synth.collate(records)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return hb, (t(hb.x), t(hb.edge_attr), t(hb.tupleid), t(hb.tuplefeat), t(hb.acd["X___X___1___A___0"]), t(hb.batch), hb.num_graphs)


def _port_model():
    from oracle import aten_port as P
    torch.manual_seed(3)
    m = P.NGNNPort(16, 2)
    m.eval()                     # BatchNorm per rank differs from BatchNorm over the global batch by design (no SyncBN in the reference)
    return m


def _sharded_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs = _global_records()
    lo, hi = shard_ranges([r.acd["X___X___1___A___0"].shape[1] for r in recs], world)[rank]
    hb, args = _ngnn_step_inputs(recs[lo:hi])
    model = _port_model()
    sync = FlatGradSync(model.parameters())
    sync.broadcast_params(0)
    sync.zero_grad()
    pred = model(*args)
    y = torch.from_numpy(hb.y).unsqueeze(-1)
    # weight of this shard: the averaged gradient must be the gradient of the GLOBAL mean loss whatever the shard sizes are
    loss = torch.nn.functional.l1_loss(y, pred) * (world * hb.num_graphs / len(recs))
    loss.backward()
    sync.sync()
    ret[rank] = (lo, hi, sync.flat.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_global_stream_sharded_by_message_count_equals_single_process():
    """bench.py --global-stream: every rank collates its own contiguous graph range (offsets restart at 0, plans are rank-local),
    ranges balanced by message count (unequal graph counts), losses weighted by shard size: the flat all-reduced gradient equals
    the gradient of the global mean loss computed by one process on the whole batch.  The model is the CPU port of the
    benchmark's NGNN (oracle/aten_port.py, test infrastructure)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(2, port, ret), nprocs=2, join=True)
    (lo0, hi0, flat0), (lo1, hi1, flat1) = ret[0], ret[1]
    assert lo0 == 0 and hi0 == lo1 and hi1 == 12 and 0 < hi0 < 12
    torch.testing.assert_close(flat0, flat1, rtol=0, atol=0)          # both ranks hold the same averaged gradient
    hb, args = _ngnn_step_inputs(_global_records())
    model = _port_model()
    sync = FlatGradSync(model.parameters())
    sync.zero_grad()
    torch.nn.functional.l1_loss(torch.from_numpy(hb.y).unsqueeze(-1), model(*args)).backward()
    sync.sync()
    torch.testing.assert_close(flat0, sync.flat, rtol=1e-5, atol=1e-6)


def _accum_worker(rank, world, port, ret):
    """gradient accumulation over two micro-batches with overlap: the first backward runs under no_sync(); rank 1 starts from
    DIFFERENT parameters, so the flat broadcast must bring them in line first"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y = _data()
    lo, hi = shard_ranges(np.ones(40), world)[rank]
    mid = (lo + hi) // 2
    model = _model()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)                                   # not rank 0's parameters
    sync = FlatGradSync(model.parameters(), overlap=True, buckets=2)
    sync.broadcast_params(0)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for _ in range(3):
        sync.zero_grad()
        with sync.no_sync():
            (torch.nn.functional.mse_loss(model(x[lo:mid]), y[lo:mid], reduction="sum") / 20.0).backward()
        (torch.nn.functional.mse_loss(model(x[mid:hi]), y[mid:hi], reduction="sum") / 20.0).backward()
        sync.sync()
        opt.step()
    ret[f"params{rank}"] = [p.detach().clone() for p in model.parameters()]
    if rank == 0:
        ret["calls"], ret["broadcasts"] = sync.allreduce_calls, sync.broadcast_calls
    # without no_sync() the second backward is refused, with a pointer to it
    sync.zero_grad()
    model(x[lo:mid]).sum().backward()
    try:
        model(x[mid:hi]).sum().backward()
        ret[f"raised{rank}"] = False
    except RuntimeError as e:
        ret[f"raised{rank}"] = "no_sync" in str(e)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_under_no_sync_and_the_flat_broadcast():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_accum_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["calls"] == 6 and ret["broadcasts"] == 1              # two ranges per step; ONE broadcast for all (f32) parameters
    assert ret["raised0"] and ret["raised1"]
    x, y = _data()
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(model(x), y, reduction="sum").div(40.0).backward()
        opt.step()
    for a, b, c in zip(ret["params0"], ret["params1"], model.parameters()):
        torch.testing.assert_close(a, c.detach(), rtol=1e-5, atol=1e-6)
        assert torch.equal(a, b)

"""
The N > 1 path ON THE HIP KERNELS with more than one rank, on a one-GPU box: two processes, both on cuda:0, rendezvous over gloo
(RCCL refuses two ranks on one device; the collective's transport is not what is tested).  Each rank runs bench.py's
--global-stream step on pygho_amd.ngnn.SpModel: the same global batch of graph records, its own contiguous range from
parallel.shard_ranges (balanced by message count, so the graph counts differ), rank-local collation and plans, the loss weighted by
its share, the gradient exchanged by FlatGradSync in its overlapping form (two ranges launched from backward hooks on a side
stream).  Asserted: both ranks end with the same averaged gradient, and it equals the gradient one process computes on the
concatenated batch to 1e-5.  BatchNorm runs in eval mode: the reference has no SyncBN, per-rank batch statistics differ from
global ones by design.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

KEY = "X___X___1___A___0"
N_GRAPHS, HIDDEN, LAYERS = 96, 64, 3


def _records():
    from pygho_amd import synth
    rng = np.random.default_rng(21)
    return [synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(N_GRAPHS)]


def _model(dev):
    from pygho_amd.ngnn import SpModel
    torch.manual_seed(5)
    m = SpModel(1, LAYERS, HIDDEN, act_dtype=None).to(dev)       # f32 end to end: the comparison is about the exchange, not rounding
    m.eval()
    return m


def _step(model, sync, recs, weight, dev):
    from pygho_amd import synth
    hb = synth.collate(recs)
    dd = synth.to_datadict(hb, dev)
    sync.zero_grad()
    pred = model(dd)
    loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float()) * weight
    loss.backward()
    sync.sync()
    return hb


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pygho_amd import _native
    from pygho_amd.parallel import FlatGradSync, shard_ranges
    _native.lib()                                                 # the HIP extension must load in every rank: no fallback
    recs = _records()
    lo, hi = shard_ranges([r.acd[KEY].shape[1] for r in recs], world)[rank]
    model = _model(dev)
    sync = FlatGradSync(model.parameters(), overlap=True, buckets=2)
    sync.broadcast_params(0)
    hb = _step(model, sync, recs[lo:hi], world * (hi - lo) / len(recs), dev)
    torch.cuda.synchronize(dev)
    ret[rank] = (lo, hi, hb.num_graphs, sync.flat.detach().cpu(), sync.allreduce_calls)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_single_process_on_the_hip_path():
    assert torch.cuda.is_available(), "needs the MI355X"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    (lo0, hi0, n0, flat0, calls0), (lo1, hi1, n1, flat1, calls1) = ret[0], ret[1]
    assert lo0 == 0 and hi0 == lo1 and hi1 == N_GRAPHS and n0 + n1 == N_GRAPHS
    assert calls0 == calls1 == 2                                  # two ranges per step
    assert torch.equal(flat0, flat1), "both ranks must hold the same averaged gradient"
    # one process, the whole batch
    from pygho_amd.parallel import FlatGradSync
    dev = torch.device("cuda", 0)
    model = _model(dev)
    sync = FlatGradSync(model.parameters())
    _step(model, sync, _records(), 1.0, dev)
    ref = sync.flat.detach().cpu()
    assert float(ref.abs().max()) > 0
    torch.testing.assert_close(flat0, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max()) + 1e-7)


def _stepper(model, sync, capturable):
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=capturable, fused=True)

    def step(dd):
        sync.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        sync.sync()
        opt.step()
        return loss.detach()
    return step


def _captured_worker(rank, world, port, backend, ret):
    import copy
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.graphs import SlotStep
    from pygho_amd.ngnn import SpModel
    from pygho_amd.parallel import FlatGradSync
    store = DeviceGraphStore(_records(), dev)
    torch.manual_seed(3)
    model = SpModel(1, 2, 128, act_dtype=torch.bfloat16).to(dev)
    ref = copy.deepcopy(model)
    ids = [np.random.default_rng(40 + 7 * rank + k).permutation(N_GRAPHS)[:48] for k in range(6)]
    sync_c = FlatGradSync(model.parameters(), overlap=True, buckets=2)
    ss = SlotStep(store, 48, _stepper(model, sync_c, True), warmup_ids=ids[0], warmup=3, sync=sync_c)
    if backend == "nccl":
        calls_after_setup = sync_c.allreduce_calls
        got = [ss.run(i).clone() for i in ids[1:]]
        sync_e = FlatGradSync(ref.parameters(), overlap=True, buckets=2)
        eager = _stepper(ref, sync_e, True)
        for _ in range(3):                                      # the three warm-up steps on ids[0] (the capture itself executes nothing)
            eager(store.collate(ids[0]))
        want = [eager(store.collate(i)).clone() for i in ids[1:]]
        torch.cuda.synchronize(dev)
        ret[rank] = dict(captured=ss.captured, replays=ss.replays, eager_steps=ss.eager_steps, per_step=ss._calls_per_step,
                         calls_c=sync_c.allreduce_calls - calls_after_setup, calls_e=sync_e.allreduce_calls - 6,
                         losses_equal=all(torch.equal(a, b) for a, b in zip(got, want)),
                         params_equal=all(torch.equal(p, q) for p, q in zip(model.parameters(), ref.parameters())),
                         flat_equal=bool(torch.equal(sync_c.flat, sync_e.flat)))
    else:
        losses = [float(ss.run(i)) for i in ids[1:3]]
        torch.cuda.synchronize(dev)
        ret[rank] = dict(captured=ss.captured, why=ss.why_eager, replays=ss.replays, eager_steps=ss.eager_steps, finite=all(np.isfinite(losses)),
                         flat=sync_c.flat.detach().cpu(), calls=sync_c.allreduce_calls)
    del ss                                                      # the graph holds captured collectives: gone before their communicator
    torch.cuda.synchronize(dev)
    dist.barrier()
    dist.destroy_process_group()


def test_captured_slot_step_with_the_rccl_exchange_inside_equals_the_eager_loop():
    """`graphs.SlotStep(..., sync=FlatGradSync)` over a ONE-rank RCCL group -- the collectives that the backward hooks
    issue on the side stream are captured inside the step's HIP graph.  Five replayed batches against the eager loop on the exactly
    sized batches: same losses, same parameters after AdamW, same flat gradient, bit for bit; the collective count follows the replays."""
    assert torch.cuda.is_available(), "needs the MI355X"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_captured_worker, args=(1, port, "nccl", ret), nprocs=1, join=True)
    r = ret[0]
    assert r["captured"] and r["replays"] == 5 and r["eager_steps"] == 0 and r["per_step"] == 2
    assert r["calls_c"] == r["calls_e"] == 10
    assert r["losses_equal"] and r["params_equal"] and r["flat_equal"]


def test_slot_step_under_gloo_runs_every_step_eagerly_and_says_so():
    """a host-staged backend cannot be captured: two gloo ranks on one GPU keep the eager path (`captured` False, `why_eager` names the
    backend), exchange two ranges per step and end with the same averaged gradient"""
    assert torch.cuda.is_available(), "needs the MI355X"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_captured_worker, args=(2, port, "gloo", ret), nprocs=2, join=True)
    for r in (ret[0], ret[1]):
        assert r["captured"] is False and "gloo" in r["why"] and r["replays"] == 0 and r["eager_steps"] == 2 and r["finite"]
        assert r["calls"] == 4
    assert torch.equal(ret[0]["flat"], ret[1]["flat"])

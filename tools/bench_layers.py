#!/usr/bin/env python3
"""Forward + backward time of ONE layer of each shipped subgraph-GNN family on the sparse backend (north_star:
NGNNConv, SSWLConv, SUNConv, I2Conv), ZINC-shape 2-tuple batches (I2Conv: the 3-tuple stress shape), bf16 autocast.

    python tools/bench_layers.py [--graphs 8192] [--profile LAYER]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import synth                                    # noqa: E402
from pygho_amd.honn import Conv                                 # noqa: E402
from pygho_amd.honn.SpOperator import parse_precomputekey       # noqa: E402

MLP = {"numlayer": 1, "tailact": True, "norm": "bn", "act": "silu", "dp": 0.0}


_BATCHES = {}


def batch(graphs, kind, keys, seed=11):
    """`graphs` DISTINCT synthetic graphs (no tiled pattern), generated once per (size, kind, keys) and shared by the cases"""
    k = (graphs, kind, tuple(keys), seed)
    if k not in _BATCHES:
        _BATCHES.clear()                                   # one host batch alive at a time
        _BATCHES[k] = synth.make_batch(graphs, kind, seed=seed, keys=tuple(keys))
    return _BATCHES[k]


def build(name, h, dev, aggr="sum"):
    torch.manual_seed(0)
    if name == "NGNNConv":
        return Conv.NGNNConv(h, h, aggr, "SS", dict(MLP)).to(dev)
    if name == "SSWLConv":
        return Conv.SSWLConv(h, h, "sum", "SS", dict(MLP)).to(dev)
    if name == "SUNConv":
        return Conv.SUNConv(h, h, "sum", "mean", "SS", dict(MLP), dict(MLP)).to(dev)
    if name == "I2Conv":
        return Conv.I2Conv(h, h, "sum", "SS", dict(MLP)).to(dev)
    raise ValueError(name)


def case(name, graphs, dev, profile=False, aggr="sum", kernels=False, residual_lookup=False):
    """`residual_lookup`: the layer as the MODEL LOOP runs it -- `X.add(layer(A, X), True)` through `forward_residual`, with A's values
    an embedding lookup of the edge feature (what `InputEncoderSp` produces): the form that takes the fused block kernels
    (csrc/seg_fused.hip for NGNNConv); otherwise `layer.forward` on random edge values (the operator path alone)"""
    kind, h = ("i2", 256) if name == "I2Conv" else ("zinc", 128)
    layer = build(name, h, dev, aggr)
    keys = tuple(parse_precomputekey(layer))
    hb = batch(graphs, kind, keys)
    dd = synth.to_datadict(hb, dev, kind)
    from pygho_amd import SparseTensor
    X0, A0 = dd["X"], dd["A"]
    xv = torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16).requires_grad_(True)
    av = torch.randn(A0.nnz, h, device=dev).to(torch.bfloat16)
    if residual_lookup:
        from pygho_amd.ngnn import IndexEmbedding
        torch.manual_seed(1)
        emb = IndexEmbedding(16, h, torch.bfloat16).to(dev)
    A = SparseTensor(A0.indices, av, list(A0.shape[:A0.sparse_dim]) + [h], True)
    w = torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16)
    # the batch's message plans, INCLUDING the by-edge gradient's scatter plans, exist before the timed region -- as for a batch collated
    # from a `DeviceGraphStore`, or one a loop prepared (`SpModel.prepare`): the dispatcher itself never plans (round 5)
    from pygho_amd import _ops
    from pygho_amd.synth import parse_key
    for k in keys:
        roles = parse_key(k)
        rows = lambda r: int(X0.nnz) if r[0] == "X" else int(A0.nnz)
        plan = _ops.message_plan(dd[k + "___acd"], rows(roles[0]), rows(roles[1]), rows(roles[3]))
        _ops.scatter_plan(plan)
        if residual_lookup:
            _ops.fused_plan(plan)                  # the fused block forward's chunk list (None when the pattern is outside its limits)

    def step():
        xv.grad = None
        for p in layer.parameters():
            p.grad = None
        X = SparseTensor(X0.indices, xv, list(X0.shape[:X0.sparse_dim]) + [h], True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if residual_lookup:
                emb.weight.grad = None
                Al = A0.tuplewiseapply(lambda v: emb(v))            # values carry their provenance (table, index): honn/Conv._residual_update
                out = layer.forward_residual(Al, X, dd)
            else:
                out = layer(A, X, dd)
        out.values.backward(w)

    for _ in range(6):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if profile:
        from torch.profiler import profile as prof_, ProfilerActivity
        with prof_(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as pr:
            for _ in range(5):
                step()
            torch.cuda.synchronize()
        print(pr.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=90), file=sys.stderr)
    msgs = {k: int(dd[k + "___acd"].shape[1]) for k in keys}
    res = {"op": f"{name} SS layer fwd+bwd" + ("" if aggr == "sum" else f" (aggr={aggr})") + (" + residual, edge values = embedding lookup" if residual_lookup else ""),
           "graphs": hb.num_graphs, "tuples": int(X0.nnz),
           "d": h, "dtype": "bfloat16", "msg_edges": msgs, "ms": ms, "graphs_per_s": hb.num_graphs / ms * 1e3}
    if kernels:                                           # per-launch HIP-event times of the segment kernels of one more step
        from pygho_amd import _ops
        timer = _ops.LaunchTimer()
        with timer:
            for _ in range(3):
                step()
        torch.cuda.synchronize()
        res["kernels"] = {k: {"launches": v[0], "avg_ms": v[1], "algorithmic_bytes": v[2], "GBps": v[2] / (v[1] * 1e-3) / 1e9,
                              "frac": v[2] / (v[1] * 1e-3) / 1e9 / 8000.0} for k, v in timer.summary().items()}
    return res


def model_case(conv, graphs, dev):
    """one training step (forward, backward, AdamW) of the 6-layer example/zinc.py model of a layer family (pygho_amd.models.SpModel),
    hidden 128 (I2GNN: 3-tuple batches), bf16 activations, resident batch."""
    from pygho_amd.models import SpModel
    kind = "i2" if conv == "I2GNN" else "zinc"
    torch.manual_seed(0)
    model = SpModel(conv, num_layer=6, hiddim=128, act_dtype=torch.bfloat16).to(dev)
    keys = tuple(parse_precomputekey(model))
    hb = batch(graphs, kind, keys)
    dd = synth.to_datadict(hb, dev, kind)
    y = dd["y"].unsqueeze(-1)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        torch.nn.functional.l1_loss(y, pred.float()).backward()
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    return {"op": f"{conv} model train step (6 layers, hidden 128, bf16)", "graphs": hb.num_graphs, "ms": ms,
            "graphs_per_s": hb.num_graphs / ms * 1e3}


def dense_model_case(conv, graphs, dev):
    """the same training step on the dense layout (pygho_amd.models.MaModel: padded (b, n, n) MaskedTensors, zinc.py:150-236)"""
    from pygho_amd.models import MaModel
    torch.manual_seed(0)
    model = MaModel(conv, num_layer=6, hiddim=128, act_dtype=torch.bfloat16).to(dev)
    hb = synth.make_batch(graphs, "zinc", seed=11)
    dd = synth.to_dense_datadict(hb, dev)
    y = dd["y"].unsqueeze(-1)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dict(dd))
        torch.nn.functional.l1_loss(y, pred.float()).backward()
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    return {"op": f"{conv} dense-layout model train step (6 layers, hidden 128, bf16)", "graphs": hb.num_graphs,
            "padded_nodes": int(dd["x"].shape[1]), "ms": ms, "graphs_per_s": hb.num_graphs / ms * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=8192)
    ap.add_argument("--profile", default="")
    ap.add_argument("--models", action="store_true", help="also time a 6-layer model train step per layer family")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in ("NGNNConv", "SSWLConv", "SUNConv", "I2Conv"):
        g = args.graphs if name != "I2Conv" else max(256, args.graphs // 4)
        print(json.dumps(case(name, g, dev, profile=args.profile == name)), flush=True)
    if args.models:
        for conv in ("NGNN", "SSWL", "DSSGNN", "GNNAK", "SUN", "PPGN", "I2GNN"):
            g = args.graphs // 2 if conv in ("SUN", "PPGN", "GNNAK") else (max(256, args.graphs // 8) if conv == "I2GNN" else args.graphs)
            print(json.dumps(model_case(conv, g, dev)), flush=True)
        for conv in ("NGNN", "SSWL", "DSSGNN", "GNNAK", "SUN", "PPGN"):
            print(json.dumps(dense_model_case(conv, min(args.graphs, 1024), dev)), flush=True)


if __name__ == "__main__":
    main()

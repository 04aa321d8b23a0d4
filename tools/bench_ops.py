#!/usr/bin/env python3
"""
Operator-level micro-benchmarks for the other BASELINE.json configs (not the headline line of bench.py):
  config 2  spspmm forward, ZINC-shape 2-tuples, d=128 (bf16 / f32)
  config 5  spspmm forward, I2-shape 3-tuples, d=256 bf16
  config 3  masked batched einsum mamamm(X,2,A,1) on the MFMA kernel, (b, 37, 37, 128) bf16 / f32, and its backward
Each line: median kernel time (HIP events), algorithmic bytes, GB/s and fraction of the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth  # noqa: E402

PEAK = 8000.0


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def spspmm_case(kind, graphs, d, dtype, dev):
    base = 1024 if kind == "zinc" else 128
    key = "X___X___1___A___0" if kind == "zinc" else "X___X___2___A___0"
    hb = synth.replicate(synth.make_batch(min(graphs, base), kind, seed=1), max(1, graphs // base))
    acd = torch.from_numpy(hb.acd[key]).to(dev)
    nt, ne, m = hb.num_tuples, hb.num_edges, acd.shape[1]
    X = torch.randn(nt, d, device=dev).to(dtype)
    A = torch.randn(ne, d, device=dev).to(dtype)
    plan = _ops.message_plan(acd, nt, nt, ne)
    ms = timed(lambda: _ops.seg_gmr(nt, X, A, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum"))
    nbytes = X.element_size() * d * (2 * nt + ne) + 8 * m + 4 * (nt + 1)
    return {"op": f"spspmm fwd {kind}", "graphs": hb.num_graphs, "d": d, "dtype": str(dtype).split(".")[-1], "msg_edges": m,
            "ms": ms, "alg_MB": nbytes / 1e6, "GBps": nbytes / ms / 1e6, "frac_hbm": nbytes / ms / 1e6 / PEAK,
            "G_msg_edges_per_s": m / ms / 1e6}


def pooling_case(graphs, d, dtype, dev):
    """a5 / a11: tuple-wise segment reduce (subgraph pooling X.sum(dim 1): nnz rows -> N rows, sorted index) and a8 unpooling."""
    hb = synth.replicate(synth.make_batch(min(graphs, 1024), "zinc", seed=1), max(1, graphs // 1024))
    row = torch.from_numpy(hb.tupleid[0]).to(dev)
    nt, n = hb.num_tuples, hb.num_nodes
    X = torch.randn(nt, d, device=dev).to(dtype)
    plan = _ops.cached_plan(row, n, "bench")
    ms = timed(lambda: _ops.seg_reduce_rows(X, plan, "sum"))
    nbytes = X.element_size() * d * (nt + n) + 4 * (n + 1)
    x = torch.randn(n, d, device=dev).to(dtype)
    r32 = _ops.narrow_i32(row)
    ms_g = timed(lambda: _ops.row_gather(x, r32))
    nb_g = X.element_size() * d * (nt + n) + 4 * nt
    return [{"op": "segment reduce (subgraph pooling)", "graphs": hb.num_graphs, "rows_in": nt, "rows_out": n, "d": d,
             "dtype": str(dtype).split(".")[-1], "ms": ms, "GBps": nbytes / ms / 1e6, "frac_hbm": nbytes / ms / 1e6 / PEAK},
            {"op": "row gather (unpooling)", "graphs": hb.num_graphs, "rows_in": n, "rows_out": nt, "d": d,
             "dtype": str(dtype).split(".")[-1], "ms": ms_g, "GBps": nb_g / ms_g / 1e6, "frac_hbm": nb_g / ms_g / 1e6 / PEAK}]


def spmm_case(graphs, d, dtype, dev):
    """a17: node-level message passing out[t] = sum_e val[e] * X[src[e]] over the batch adjacency (E edges, N nodes)."""
    hb = synth.replicate(synth.make_batch(min(graphs, 1024), "zinc", seed=1), max(1, graphs // 1024))
    ei = torch.from_numpy(hb.edge_index).to(dev)
    n, e = hb.num_nodes, hb.num_edges
    val = torch.randn(e, d, device=dev).to(dtype)
    X = torch.randn(n, d, device=dev).to(dtype)
    _ops.spmm_values(val, X, ei[1].contiguous(), ei[0].contiguous(), n, "sum")
    ms = timed(lambda: _ops.spmm_values(val, X, ei[1].contiguous(), ei[0].contiguous(), n, "sum"))
    nbytes = X.element_size() * d * (e + 2 * n) + 4 * e + 4 * (n + 1)
    return {"op": "spmm (node-level message passing)", "graphs": hb.num_graphs, "edges": e, "nodes": n, "d": d,
            "dtype": str(dtype).split(".")[-1], "ms": ms, "GBps": nbytes / ms / 1e6, "frac_hbm": nbytes / ms / 1e6 / PEAK}


def graph_step_case(graphs, dev):
    """launch-bound regime: the NGNN training step on a small batch, eager vs captured into a HIP graph (pygho_amd.graphs)."""
    import time
    from pygho_amd.graphs import GraphedStep
    from pygho_amd.ngnn import SpModel
    hb = synth.make_batch(graphs, "zinc", seed=77)
    dd = synth.to_datadict(hb, dev)
    y = dd["y"].unsqueeze(-1)
    torch.manual_seed(0)
    model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(y, pred.float())
        loss.backward()
        opt.step()
        return loss.detach()

    def wall(fn, n=20):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    eager = wall(step)
    gs = GraphedStep(step)
    graphed = wall(gs.replay)
    return {"op": "NGNN train step, eager vs HIP graph", "graphs": graphs, "eager_ms": eager, "graph_ms": graphed,
            "graphs_per_s_eager": graphs / eager * 1e3, "graphs_per_s_graph": graphs / graphed * 1e3}


def sampler_case(kind, graphs, dev):
    """device tuple sampler (SpTupleSampler.py:91-173) on a whole batch: k-hop (ZINC shape) or pair-rooted (I2 shape) tuples +
    distance features from the edge list.  Wall time per batch (three host syncs size the outputs)."""
    import time
    from pygho_amd.hodata import i2_sample, khop_sample
    base = 1024 if kind == "zinc" else 128
    hb = synth.replicate(synth.make_batch(min(graphs, base), kind, seed=1), max(1, graphs // base))
    ei, nb = torch.from_numpy(hb.edge_index).to(dev), torch.from_numpy(hb.batch).to(dev)
    fn = khop_sample if kind == "zinc" else i2_sample
    tid, tf = fn(ei, hb.num_nodes, 3, nb)
    assert tid.shape[1] == hb.num_tuples
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        fn(ei, hb.num_nodes, 3, nb)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ms = sorted(ts)[len(ts) // 2]
    return {"op": f"device tuple sampler ({'KhopSampler' if kind == 'zinc' else 'I2Sampler'}, hop 3) {kind}", "graphs": hb.num_graphs,
            "nodes": hb.num_nodes, "edges": hb.num_edges, "tuples": hb.num_tuples, "ms": ms, "tuples_per_s": hb.num_tuples / ms * 1e3}


def planner_case(kind, graphs, dev):
    """device planner (Spspmm.py:57-222): tuple pattern x adjacency -> (tarind, bcd) -> acd on the tuple pattern.
    Wall time per batch (includes the two host syncs that size the outputs)."""
    import time
    from pygho_amd.backend import Spspmm
    base = 1024 if kind == "zinc" else 128
    hb = synth.replicate(synth.make_batch(min(graphs, base), kind, seed=1), max(1, graphs // base))
    tup = torch.from_numpy(hb.tupleid).to(dev)
    adj = torch.from_numpy(hb.edge_index).to(dev)
    dim1 = tup.shape[0] - 1

    def run():
        ind, bcd = Spspmm.spspmm_ind(tup, dim1, adj, 0, is_k2_sorted=True)
        return Spspmm.filterind(tup, ind, bcd)
    acd = run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ms = sorted(ts)[len(ts) // 2]
    return {"op": f"planner spspmm_ind+filterind {kind}", "graphs": hb.num_graphs, "tuples": int(tup.shape[1]),
            "msg_edges": int(acd.shape[1]), "ms": ms, "M_msg_edges_per_s": acd.shape[1] / ms / 1e3}


def sunconv_case(b, n, d, dtype, dev):
    """BASELINE config 3, second half: one SUNConv layer on the dense MaskedTensor path (mode "DD": masked bmm on the matrix
    cores, masked pooling / broadcast kernels, fused BatchNorm MLPs), forward + backward, padded ZINC-shape batch."""
    from pygho_amd import MaskedTensor
    from pygho_amd.honn import Conv
    dn = synth.make_dense_batch(b, seed=2, hidden=1, nmax=n)           # every graph distinct (masks from the generator, values drawn on the device)
    xm, am = torch.from_numpy(dn["Xmask"]).to(dev), torch.from_numpy(dn["Amask"]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(2)
    vals = lambda mask: (torch.randn(mask.shape + (d,), device=dev, generator=gen) * mask.unsqueeze(-1)).to(dtype)
    mlp = {"numlayer": 1, "tailact": True, "norm": "bn", "act": "silu", "dp": 0.0}
    torch.manual_seed(0)
    layer = Conv.SUNConv(d, d, "sum", "mean", "DD", dict(mlp), dict(mlp)).to(dev)
    Xraw = vals(xm).requires_grad_(True)
    X = MaskedTensor(Xraw, xm, 0.0, True)
    A = MaskedTensor(vals(am), am, 0.0, True)
    w = torch.randn_like(Xraw)

    def step():
        Xraw.grad = None
        for p in layer.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            out = layer(A, X, {})
        out.data.backward(w)                   # upstream gradient handed in directly: only the layer is timed
    ms = timed(step, reps=10)
    bb = Xraw.shape[0]
    return {"op": "SUNConv DD layer fwd+bwd", "b": bb, "n": n, "d": d, "dtype": str(dtype).split(".")[-1], "ms": ms,
            "graphs_per_s": bb / ms * 1e3}


def collate_case(graphs, dev):
    """on-device mini-batch collation (hodata/SpData.py:56-112) from the int32 graph store: wall time per batch."""
    import time
    import numpy as np
    from pygho_amd.collate import DeviceGraphStore
    rng = np.random.default_rng(3)
    recs = [synth.make_graph(rng, "zinc") for _ in range(1024)]
    store = DeviceGraphStore(recs, dev)
    sel = torch.from_numpy(rng.integers(0, len(recs), graphs)).to(dev)
    dd = store.collate(sel)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        dd = store.collate(sel)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    key = next(k for k in dd if k.endswith("acd"))
    out_bytes = 8 * (dd["A"].indices.numel() + dd["X"].indices.numel() + dd[key].numel() + dd["X"].nnz + dd["A"].nnz + 2 * dd["num_nodes"])
    return {"op": "device collate zinc", "graphs": graphs, "tuples": dd["X"].nnz, "msg_edges": int(dd[key].shape[1]),
            "ms": sorted(ts)[len(ts) // 2], "int64_MB_not_shipped_over_PCIe": out_bytes / 1e6}


def dense_collate_case(b, n, d, dev):
    """padded-batch construction of the dense layout on the device (hodata/MaData.py:217-255 batch2dense): ragged per-graph
    tuple features (n_g x n_g x d, bf16), node features and a COO adjacency -> (b, n, n, d) MaskedTensors."""
    import types
    from pygho_amd.hodata import batch2dense
    dn = synth.make_dense_batch(min(b, 256), seed=4, hidden=d, nmax=n)
    rep = max(1, b // min(b, 256))
    nm = np.tile(dn["nodemask"], (rep, 1))
    counts = nm.sum(1).astype(np.int64)
    bb = counts.shape[0]
    ptr = np.concatenate(([0], np.cumsum(counts)))
    tptr = np.concatenate(([0], np.cumsum(counts * counts)))
    gen = torch.Generator(device="cpu").manual_seed(0)
    tf = torch.randn((int(tptr[-1]), d), generator=gen).to(torch.bfloat16).to(dev)
    Am = np.tile(dn["Amask"], (rep, 1, 1))
    eb, er, ec = np.nonzero(Am)
    ea = torch.randn((eb.shape[0], d), generator=gen).to(torch.bfloat16).to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    base = dict(x=t(np.zeros(int(ptr[-1]), dtype=np.int64)), ptr=t(ptr), edge_index=t(np.stack((er, ec))), edge_index_batch=t(eb),
                edge_attr=ea, tuplefeat=tf, tupleshape=t(np.stack((counts, counts), 1)), tuplefeat_ptr=t(tptr))

    def run():
        return batch2dense(types.SimpleNamespace(**base), batch_size=bb, max_num_nodes=n, denseadj=True)
    ms = timed(run, reps=20)
    out_bytes = 2 * bb * n * n * d * 2 + 2 * bb * n * n
    return {"op": "device batch2dense (padded dense batch)", "b": bb, "n": n, "d": d, "dtype": "bfloat16", "ms": ms,
            "tuples": int(tptr[-1]), "edges": int(eb.shape[0]), "written_MB": out_bytes / 1e6, "GBps_written": out_bytes / ms / 1e6}


def fresh_batch_case(graphs, dev):
    """training with a NEW batch every step (the bench step re-uses one resident batch): plain device collation + first-use plan
    construction inside the step, against collate.BatchPrefetcher (collation and SpModel.prepare one batch ahead on a side stream)."""
    import time
    from pygho_amd.collate import BatchPrefetcher, DeviceGraphStore
    from pygho_amd.ngnn import SpModel
    rng = np.random.default_rng(0)
    recs = [synth.make_graph(rng, "zinc", 3, ("X___X___1___A___0",)) for _ in range(min(2048, graphs))]
    store = DeviceGraphStore(recs * max(2, 2 * graphs // len(recs)), dev)
    torch.manual_seed(0)
    model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)

    def step(dd):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float()).backward()
        opt.step()
    gen = torch.Generator().manual_seed(0)
    ids = [torch.randperm(store.num_graphs, generator=gen)[:graphs] for _ in range(14)]
    res = {}
    resident = store.collate(ids[0])
    for name, src in (("resident", lambda: (resident for _ in ids)), ("plain", lambda: (store.collate(i) for i in ids)),
                      ("prefetch", lambda: BatchPrefetcher(store, ids, model.prepare))):
        n = 0
        for k, dd in enumerate(src()):
            if k == 4:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            step(dd)
            n += 1
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / (n - 4) * 1e3
    return {"op": "NGNN train step, new batch every step", "graphs": graphs, "resident_batch_ms": res["resident"],
            "plain_collate_ms": res["plain"], "prefetch_prepare_ms": res["prefetch"],
            "graphs_per_s_prefetch": graphs / res["prefetch"] * 1e3}


def mamamm_case(b, n, d, dtype, dev):
    from pygho_amd import MaskedTensor
    from pygho_amd.backend.Mamamm import mamamm
    dn = synth.make_dense_batch(min(b, 256), seed=2, hidden=d, nmax=n)
    rep = max(1, b // min(b, 256))
    t = lambda a, dt=None: torch.from_numpy(a).to(dev).repeat((rep,) + (1,) * (a.ndim - 1)).to(dt) if dt else torch.from_numpy(a).to(dev).repeat((rep,) + (1,) * (a.ndim - 1))
    X = MaskedTensor(t(dn["X"], dtype), t(dn["Xmask"]), 0.0, True)
    A = MaskedTensor(t(dn["A"], dtype), t(dn["Amask"]), 0.0, True)
    ms = timed(lambda: mamamm(X, 2, A, 1, X.mask))
    bb = X.shape[0]
    es = X.raw.element_size()
    tensor = bb * n * n * d * es
    dense = 3 * tensor + bb * n * n                       # three full tensors + mask
    xv, av = float(X.mask.float().mean()), float(A.mask.float().mean())
    move = (xv + av) * tensor + tensor                    # has to move: unmasked operand rows in, every output row out
    flops = 2 * bb * d * n ** 3
    Xg = MaskedTensor(X.raw.clone().requires_grad_(True), X.mask, 0.0, True)
    Ag = MaskedTensor(A.raw.clone().requires_grad_(True), A.mask, 0.0, True)
    out = mamamm(Xg, 2, Ag, 1, X.mask).data
    g = torch.randn_like(out)
    msb = timed(lambda: torch.autograd.grad(out, (Xg.raw, Ag.raw), g, retain_graph=True), reps=10)
    res = [{"op": "mamamm(X,2,A,1) fwd (sparse adjacency mask -> neighbour lists)", "b": bb, "n": n, "d": d,
            "dtype": str(dtype).split(".")[-1], "ms": ms, "has_to_move_MB": move / 1e6, "GBps": move / ms / 1e6,
            "frac_hbm": move / ms / 1e6 / PEAK, "dense_MB": dense / 1e6, "frac_hbm_on_dense_bytes": dense / ms / 1e6 / PEAK,
            "TFLOPs": flops / ms / 1e9, "bwd_ms": msb}]
    # dense x dense (PPGN / 2-FWL): both operands carry the node-pair mask -> the matrix-core kernel
    Y = MaskedTensor(torch.randn_like(X.raw) * X.mask.unsqueeze(-1).to(X.raw.dtype), X.mask, 0.0, True)   # a second tensor: X X would alias
    ms2 = timed(lambda: mamamm(X, 2, Y, 1, X.mask))
    move2 = 2 * xv * tensor + tensor
    Xh = MaskedTensor(X.raw.clone().requires_grad_(True), X.mask, 0.0, True)
    Yh = MaskedTensor(Y.raw.clone().requires_grad_(True), X.mask, 0.0, True)
    out2 = mamamm(Xh, 2, Yh, 1, X.mask).data
    msb2 = timed(lambda: torch.autograd.grad(out2, (Xh.raw, Yh.raw), g, retain_graph=True), reps=10)
    res.append({"op": "mamamm(X,2,Y,1) fwd (dense x dense -> matrix cores)", "b": bb, "n": n, "d": d,
                "dtype": str(dtype).split(".")[-1], "ms": ms2, "has_to_move_MB": move2 / 1e6, "GBps": move2 / ms2 / 1e6,
                "frac_hbm": move2 / ms2 / 1e6 / PEAK, "dense_MB": dense / 1e6, "frac_hbm_on_dense_bytes": dense / ms2 / 1e6 / PEAK,
                "TFLOPs": flops / ms2 / 1e9, "bwd_ms": msb2})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--i2-only", action="store_true", help="only the I2-shape spspmm case (profiling runs)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")

    class _Emit(list):                     # every line is printed as soon as its case is done
        def append(self, r):
            print(json.dumps(r), flush=True)

        def extend(self, rs):
            for r in rs:
                self.append(r)
    out = _Emit()
    if args.i2_only:
        print(json.dumps(spspmm_case("i2", 256 if args.quick else 2048, 256, torch.bfloat16, dev)))
        return
    out.append(spspmm_case("zinc", 1024 if args.quick else 8192, 128, torch.bfloat16, dev))
    out.append(spspmm_case("zinc", 1024 if args.quick else 8192, 128, torch.float32, dev))
    out.append(spspmm_case("zinc", 128, 128, torch.bfloat16, dev))
    out.append(spspmm_case("i2", 256 if args.quick else 2048, 256, torch.bfloat16, dev))
    out.extend(mamamm_case(128, 37, 128, torch.bfloat16, dev))
    out.extend(mamamm_case(128 if args.quick else 1024, 37, 128, torch.bfloat16, dev))
    out.extend(mamamm_case(128 if args.quick else 1024, 37, 128, torch.float32, dev))
    out.extend(pooling_case(1024 if args.quick else 8192, 128, torch.bfloat16, dev))
    out.append(spmm_case(1024 if args.quick else 8192, 128, torch.bfloat16, dev))
    out.append(sunconv_case(128, 37, 128, torch.bfloat16, dev))
    out.append(sunconv_case(128 if args.quick else 1024, 37, 128, torch.bfloat16, dev))
    out.append(graph_step_case(1024, dev))
    out.append(fresh_batch_case(1024 if args.quick else 8192, dev))
    out.append(sampler_case("zinc", 1024 if args.quick else 8192, dev))
    out.append(sampler_case("i2", 256 if args.quick else 2048, dev))
    out.append(planner_case("zinc", 1024 if args.quick else 8192, dev))
    out.append(collate_case(1024 if args.quick else 8192, dev))
    out.append(dense_collate_case(128 if args.quick else 1024, 37, 128, dev))


if __name__ == "__main__":
    main()

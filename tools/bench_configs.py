#!/usr/bin/env python3
"""
The other measured configurations of BASELINE.json, as ONE object that bench.py puts into its JSON line (`configs`):

  config 5  3-tuple SparseTensor (I2-style), hidden 256 bf16: the spspmm aggregation launch at 2048 I2-shape graphs
            (pygho/backend/Spspmm.py:307-321 with key X___X___2___A___0) and one I2Conv layer forward + backward
            (pygho/honn/Conv.py:107-147)
  config 3  dense MaskedTensor path at (1024, 37, 37, 128) bf16: mamamm(X, 2, A, 1) and mamamm(X, 2, Y, 1)
            (pygho/backend/Mamamm.py:35-64) and one SUNConv("DD") layer forward + backward (pygho/honn/Conv.py:301-362)
  layers    one forward + backward of the other shipped sparse layers (NGNNConv, SSWLConv, SUNConv SS) at 8192 ZINC-shape graphs

Every kernel entry: {kernel, launches, avg_ms (HIP events on the launch stream), algorithmic_bytes, frac (of the 8 TB/s HBM
peak), traffic (HBM bytes per launch from the committed rocprofv3 --pmc passes of THIS script, keyed on the kernel sources; null
when the sources changed since)}.

    python tools/bench_configs.py [--quick]          # prints the object; tools/profile_configs.sh profiles this command
"""
import argparse
import glob
import hashlib
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

PEAK = 8000.0
MLP = {"numlayer": 1, "tailact": True, "norm": "bn", "act": "silu", "dp": 0.0}


def source_hash(names):
    h = hashlib.sha256()
    for name in names:
        h.update(open(os.path.join(REPO, "pygho_amd", "csrc", name), "rb").read())
    return h.hexdigest()


# kernel family -> the sources its traffic figure is tied to
SOURCES = {
    "seg": ("common.h", "seg_reduce.hip", "seg_window.hip", "seg_tile.hip"),
    "bmm": ("common.h", "masked_bmm.hip", "masked_bmm_blocks.h", "masked_bmm_lists.hip"),
}


def committed_traffic(key, family):
    """bytes per launch from profiles/*_configs_traffic.json (tools/collect_config_traffic.py) for entry `key`, or (None, why)"""
    want = source_hash(SOURCES[family])
    why = "no profiles/*_configs_traffic.json entry"
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*_configs_traffic.json")), reverse=True):
        try:
            ent = json.load(open(f)).get(key)
        except Exception:
            continue
        if ent is None:
            continue
        if ent.get("kernel_source_sha256") != want:
            why = f"{os.path.basename(f)} was measured on other kernel sources (stale)"
            continue
        return ent["traffic_bytes_per_launch"], os.path.basename(f)
    return None, why


def events_ms(fn, reps=20, warm=3):
    """mean and median launch time by HIP events recorded on the stream the op is launched on (torch's current stream)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ts) / len(ts), ts[len(ts) // 2]


def kernel_entry(key, family, kernel, launches, avg_ms, median_ms, alg_bytes, **extra):
    traffic, src = committed_traffic(key, family)
    e = {"kernel": kernel, "launches": launches, "avg_ms": avg_ms, "median_ms": median_ms, "algorithmic_bytes": alg_bytes,
         "achieved_GBps": alg_bytes / avg_ms / 1e6, "frac": alg_bytes / avg_ms / 1e6 / PEAK, "traffic": traffic, "traffic_source": src}
    e.update(extra)
    return e


def config5(dev, quick=False, kernels_only=False):
    """I2-shape 3-tuple stress: the aggregation launch and the layer around it"""
    from pygho_amd import _ops, synth
    import bench_layers
    graphs, d = (256 if quick else 2048), 256
    key = "X___X___2___A___0"
    hb = synth.make_batch(graphs, "i2", seed=1)                        # every graph distinct: no tiled index pattern
    acd = torch.from_numpy(hb.acd[key]).to(dev)
    nt, ne, m = hb.num_tuples, hb.num_edges, acd.shape[1]
    X = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    A = torch.randn(ne, d, device=dev).to(torch.bfloat16)
    plan = _ops.message_plan(acd, nt, nt, ne)
    timer = _ops.LaunchTimer()
    with timer:
        _ops.seg_gmr(nt, X, A, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum")
    torch.cuda.synchronize()
    variant = next(iter(timer.summary()))
    kern = "seg_gmr_tile_kernel<bf16,8>" if ",tiled" in variant else ("seg_gmr_window_kernel<bf16>" if ",window" in variant
                                                                      else "seg_gmr_fast_kernel<bf16,SUM,BOTH>")
    reps = 30
    avg, med = events_ms(lambda: _ops.seg_gmr(nt, X, A, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum"), reps)
    nbytes = 2 * d * (2 * nt + ne) + 8 * m + 4 * (nt + 1)             # every operand row once, every output row once, int32 indices
    out = {"workload": f"3-tuple SparseTensor (I2GNN-style) on subgraph-count-shape synthetic batches, hidden={d} bf16, {hb.num_graphs} graphs",
           "graphs": hb.num_graphs, "tuples": nt, "edges": ne, "msg_edges": m,
           "spspmm_fwd": kernel_entry("config5_spspmm_fwd", "seg", kern, reps, avg, med, nbytes, bytes_per_msg_edge=nbytes / m,
                                      msg_edges_per_s=m / avg * 1e3, dispatch=variant)}
    del X, A, plan, acd
    if kernels_only:
        return out
    r = bench_layers.case("I2Conv", graphs, dev)
    out["I2Conv_layer_fwd_bwd"] = {"ms": r["ms"], "graphs": r["graphs"], "graphs_per_s": r["graphs_per_s"], "tuples": r["tuples"], "d": r["d"],
                                   "msg_edges": r["msg_edges"]}
    return out


def config3(dev, quick=False, kernels_only=False):
    """dense MaskedTensor path: the two contractions and the SUNConv DD layer"""
    from pygho_amd import MaskedTensor, synth
    from pygho_amd.backend.Mamamm import mamamm
    import bench_ops
    b, n, d, dt = (128 if quick else 1024), 37, 128, torch.bfloat16
    dn = synth.make_dense_batch(b, seed=2, hidden=1, nmax=n)           # every graph distinct (the masks; values are drawn on the device)
    xm, am = torch.from_numpy(dn["Xmask"]).to(dev), torch.from_numpy(dn["Amask"]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(2)
    vals = lambda mask: (torch.randn(mask.shape + (d,), device=dev, generator=gen) * mask.unsqueeze(-1)).to(dt)
    X = MaskedTensor(vals(xm), xm, 0.0, True)
    A = MaskedTensor(vals(am), am, 0.0, True)
    Y = MaskedTensor(torch.randn_like(X.raw) * X.mask.unsqueeze(-1).to(dt), X.mask, 0.0, True)
    bb = X.shape[0]
    tensor = bb * n * n * d * 2
    dense = 3 * tensor + bb * n * n                                   # three full tensors + the mask
    xv, av = float(X.mask.float().mean()), float(A.mask.float().mean())
    flops = 2.0 * bb * d * n ** 3
    out = {"workload": f"SUNConv dense MaskedTensor path on padded ZINC-shape batches, ({bb}, {n}, {n}, {d}) bf16",
           "shape": [bb, n, n, d], "X_valid_fraction": xv, "A_valid_fraction": av}
    reps = 30
    for name, key, other, valid, kern in (("mamamm_X2_A1", "config3_mamamm_XA", A, av, "masked_bmm (adjacency-masked operand)"),
                                          ("mamamm_X2_Y1", "config3_mamamm_XY", Y, xv, "masked_bmm_blocks_kernel<bf16> (dense x dense)")):
        avg, med = events_ms(lambda: mamamm(X, 2, other, 1, X.mask), reps)
        move = (xv + valid) * tensor + tensor                        # has to move: unmasked operand rows in, every output row out
        out[name] = kernel_entry(key, "bmm", kern, reps, avg, med, move, dense_bytes=dense, frac_on_dense_bytes=dense / avg / 1e6 / PEAK,
                                 TFLOPs=flops / avg / 1e9, frac_mfma_peak_bf16_dense=flops / avg / 1e9 / 2500.0,
                                 note="algorithmic_bytes = the bytes that have to move (unmasked operand rows + every output row); "
                                      "dense_bytes = SURVEY 8(d)'s 3 b n^2 d s + b n^2")
    del X, A, Y
    if kernels_only:
        return out
    r = bench_ops.sunconv_case(b, n, d, dt, dev)
    out["SUNConv_DD_layer_fwd_bwd"] = {"ms": r["ms"], "graphs": r["b"], "graphs_per_s": r["graphs_per_s"], "n": n, "d": d}
    return out


def layers(dev, quick=False):
    """the other shipped sparse layers, forward + backward, ZINC-shape 2-tuples, hidden 128 bf16"""
    import bench_layers
    graphs = 1024 if quick else 8192
    out = {}
    for name in ("NGNNConv", "SSWLConv", "SUNConv"):
        r = bench_layers.case(name, graphs, dev)
        out[f"{name}_SS_layer_fwd_bwd"] = {"ms": r["ms"], "graphs": r["graphs"], "graphs_per_s": r["graphs_per_s"], "tuples": r["tuples"],
                                           "d": r["d"], "msg_edges": r["msg_edges"]}
    # NGNNConv as the model loop runs it (example/minimal.py:76-79): + residual, edge values = an embedding lookup -- the fused block
    # kernels (round 5: Linear -> BatchNorm -> act inside the forward aggregation, csrc/seg_fused.hip), with their per-launch figures
    r = bench_layers.case("NGNNConv", graphs, dev, kernels=True, residual_lookup=True)
    out["NGNNConv_SS_residual_lookup_layer_fwd_bwd"] = {"ms": r["ms"], "graphs": r["graphs"], "graphs_per_s": r["graphs_per_s"], "tuples": r["tuples"],
                                                        "d": r["d"], "msg_edges": r["msg_edges"],
                                                        "kernels": {k: v for k, v in r["kernels"].items() if k.startswith(("seg_fused", "seg_gmr"))}}
    # the reference's other first-class aggregation (pygho/backend/utils.py:44-56, --aggr in example/zinc.py): max, with the
    # per-launch figures of its backward (share pass + the two gradient plans on the 16-byte-per-lane kernels)
    r = bench_layers.case("NGNNConv", graphs, dev, aggr="max", kernels=True)
    out["NGNNConv_SS_max_layer_fwd_bwd"] = {"ms": r["ms"], "graphs": r["graphs"], "graphs_per_s": r["graphs_per_s"], "tuples": r["tuples"],
                                            "d": r["d"], "msg_edges": r["msg_edges"],
                                            "kernels": {k: v for k, v in r["kernels"].items() if k.startswith(("seg_ext", "seg_gmr"))}}
    return out


def ops(dev, quick=False):
    """a17 `spmm` (pygho/backend/Spmm.py:6-44: node-level message passing out[t] = sum_e val[e] * X[src[e]] over the batch adjacency):
    forward and, through autograd, both gradient launches, ZINC-shape adjacency of 8192 distinct graphs, d = 128 bf16.  The working
    set (E + 2 N rows = 200 MB) is BELOW the 256 MB Infinity Cache: the fraction is on algorithmic bytes, not an HBM claim."""
    from pygho_amd import _ops
    import bench_layers
    graphs, d = (1024 if quick else 8192), 128
    hb = bench_layers.batch(graphs, "zinc", ("X___X___1___A___0",))
    ei = torch.from_numpy(hb.edge_index).to(dev)
    src, tar = ei[1].contiguous(), ei[0].contiguous()
    n, e = hb.num_nodes, hb.num_edges
    val = torch.randn(e, d, device=dev).to(torch.bfloat16).requires_grad_(True)
    X = torch.randn(n, d, device=dev).to(torch.bfloat16).requires_grad_(True)
    w = torch.randn(n, d, device=dev).to(torch.bfloat16)

    def step():
        val.grad = X.grad = None
        _ops.spmm_values(val, X, src, tar, n, "sum").backward(w)
    for _ in range(3):
        step()
    timer = _ops.LaunchTimer()
    with timer:
        for _ in range(10):
            step()
    torch.cuda.synchronize()
    return {"spmm_fwd_bwd": {"graphs": hb.num_graphs, "edges": e, "nodes": n, "d": d, "dtype": "bfloat16",
                             "working_set_MB": 2 * d * (e + 2 * n) / 1e6,
                             "launches": {k: {"launches": v[0], "avg_ms": v[1], "algorithmic_bytes": v[2], "frac": v[2] / (v[1] * 1e-3) / 1e9 / PEAK}
                                          for k, v in timer.summary().items()}}}


def run(dev, quick=False, kernels_only=False):
    """`kernels_only`: the three kernel entries alone (the command of the --pmc passes: per-kernel counter means stay unmixed)"""
    out = {"config5": config5(dev, quick, kernels_only)}
    torch.cuda.empty_cache()
    out["config3"] = config3(dev, quick, kernels_only)
    torch.cuda.empty_cache()
    if not kernels_only:
        out["layers"] = layers(dev, quick)
        torch.cuda.empty_cache()
        out["ops"] = ops(dev, quick)
        torch.cuda.empty_cache()
    out["note"] = ("outside `value`: per-kernel roofline figures of BASELINE configs 3 and 5 and one forward + backward of every shipped "
                   "layer; HIP events around each launch, tools/bench_configs.py; rocprofv3 summaries of this script under profiles/")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--kernels-only", action="store_true")
    args = ap.parse_args()
    print(json.dumps(run(torch.device("cuda:0"), args.quick, args.kernels_only)))

#!/usr/bin/env python3
"""A/B of the segment kernels (fast / window / tiled) on the forward, by-tuple and by-edge plans of the I2 and ZINC shapes:
bit-equality against the fast kernel and median launch time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth  # noqa: E402
from pygho_amd._native import AGGR_CODE, check, dtype_code, lib, ptr, stream_ptr  # noqa: E402


def timed(fn, reps=20):
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


def fast(out, lhs, rhs, sp, li, ri, n, aggr="sum"):
    d = lhs.shape[1]
    dev = lhs.device
    check(lib().pygho_seg_gather_mul_reduce(ptr(out), ptr(lhs), ptr(rhs), ptr(sp), ptr(li), ptr(ri), None, n, d, d, d, lhs.shape[0],
                                            rhs.shape[0], dtype_code(lhs), AGGR_CODE[aggr], stream_ptr(dev)), "fast")


def window(out, lhs, rhs, sp, li, ri, n, aggr="sum"):
    d = lhs.shape[1]
    dev = lhs.device
    check(lib().pygho_seg_gather_mul_reduce_window(ptr(out), None, ptr(lhs), ptr(rhs), ptr(sp), ptr(li), ptr(ri), None, n, d,
                                                   lhs.shape[0], rhs.shape[0], dtype_code(lhs), AGGR_CODE[aggr], stream_ptr(dev)), "window")


def tiled(out, lhs, rhs, sp, li, ri, n, w, tp, aggr="sum"):
    d = lhs.shape[1]
    dev = lhs.device
    check(lib().pygho_seg_gather_mul_reduce_tiled(ptr(out), None, ptr(lhs), ptr(rhs), ptr(sp), ptr(li), ptr(ri), None, ptr(tp[0]), ptr(tp[1]),
                                                  n, d, lhs.shape[0], rhs.shape[0], w, dtype_code(lhs), AGGR_CODE[aggr], stream_ptr(dev)), "tiled")


def case(kind, graphs, d, dtype, dev, wins=(24, 32)):
    base = 1024 if kind == "zinc" else 128
    key = "X___X___1___A___0" if kind == "zinc" else "X___X___2___A___0"
    hb = synth.replicate(synth.make_batch(min(graphs, base), kind, seed=1), max(1, graphs // base))
    acd = torch.from_numpy(hb.acd[key]).to(dev)
    nt, ne, m = hb.num_tuples, hb.num_edges, acd.shape[1]
    X = torch.randn(nt, d, device=dev).to(dtype)
    A = torch.randn(ne, d, device=dev).to(dtype)
    G = torch.randn(nt, d, device=dev).to(dtype)
    plan = _ops.message_plan(acd, nt, nt, ne)
    pc, a_c, d_c = plan.by_c()
    pd, a_d, c_d = plan.by_d()
    es = X.element_size()
    cases = [("fwd", nt, X, A, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd), ("by_c", nt, G, A, pc.seg_ptr, a_c, d_c),
             ("by_d", ne, G, X, pd.seg_ptr, a_d, c_d)]
    for name, n, lhs, rhs, sp, li, ri in cases:
        nbytes = es * d * (lhs.shape[0] + rhs.shape[0] + n) + 8 * m + 4 * (n + 1)
        ref = torch.empty((n, d), dtype=dtype, device=dev)
        fast(ref, lhs, rhs, sp, li, ri, n)
        rec = {"shape": kind, "graphs": hb.num_graphs, "d": d, "dtype": str(dtype).split(".")[-1], "plan": name, "segments": n, "messages": m,
               "alg_MB": nbytes / 1e6}
        rec["fast_ms"] = timed(lambda: fast(ref, lhs, rhs, sp, li, ri, n))
        rb = d * es
        if rb % 16 == 0 and rb <= 1024:
            o = torch.empty_like(ref)
            try:
                window(o, lhs, rhs, sp, li, ri, n)
                rec["window_equal"] = bool(torch.equal(o.view(torch.int16 if es == 2 else torch.int32), ref.view(torch.int16 if es == 2 else torch.int32)))
                rec["window_ms"] = timed(lambda: window(o, lhs, rhs, sp, li, ri, n))
            except RuntimeError as e:
                rec["window_err"] = str(e)[:80]
        if rb in (256, 512, 1024):
            for w in wins:
                tp = _ops.tile_plan(sp, li, n, w)
                cnt = tp[0].to(torch.int64)
                ntile = int(cnt.sum())
                o = torch.full_like(ref, float("nan"))
                tiled(o, lhs, rhs, sp, li, ri, n, w, tp)
                eq = bool(torch.equal(o.view(torch.int16 if es == 2 else torch.int32), ref.view(torch.int16 if es == 2 else torch.int32)))
                rec[f"tile{w}_equal"] = eq
                rec[f"tile{w}_tiles"] = ntile
                rec[f"tile{w}_segs_per_tile"] = n / max(ntile, 1)
                rec[f"tile{w}_ms"] = timed(lambda: tiled(o, lhs, rhs, sp, li, ri, n, w, tp))
                rec[f"tile{w}_frac"] = nbytes / rec[f"tile{w}_ms"] / 1e6 / 8000.0
        rec["fast_frac"] = nbytes / rec["fast_ms"] / 1e6 / 8000.0
        if "window_ms" in rec:
            rec["window_frac"] = nbytes / rec["window_ms"] / 1e6 / 8000.0
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    which = sys.argv[1:] or ["i2", "zinc"]
    if "i2" in which:
        case("i2", 2048, 256, torch.bfloat16, dev)
        case("i2", 2048, 128, torch.bfloat16, dev)
        case("i2", 2048, 128, torch.float32, dev)
    if "zinc" in which:
        case("zinc", 8192, 128, torch.bfloat16, dev)
        case("zinc", 8192, 128, torch.float32, dev)

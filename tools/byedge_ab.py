#!/usr/bin/env python3
"""A/B of the by-edge gradient forms at BASELINE size: the scatter kernel (csrc/seg_scatter.hip) against the gather form on the
window kernel, same plan, same operands, HIP events around every launch; optional second shape (I2, d = 256).

    python tools/byedge_ab.py [--graphs 8192] [--i2]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth                    # noqa: E402
from pygho_amd import segment as S                   # noqa: E402


def events_ms(fn, reps=30, warm=5):
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
    return sum(ts) / len(ts), ts[len(ts) // 2], ts[0]


def case(kind, graphs, d, key, dev):
    hb = synth.make_batch(graphs, kind, seed=1000)
    acd = torch.from_numpy(hb.acd[key]).to(dev)
    nt, ne = hb.num_tuples, hb.num_edges
    plan = _ops.message_plan(acd, nt, nt, ne)
    g = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    h = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    add = torch.randn(ne, d, device=dev).to(torch.bfloat16)
    p, a_g, c_g = plan.by_d()
    sp = S.scatter_plan(plan)
    out = {"kind": kind, "graphs": graphs, "d": d, "tuples": nt, "edges": ne, "msgs": int(acd.shape[1]),
           "scatter_plan": None if sp is None else {"blocks": sp.n_blocks, "chunks": sp.n_chunks, "max_edges": sp.max_edges,
                                                    "msgs_per_chunk": acd.shape[1] / sp.n_chunks}}
    for name, addend in (("plain", None), ("res", add)):
        nbytes = 2 * d * (2 * nt + ne * (2 if addend is not None else 1)) + 8 * acd.shape[1] + 4 * (ne + 1)
        gather = events_ms(lambda: _ops.seg_gmr(ne, g, h, p.seg_ptr, a_g, c_g, "sum", None, addend=addend))
        out[f"gather_{name}_ms"] = gather
        out[f"gather_{name}_TBps"] = nbytes / gather[0] / 1e9
        if sp is not None:
            sc = events_ms(lambda: S.by_edge_product(plan, g, h, None, addend))
            out[f"scatter_{name}_ms"] = sc
            out[f"scatter_{name}_TBps"] = nbytes / sc[0] / 1e9
            out[f"equal_{name}"] = bool(torch.equal(S.by_edge_product(plan, g, h, None, addend),
                                                    _ops.seg_gmr(ne, g, h, p.seg_ptr, a_g, c_g, "sum", None, addend=addend)))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=8192)
    ap.add_argument("--i2", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    print(json.dumps(case("zinc", args.graphs, 128, "X___X___1___A___0", dev)), flush=True)
    if args.i2:
        print(json.dumps(case("i2", args.graphs // 4, 256, "X___X___2___A___0", dev)), flush=True)

#!/usr/bin/env python3
"""A/B of the fused block forward (csrc/seg_fused.hip: Linear -> BatchNorm -> act folded into the aggregation's load path) against the
two launches it replaces (rowblock_linear_bn_act, then seg_gmr with the residual row), at the BASELINE shape (8192 ZINC-shape graphs,
width 128 bf16): same bits, HIP-event means.  usage: fused_fwd_ab.py [graphs] [reps] [--no-h]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth          # noqa: E402
from pygho_amd import segment as S         # noqa: E402
from pygho_amd import blocks as B          # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
graphs = int(args[0]) if args else 8192
reps = int(args[1]) if len(args) > 1 else 20
want_h = "--no-h" not in sys.argv
dev = torch.device("cuda:0")
d, key = 128, "X___X___1___A___0"
hb = synth.make_batch(graphs, "zinc", seed=1000)
acd = torch.from_numpy(hb.acd[key]).to(dev)
nt, ne = hb.num_tuples, hb.num_edges
plan = _ops.message_plan(acd, nt, nt, ne)
ea = torch.from_numpy(hb.edge_attr).to(dev).long()
torch.manual_seed(0)
x = torch.randn(nt, d, device=dev).to(torch.bfloat16)
wl = (torch.randn(d, d, device=dev) / d ** 0.5).to(torch.bfloat16)
bias = (torch.randn(d, device=dev) * 0.1).to(torch.bfloat16)
scale = (torch.rand(d, device=dev) + 0.5).float()
shift = (torch.randn(d, device=dev) * 0.1).float()
table = torch.randn(16, d, device=dev).to(torch.bfloat16)
look_fwd = plan.lookup(ea)[0]
fp = S.fused_plan(plan)
assert fp is not None, "the plan is outside the fused kernel's limits"
res = {"graphs": graphs, "tuples": nt, "messages": plan.m, "chunks": fp.n_chunks, "messages_per_chunk": plan.m / fp.n_chunks,
       "rows_per_chunk": nt / fp.n_chunks, "stores_h": want_h}


def separate():
    h = B.rowblock_linear_bn_act(x, wl, bias, scale, shift, "silu")
    return S.seg_gmr(nt, h, table, plan.fwd.seg_ptr, plan.c_fwd, look_fwd, "sum", addend=x), h


def fused():
    return S.fused_forward(x, wl, bias, scale, shift, "silu", table, look_fwd, plan, fp, "sum", True, want_h)


o_ref, h_ref = separate()
o, h = fused()
torch.cuda.synchronize()
res["out_bit_identical"] = bool(torch.equal(o, o_ref))
if not res["out_bit_identical"]:
    bad = (o != o_ref).any(1)
    res["rows_differing"] = int(bad.sum())
    res["max_abs_diff"] = float((o.float() - o_ref.float()).abs().max())
    res["first_bad_rows"] = bad.nonzero().flatten()[:8].tolist()
if want_h:
    used = torch.unique(plan.c_fwd.long())
    res["h_rows_read_by_messages"] = int(used.numel())
    res["h_bit_identical_on_those"] = bool(torch.equal(h[used], h_ref[used]))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res["separate_ms"] = timed(separate)
res["linear_bn_act_ms"] = timed(lambda: B.rowblock_linear_bn_act(x, wl, bias, scale, shift, "silu"))
res["seg_gmr_ms"] = timed(lambda: S.seg_gmr(nt, h_ref, table, plan.fwd.seg_ptr, plan.c_fwd, look_fwd, "sum", addend=x))
res["fused_ms"] = timed(fused)
nbytes = 2 * d * nt * (3 if want_h else 2) + 12 * plan.m + 4 * (nt + 1)
res["fused_has_to_move_GB"] = nbytes / 1e9
res["fused_frac_of_8TBs"] = nbytes / res["fused_ms"] / 1e6 / 8000
print(json.dumps(res))

#!/usr/bin/env python3
"""A/B of the fused backward (csrc/seg_dual.hip: by-tuple AND by-edge gradient of a layer's aggregation in one pass) against the two
launches it replaces (seg_gmr over the by-c plan, seg_scatter), at the BASELINE shape (8192 ZINC-shape graphs, width 128 bf16): same
bits, HIP-event means.  usage: dual_bwd_ab.py [graphs] [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth          # noqa: E402
from pygho_amd import segment as S         # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
graphs = int(args[0]) if args else 8192
reps = int(args[1]) if len(args) > 1 else 20
dev = torch.device("cuda:0")
d, key = 128, "X___X___1___A___0"
hb = synth.make_batch(graphs, "zinc", seed=1000)
acd = torch.from_numpy(hb.acd[key]).to(dev)
nt, ne = hb.num_tuples, hb.num_edges
_ops.SEG_SCATTER_MIN_MESSAGES = 0
plan = _ops.message_plan(acd, nt, nt, ne)
ea = torch.from_numpy(hb.edge_attr).to(dev).long()
torch.manual_seed(0)
g = torch.randn(nt, d, device=dev).to(torch.bfloat16)
h = torch.randn(nt, d, device=dev).to(torch.bfloat16)
table = torch.randn(16, d, device=dev).to(torch.bfloat16)
addend = torch.randn(ne, d, device=dev).to(torch.bfloat16)
look_fwd, look_byc = plan.lookup(_ops.flat_index(ea))[:2]
sp = S.scatter_plan(plan)
assert sp is not None and sp.cgap is not None, "no aligned plan"
pc, a_byc, _ = plan.by_c()
res = {"graphs": graphs, "tuples": nt, "messages": plan.m, "edges": ne, "chunks": sp.n_chunks, "messages_per_chunk": plan.m / sp.n_chunks,
       "max_edges": sp.max_edges, "covers_c": sp.covers_c}


def by_tuple():
    return _ops.seg_gmr(nt, g, table, pc.seg_ptr, a_byc, look_byc, "sum")


def by_edge(add):
    return _ops.by_edge_product(plan, g, h, None, addend=addend if add else None)


def dual(add):
    return _ops.dual_backward(plan, g, h, table, look_byc, addend=addend if add else None)


for add in (False, True):
    gh, gr = dual(add)
    torch.cuda.synchronize()
    res[f"bit_identical{'_chained' if add else ''}"] = bool(torch.equal(gh, by_tuple()) and torch.equal(gr, by_edge(add)))


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


res["by_tuple_ms"] = timed(by_tuple)
res["by_edge_ms"] = timed(lambda: by_edge(False))
res["by_edge_chained_ms"] = timed(lambda: by_edge(True))
res["dual_ms"] = timed(lambda: dual(False))
res["dual_chained_ms"] = timed(lambda: dual(True))
# the table-gradient form (what the training step runs): gh bit for bit, the table's gradient against the per-edge gradient summed in f64
gh_tg, g_tab = S.dual_backward_tg(plan, g, h, table, look_fwd, look_byc)
res["tg_gh_bit_identical"] = bool(torch.equal(gh_tg, by_tuple()))
ref_tab = torch.zeros(16, d, dtype=torch.float64, device=dev).index_add_(0, ea, by_edge(False).double())
res["tg_table_grad_max_rel_err_vs_summed_per_edge_rows"] = float((g_tab.double() - ref_tab).abs().max() / ref_tab.abs().max())
res["dual_tg_ms"] = timed(lambda: S.dual_backward_tg(plan, g, h, table, look_fwd, look_byc))
nb_tg = 2 * d * 3 * nt + 16 * plan.m + 4 * (nt + 1) + 20 * sp.n_chunks
res["dual_tg_has_to_move_GB"] = nb_tg / 1e9
res["dual_tg_frac_of_8TBs"] = nb_tg / res["dual_tg_ms"] / 1e6 / 8000
nbytes = 2 * d * (3 * nt + 2 * ne) + 12 * plan.m + 4 * (nt + 1) + 20 * sp.n_chunks
res["dual_chained_has_to_move_GB"] = nbytes / 1e9
res["dual_chained_frac_of_8TBs"] = nbytes / res["dual_chained_ms"] / 1e6 / 8000
print(json.dumps(res))

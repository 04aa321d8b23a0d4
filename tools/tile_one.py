#!/usr/bin/env python3
"""One segment kernel (fast / window / tiled) at the I2 or ZINC forward plan, N launches (median by HIP events); index
substitution modes for upper-bound runs (rhs0: every rhs gather hits one row; lhs_self: lhs index = output slot).
usage: tile_one.py <i2|zinc> <kernel: fast|window|tiled> <W> <mode: real|rhs0|lhs_self> [reps]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops, synth
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tile_ab import fast, window, tiled, timed
kind, kern, w, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
d, graphs, base, key = (256, 2048, 128, "X___X___2___A___0") if kind == "i2" else (128, 8192, 1024, "X___X___1___A___0")
hb = synth.replicate(synth.make_batch(base, kind, seed=1), graphs // base)
acd = torch.from_numpy(hb.acd[key]).to(dev)
nt, ne, m = hb.num_tuples, hb.num_edges, acd.shape[1]
X = torch.randn(nt, d, device=dev).to(torch.bfloat16)
A = torch.randn(ne, d, device=dev).to(torch.bfloat16)
plan = _ops.message_plan(acd, nt, nt, ne)
sp, li, ri = plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd
if mode == "rhs0":
    ri = torch.zeros_like(ri)
if mode == "lhs_self":
    li = _ops.narrow_i32(acd[0].contiguous())
out = torch.empty((nt, d), dtype=torch.bfloat16, device=dev)
if kern == "fast":
    fn = lambda: fast(out, X, A, sp, li, ri, nt)
elif kern == "window":
    fn = lambda: window(out, X, A, sp, li, ri, nt)
else:
    tp = _ops.tile_plan(sp, li, nt, w)
    fn = lambda: tiled(out, X, A, sp, li, ri, nt, w, tp)
ms = timed(fn, reps)
nbytes = 2 * d * (2 * nt + ne) + 8 * m + 4 * (nt + 1)
print(json.dumps({"kind": kind, "kernel": kern, "W": w, "mode": mode, "ms": ms, "frac": nbytes / ms / 1e6 / 8000, "alg_GB": nbytes / 1e9}))

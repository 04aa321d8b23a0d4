#!/usr/bin/env python3
"""
Micro-benchmark of the kernels of the dense (MaskedTensor) path at the config-3 shape (b, n, n, d) = (1024, 37, 37, 128) bf16, padded
ZINC-shape batch (X mask = node-mask outer product, A mask = adjacency): masked_bmm forward, masked fill / reductions / broadcast,
masked_pair_combine.  Also the command the PMC passes of profiles/r01_pmc_masked.md were taken on.

    python tools/masked_bench.py [--reps 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import MaskedTensor, _ops, synth            # noqa: E402
from pygho_amd.backend.Mamamm import mamamm                  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--b", type=int, default=1024)
    ap.add_argument("--only", default="", help="run only the cases whose name contains this substring (profiling passes)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, d, dt = 37, 128, torch.bfloat16
    dn = synth.make_dense_batch(256, seed=2, hidden=d, nmax=n)
    rep = args.b // 256
    t = lambda a, to=None: (lambda v: v.to(to) if to else v)(torch.from_numpy(a).to(dev).repeat((rep,) + (1,) * (a.ndim - 1)))
    xraw, xm = t(dn["X"], dt), t(dn["Xmask"])
    araw, am = t(dn["A"], dt), t(dn["Amask"])
    X = MaskedTensor(xraw, xm, 0.0, True)
    A = MaskedTensor(araw, am, 0.0, True)
    b = xraw.shape[0]
    tensor = b * n * n * d * 2
    valid = float(xm.float().mean())
    node = torch.randn((b, n, d), device=dev).to(dt)
    out = []

    def rec(name, fn, alg_bytes, real_bytes=None, flops=None):
        if args.only and args.only not in name:
            return
        ms = timed(fn, args.reps)
        # frac_hbm is quoted on the bytes that HAVE TO MOVE (unmasked operand rows in + every output row out); the dense
        # figure of SURVEY.md 8(d) (three full tensors + mask) counts padded rows no mask-aware kernel fetches and is kept
        # as a second field only
        move = real_bytes if real_bytes is not None else alg_bytes
        r = {"kernel": name, "ms": ms, "has_to_move_MB": move / 1e6, "GBps": move / ms / 1e6, "frac_hbm": move / ms / 1e6 / 8000.0,
             "dense_MB": alg_bytes / 1e6, "frac_hbm_on_dense_bytes": alg_bytes / ms / 1e6 / 8000.0}
        if flops is not None:
            r["TFLOPs"] = flops / ms / 1e9
            r["frac_mfma_peak_bf16_dense"] = flops / ms / 1e9 / 2500.0
        out.append(r)

    a_valid = float(am.float().mean())
    need = (valid + a_valid) * tensor + tensor                  # unmasked operand rows in, every output row out
    flops = 2.0 * b * d * n ** 3
    # the adjacency mask is 3.6 % dense: mamamm dispatches to the neighbour-list kernel; the matrix-core kernel is timed on the
    # same inputs with the dispatch switched off (dense x dense contractions always take it)
    rec("masked_bmm_lists_kernel<bf16> (mamamm(X,2,A,1) fwd, sparse-mask dispatch)", lambda: mamamm(X, 2, A, 1, xm),
        3 * tensor + b * n * n, need, flops)

    def dense_path():
        _ops.USE_BMM_LISTS = False
        try:
            return mamamm(X, 2, A, 1, xm)
        finally:
            _ops.USE_BMM_LISTS = True
    rec("masked_bmm matrix-core kernel <bf16> (same contraction X A, dispatch to the lists switched off)", dense_path,
        3 * tensor + b * n * n, need, flops)
    # dense x dense: the PPGN / 2-FWL contraction X X (both operands 40 % valid) -- the case the matrix-core kernel serves
    # (two DIFFERENT tensors: X X would let the second operand's rows hit in L2 and halve the read volume)
    Y = MaskedTensor(torch.randn_like(xraw) * xm.unsqueeze(-1).to(dt), xm, 0.0, True)
    rec("masked_bmm matrix-core kernel <bf16> (mamamm(X,2,Y,1): dense x dense, PPGN)", lambda: mamamm(X, 2, Y, 1, xm),
        3 * tensor + b * n * n, 2 * valid * tensor + tensor, flops)
    xf, yf = MaskedTensor(xraw.float(), xm, 0.0, True), MaskedTensor(Y.raw.float(), xm, 0.0, True)
    rec("masked_bmm matrix-core kernel <f32> (mamamm(X,2,Y,1))", lambda: mamamm(xf, 2, yf, 1, xm),
        2 * (3 * tensor) + b * n * n, 2 * (2 * valid * tensor + tensor), flops)
    del xf, yf
    rec("masked_fill_vec_kernel", lambda: _ops.masked_fill(xraw, xm, 0.0), 2 * tensor, (1 + valid) * tensor)
    rec("masked_reduce_vec_kernel (sum over dim 1)", lambda: _ops.masked_reduce(xraw, xm, 1, "sum"), tensor, valid * tensor)
    rec("masked_reduce_vec_kernel (sum over dim 2)", lambda: _ops.masked_reduce(xraw, xm, 2, "sum"), tensor, valid * tensor)
    rec("masked_broadcast_vec_kernel (unpool dim 1)", lambda: _ops.masked_broadcast(node, xm, 1, 0.0, 2), tensor)
    rec("masked_pair_combine_kernel (base + u + v, diag select)",
        lambda: _ops.masked_pair_combine(xraw, node, node, node, True, xm, tuple(xraw.shape), dt, dev), 2 * tensor, (1 + valid) * tensor)
    rec("masked_pair_combine_kernel (views gradient: u + v + diag)",
        lambda: _ops.masked_pair_combine(None, node, node, node, False, xm, tuple(xraw.shape), dt, dev), tensor)
    meta = {"shape": [b, n, n, d], "dtype": "bfloat16", "tensor_MB": tensor / 1e6, "X_valid_fraction": valid, "A_valid_fraction": a_valid,
            "bmm_variant": os.environ.get("PYGHO_BMM_VARIANT", "blocks (default)")}
    print(json.dumps({"meta": meta, "kernels": out}))


if __name__ == "__main__":
    main()

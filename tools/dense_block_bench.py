#!/usr/bin/env python3
"""
Micro-benchmark of the dense half of the fused tuple-wise block at the headline size (nnz = 1.79 M rows, d = 128, bf16):
rowblock_linear (+ BatchNorm statistics / residual epilogue) against the library GEMM + separate passes, and the one-pass
backward kernels.  Also the command the PMC passes of profiles/r01_pmc_dense_block.md were taken on.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygho_amd import _ops  # noqa: E402


def timed(fn, reps=20):
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
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m, d, dt = 1_793_216, 128, torch.bfloat16
    x = torch.randn(m, d, device=dev).to(dt)
    w = (torch.randn(d, d, device=dev) / d ** 0.5).to(dt)
    b = torch.randn(d, device=dev).to(dt)
    g = torch.randn(m, d, device=dev).to(dt)
    gh = torch.randn(m, d, device=dev).to(dt)
    stream = 2 * m * d  # bytes of one (m, d) bf16 tensor
    out = []
    shift = torch.nn.functional.linear(x[:1], w, b).float().reshape(-1)
    t = timed(lambda: _ops.rowblock_linear(x, w, b, stats_shift=shift))
    out.append({"op": "rowblock_linear + BN statistics", "ms": t, "alg_MB": 2 * stream / 1e6, "GBps": 2 * stream / t / 1e6})
    t = timed(lambda: torch.nn.functional.linear(x, w, b))
    out.append({"op": "library GEMM (same product)", "ms": t, "alg_MB": 2 * stream / 1e6, "GBps": 2 * stream / t / 1e6})
    bn = torch.nn.BatchNorm1d(d).to(dev)
    pre, _ = _ops.rowblock_linear(x, w, b)
    _, _, _, saved = _ops._bn_forward(pre, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, 1e-5, "silu")
    t = timed(lambda: _ops.bn_bwd_linear(pre, gh, saved, True, "silu", w, g, True, x=x))
    out.append({"op": "BN-backward sums + bn_bwd_linear_dw (gX, gW, gb)", "ms": t, "alg_MB": 7 * stream / 1e6, "GBps": 7 * stream / t / 1e6})
    t = timed(lambda: _ops.weight_grad_splitk(g, x, torch.float32, want_colsum=True))
    out.append({"op": "weight_grad (gW, gb)", "ms": t, "alg_MB": 2 * stream / 1e6, "GBps": 2 * stream / t / 1e6})
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()

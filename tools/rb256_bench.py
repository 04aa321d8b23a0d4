"""width-256 streaming passes (csrc/rowblock_linear.hip, column-split) at BASELINE config 5's row count against the library path they
replace: ms per launch and GB/s of algorithmic bytes.  python tools/rb256_bench.py [--rows 2386620]"""
import argparse
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2386620)
    ap.add_argument("--d", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m, d, dt = args.rows, args.d, torch.bfloat16
    torch.manual_seed(0)
    x = (torch.randn(m, d, device=dev) * 0.9).to(dt)
    w = (torch.randn(d, d, device=dev) / d ** 0.5).to(dt)
    b = (torch.randn(d, device=dev) * 0.2).to(dt)
    gh = torch.randn(m, d, device=dev).to(dt)
    g = torch.randn(m, d, device=dev).to(dt)
    bn = torch.nn.BatchNorm1d(d).to(dev).train()
    S = m * d * 2
    out = {"rows": m, "d": d, "stream_GB": S / 1e9}

    def rec(name, ms, streams):
        out[name] = {"ms": round(ms, 4), "streams": streams, "GBps": round(streams * S / ms / 1e6, 1)}
    pre, partial = _ops.rowblock_linear(x, w, b, stats_shift=True)
    _, _, _, saved = _ops._bn_forward(pre, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps, "silu", None, partial)
    scale, shift = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    s1, s2 = _ops.rowblock_linear_bwd_sums(x, w, b, gh, saved, "silu")
    rec("linear_store", timed(lambda: _ops.rowblock_linear(x, w, b)), 2)
    rec("linear_store_stats", timed(lambda: _ops.rowblock_linear(x, w, b, stats_shift=True)), 2)
    rec("stats_only", timed(lambda: _ops.rowblock_linear(x, w, b, stats_shift=True, store=False)), 1)
    rec("linear_bn_act", timed(lambda: _ops.rowblock_linear_bn_act(x, w, b, scale, shift, "silu")), 2)
    rec("linear_bn_act_residual", timed(lambda: _ops.rowblock_linear_bn_act(x, w, b, scale, shift, "silu", g)), 3)
    rec("bwd_sums", timed(lambda: _ops.rowblock_linear_bwd_sums(x, w, b, gh, saved, "silu")), 2)
    rec("bwd_apply", timed(lambda: _ops.rowblock_linear_bwd_apply(x, w, b, gh, saved, (s1, s2), "silu", True, True)), 3)
    rec("gx_with_residual", timed(lambda: _ops.rowblock_linear(gh, w.t().contiguous(), None, addend=g)), 3)
    # the library path of round 4
    rec("lib_linear", timed(lambda: torch.nn.functional.linear(x, w, b)), 2)
    rec("lib_bn_stats+apply", timed(lambda: _ops._bn_forward(pre, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps, "silu")), 3)
    rec("lib_bn_backward", timed(lambda: _ops._bn_backward(pre, gh, saved, True, "silu", want_colsum=True)), 5)
    rec("lib_gx", timed(lambda: gh @ w), 2)
    rec("lib_gx_add", timed(lambda: (gh @ w).add_(g)), 5)
    rec("lib_dw", timed(lambda: torch.mm(gh.t(), x, out_dtype=torch.float32), reps=5), 2)
    _ops.FORCE_DW_BLOCKS = True
    rec("dw_2x2_blocks", timed(lambda: _ops.weight_grad_splitk(gh, x, torch.float32)), 4)
    _ops.FORCE_DW_BLOCKS = False
    rec("lib_dw_batched_splitk", timed(lambda: _ops.weight_grad_splitk(gh, x, torch.float32), reps=5), 2)
    rec("bn_act_fwd_generic", timed(lambda: _ops._bn_forward(pre, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps, "silu", None, partial)), 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

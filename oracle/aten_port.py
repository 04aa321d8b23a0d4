"""
CPU port of the reference's hot path as the SAME ATen op sequence it issues  --  TEST / BASELINE
INFRASTRUCTURE ONLY (nothing under pygho_amd/ may import this).

The reference's "kernels" are ATen calls from Python; its CPU performance is therefore the performance of
this op sequence on torch-CPU.  ``bench.py``'s ``cpu_baseline`` leg times it on the GPU box's host cores
(kind "port").  Every function is pinned against the golden fixtures in tests/test_oracle_golden.py.

  scatter_reduce   pygho/backend/utils.py:44-56     zeros + scatter_reduce_(expanded int64 index, include_self=False)
  spspmm_values    pygho/backend/Spspmm.py:309-315  index, index, mul, scatter_reduce
  spmm_values      pygho/backend/Spmm.py:40-44      mul, index, scatter_reduce
  NGNNPort         example/minimal.py:37-85 + honn/Conv.py:53-58 (6 x [MLP, spspmm] + residual, pooling, readout)
"""
import torch
import torch.nn as nn
from torch import Tensor


def scatter_reduce(src: Tensor, ind: Tensor, dim_size: int, aggr: str) -> Tensor:
    red = {"sum": "sum", "mean": "mean", "max": "amax", "min": "amin"}[aggr]
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    index = ind.reshape((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    return out.scatter_reduce_(0, index, src, red, include_self=False)


def spspmm_values(valA, valB, acd: Tensor, n_out: int, aggr: str = "sum") -> Tensor:
    if valA is None:
        msg = valB[acd[2]]
    elif valB is None:
        msg = valA[acd[1]]
    else:
        msg = valA[acd[1]] * valB[acd[2]]
    return scatter_reduce(msg, acd[0], n_out, aggr)


def spspmm_values_chunked(valA: Tensor, valB: Tensor, out_idx: Tensor, a_idx: Tensor, b_idx: Tensor, n_out: int, aggr: str = "sum",
                          chunk: int = 1 << 20, a_rowscale: Tensor = None) -> Tensor:
    """the same value computation (Spspmm.py:309-315: index, index, mul, scatter-reduce) for message counts whose (M, d)
    temporaries do not fit at once: f32 products accumulated with index_add_ chunk by chunk.  index_add_ adds the rows of a chunk in
    index order, so every output row is summed in MESSAGE ORDER exactly like the one-shot scatter_reduce_ (checked against it in
    tests/test_oracle_golden.py); `mean` divides by the message count as scatter_reduce_(include_self=False) does.
    (out_idx, a_idx, b_idx) = (acd[0], acd[1], acd[2]) for the forward, permuted for the two gradient plans.
    `a_rowscale` (f32, one factor per row of valA): every message is a_rowscale[a] * (valA[a] * valB[b]) -- the gradient of `mean`
    with the division by the count applied PER MESSAGE after the product, which is the association the device kernels use
    (autograd's own is (g / count) * v: one rounding elsewhere, same value to 1 ulp)."""
    assert aggr in ("sum", "mean", "max", "min") and valA.dtype == torch.float32 and valB.dtype == torch.float32
    if aggr in ("max", "min"):
        # scatter_reduce_(amax | amin, include_self=False) into zeros (utils.py:50-55), chunk by chunk: the running extremum starts at
        # -inf / +inf and rows without a message end as 0, which is what include_self=False leaves in the zero-initialised output
        assert a_rowscale is None
        out = torch.full((n_out, valA.shape[1]), float("-inf") if aggr == "max" else float("inf"), dtype=torch.float32)
        for lo in range(0, out_idx.numel(), chunk):
            sl = slice(lo, lo + chunk)
            out.index_reduce_(0, out_idx[sl], valA[a_idx[sl]] * valB[b_idx[sl]], "amax" if aggr == "max" else "amin", include_self=True)
        out[torch.bincount(out_idx, minlength=n_out) == 0] = 0.0
        return out
    out = torch.zeros((n_out, valA.shape[1]), dtype=torch.float32)
    for lo in range(0, out_idx.numel(), chunk):
        sl = slice(lo, lo + chunk)
        prod = valA[a_idx[sl]] * valB[b_idx[sl]]
        if a_rowscale is not None:
            prod = a_rowscale[a_idx[sl]].unsqueeze(1) * prod
        out.index_add_(0, out_idx[sl], prod)
    if aggr == "mean":
        cnt = torch.bincount(out_idx, minlength=n_out).clamp_min(1).to(torch.float32)
        out = out / cnt.unsqueeze(1)
    return out


def spspmm_extremum_grads_chunked(valA: Tensor, valB: Tensor, acd: Tensor, n_out: int, fwd: Tensor, gout: Tensor,
                                  store_dtype: torch.dtype = torch.float32, chunk: int = 1 << 20):
    """gradients of spspmm_values(..., aggr = max | min) wrt valA and valB, chunk by chunk, as autograd computes them for the
    reference's op sequence (Spspmm.py:309-315 + utils.py:50-55): scatter_reduce_(amax)'s backward hands every message that attains
    the extremum grad / N_to_distribute (N = number of ties, the quotient in the gradient's dtype), mul's backward multiplies by the
    other operand, index's backward adds the rows up in message order.  `fwd` = the forward result AS STORED (rounded to
    `store_dtype`), `store_dtype` = the dtype the values live in on the device: the message compared with `fwd` and the share
    grad / N are rounded to it; the sums are f32 (what the device kernels accumulate in)."""
    rnd = (lambda t: t) if store_dtype == torch.float32 else (lambda t: t.to(store_dtype).float())
    a, c, d = acd[0], acd[1], acd[2]
    # N_to_distribute = (self == result) + sum over messages of (src == result[index])  (torch's scatter_reduce_backward): `self` is
    # the zero-initialised output of utils.py:44-49, so an extremum that is exactly 0 counts one more "tie" -- the (masked-out)
    # initial value -- although include_self=False.  Part of the reference's numerics; reproduced.
    ties = (fwd == 0).float()
    for lo in range(0, a.numel(), chunk):
        sl = slice(lo, lo + chunk)
        ties.index_add_(0, a[sl], (rnd(valA[c[sl]] * valB[d[sl]]) == fwd[a[sl]]).float())
    share = rnd(gout / ties.clamp_min(1.0))
    gA, gB = torch.zeros_like(valA), torch.zeros_like(valB)
    for lo in range(0, a.numel(), chunk):
        sl = slice(lo, lo + chunk)
        av, bv = valA[c[sl]], valB[d[sl]]
        hit = (rnd(av * bv) == fwd[a[sl]]).float() * share[a[sl]]
        gA.index_add_(0, c[sl], hit * bv)
        gB.index_add_(0, d[sl], hit * av)
    return gA, gB


def spmm_values(valA, X: Tensor, src: Tensor, tar: Tensor, n_tar: int, aggr: str = "sum") -> Tensor:
    msg = X[src] if valA is None else valA * X[src]
    return scatter_reduce(msg, tar, n_tar, aggr)


def _mlp(h_in, h_out, tailact=True):
    layers = [nn.Linear(h_in, h_out)]
    if tailact:
        layers += [nn.BatchNorm1d(h_out), nn.SiLU(inplace=True)]
    return nn.Sequential(*layers)


def port_key(ref_key: str) -> str:
    """state_dict key of the reference's model (example/minimal.py:37-85 with honn/utils.py MLP naming) -> NGNNPort key."""
    import re
    k = ref_key.replace("data_encoder.", "")
    k = re.sub(r"^subggnns\.(\d+)\.lin\.lins\.0\.", r"convs.\1.0.", k)
    k = re.sub(r"^subggnns\.(\d+)\.lin\.lins\.1\.norm\.", r"convs.\1.1.", k)
    k = re.sub(r"^poolmlp\.lins\.0\.", "poolmlp.0.", k)
    k = re.sub(r"^poolmlp\.lins\.1\.norm\.", "poolmlp.1.", k)
    k = re.sub(r"^pred_lin\.lins\.(\d+)\.norm\.", r"pred.\1.", k)
    k = re.sub(r"^pred_lin\.lins\.(\d+)\.", r"pred.\1.", k)
    return k


class NGNNPort(nn.Module):
    """functional twin of pygho_amd.ngnn.SpModel on plain CPU tensors (same layer stack and sizes)."""

    def __init__(self, hiddim: int = 128, num_layer: int = 6):
        super().__init__()
        self.x_encoder = nn.Embedding(32, hiddim)
        self.ea_encoder = nn.Embedding(16, hiddim)
        self.tuplefeat_encoder = nn.Embedding(16, hiddim)
        self.lin_tupleinit0 = nn.Linear(hiddim, hiddim)
        self.lin_tupleinit1 = nn.Linear(hiddim, hiddim)
        self.convs = nn.ModuleList([_mlp(hiddim, hiddim) for _ in range(num_layer)])
        self.poolmlp = _mlp(hiddim, hiddim)
        self.pred = nn.Sequential(nn.Linear(hiddim, hiddim), nn.BatchNorm1d(hiddim), nn.SiLU(inplace=True), nn.Linear(hiddim, 1))

    def forward(self, x, edge_attr, tupleid, tuplefeat, acd, batch, num_graphs):
        n, nnz = x.shape[0], tupleid.shape[1]
        xh = self.x_encoder(x)
        Av = self.ea_encoder(edge_attr)
        Xv = self.tuplefeat_encoder(tuplefeat)
        Xv = self.lin_tupleinit0(xh)[tupleid[0]] * self.lin_tupleinit1(xh)[tupleid[1]] * Xv
        for mlp in self.convs:
            Xv = Xv + spspmm_values(mlp(Xv), Av, acd, nnz, "sum")
        xn = scatter_reduce(Xv, tupleid[0], n, "mean")
        hg = scatter_reduce(self.poolmlp(xn), batch, num_graphs, "sum")
        return self.pred(hg)

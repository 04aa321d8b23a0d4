"""
Dense (MaskedTensor) path: masked fill / reduce / broadcast, padded-batch builders, the pair kernels of the SUN / GNNAK layers and
the masked batched contraction (matrix-core kernels, neighbour lists, per-element extents) with their autograd Functions.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from ._native import AGGR_CODE, DTYPE_CODE, check, dtype_code, lib, ptr, require_device, stream_ptr

from .plans import *          # noqa: F401,F403
from .plans import _I32, _fetch, _flag
from .segment import *        # noqa: F401,F403
from .segment import _as2d, _ScatterReduce


def _blocks():
    """the fused-block layer (row-block Linear, split-K weight gradient) sits above this module and imports it: late binding"""
    from . import blocks
    return blocks


# --------------------------------------------------------------------------
# masked (dense) path
# --------------------------------------------------------------------------
def _mask_u8(mask: Tensor) -> Tensor:
    """bool mask as a uint8 view (no copy), cached on the mask tensor object."""
    c = getattr(mask, "_pygho_u8", None)
    if c is None:
        c = unbased(mask.contiguous().view(torch.uint8)) if mask.dtype == torch.bool else mask.contiguous().to(torch.uint8)
        if c is mask:
            return c                                         # already uint8 and contiguous: nothing to cache (not on itself)
        try:
            mask._pygho_u8 = c
        except Exception:
            pass
    return c


def _rows_d(data: Tensor, masked_dim: int) -> Tuple[int, int]:
    rows = 1
    for s in data.shape[:masked_dim]:
        rows *= s
    d = 1
    for s in data.shape[masked_dim:]:
        d *= s
    return rows, d


class _MaskedFill(torch.autograd.Function):
    """out = mask ? data : value; gradient flows through the unmasked entries only."""

    @staticmethod
    def forward(ctx, data: Tensor, mask: Tensor, value: float):
        dev = require_device(data, mask)
        data = data.contiguous()
        rows, d = _rows_d(data, mask.dim())
        out = torch.empty_like(data)
        m8 = _mask_u8(mask)
        check(lib().pygho_masked_fill(ptr(out), ptr(data), ptr(m8), float(value), rows, d, dtype_code(data),
                                      stream_ptr(dev)), "masked_fill")
        ctx.mask = mask
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        return _MaskedFill.apply(g, ctx.mask, 0.0), None, None


def masked_fill(data: Tensor, mask: Tensor, value: float) -> Tensor:
    return _MaskedFill.apply(data, mask, value)


class _MaskedReduce(torch.autograd.Function):
    """reduce ONE masked dim `dim` of data (masked dims first, dense dims last)."""

    @staticmethod
    def forward(ctx, data: Tensor, mask: Tensor, dim: int, aggr: str):
        dev = require_device(data, mask)
        data = data.contiguous()
        md = mask.dim()
        outer = 1
        for s in data.shape[:dim]:
            outer *= s
        r = data.shape[dim]
        inner = 1
        for s in data.shape[dim + 1:md]:
            inner *= s
        d = 1
        for s in data.shape[md:]:
            d *= s
        oshape = tuple(data.shape[:dim]) + tuple(data.shape[dim + 1:])
        mshape = tuple(mask.shape[:dim]) + tuple(mask.shape[dim + 1:])
        out = torch.empty(oshape, dtype=data.dtype, device=dev)
        omask = torch.empty(mshape, dtype=torch.uint8, device=dev)
        m8 = _mask_u8(mask)
        check(lib().pygho_masked_reduce(ptr(out), ptr(omask), ptr(data), ptr(m8), outer, r, inner, d, dtype_code(data),
                                        AGGR_CODE[aggr], stream_ptr(dev)), "masked_reduce")
        ctx.meta = (m8, outer, r, inner, d, aggr, tuple(data.shape))
        ctx.save_for_backward(*((data, out) if aggr in ("max", "min") else ()))
        ctx.mark_non_differentiable(omask)
        return out, omask

    @staticmethod
    def backward(ctx, g: Tensor, _gm):
        m8, outer, r, inner, d, aggr, shape = ctx.meta
        g = g.contiguous()
        data = fwd = None
        if aggr in ("max", "min"):
            data, fwd = ctx.saved_tensors
        gdata = torch.empty(shape, dtype=g.dtype, device=g.device)
        check(lib().pygho_masked_reduce_bwd(ptr(gdata), ptr(g), ptr(data), ptr(fwd), ptr(m8), outer, r, inner, d,
                                            dtype_code(g), AGGR_CODE[aggr], stream_ptr(g.device)), "masked_reduce_bwd")
        return gdata, None, None, None


def masked_reduce(data: Tensor, mask: Tensor, dim: int, aggr: str) -> Tuple[Tensor, Tensor]:
    """(reduced data, reduced mask as bool) over one masked dim."""
    out, om = _MaskedReduce.apply(data, mask, dim, aggr)
    return out, om.view(torch.bool)


class _MaskedBroadcast(torch.autograd.Function):
    """out[o, k, i] = mask[o, k, i] ? src[o, i] : value   (unpooling along one new masked dim)."""

    @staticmethod
    def forward(ctx, src: Tensor, mask: Tensor, dim: int, value: float, src_masked_dim: int):
        dev = require_device(src, mask)
        src = src.contiguous()
        outer = 1
        for s in src.shape[:dim]:
            outer *= s
        inner = 1
        for s in src.shape[dim:src_masked_dim]:
            inner *= s
        d = 1
        for s in src.shape[src_masked_dim:]:
            d *= s
        r = mask.shape[dim]
        oshape = tuple(src.shape[:dim]) + (r,) + tuple(src.shape[dim:])
        out = torch.empty(oshape, dtype=src.dtype, device=dev)
        m8 = _mask_u8(mask)
        check(lib().pygho_masked_broadcast(ptr(out), ptr(src), ptr(m8), float(value), outer, r, inner, d,
                                           dtype_code(src), stream_ptr(dev)), "masked_broadcast")
        ctx.meta = (mask, dim)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        mask, dim = ctx.meta
        gs, _ = _MaskedReduce.apply(g, mask, dim, "sum")
        return gs, None, None, None, None


def masked_broadcast(src: Tensor, mask: Tensor, dim: int, value: float, src_masked_dim: int) -> Tensor:
    return _MaskedBroadcast.apply(src, mask, dim, value, src_masked_dim)


def pad_stack(src: Tensor, start: Tensor, shape: Tensor, max_shape) -> Tuple[Tensor, Tensor]:
    """ragged per-graph grids -> padded (nb, *max_shape, *dense) + bool mask (`pygho_pad_stack`; hodata/MaData.py:108-214).
    graph b owns the rows [start[b], start[b] + prod(shape[b])) of src as a row-major grid of shape[b]."""
    dev = require_device(src, start, shape)
    src = src.contiguous()
    start = start.to(torch.int64).contiguous()
    shape = shape.to(torch.int64).contiguous()
    nb, nd = shape.shape
    assert len(max_shape) == nd and 1 <= nd <= 3, "1 to 3 grid dims"
    assert start.numel() == nb + 1
    m = [1] * (3 - nd) + [int(v) for v in max_shape]
    tail = tuple(src.shape[1:])
    row_bytes = src.element_size()
    for t in tail:
        row_bytes *= t
    out = torch.empty((nb,) + tuple(int(v) for v in max_shape) + tail, dtype=src.dtype, device=dev)
    mask = torch.empty((nb,) + tuple(int(v) for v in max_shape), dtype=torch.uint8, device=dev)
    check(lib().pygho_pad_stack(ptr(out), ptr(mask), ptr(src), ptr(start), ptr(shape), nb, nd, m[0], m[1], m[2], row_bytes,
                                src.shape[0], stream_ptr(dev)), "pad_stack")
    return out, mask.view(torch.bool)


def dense_adj(edge_index: Tensor, edge_batch: Tensor, edge_attr: Tensor, n: int, nb: int, filled_value=0) -> Tuple[Tensor, Tensor]:
    """(nb, n, n, *dense) filled with `filled_value`, edge_attr scattered at (edge_batch, edge_index[0], edge_index[1]), + mask
    (`pygho_dense_adj`; hodata/MaData.py:25-72)."""
    dev = require_device(edge_index, edge_batch, edge_attr)
    edge_attr = edge_attr.contiguous()
    eb = edge_batch.to(torch.int64).contiguous()
    er, ec = edge_index[0].to(torch.int64).contiguous(), edge_index[1].to(torch.int64).contiguous()
    tail = tuple(edge_attr.shape[1:])
    es = edge_attr.element_size()
    row_bytes = es
    for t in tail:
        row_bytes *= t
    out = torch.empty((nb, n, n) + tail, dtype=edge_attr.dtype, device=dev)
    mask = torch.empty((nb, n, n), dtype=torch.uint8, device=dev)
    view = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[es]
    bits = int(torch.tensor([filled_value], dtype=edge_attr.dtype).view(view).item()) & ((1 << (8 * es)) - 1)
    check(lib().pygho_dense_adj(ptr(out), ptr(mask), ptr(edge_attr), ptr(eb), ptr(er), ptr(ec), eb.numel(), nb, n, row_bytes,
                                bits, es, stream_ptr(dev)), "dense_adj")
    return out, mask.view(torch.bool)


def pair_combine_supported(data: Tensor) -> bool:
    return (data.is_cuda and data.dim() == 4 and data.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and (data.shape[-1] * data.element_size()) % 16 == 0 and data.shape[-1] * data.element_size() <= 4096)


def masked_pair_combine(base: Optional[Tensor], row_term: Optional[Tensor], col_term: Optional[Tensor],
                        diag_term: Optional[Tensor], replace_diag: bool, mask: Optional[Tensor], shape, dtype, device) -> Tensor:
    """out[b,i,j] = mask ? ((base[b,i,j] + row_term[b,i]) + col_term[b,j]) : 0; diag_term[b,i] is added on / replaces the
    diagonal (no autograd; `pygho_masked_pair_combine`)."""
    nb, n1, n2, d = shape
    ops = [None if t is None else t.contiguous() for t in (base, row_term, col_term, diag_term)]
    for t in ops:
        assert t is None or (t.dtype == dtype and t.device == device)
    out = torch.empty(shape, dtype=dtype, device=device)
    m8 = None if mask is None else _mask_u8(mask)
    check(lib().pygho_masked_pair_combine(ptr(out), ptr(ops[0]), ptr(ops[1]), ptr(ops[2]), ptr(ops[3]), 1 if replace_diag else 0,
                                          ptr(m8), nb, n1, n2, d, DTYPE_CODE[dtype], stream_ptr(device)), "masked_pair_combine")
    return out


def _diag_rows(data: Tensor) -> Tensor:
    """(b, n1, n2, d) -> (b, min(n1, n2), d): rows (b, i, i)."""
    return torch.diagonal(data, 0, 1, 2).movedim(-1, 1)


class _PairViews(torch.autograd.Function):
    """(data (b, n1, n2, d), mask) -> (diagonal rows, sum over dim 1, sum over dim 2): the three node-level views a
    subgraph layer takes of a 2-D representation; their gradients return to the tuple level in ONE pass."""

    @staticmethod
    def forward(ctx, data: Tensor, mask: Tensor, want_dim2: bool = True, chain: bool = False):
        require_device(data, mask)
        data = data.contiguous()
        dmask = torch.diagonal(mask, 0, 1, 2)
        dg = torch.where(dmask.unsqueeze(-1), _diag_rows(data), torch.zeros((), dtype=data.dtype, device=data.device))
        s1, _ = _MaskedReduce.apply(data, mask, 1, "sum")
        s2 = _MaskedReduce.apply(data, mask, 2, "sum")[0] if want_dim2 else data.new_empty((0,) + tuple(data.shape[2:]))
        ctx.mask = mask
        ctx.meta = (tuple(data.shape), data.dtype, want_dim2)
        ctx.set_materialize_grads(False)
        if chain:       # `data` again as an output: a later consumer's gradient arrives here and rides in the combine pass as its base
            return dg.contiguous(), s1, s2, data.view_as(data)
        return dg.contiguous(), s1, s2

    @staticmethod
    def backward(ctx, g_dg, g_s1, g_s2, g_chain=None):
        shape, dtype, want_dim2 = ctx.meta
        dev = ctx.mask.device
        cast = lambda t: None if t is None else t.to(dtype).contiguous()
        if not want_dim2:
            g_s2 = None
        if g_dg is None and g_s1 is None and g_s2 is None:
            return g_chain, None, None, None
        return masked_pair_combine(cast(g_chain), cast(g_s2), cast(g_s1), cast(g_dg), False, ctx.mask, shape, dtype, dev), None, None, None


def pair_views(data: Tensor, mask: Tensor, want_dim2: bool = True, chain: bool = False):
    """`chain`: also returns `data` as an autograd output (same storage) for the consumer that comes after the views, so that its
    gradient is summed with the views' gradients inside their one combine pass instead of by a separate accumulation."""
    return _PairViews.apply(data, mask, want_dim2, chain)


def _dense_linear(flat: Tensor, w_in_out: Tensor, addend: Optional[Tensor] = None) -> Tensor:
    """flat @ w (+ addend) with w stored (in, out): the streaming MFMA kernel when its shape is supported, else the library."""
    if _blocks().rowblock_linear_supported(flat, w_in_out.shape[1]) and w_in_out.shape[0] == w_in_out.shape[1]:
        return _blocks().rowblock_linear(flat, w_in_out.t().contiguous(), None, addend)[0]
    out = flat @ w_in_out
    return out if addend is None else out + addend


class _PairLinearMix(torch.autograd.Function):
    """out[b,i,j] = mask ? (i == j ? dg[b,i] : ((x[b,i,j] @ w_x + y[b,i,j] @ w_y) + u[b,i]) + v[b,j]) : 0

    The recombination step of SUNConv (reference Conv.py:338-362) after the linear map has been pulled through the
    broadcasts: two tuple-level GEMMs (the second with the first in its epilogue), then one pass adding the node-level
    terms and selecting the diagonal.  Backward: one masked copy (off-diagonal part of g), two masked reductions, two
    input-gradient GEMMs, two weight gradients."""

    @staticmethod
    def forward(ctx, x, y, w_x, w_y, u, v, dg, mask):
        require_device(x, y, mask)
        x, y = x.contiguous(), y.contiguous()
        shape = tuple(x.shape)
        d = shape[-1]
        a = _dense_linear(x.reshape(-1, d), w_x)
        ab = _dense_linear(y.reshape(-1, d), w_y, a)
        out = masked_pair_combine(ab.reshape(shape), u, v, dg, True, mask, shape[:3] + (w_x.shape[1],), x.dtype, x.device)
        ctx.save_for_backward(x, y, w_x, w_y)
        ctx.mask = mask
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, w_x, w_y = ctx.saved_tensors
        mask = ctx.mask
        g = g.contiguous()
        d_in, d_out = w_x.shape
        n1, n2 = mask.shape[1], mask.shape[2]
        off = getattr(mask, "_pygho_offdiag", None)
        if off is None:
            eye = torch.eye(n1, n2, dtype=torch.bool, device=mask.device)
            off = mask & ~eye
            try:
                mask._pygho_offdiag = off
            except Exception:
                pass
        goff = _MaskedFill.apply(g, off, 0.0)
        gu, _ = _MaskedReduce.apply(goff, off, 2, "sum")
        gv, _ = _MaskedReduce.apply(goff, off, 1, "sum")
        dmask = torch.diagonal(mask, 0, 1, 2)
        gdg = torch.where(dmask.unsqueeze(-1), _diag_rows(g), torch.zeros((), dtype=g.dtype, device=g.device))
        gf = goff.reshape(-1, d_out)
        xf, yf = x.reshape(-1, d_in), y.reshape(-1, d_in)
        gx = _dense_linear(gf, w_x.t()).reshape(x.shape) if ctx.needs_input_grad[0] else None
        gy = _dense_linear(gf, w_y.t()).reshape(y.shape) if ctx.needs_input_grad[1] else None
        gwx = _blocks().weight_grad_splitk(gf, xf, w_x.dtype).t() if ctx.needs_input_grad[2] else None
        gwy = _blocks().weight_grad_splitk(gf, yf, w_y.dtype).t() if ctx.needs_input_grad[3] else None
        return gx, gy, gwx, gwy, gu, gv, gdg, None


def pair_linear_mix(x, y, w_x, w_y, u, v, dg, mask):
    return _PairLinearMix.apply(x, y, w_x, w_y, u, v, dg, mask)


def pair_gather_combine(base: Optional[Tensor], row_term: Optional[Tensor], col_term: Optional[Tensor],
                        diag_term: Optional[Tensor], replace_diag: bool, ri32: Tensor, ci32: Tensor, d: int, dtype, device) -> Tensor:
    """sparse twin of masked_pair_combine: out[t] = (base[t] + row_term[ri[t]]) + col_term[ci[t]], diag_term[ri[t]] added on /
    replacing the tuples with ri == ci (no autograd; `pygho_pair_gather_combine`)."""
    ops = [None if t is None else t.contiguous() for t in (base, row_term, col_term, diag_term)]
    for t in ops:
        assert t is None or (t.dtype == dtype and t.device == device)
    n_rows = ri32.numel()
    out = torch.empty((n_rows, d), dtype=dtype, device=device)
    check(lib().pygho_pair_gather_combine(ptr(out), ptr(ops[0]), ptr(ops[1]), ptr(ops[2]), ptr(ops[3]), 1 if replace_diag else 0,
                                          ptr(ri32), ptr(ci32), n_rows, d, DTYPE_CODE[dtype], stream_ptr(device)),
          "pair_gather_combine")
    return out


def pair_gather_supported(values: Tensor, width: Optional[int] = None, dtype: Optional[torch.dtype] = None) -> bool:
    """rows of `width` elements of `dtype` (default: `values`' own) are whole 16-byte pieces of at most 4 KB.  Callers that run
    the pair kernels in the autocast dtype pass it: an f32 `values` under bf16 autocast has 2-byte rows."""
    dtype = values.dtype if dtype is None else dtype
    width = values.shape[1] if width is None and values.dim() == 2 else width
    return (values.is_cuda and values.dim() == 2 and dtype in (torch.float32, torch.bfloat16, torch.float16)
            and (width * dtype.itemsize) % 16 == 0 and width * dtype.itemsize <= 4096)


def _matched_rows(src: Tensor, pos: Tensor) -> Tensor:
    """out[r] = pos[r] >= 0 ? src[pos[r]] : 0 (no autograd)."""
    return row_gather(src, narrow_i32(pos.clamp_min(0)), (pos >= 0).to(_I32))


class _SparsePairViews(torch.autograd.Function):
    """(values of a sparse 2-D representation) -> (diagonal rows (n, d), sum over tuples sharing index 0, sum over tuples
    sharing index 1); the three gradients return to the tuples in ONE gather pass."""

    @staticmethod
    def forward(ctx, values: Tensor, ri: Tensor, ci: Tensor, diag_pos: Tensor, n: int, want_rows: bool = True, chain: bool = False):
        require_device(values, ri, ci, diag_pos)
        values = values.contiguous()
        dg = _matched_rows(values, diag_pos)
        # want_rows = False: the per-i sums are not needed by the caller (SUNConv takes only the diagonal and the per-j sums of
        # the aggregated representation): one pooling pass less, and an empty placeholder in its place
        s_r = (_ScatterReduce.apply(values, cached_plan(ri, n, "scatter"), narrow_i32(ri), "sum") if want_rows
               else values.new_empty((0, values.shape[1])))
        s_c = _ScatterReduce.apply(values, cached_plan(ci, n, "scatter"), narrow_i32(ci), "sum")
        ctx.idx = (narrow_i32(ri), narrow_i32(ci))
        ctx.meta = (values.shape[1], values.dtype, want_rows)
        ctx.set_materialize_grads(False)
        if chain:       # see _PairViews
            return dg, s_r, s_c, values.view_as(values)
        return dg, s_r, s_c

    @staticmethod
    def backward(ctx, g_dg, g_r, g_c, g_chain=None):
        d, dtype, want_rows = ctx.meta
        ri32, ci32 = ctx.idx
        cast = lambda t: None if t is None else t.to(dtype).contiguous()
        if not want_rows:
            g_r = None
        if g_dg is None and g_r is None and g_c is None:
            return g_chain, None, None, None, None, None, None
        return (pair_gather_combine(cast(g_chain), cast(g_r), cast(g_c), cast(g_dg), False, ri32, ci32, d, dtype, ri32.device),
                None, None, None, None, None, None)


def sparse_pair_views(values: Tensor, ri: Tensor, ci: Tensor, diag_pos: Tensor, n: int, want_rows: bool = True, chain: bool = False):
    return _SparsePairViews.apply(values, ri, ci, diag_pos, n, want_rows, chain)


class _SparsePairBroadcast(torch.autograd.Function):
    """out[t] = u[i] (+ v[j]) for the tuple t = (i, j): two node-level tensors broadcast onto a sparse 2-D pattern and added in one
    pass (`pygho_pair_gather_combine`); the gradients are the two segment sums of the output gradient."""

    @staticmethod
    def forward(ctx, u, v, ri, ci, n):
        require_device(u, v, ri, ci)
        ri32, ci32 = narrow_i32(ri), narrow_i32(ci)
        ctx.idx = (ri, ci, n, v is not None)
        return pair_gather_combine(None, u, v, None, False, ri32, ci32, u.shape[1], u.dtype, u.device)

    @staticmethod
    def backward(ctx, g):
        ri, ci, n, has_v = ctx.idx
        g = g.contiguous()
        gu = seg_reduce_rows(g, cached_plan(ri, n, "scatter"), "sum") if ctx.needs_input_grad[0] else None
        gv = seg_reduce_rows(g, cached_plan(ci, n, "scatter"), "sum") if has_v and ctx.needs_input_grad[1] else None
        return gu, gv, None, None, None


def sparse_pair_broadcast(u: Tensor, v: Optional[Tensor], ri: Tensor, ci: Tensor, n: int) -> Tensor:
    return _SparsePairBroadcast.apply(u.contiguous(), None if v is None else v.contiguous(), ri, ci, n)


class _SparsePairLinearMix(torch.autograd.Function):
    """out[t] = (i == j) ? dg[i] : ((x[t] @ w_x + y[t] @ w_y) + u[i]) + v[j] for the tuple t = (i, j): `_PairLinearMix` on the
    sparse layout (SUNConv mode "SS")."""

    @staticmethod
    def forward(ctx, x, y, w_x, w_y, u, v, dg, ri, ci, diag_pos, n):
        require_device(x, y, ri, ci)
        x, y = x.contiguous(), y.contiguous()
        ri32, ci32 = narrow_i32(ri), narrow_i32(ci)
        ab = _dense_linear(y, w_y, _dense_linear(x, w_x))
        out = pair_gather_combine(ab, u, v, dg, True, ri32, ci32, w_x.shape[1], x.dtype, x.device)
        ctx.save_for_backward(x, y, w_x, w_y)
        ctx.idx = (ri, ci, diag_pos, n)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, w_x, w_y = ctx.saved_tensors
        ri, ci, diag_pos, n = ctx.idx
        ri32, ci32 = narrow_i32(ri), narrow_i32(ci)
        g = g.contiguous()
        d_out = w_x.shape[1]
        zeros = torch.zeros((n, d_out), dtype=g.dtype, device=g.device)
        goff = pair_gather_combine(g, None, None, zeros, True, ri32, ci32, d_out, g.dtype, g.device)   # diagonal tuples zeroed
        gu = _ScatterReduce.apply(goff, cached_plan(ri, n, "scatter"), ri32, "sum")
        gv = _ScatterReduce.apply(goff, cached_plan(ci, n, "scatter"), ci32, "sum")
        gdg = _matched_rows(g, diag_pos)
        gx = _dense_linear(goff, w_x.t()) if ctx.needs_input_grad[0] else None
        gy = _dense_linear(goff, w_y.t()) if ctx.needs_input_grad[1] else None
        gwx = _blocks().weight_grad_splitk(goff, x, w_x.dtype).t() if ctx.needs_input_grad[2] else None
        gwy = _blocks().weight_grad_splitk(goff, y, w_y.dtype).t() if ctx.needs_input_grad[3] else None
        return gx, gy, gwx, gwy, gu, gv, gdg, None, None, None, None


def sparse_pair_linear_mix(x, y, w_x, w_y, u, v, dg, ri, ci, diag_pos, n):
    return _SparsePairLinearMix.apply(x, y, w_x, w_y, u, v, dg, ri, ci, diag_pos, n)


USE_BMM_BLOCKS = True         # rows of whole 256-B multiples, k <= 64: the multi-block matrix-core kernel serves every mask pattern
USE_BMM_LISTS = True          # (other shapes) masked contraction with a sparse-masked operand: neighbour-list kernel instead of the dense MFMA one
BMM_LIST_DENSITY = 0.15       # ... when at most this fraction of that operand's positions is unmasked


USE_BMM_EXTENTS = True        # matrix-core contraction: stage / multiply only up to the last unmasked row, k and column of each batch element


def _mask_extents(amask, bmask, omask, nb, ni, nk, nj, a_kfirst: bool, b_kfirst: bool) -> Optional[Tensor]:
    """(nb, 3) int32 (ei, ek, ej) per batch element (`pygho_mask_extents`), cached on the first mask of the triple.  An entry holds
    its other masks WEAKLY (strong references between two masks' caches -- forward keyed on one, backward on the other -- were a
    cycle that kept both masks' memory until the cyclic collector ran) and is valid only while those references are alive and are
    the very objects asked about (so a recycled `id` cannot hit); None when no mask is given."""
    import weakref
    holder = amask if amask is not None else (bmask if bmask is not None else omask)
    if holder is None:
        return None
    cache = getattr(holder, "_pygho_extents", None)
    if cache is None:
        cache = {}
        try:
            holder._pygho_extents = cache
        except Exception:
            pass
    masks = (amask, bmask, omask)
    ver = lambda m: None if m is None else (id(m), m._version)
    key = (ver(amask), ver(bmask), ver(omask), ni, nk, nj, a_kfirst, b_kfirst)
    hit = cache.get(key)
    if hit is not None and not all((r is None and (m is None or m is holder)) or (r is not None and r() is m) for r, m in zip(hit[1], masks)):
        hit = None
    if hit is None:
        dev = holder.device
        ext = torch.empty((nb, 3), dtype=torch.int32, device=dev)
        check(lib().pygho_mask_extents(ptr(ext), ptr(amask), ptr(bmask), ptr(omask), nb, ni, nk, nj, 1 if a_kfirst else 0,
                                       1 if b_kfirst else 0, stream_ptr(dev)), "mask_extents")
        if len(cache) > 8:
            cache.clear()
        hit = cache[key] = (ext, tuple(None if (m is None or m is holder) else weakref.ref(m) for m in masks))
    return hit[0]


def _mask_density(m8: Optional[Tensor]) -> float:
    """unmasked fraction of a uint8 mask, computed once per mask tensor object (one small reduction + one sync per batch)."""
    if m8 is None:
        return 1.0
    c = getattr(m8, "_pygho_density", None)
    if c is None or c[0] != m8._version:
        c = (m8._version, float(m8.sum(dtype=torch.int64).item()) / max(1, m8.numel()))
        try:
            m8._pygho_density = c
        except Exception:
            pass
    return c[1]


def _mask_lists(m8: Tensor, nb: int, nk: int, nc: int, k_first: bool):
    """(list (nb, nc, roundup4(nk)) int16, -1 terminated; count (nb, nc) int32) of the unmasked k per (b, c), cached on the mask."""
    cache = getattr(m8, "_pygho_lists", None)
    if cache is None:
        cache = {}
        try:
            m8._pygho_lists = cache
        except Exception:
            pass
    key = (m8._version, nk, nc, k_first)
    if key not in cache:
        dev = m8.device
        lst = torch.empty((nb, nc, (nk + 3) & ~3), dtype=torch.int16, device=dev)       # -1 terminated rows, 8-byte groups
        cnt = torch.empty((nb, nc), dtype=torch.int32, device=dev)
        check(lib().pygho_mask_lists(ptr(lst), ptr(cnt), ptr(m8), nb, nk, nc, 1 if k_first else 0, stream_ptr(dev)), "mask_lists")
        cache[key] = (lst, cnt)                                  # column and row lists of one mask coexist (forward / backward)
    return cache[key]


def _bmm_launch(A: Tensor, B: Tensor, amask, bmask, omask, nb, ni, nk, nj, d, a_kfirst: bool, b_kfirst: bool) -> Tensor:
    dev = require_device(A, B, amask, bmask, omask)
    out = torch.empty((nb, ni, nj, d), dtype=A.dtype, device=dev)
    # the multi-block matrix-core kernel (csrc/masked_bmm_blocks.h) serves every contraction whose rows are whole 256-B
    # multiples and whose contracted dim fits its 64-bit row bitmasks -- including a sparse operand or output mask, where it
    # beats the neighbour-list kernels below (forward 156 vs 185 us, forward + both gradients 0.47 vs 0.55 ms at
    # (1024, 37, 37, 128) bf16) and needs no density probe (a reduction + a host synchronisation per new mask)
    blocks_ok = USE_BMM_BLOCKS and (d * A.element_size()) % 256 == 0 and nk <= 64
    if (USE_BMM_LISTS and not blocks_ok and 0 < nk <= 32767 and nb * ni * nj > 0 and (d * A.element_size()) % 16 == 0
            and d * A.element_size() <= 4096 and A.dtype in (torch.float32, torch.bfloat16, torch.float16)):
        da, db = _mask_density(amask), _mask_density(bmask)
        if min(da, db) <= BMM_LIST_DENSITY:
            on_j = db <= da                                      # the sparser operand supplies the lists
            if on_j:
                lst, cnt = _mask_lists(bmask, nb, nk, nj, b_kfirst)
                dense_mask = amask
            else:
                lst, cnt = _mask_lists(amask, nb, nk, ni, a_kfirst)
                dense_mask = bmask
            check(lib().pygho_masked_bmm_lists(ptr(out), ptr(A), ptr(B), ptr(dense_mask), ptr(omask), ptr(lst), ptr(cnt),
                                               1 if on_j else 0, nb, ni, nk, nj, d, 1 if a_kfirst else 0, 1 if b_kfirst else 0,
                                               dtype_code(A), stream_ptr(dev)), "masked_bmm_lists")
            return out
        # output-sparse: two dense operands, few outputs wanted (the gradient of an adjacency's values)
        if (omask is not None and nb * ni * nk * d * A.element_size() < 2 ** 31 - 1 and nb * nk * nj * d * A.element_size() < 2 ** 31 - 1
                and ni <= 32767 and _mask_density(omask) <= BMM_LIST_DENSITY):
            lst, cnt = _mask_lists(omask, nb, ni, nj, True)
            maxc = getattr(cnt, "_pygho_max", None)
            if maxc is None:
                maxc = int(cnt.max().item()) if cnt.numel() else 0
                cnt._pygho_max = maxc
            out.zero_()
            check(lib().pygho_masked_bmm_outlists(ptr(out), ptr(A), ptr(B), ptr(amask), ptr(bmask), ptr(lst), maxc, nb, ni, nk, nj, d,
                                                  1 if a_kfirst else 0, 1 if b_kfirst else 0, dtype_code(A), stream_ptr(dev)),
                  "masked_bmm_outlists")
            return out
    ext = _mask_extents(amask, bmask, omask, nb, ni, nk, nj, a_kfirst, b_kfirst) if USE_BMM_EXTENTS else None
    if ext is not None:
        check(lib().pygho_masked_bmm_clipped(ptr(out), ptr(A), ptr(B), ptr(amask), ptr(bmask), ptr(omask), ptr(ext), nb, ni, nk, nj, d,
                                             1 if a_kfirst else 0, 1 if b_kfirst else 0, dtype_code(A), stream_ptr(dev)),
              "masked_bmm_clipped")
        return out
    check(lib().pygho_masked_bmm(ptr(out), ptr(A), ptr(B), ptr(amask), ptr(bmask), ptr(omask), nb, ni, nk, nj, d,
                                 1 if a_kfirst else 0, 1 if b_kfirst else 0, dtype_code(A), stream_ptr(dev)), "masked_bmm")
    return out


class _MaskedBmm(torch.autograd.Function):
    """out[b,i,j,:] = omask ? sum_k A[b,i,k,:] * B[b,k,j,:] : 0 on the matrix cores; A stored (b,i,k,d) or
    k-first (b,k,i,d), B stored (b,k,j,d) (k-first) or (b,j,k,d).  Masks are uint8 or None (= all valid)."""

    @staticmethod
    def forward(ctx, A, B, amask, bmask, omask, dims, a_kfirst, b_kfirst):
        nb, ni, nk, nj, d = dims
        out = _bmm_launch(A, B, amask, bmask, omask, nb, ni, nk, nj, d, a_kfirst, b_kfirst)
        ctx.save_for_backward(A, B)
        ctx.meta = (amask, bmask, omask, dims, a_kfirst, b_kfirst)
        return out

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        amask, bmask, omask, (nb, ni, nk, nj, d), akf, bkf = ctx.meta
        g = g.contiguous()
        gA = gB = None
        if ctx.needs_input_grad[0]:
            if not akf:   # gA[b,i,k] = sum_j g[b,i,j] * B[k,j]
                gA = _bmm_launch(g, B, omask, bmask, amask, nb, ni, nj, nk, d, False, not bkf)
            else:         # gA[b,k,i] = sum_j B[k,j] * g[b,i,j]
                gA = _bmm_launch(B, g, bmask, omask, amask, nb, nk, nj, ni, d, not bkf, False)
        if ctx.needs_input_grad[1]:
            if bkf:       # gB[b,k,j] = sum_i A[i,k] * g[b,i,j]
                gB = _bmm_launch(A, g, amask, omask, bmask, nb, nk, ni, nj, d, not akf, True)
            else:         # gB[b,j,k] = sum_i g[b,i,j] * A[i,k]
                gB = _bmm_launch(g, A, omask, amask, bmask, nb, nj, ni, nk, d, True, not akf)
        return gA, gB, None, None, None, None, None, None


def masked_bmm(A, B, amask, bmask, omask, nb, ni, nk, nj, d, a_kfirst, b_kfirst) -> Tensor:
    """channel-innermost batched contraction; pads d up to the kernel's channel granule when needed."""
    gran = 8 if A.dtype in (torch.bfloat16, torch.float16) else 4
    if A.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise TypeError(f"pygho_amd: masked_bmm supports bf16 / f16 / f32, got {A.dtype}")
    pad = (-d) % gran
    if pad:
        A = torch.nn.functional.pad(A, (0, pad))
        B = torch.nn.functional.pad(B, (0, pad))
    out = _MaskedBmm.apply(A.contiguous(), B.contiguous(), amask, bmask, omask, (nb, ni, nk, nj, d + pad), a_kfirst, b_kfirst)
    return out[..., :d] if pad else out

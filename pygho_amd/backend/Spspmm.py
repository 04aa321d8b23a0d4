"""
Sparse x sparse contraction and its index planner.  Mirror of ``pygho/backend/Spspmm.py``
(file:line cited per function).  Value computation is ONE fused HIP kernel per call
(gather * gather -> segment reduce, no (M, d) temporaries); the planner functions are integer
HIP kernels (searchsorted / radix sort / run ids), bit-exact with the reference after
canonicalisation of the column order inside one output segment.
"""
import warnings
from typing import Callable, Optional, Tuple

import torch
from torch import LongTensor, Tensor

from .. import _ops
from .SpTensor import SparseTensor, decodehash, indicehash
from .utils import torch_scatter_reduce


def ptr2batch(ptr: LongTensor, dim_size: int = None) -> LongTensor:
    """batch[ptr[i]:ptr[i+1]] = i.  Reference: Spspmm.py:9-31."""
    assert ptr.ndim == 1, "ptr should be 1-d"
    if __debug__:
        assert ptr[0] == 0 and bool(torch.all(torch.diff(ptr) >= 0)), "should put in a ptr tensor"
        assert ptr[-1] == dim_size, "dim_size should match ptr"
    return deg2batch(torch.diff(ptr), dim_size)


def deg2batch(deg: LongTensor, dim_size: int = None) -> LongTensor:
    """repeat i deg[i] times.  Reference: Spspmm.py:34-54."""
    assert deg.ndim == 1, "ptr should be 1-d"
    if __debug__:
        assert bool(torch.all(deg >= 0)), "should put in a degree tensor"
    zeros = torch.zeros_like(deg)
    c, _ = _ops.expand_pairs(zeros, deg)
    return c


def spspmm_ind(ind1: LongTensor, dim1: int, ind2: LongTensor, dim2: int,
               is_k2_sorted: bool = False) -> Tuple[LongTensor, LongTensor]:
    """
    All pairs (c, d) with ``ind1[dim1, c] == ind2[dim2, d]`` and the pattern of the product.
    Returns ``(tarind, bcd)``: ``tarind`` (sd1+sd2-2, nnz_out) sorted unique pattern of the
    concatenated remaining coordinates, ``bcd`` (3, M) with ``out.val[b] += v1[c] * v2[d]``.
    Reference: Spspmm.py:57-143.  ``bcd`` is returned in canonical order (sorted by b, then c, then
    d): the reference only guarantees sorted ``b`` (its final argsort is unstable).
    """
    assert 0 <= dim1 < ind1.shape[0], f"ind1's reduced dim {dim1} is out of range"
    assert 0 <= dim2 < ind2.shape[0], f"ind2's reduced dim {dim2} is out of range"
    sd1, sd2 = ind1.shape[0], ind2.shape[0]
    k1, k2 = ind1[dim1].contiguous(), ind2[dim2].contiguous()
    if dim2 != 0 and not is_k2_sorted:
        k2s, perm2 = _ops.sort_with_perm(k2)
    else:
        if __debug__:
            assert bool(torch.all(torch.diff(k2) >= 0)), "ind2[0] should be sorted"
        k2s, perm2 = k2, None
    lower, upper = _ops.search_bounds(k2s, k1)                       # Spspmm.py:114-116
    c, d = _ops.expand_pairs(lower, upper - lower)                   # Spspmm.py:119-129
    if perm2 is not None:
        d = _ops.widen_gather(perm2, d)                              # Spspmm.py:104
    combined = _ops.product_hash(ind1, dim1, ind2, dim2, c, d)       # Spspmm.py:132-135
    uniq, plan, inv = _ops.unique_plan(combined)                      # Spspmm.py:136-140
    tarind = decodehash(uniq, sd1 + sd2 - 2)
    # stable grouping by b keeps the (c, d) enumeration order inside a segment: c ascending, and for
    # equal c ascending position in the sorted k2 run -> canonical once d is ascending too
    bcd = _ops.plan_triples(inv, c, d, plan.perm)
    if perm2 is not None:
        bcd = _canonical(bcd)
    return tarind, bcd


def _canonical(t: LongTensor) -> LongTensor:
    """sort the columns of a (3, M) triple array by (row 0, row 1, row 2)."""
    if t.shape[1] == 0:
        return t
    for r in (2, 1, 0):                          # LSD passes of a stable sort
        _, p = _ops.sort_with_perm(t[r])
        t = _ops.gather_cols(t, p)
    return t


def spsphadamard_ind(tar_ind: LongTensor, ind: LongTensor) -> LongTensor:
    """b2a[i] = slot of ``ind[:, i]`` in the sorted coalesced pattern ``tar_ind`` or -1.
    Reference: Spspmm.py:146-183."""
    assert tar_ind.shape[0] == ind.shape[0]
    th = indicehash(tar_ind)
    if __debug__:
        assert bool(torch.all(torch.diff(th) > 0)), "tar_ind should be sorted and coalesce"
    return _ops.sorted_match(th, indicehash(ind))


def filterind(tar_ind: LongTensor, ind: LongTensor, bcd: LongTensor) -> LongTensor:
    """restrict a product plan to a target pattern: ``acd`` with columns whose output slot is not in
    ``tar_ind`` dropped; column order preserved (so ``acd[0]`` stays sorted).
    Reference: Spspmm.py:186-222."""
    b2a = spsphadamard_ind(tar_ind, ind)
    keep = _ops.nonneg_positions(b2a, via=bcd[0])
    acd = _ops.gather_cols(bcd, keep)
    acd[0] = _ops.gather_cols(b2a, acd[0])
    return acd


def spsphadamard(A: SparseTensor, B: SparseTensor, b2a: Optional[LongTensor] = None) -> SparseTensor:
    """element-wise product on B's entries that also exist in A.  Reference: Spspmm.py:225-267."""
    assert A.is_coalesced(), "A should be coalesced"
    assert B.is_coalesced(), "B should be coalesced"
    assert A.sparseshape == B.sparseshape, "A, B should be of the same sparse shape"
    if b2a is None:
        b2a = spsphadamard_ind(A.indices, B.indices)
    sel = _ops.nonneg_positions(b2a)
    if A.values is None:
        retval = _ops.gather_rows(B.values, sel)
    elif B.values is None:
        retval = _ops.gather_rows(A.values, _ops.gather_cols(b2a, sel))
    else:
        retval = _ops.gather_rows(A.values, _ops.gather_cols(b2a, sel)) * _ops.gather_rows(B.values, sel)
    return SparseTensor(_ops.gather_cols(B.indices, sel), retval, shape=A.sparseshape + retval.shape[1:], is_coalesced=True)


def _product_shape(A: SparseTensor, dim1: int, B: SparseTensor, dim2: int, retval: Tensor):
    return (A.sparseshape[:dim1] + A.sparseshape[dim1 + 1:] + B.sparseshape[:dim2] + B.sparseshape[dim2 + 1:] +
            tuple(retval.shape[1:]))


def spspmm(A: SparseTensor, dim1: int, B: SparseTensor, dim2: int, aggr: str = "sum",
           bcd: Optional[LongTensor] = None, tar_ind: Optional[LongTensor] = None,
           acd: Optional[LongTensor] = None) -> SparseTensor:
    """
    ``out[a] = (+)_{(a,c,d)} A.val[c] * B.val[d]`` with (+) in sum | mean | max | min; either operand
    may be value-less (pattern only).  Reference: Spspmm.py:270-331.  With ``acd`` and ``tar_ind`` given
    (the precomputed fast path, Spspmm.py:307-321) this is ONE HIP kernel launch.
    """
    assert A.is_coalesced(), "A should be coalesced"
    assert B.is_coalesced(), "B should be coalesced"
    if acd is not None:
        assert tar_ind is not None
        retval = _ops.message_reduce(A.values, B.values, acd, tar_ind.shape[1], A.nnz, B.nnz, aggr)
        return SparseTensor(tar_ind, retval, shape=_product_shape(A, dim1, B, dim2, retval), is_coalesced=True)
    warnings.warn("acd is not found")
    ind = None
    if bcd is None:
        ind, bcd = spspmm_ind(A.indices, dim1, B.indices, dim2)
    if tar_ind is not None:
        if ind is None:
            # the reference leaves `ind` unbound here (UnboundLocalError, Spspmm.py:324-327); the pattern
            # `bcd` refers to is recomputed instead
            ind, _ = spspmm_ind(A.indices, dim1, B.indices, dim2)
        acd = filterind(tar_ind, ind, bcd)
        return spspmm(A, dim1, B, dim2, aggr, acd=acd, tar_ind=tar_ind)
    warnings.warn("tar_ind is not found")
    if ind is None:
        ind, _ = spspmm_ind(A.indices, dim1, B.indices, dim2)
    return spspmm(A, dim1, B, dim2, aggr, acd=bcd, tar_ind=ind)


def spspmpnn(A: SparseTensor, dim1: int, B: SparseTensor, dim2: int, C: SparseTensor, acd: LongTensor,
             message_func: Callable[[Tensor, Tensor, Tensor, LongTensor], Tensor], aggr: str = "sum") -> SparseTensor:
    """
    Message passing with a user message function ``message_func(A.val[c], B.val[d], C.val[a], a)``
    reduced into C's pattern.  Reference: Spspmm.py:334-380.  The three operand gathers and the final
    segment reduce are HIP kernels; the message function itself is arbitrary Python on (M, d) tensors.
    """
    a, c, d = acd[0], acd[1], acd[2]
    mult = message_func(None if A.values is None else _ops.gather_rows(A.values, c),
                        None if B.values is None else _ops.gather_rows(B.values, d),
                        None if C.values is None else _ops.gather_rows(C.values, a), a)
    tar_ind = C.indices
    retval = torch_scatter_reduce(0, mult, a, tar_ind.shape[1], aggr)
    return SparseTensor(tar_ind, retval, shape=_product_shape(A, dim1, B, dim2, retval), is_coalesced=True)

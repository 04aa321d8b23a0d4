"""
CPU oracle for the PygHO sparse / masked operator path  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the algorithms in the reference's
``pygho/backend`` (file:line cited per function).  It is the checker that the
parity tests, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg compare the HIP path against.  Nothing under ``pygho_amd/`` may import it:
the product path runs on the HIP extension or fails loudly.

Parity pinning: every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by importing the reference itself in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and
against the known-answer vectors the reference's own tests hold
(``tests/test_backend_sparse.py:94-99`` ptr2batch, ``tests/test_backend_masked.py:45-50``
filterinf, docstring examples of ``Spspmm.py``).

Conventions
  * index arrays are int64, shape (sparse_dim, nnz); values are (nnz, *dense).
  * planner outputs (bcd / acd) are returned in CANONICAL order: columns sorted
    lexicographically by (b|a, c, d).  The reference's order inside one output
    segment is not canonical (its final argsort is unstable), so
    bit-exact comparisons are defined on this form.
  * MaskedTensor semantics follow the reference's DOCUMENTED behaviour
    (docs/BasicDataStructure.md:12 "unused elements do not affect the output");
    the reference constructor's missing fill (MaTensor.py:107-120) and the
    min-uses-amax slip (MaTensor.py:203) are deliberate deviations, recorded in
    DESIGN.md and pinned by a fixture.
"""
from __future__ import annotations

import numpy as np

I64 = np.int64


# --------------------------------------------------------------------------
# index hashing                                     reference: SpTensor.py:10-164
# --------------------------------------------------------------------------
def hash_bits(sparse_dim: int) -> int:
    """bits per coordinate of the order-preserving pack (SpTensor.py:36)."""
    return 63 // sparse_dim


def indicehash(ind: np.ndarray) -> np.ndarray:
    """(sd, nnz) -> (nnz,) most-significant-first bit pack; SpTensor.py:10-44."""
    ind = np.asarray(ind, dtype=I64)
    assert ind.ndim == 2
    assert (ind >= 0).all(), "indice cannot be negative"
    sd = ind.shape[0]
    if sd == 1:
        return ind[0]
    bits = hash_bits(sd)
    if ind.size:
        assert int(ind.max()) < (1 << bits), "too large indice, hash is not injective"
    out = np.zeros(ind.shape[1], dtype=I64)
    for row in range(sd):
        out |= ind[row] << I64(bits * (sd - 1 - row))
    return out


def decodehash(h: np.ndarray, sparse_dim: int) -> np.ndarray:
    """inverse of indicehash; SpTensor.py:47-87."""
    h = np.asarray(h, dtype=I64)
    if sparse_dim == 1:
        return h[None, :]
    bits = hash_bits(sparse_dim)
    low = I64((1 << bits) - 1)
    return np.stack([(h >> I64(bits * (sparse_dim - 1 - row))) & low
                     for row in range(sparse_dim)])


def _radix_steps(dimsize: np.ndarray) -> np.ndarray:
    dimsize = np.asarray(dimsize, dtype=I64)
    steps = np.ones_like(dimsize)
    for i in range(len(dimsize) - 2, -1, -1):
        steps[i] = steps[i + 1] * dimsize[i + 1]
    return steps


def indicehash_tight(ind: np.ndarray, dimsize: np.ndarray) -> np.ndarray:
    """mixed-radix flatten; SpTensor.py:90-127."""
    ind = np.asarray(ind, dtype=I64)
    dimsize = np.asarray(dimsize, dtype=I64)
    assert ind.ndim == 2 and dimsize.ndim == 1 and dimsize.shape[0] == ind.shape[0]
    assert (ind >= 0).all(), "indice cannot be negative"
    if ind.shape[1]:
        assert (ind.max(axis=1) < dimsize).all(), "indice exceeds dimsize"
    if ind.shape[0] == 1:
        return ind[0]
    return (_radix_steps(dimsize)[:, None] * ind).sum(axis=0)


def decodehash_tight(h: np.ndarray, dimsize: np.ndarray) -> np.ndarray:
    """mixed-radix unflatten; SpTensor.py:130-164."""
    h = np.asarray(h, dtype=I64)
    dimsize = np.asarray(dimsize, dtype=I64)
    if dimsize.shape[0] == 1:
        return h[None, :]
    steps = _radix_steps(dimsize)
    return np.stack([(h // steps[i]) % dimsize[i] for i in range(len(dimsize))])


# --------------------------------------------------------------------------
# scatter / segment reduce                          reference: utils.py:6-56
# --------------------------------------------------------------------------
def scatter_reduce(src: np.ndarray, ind: np.ndarray, dim_size: int, aggr: str) -> np.ndarray:
    """
    out[ind[m]] (+)= src[m] along dim 0, output starts from zeros and segments
    that receive nothing stay 0 for EVERY aggr (scatter_reduce_ with
    include_self=False, utils.py:50-55).  'mean' divides by the segment count
    (floor division for integer dtypes); 'max'/'min' are amax/amin.
    """
    src = np.asarray(src)
    ind = np.asarray(ind, dtype=I64)
    assert ind.ndim == 1 and src.shape[0] == ind.shape[0]
    out = np.zeros((dim_size,) + src.shape[1:], dtype=src.dtype)
    if src.shape[0] == 0:
        return out
    cnt = np.bincount(ind, minlength=dim_size)
    has = cnt > 0
    if aggr in ("sum", "mean"):
        np.add.at(out, ind, src)
        if aggr == "mean":
            c = np.maximum(cnt, 1).reshape((-1,) + (1,) * (src.ndim - 1))
            if np.issubdtype(src.dtype, np.integer):
                out = np.floor_divide(out, c.astype(src.dtype))
            else:
                out = (out / c.astype(src.dtype)).astype(src.dtype)
    elif aggr in ("max", "amax", "min", "amin"):
        is_max = aggr in ("max", "amax")
        if np.issubdtype(src.dtype, np.integer):
            info = np.iinfo(src.dtype)
            init = info.min if is_max else info.max
        else:
            init = -np.inf if is_max else np.inf
        tmp = np.full_like(out, init)
        (np.maximum if is_max else np.minimum).at(tmp, ind, src)
        out[has] = tmp[has]
    elif aggr == "prod":
        tmp = np.ones_like(out)
        np.multiply.at(tmp, ind, src)
        out[has] = tmp[has]
    else:
        raise ValueError(f"unknown aggr {aggr}")
    return out


# --------------------------------------------------------------------------
# coalesce                                          reference: SpTensor.py:167-197
# --------------------------------------------------------------------------
def coalesce(ind: np.ndarray, val, reduce: str = "sum"):
    """hash -> unique(return_inverse) -> decode; values scatter-reduced."""
    ind = np.asarray(ind, dtype=I64)
    sd = ind.shape[0]
    uniq, inv = np.unique(indicehash(ind), return_inverse=True)
    new_ind = decodehash(uniq, sd)
    if val is None:
        return new_ind, None
    return new_ind, scatter_reduce(np.asarray(val), inv.reshape(-1), uniq.shape[0], reduce)


# --------------------------------------------------------------------------
# ptr / degree expansion                            reference: Spspmm.py:9-54
# --------------------------------------------------------------------------
def ptr2batch(ptr: np.ndarray, dim_size: int | None = None) -> np.ndarray:
    ptr = np.asarray(ptr, dtype=I64)
    assert ptr.ndim == 1 and ptr[0] == 0 and (np.diff(ptr) >= 0).all()
    if dim_size is not None:
        assert ptr[-1] == dim_size
    return np.repeat(np.arange(len(ptr) - 1, dtype=I64), np.diff(ptr))


def deg2batch(deg: np.ndarray, dim_size: int | None = None) -> np.ndarray:
    deg = np.asarray(deg, dtype=I64)
    assert deg.ndim == 1 and (deg >= 0).all()
    return np.repeat(np.arange(len(deg), dtype=I64), deg)


# --------------------------------------------------------------------------
# planner                                           reference: Spspmm.py:57-222
# --------------------------------------------------------------------------
def canonical_triples(t: np.ndarray) -> np.ndarray:
    """sort the columns of a (3, M) triple array lexicographically by rows 0,1,2."""
    t = np.asarray(t, dtype=I64)
    order = np.lexsort((t[2], t[1], t[0]))
    return t[:, order]


def spspmm_ind(ind1: np.ndarray, dim1: int, ind2: np.ndarray, dim2: int):
    """
    Enumerate every pair (c, d) with ind1[dim1, c] == ind2[dim2, d]
    (Spspmm.py:57-143).  Returns (tarind, bcd): tarind (sd1+sd2-2, nnz_out) is
    the sorted unique pattern of the concatenated remaining coordinates and
    bcd[0] the pattern slot of each pair.  bcd is canonical (see module doc).
    """
    ind1 = np.asarray(ind1, dtype=I64)
    ind2 = np.asarray(ind2, dtype=I64)
    sd1, sd2 = ind1.shape[0], ind2.shape[0]
    assert 0 <= dim1 < sd1 and 0 <= dim2 < sd2
    k1, k2 = ind1[dim1], ind2[dim2]
    perm = np.argsort(k2, kind="stable")               # Spspmm.py:101-105
    k2s = k2[perm]
    lo = np.searchsorted(k2s, k1, side="left")          # Spspmm.py:114-116
    hi = np.searchsorted(k2s, k1, side="right")
    cnt = hi - lo
    c = np.repeat(np.arange(ind1.shape[1], dtype=I64), cnt)
    start = np.concatenate(([0], np.cumsum(cnt)))[:-1]
    rank = np.arange(c.shape[0], dtype=I64) - start[c]  # Spspmm.py:127-129
    d = perm[lo[c] + rank]
    rest1 = np.delete(ind1, dim1, axis=0)[:, c]
    rest2 = np.delete(ind2, dim2, axis=0)[:, d]
    combined = indicehash(np.concatenate((rest1, rest2), axis=0))   # :132-135
    uniq, b = np.unique(combined, return_inverse=True)              # :136-140
    tarind = decodehash(uniq, sd1 + sd2 - 2)
    bcd = canonical_triples(np.stack((b.reshape(-1).astype(I64), c, d)))
    return tarind, bcd


def spsphadamard_ind(tar_ind: np.ndarray, ind: np.ndarray) -> np.ndarray:
    """b2a[i] = slot of ind[:, i] in the sorted pattern tar_ind, -1 if absent
    (Spspmm.py:146-183)."""
    tar_ind = np.asarray(tar_ind, dtype=I64)
    ind = np.asarray(ind, dtype=I64)
    assert tar_ind.shape[0] == ind.shape[0]
    th = indicehash(tar_ind)
    assert (np.diff(th) > 0).all(), "tar_ind should be sorted and coalesce"
    h = indicehash(ind)
    if th.shape[0] == 0:
        return np.full(h.shape, -1, dtype=I64)
    pos = np.maximum(np.searchsorted(th, h, side="right") - 1, 0)
    return np.where(th[pos] == h, pos, -1).astype(I64)


def filterind(tar_ind: np.ndarray, ind: np.ndarray, bcd: np.ndarray) -> np.ndarray:
    """restrict the product pattern to a target pattern (Spspmm.py:186-222);
    returned canonical, hence acd[0] sorted."""
    bcd = np.asarray(bcd, dtype=I64)
    b2a = spsphadamard_ind(tar_ind, ind)
    a = b2a[bcd[0]]
    keep = a >= 0
    return canonical_triples(np.stack((a[keep], bcd[1][keep], bcd[2][keep])))


# --------------------------------------------------------------------------
# value ops on sparse operands
# --------------------------------------------------------------------------
def spsphadamard(indA, valA, indB, valB, b2a=None):
    """(A (.) B) on B's matched entries; Spspmm.py:225-267."""
    indB = np.asarray(indB, dtype=I64)
    if b2a is None:
        b2a = spsphadamard_ind(indA, indB)
    m = b2a >= 0
    if valA is None:
        val = valB[m]
    elif valB is None:
        val = valA[b2a[m]]
    else:
        val = valA[b2a[m]] * valB[m]
    return indB[:, m], val


def spspmm_values(valA, valB, acd: np.ndarray, n_out: int, aggr: str = "sum"):
    """out[a] = (+)_{(a,c,d)} A.val[c] * B.val[d]; either operand may be
    value-less (pattern only).  Spspmm.py:307-315."""
    acd = np.asarray(acd, dtype=I64)
    if valA is None:
        msg = valB[acd[2]]
    elif valB is None:
        msg = valA[acd[1]]
    else:
        msg = valA[acd[1]] * valB[acd[2]]
    return scatter_reduce(msg, acd[0], n_out, aggr)


def spspmm_values_grad(valA, valB, acd, n_out, aggr, grad_out):
    """analytic gradients of spspmm_values wrt valA / valB for sum | mean | max | min.
    max/min follow torch's scatter_reduce(amax/amin) rule: the gradient of a
    segment is shared equally between the entries that attain the extremum."""
    acd = np.asarray(acd, dtype=I64)
    a, c, d = acd
    gA = None if valA is None else np.zeros_like(valA)
    gB = None if valB is None else np.zeros_like(valB)
    gm = grad_out[a]
    if aggr == "mean":
        cnt = np.maximum(np.bincount(a, minlength=n_out), 1).astype(grad_out.dtype)
        gm = gm / cnt[a].reshape((-1,) + (1,) * (grad_out.ndim - 1))
    elif aggr in ("max", "min"):
        msg = (valB[d] if valA is None else valA[c] if valB is None else valA[c] * valB[d])
        out = scatter_reduce(msg, a, n_out, aggr)
        hit = (msg == out[a]).astype(grad_out.dtype)
        # torch's scatter_reduce_backward: N_to_distribute = (self == result) + scatter_add(src == result[index]); `self` is the
        # zero-initialised output (utils.py:44-49), so an extremum of exactly 0 has one extra tie although include_self=False
        ties = (out == 0).astype(out.dtype)
        np.add.at(ties, a, hit)
        gm = gm * hit / np.maximum(ties[a], 1)
    if gA is not None:
        np.add.at(gA, c, gm if valB is None else gm * valB[d])
    if gB is not None:
        np.add.at(gB, d, gm if valA is None else gm * valA[c])
    return gA, gB


def spspmpnn_values(valA, valB, valC, acd, n_out, message_func, aggr="sum"):
    """generalised message function variant; Spspmm.py:370-374."""
    acd = np.asarray(acd, dtype=I64)
    msg = message_func(None if valA is None else valA[acd[1]],
                       None if valB is None else valB[acd[2]],
                       None if valC is None else valC[acd[0]], acd[0])
    return scatter_reduce(msg, acd[0], n_out, aggr)


def spmm(indA, valA, shapeA, dim1: int, X, aggr: str = "sum"):
    """2-D sparse x dense; Spmm.py:31-44."""
    indA = np.asarray(indA, dtype=I64)
    assert indA.shape[0] == 2
    src, tar, n_tar = (indA[0], indA[1], shapeA[1]) if dim1 == 0 else (indA[1], indA[0], shapeA[0])
    msg = X[src] if valA is None else valA * X[src]
    return scatter_reduce(msg, tar, n_tar, aggr)


# --------------------------------------------------------------------------
# SparseTensor methods                               reference: SpTensor.py:304-524
# --------------------------------------------------------------------------
def sp_reduce_to_dense(ind, val, shape, dims, reduce: str):
    """pool away sparse dims `dims`, dense result; SpTensor.py:382-409."""
    ind = np.asarray(ind, dtype=I64)
    sd = ind.shape[0]
    keep = [i for i in range(sd) if i not in list(dims)]
    if len(keep) == 1:
        return scatter_reduce(val, ind[keep[0]], shape[keep[0]], reduce)
    kshape = tuple(shape[i] for i in keep)
    flat = indicehash_tight(ind[keep], np.array(kshape, dtype=I64))
    out = scatter_reduce(val, flat, int(np.prod(kshape)), reduce)
    return out.reshape(kshape + out.shape[1:])


def sp_reduce_to_sparse(ind, val, shape, dims, reduce: str):
    """pool away sparse dims, sparse (re-coalesced) result; SpTensor.py:368-380."""
    ind = np.asarray(ind, dtype=I64)
    keep = [i for i in range(ind.shape[0]) if i not in list(dims)]
    nind, nval = coalesce(ind[keep], val, reduce)
    return nind, nval, tuple(shape[i] for i in keep) + tuple(val.shape[1:])


def sp_diag_to_dense(ind, val, shape, dims):
    """values at (i, i, ...) over all sparse dims, zero where absent;
    SpTensor.py:326-335 (the full-diagonal branch, the one the operators use)."""
    ind = np.asarray(ind, dtype=I64)
    dims = sorted(set(dims))
    if len(dims) < ind.shape[0]:
        return sp_diag_partial_to_dense(ind, val, shape, dims)
    n = shape[dims[0]]
    dh = indicehash(np.tile(np.arange(n, dtype=I64), (len(dims), 1)))
    sh = indicehash(ind[dims])
    out = np.zeros((n,) + val.shape[1:], dtype=val.dtype)
    if sh.shape[0] == 0:
        return out
    # NOTE: the reference tests `matchidx < 0` only (SpTensor.py:330-334) and so
    # returns the neighbouring entry's value when (i,i) is absent but a smaller
    # hash exists.  The documented contract ("diagonal elements") is an exact
    # match; the oracle implements the exact match and the fixture generator
    # only uses patterns that contain their full diagonal (as every shipped
    # sampler produces), where both agree.
    pos = np.maximum(np.searchsorted(sh, dh, side="right") - 1, 0)
    ok = sh[pos] == dh
    out[ok] = val[pos[ok]]
    return out


def sp_diag_partial_to_dense(ind, val, shape, dims):
    """diagonal over SOME sparse dims, dense result: out[.., i at dims[0]'s slot, ..] = value at the entry whose coordinates in
    `dims` all equal i, zero where absent; the dims in dims[1:] disappear.  This is the documented intent of SpTensor.py:337-352 --
    that branch itself cannot run: `nsparse_shape + self.denseshape` adds a list and a tuple (TypeError, :346) [probe], and its
    lookup keeps ONE entry per i.  PARITY UNPINNED for this function (no reference output exists); pinned by construction only."""
    ind = np.asarray(ind, dtype=I64)
    keep = [i for i in range(ind.shape[0]) if i not in dims[1:]]
    out = np.zeros(tuple(shape[i] for i in keep) + val.shape[1:], dtype=val.dtype)
    on = np.all(ind[dims] == ind[dims[0]], axis=0)
    out[tuple(ind[k][on] for k in keep)] = val[on]
    return out


def sp_unpooling(self_ind, self_val, tar_ind, dims):
    """scatter a lower-order sparse tensor into the pattern of tarX; SpTensor.py:447-468."""
    self_ind = np.asarray(self_ind, dtype=I64)
    tar_ind = np.asarray(tar_ind, dtype=I64)
    if isinstance(dims, int):
        dims = [dims]
    keep = [i for i in range(tar_ind.shape[0]) if i not in list(dims)]
    sh = indicehash(self_ind)
    th = indicehash(tar_ind[keep])
    out = np.zeros((th.shape[0],) + self_val.shape[1:], dtype=self_val.dtype)
    if sh.shape[0] == 0:
        return out
    pos = np.maximum(np.searchsorted(sh, th, side="right") - 1, 0)
    ok = sh[pos] == th
    out[ok] = self_val[pos[ok]]
    return out


def sp_unpooling_fromdense1dim(ind, dim: int, X):
    """X[self.indices[dim]]; SpTensor.py:470-476."""
    return X[np.asarray(ind, dtype=I64)[dim]]


def sp_diagonal_flag(ind):
    """(indices[0] == indices[1]) as int64; SpTensor.py:498-501."""
    ind = np.asarray(ind, dtype=I64)
    return (ind[0] == ind[1]).astype(I64)


# --------------------------------------------------------------------------
# MaskedTensor                                       reference: MaTensor.py
# --------------------------------------------------------------------------
def filterinf(X, filled_value=0):
    """MaTensor.py:8-31."""
    X = np.array(X, copy=True)
    X[np.isinf(X)] = filled_value
    return X


def _full_mask(mask, data):
    mask = np.asarray(mask, dtype=bool)
    return mask.reshape(mask.shape + (1,) * (data.ndim - mask.ndim))


def ma_fill(data, mask, val):
    """documented fill_masked: masked-out entries become `val`; MaTensor.py:122-128."""
    return np.where(_full_mask(mask, data), data, np.asarray(val, dtype=data.dtype))


def ma_reduce(data, mask, dims, op: str, keepdim: bool = False):
    """sum / mean / max / min over masked dims; MaTensor.py:175-206 (documented
    semantics: masked-out entries never contribute, min is a true minimum,
    all-masked slices give 0).  Returns (data, mask)."""
    if isinstance(dims, int):
        dims = [dims]
    dims = tuple(dims)
    fm = _full_mask(mask, data)
    omask = np.asarray(mask, dtype=bool).any(axis=dims, keepdims=keepdim)
    if op == "sum":
        out = np.where(fm, data, 0).sum(axis=dims, keepdims=keepdim, dtype=data.dtype)
    elif op == "mean":
        s = np.where(fm, data, 0).sum(axis=dims, keepdims=keepdim, dtype=data.dtype)
        cnt = np.maximum(np.broadcast_to(fm, data.shape).sum(axis=dims, keepdims=keepdim), 1)
        out = (s / cnt).astype(data.dtype)
    elif op == "max":
        out = filterinf(np.where(fm, data, -np.inf).max(axis=dims, keepdims=keepdim)).astype(data.dtype)
    elif op == "min":
        out = filterinf(np.where(fm, data, np.inf).min(axis=dims, keepdims=keepdim)).astype(data.dtype)
    else:
        raise ValueError(op)
    return out, omask


def ma_diag(data, mask, dims):
    """diagonal over masked dims, result placed at dims[0]; MaTensor.py:208-223.  Two dims: what the reference computes (pinned by
    masked_ops.npz).  MORE than two dims: the reference's loop (:218-220) keeps using the ORIGINAL dim numbers after the first
    diagonal has removed two dims and appended one, and raises for every input tried (`diagonal dimensions cannot be identical` /
    `Dimension out of range`); restated here is the documented intent -- out[.., i at dims[0], ..] = x[.., i, .., i, .., i, ..] --
    by direct indexing (no reference output exists: pinned by construction)."""
    dims = sorted(dims)
    assert len(dims) >= 2
    if len(dims) == 2:
        td = np.diagonal(data, 0, dims[0], dims[1])
        tm = np.diagonal(mask, 0, dims[0], dims[1])
        return np.moveaxis(td, -1, dims[0]), np.moveaxis(tm, -1, dims[0])
    n = data.shape[dims[0]]
    assert all(data.shape[d] == n for d in dims)

    def take(a, nd):
        keep = [d for d in range(nd) if d not in dims[1:]]                 # dims[0] keeps its place and carries the diagonal
        out = np.empty([a.shape[d] for d in keep] + list(a.shape[nd:]), dtype=a.dtype)
        for i in range(n):
            src = tuple(i if d in dims else slice(None) for d in range(nd))
            dst = tuple(i if d == dims[0] else slice(None) for d in keep)
            out[dst] = a[src]
        return out
    return take(data, mask.ndim), take(mask, mask.ndim)


def ma_unpooling(data, dims, tar_shape):
    """insert and expand dims; MaTensor.py:225-234."""
    if isinstance(dims, int):
        dims = [dims]
    out = data
    for dm in sorted(dims):
        out = np.expand_dims(out, dm)
    shape = tuple(tar_shape[i] if i in dims else out.shape[i] for i in range(out.ndim))
    return np.broadcast_to(out, shape)


def mamamm(dataA, maskA, dim1: int, dataB, maskB, dim2: int):
    """
    Batched masked contraction; Mamamm.py:35-64.  dim 0 of both operands is the
    shared batch dim, masked dim `dim1` of A is contracted with masked dim
    `dim2` of B, remaining masked dims of A then of B follow the batch dim,
    trailing dense dims are elementwise (broadcast).  Masked-out entries are
    treated as 0 (documented fill).
    """
    tA, tB = ma_fill(dataA, maskA, 0), ma_fill(dataB, maskB, 0)
    mdA, mdB = np.asarray(maskA).ndim, np.asarray(maskB).ndim
    assert dim1 > 0 and dim2 > 0
    letters = "abcdefghijklmnopqrstuvw"
    it = iter(letters)
    batch, contr = "z", "y"
    subA = [batch] + [next(it) for _ in range(mdA - 1)]
    subB = [batch] + [next(it) for _ in range(mdB - 1)]
    subA[dim1] = contr
    subB[dim2] = contr
    outs = [batch] + [s for s in subA[1:] if s != contr] + [s for s in subB[1:] if s != contr]
    return np.einsum(f"{''.join(subA)}...,{''.join(subB)}...->{''.join(outs)}...", tA, tB)


def spmamm(indA, valA, shapeA, dim1: int, dataB, maskB, dim2: int, aggr: str = "sum"):
    """
    sparse (b, n, m[, d]) adjacency x masked (b, m, ..[, d]); Spmamm.py:42-68, to the
    DOCUMENTED semantics (the reference raises for any dense dim):
    out[b, i, ...] = (+)_k A[b,i,k] * B[b,k,...] over valid (unmasked) B entries,
    empty / all-invalid reductions give 0.
    """
    indA = np.asarray(indA, dtype=I64)
    assert indA.shape[0] == 3 and aggr != "mean"
    if dim1 == 1:
        n, bi, tar = shapeA[2], (indA[0], indA[1]), indA[2]
    elif dim1 == 2:
        n, bi, tar = shapeA[1], (indA[0], indA[2]), indA[1]
    else:
        raise NotImplementedError
    b = shapeA[0]
    tB = np.moveaxis(dataB, dim2, 1)
    tM = np.moveaxis(np.asarray(maskB, dtype=bool), dim2, 1)
    rows = tB[bi[0], bi[1]]                      # (nnz, *rest, *dense)
    valid = tM[bi[0], bi[1]]                     # (nnz, *rest)
    valid = valid.reshape(valid.shape + (1,) * (rows.ndim - valid.ndim))
    if valA is not None:
        av = valA.reshape((valA.shape[0],) + (1,) * (rows.ndim - valA.ndim) + valA.shape[1:])
        rows = av * rows
    fill = {"sum": 0.0, "max": -np.inf, "min": np.inf}[aggr]
    rows = np.where(valid, rows, fill).astype(dataB.dtype)
    out = scatter_reduce(rows, n * indA[0] + tar, b * n, aggr)
    out = out.reshape((b, n) + out.shape[1:])
    out = np.moveaxis(out, 1, dim2)
    if aggr in ("max", "min"):
        out = filterinf(out)
    return out


# --------------------------------------------------------------------------
# padded-batch builders of the dense path (reference pygho/hodata/MaData.py)
# --------------------------------------------------------------------------
def to_dense_x(nodeX: np.ndarray, ptr: np.ndarray, max_num_nodes: int | None = None):
    """MaData.py:108-147: ret[b, k] = nodeX[min(ptr[b] + k, ptr[-1] - 1)], mask[b, k] = k < ptr[b+1] - ptr[b].
    Returns (raw, mask): the raw array carries clamped neighbours in the padded slots (the reference constructs the
    MaskedTensor unfilled), consumers must look at it through the mask."""
    ptr = np.asarray(ptr, dtype=np.int64)
    counts = np.diff(ptr)
    if max_num_nodes is None:
        max_num_nodes = int(counts.max())
    k = np.arange(max_num_nodes, dtype=np.int64)[None, :]
    idx = np.minimum(ptr[:-1, None] + k, ptr[-1] - 1)
    return nodeX[idx], k < counts[:, None]


def to_dense_tuplefeat(tuplefeat: np.ndarray, tupleshape: np.ndarray, tptr: np.ndarray, max_tupleshape=None):
    """MaData.py:150-214: graph b holds a row-major (tupleshape[b]) grid starting at tptr[b];
    ret[b, i_1..i_k] = tuplefeat[min(tptr[b] + sum_j i_j * prod_{l>j} tupleshape[b, l], N - 1)],
    mask[b, i_1..i_k] = all_j (i_j < tupleshape[b, j])."""
    tupleshape = np.asarray(tupleshape, dtype=np.int64)
    nb, nd = tupleshape.shape
    if max_tupleshape is None:
        max_tupleshape = tupleshape.max(0)
    max_tupleshape = [int(v) for v in max_tupleshape]
    full = np.asarray(tptr, dtype=np.int64)[:-1].reshape([nb] + [1] * nd)
    mask = np.ones([nb] + max_tupleshape, dtype=bool)
    stride = np.ones(nb, dtype=np.int64)
    for j in range(nd - 1, -1, -1):
        ar = np.arange(max_tupleshape[j], dtype=np.int64)
        shp = [1] * (nd + 1)
        shp[j + 1] = -1
        bs = [nb] + [1] * nd
        full = full + ar.reshape(shp) * stride.reshape(bs)
        mask &= ar.reshape(shp) < tupleshape[:, j].reshape(bs)
        stride = stride * tupleshape[:, j]
    full = np.minimum(full, tuplefeat.shape[0] - 1)
    return tuplefeat[full], mask


def to_dense_adj(edge_index: np.ndarray, edge_batch: np.ndarray, edge_attr, max_num_nodes: int, batch_size: int,
                 filled_value=0):
    """MaData.py:25-72: ret = full(filled_value); ret[edge_batch, edge_index[0], edge_index[1]] = edge_attr (ones when
    absent); mask marks the written slots."""
    if edge_attr is None:
        edge_attr = np.ones(edge_batch.shape[0], dtype=np.float32)
    ret = np.full([batch_size, max_num_nodes, max_num_nodes] + list(edge_attr.shape[1:]), filled_value, dtype=edge_attr.dtype)
    mask = np.zeros((batch_size, max_num_nodes, max_num_nodes), dtype=bool)
    ret[edge_batch, edge_index[0], edge_index[1]] = edge_attr
    mask[edge_batch, edge_index[0], edge_index[1]] = True
    return ret, mask


def to_sparse_adj(edge_index: np.ndarray, edge_batch: np.ndarray, edge_attr: np.ndarray, max_num_nodes: int, batch_size: int):
    """MaData.py:75-105: indices (3, nnz) = [edge_batch; edge_index], values = edge_attr, shape (b, n, n, *dense)."""
    ind = np.concatenate((edge_batch[None, :], edge_index), axis=0)
    return ind, edge_attr, [batch_size, max_num_nodes, max_num_nodes] + list(edge_attr.shape[1:])


# --------------------------------------------------------------------------
# tuple samplers (pygho/hodata/SpTupleSampler.py)
# --------------------------------------------------------------------------
def k_hop_subgraph(roots, num_hops: int, edge_index: np.ndarray, num_nodes: int):
    """SpTupleSampler.py:12-88 (flow = 'source_to_target'): level h = the SOURCES of the edges whose target is in level
    h - 1 (levels keep duplicates and already-seen nodes, exactly like the reference); dist[v] = the smallest level v
    appears in (:68-69 assigns from the last level down to level 0); subset = sorted unique nodes."""
    col, row = edge_index[0], edge_index[1]
    levels = [np.atleast_1d(np.asarray(roots, dtype=np.int64))]
    for _ in range(num_hops):
        node_mask = np.zeros(num_nodes, dtype=bool)
        node_mask[levels[-1]] = True
        levels.append(col[node_mask[row]])
    dist = np.full(num_nodes, num_nodes + 1, dtype=np.int64)
    for h in range(num_hops, -1, -1):
        dist[levels[h]] = h
    subset = np.unique(np.concatenate(levels))
    return subset, dist[subset]


def khop_sampler(edge_index: np.ndarray, num_nodes: int, hop: int = 2):
    """KhopSampler (SpTupleSampler.py:91-126): tuples (i, j) for every j within `hop` of root i, feature = hop distance;
    coalesced (sorted by (i, j))."""
    ids, feats = [], []
    for i in range(num_nodes):
        subset, dist = k_hop_subgraph(i, hop, edge_index, num_nodes)
        ids.append(np.stack((np.full_like(subset, i), subset)))
        feats.append(dist)
    ind, val = np.concatenate(ids, axis=1), np.concatenate(feats)
    return coalesce(ind, val, "min")


def shortest_path_matrix(edge_index: np.ndarray, num_nodes: int) -> np.ndarray:
    """scipy.sparse.csgraph.shortest_path(directed=False, unweighted=True) (SpTupleSampler.py:145-150) as a BFS over the
    symmetrised edge list; unreachable = num_nodes + 1 here (the reference's cast of inf to int64 is platform noise; such
    pairs never enter a sample)."""
    adj = np.zeros((num_nodes, num_nodes), dtype=bool)
    adj[edge_index[0], edge_index[1]] = True
    adj |= adj.T
    dist = np.full((num_nodes, num_nodes), num_nodes + 1, dtype=np.int64)
    reach = np.eye(num_nodes, dtype=bool)
    dist[reach] = 0
    frontier = reach
    h = 0
    while frontier.any():
        h += 1
        nxt = ((frontier.astype(np.int64) @ adj.astype(np.int64)) > 0) & ~reach
        dist[nxt] = h
        reach |= nxt
        frontier = nxt
    return dist


def i2_sampler(edge_index: np.ndarray, num_nodes: int, hop: int = 3):
    """I2Sampler (SpTupleSampler.py:129-173): for every directed edge (i, j) the nodes within `hop` of i or j (pair-rooted
    k_hop_subgraph), features = (shortest-path distance to i, to j); coalesced (sorted by (i, j, k))."""
    full = shortest_path_matrix(edge_index, num_nodes)
    ids, feats = [], []
    for e in range(edge_index.shape[1]):
        i, j = int(edge_index[0, e]), int(edge_index[1, e])
        subset, _ = k_hop_subgraph([i, j], hop, edge_index, num_nodes)
        ids.append(np.stack((np.full_like(subset, i), np.full_like(subset, j), subset)))
        feats.append(np.stack((full[i][subset], full[j][subset]), axis=-1))
    ind, val = np.concatenate(ids, axis=1), np.concatenate(feats, axis=0)
    return coalesce(ind, val, "min")

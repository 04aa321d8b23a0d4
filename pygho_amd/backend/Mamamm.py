"""
Batched masked x masked contraction.  Mirror of ``pygho/backend/Mamamm.py`` (reference Mamamm.py:7-64):
``mamamm(A, dim1, B, dim2, mask)`` contracts masked dim ``dim1`` of A with masked dim ``dim2`` of B
(dim 0 of both is the shared batch dim), keeps A's then B's remaining masked dims, multiplies
elementwise over the trailing dense dims, and wraps the product with ``mask``.

The reference permutes both operands into channel-outermost copies and calls ``torch.matmul``; here ONE
HIP kernel transposes through LDS and runs the d independent small GEMMs on the matrix cores
(``pygho_masked_bmm``); masked-out operand entries are treated as 0 and the output is zeroed outside ``mask``.
"""
import torch
from torch import BoolTensor

from .. import _ops
from .MaTensor import MaskedTensor


def _as_bik(X: MaskedTensor, dim: int):
    """view X as (b, rows, k) or k-first (b, k, rows): returns (data4d, mask3d|None, rows, k, kfirst, rest_shape)."""
    md = X.masked_dim
    data, mask = X.raw, X.mask
    vector = md == 2                  # (b, k, *dense): a single masked dim besides batch
    if vector:
        assert dim == 1
        data, mask, md = data.unsqueeze(2), mask.unsqueeze(2), 3      # (b, k, 1)
    if dim not in (1, md - 1):        # contracted dim in the middle: bring it to the end (copy)
        data, mask = torch.movedim(data, dim, md - 1), torch.movedim(mask, dim, md - 1)
        dim = md - 1
    b = data.shape[0]
    dense = tuple(data.shape[md:])
    if dim == md - 1:
        rest = tuple(data.shape[1:md - 1])
        kfirst = False
    else:
        rest = tuple(data.shape[2:md])
        kfirst = True
    if vector:
        rest = ()
    k = data.shape[dim]
    rows = 1
    for s in rest:
        rows *= s
    d = 1
    for s in dense:
        d *= s
    shape4 = (b, k, rows, d) if kfirst else (b, rows, k, d)
    data4 = data.contiguous().reshape(shape4)
    # The mask is passed even when the data is already zero-filled: the kernels do not FETCH masked rows (a padded batch is
    # 60 % padding, an adjacency 96 %), and a sparse mask selects the neighbour-list kernel.  The (b, rows, k) view is cached on
    # the mask object so that everything derived from it (uint8 view, density, lists) is computed once per batch.
    cache = getattr(X.mask, "_pygho_bik", None)
    if cache is None:
        cache = {}
        try:
            X.mask._pygho_bik = cache
        except Exception:
            pass
    key = (X.mask._version, dim, md, vector, shape4[:3])
    if key not in cache:
        m3 = mask if (tuple(mask.shape) == shape4[:3] and mask.is_contiguous()) else mask.contiguous().reshape(shape4[:3])
        cache[key] = _ops._mask_u8(m3)
    return data4, cache[key], rows, k, kfirst, rest, dense


def mamamm(A: MaskedTensor, dim1: int, B: MaskedTensor, dim2: int, mask: BoolTensor,
           broadcast_firstdim: bool = True) -> MaskedTensor:
    assert broadcast_firstdim, "only the batched form (shared dim 0) is implemented"
    assert dim1 > 0, "0 dim of A is batch, need to be broadcasted"
    assert dim2 > 0, "0 dim of B is batch, need to be broadcasted"
    Ad, Bd = A, B
    if A.dense_dim != B.dense_dim or A.denseshape != B.denseshape:
        # broadcast the dense dims (e.g. a value-less adjacency against d-channel features)
        da, db = tuple(A.denseshape), tuple(B.denseshape)
        n = max(len(da), len(db))
        da, db = (1,) * (n - len(da)) + da, (1,) * (n - len(db)) + db
        dense = torch.broadcast_shapes(da, db)
        ra = A.raw.reshape(tuple(A.maskedshape) + da).expand(tuple(A.maskedshape) + tuple(dense))
        rb = B.raw.reshape(tuple(B.maskedshape) + db).expand(tuple(B.maskedshape) + tuple(dense))
        Ad = MaskedTensor(ra, A.mask, A.padvalue, A._is_filled_with(A.padvalue))
        Bd = MaskedTensor(rb, B.mask, B.padvalue, B._is_filled_with(B.padvalue))
    if Ad.raw.dtype != Bd.raw.dtype:
        dt = torch.promote_types(Ad.raw.dtype, Bd.raw.dtype)
        Ad = MaskedTensor(Ad.raw.to(dt), Ad.mask, Ad.padvalue, Ad._is_filled_with(Ad.padvalue))
        Bd = MaskedTensor(Bd.raw.to(dt), Bd.mask, Bd.padvalue, Bd._is_filled_with(Bd.padvalue))
    a4, am, ni, nk, akf, resta, dense = _as_bik(Ad, dim1)
    b4, bm, nj, nk2, bkf, restb, _ = _as_bik(Bd, dim2)
    assert nk == nk2, "contracted dims differ"
    nb, d = a4.shape[0], a4.shape[3]
    # the (b, i, j) uint8 view of the output mask is cached on the mask object: everything derived from a mask (density, extents,
    # neighbour lists) is cached on its uint8 view, and a fresh view per call cost a reduction + a host synchronisation + the
    # extents kernel on every contraction (profiles/r02_masked_bmm.md)
    ocache = getattr(mask, "_pygho_bik", None)
    if ocache is None:
        ocache = {}
        try:
            mask._pygho_bik = ocache
        except Exception:
            pass
    okey = ("out", mask._version, nb, ni, nj)
    if okey not in ocache:
        ocache[okey] = _ops._mask_u8(mask.contiguous().reshape(nb, ni, nj))
    om = ocache[okey]
    out = _ops.masked_bmm(a4, b4, am, bm, om, nb, ni, nk, nj, d, akf, bkf)
    out = out.reshape((nb,) + resta + restb + dense)
    return MaskedTensor(out, mask, 0.0, True)

"""
SparseTensor (b, n, m[, *dense]) x MaskedTensor contraction.  Mirror of ``pygho/backend/Spmamm.py``
(reference Spmamm.py:12-68), implemented to the DOCUMENTED semantics: the reference's version raises for
any tensor with dense dims (its validity mask does not broadcast and ``masked_fill`` is not in-place,
SURVEY.md 2.3), so there is no reference behaviour to match beyond the docstring and the einsum its own
(stale) test compares with (tests/test_backend_masked.py:111-116).

    out[b, i, ...] = (+)_k A[b, i, k] * B[b, k, ...]      over valid (unmasked) entries of B

Lowered onto the fused gather * gather -> segment-reduce kernel: one message per sparse entry of A,
grouped by the flattened target (b*n + i), gathering row (b*m + k) of B; B's masked-out rows are zeroed
(sum) / skipped through a compacted message list (max, min).
"""
from typing import Optional

import torch
from torch import BoolTensor

from .. import _ops
from .MaTensor import MaskedTensor, filterinf
from .SpTensor import SparseTensor


def spmamm(A: SparseTensor, dim1: int, B: MaskedTensor, dim2: int, mask: Optional[BoolTensor] = None,
           aggr: str = "sum") -> MaskedTensor:
    assert A.sparse_dim == 3, f"A should have 3 sparse dims, but input has {A.sparse_dim}"
    assert aggr != "mean", "not implemented"
    ind = A.indices
    if dim1 == 1:
        b, n = A.shape[0], A.shape[2]
        kk, tar = ind[1], ind[2]
    elif dim1 == 2:
        b, n = A.shape[0], A.shape[1]
        kk, tar = ind[2], ind[1]
    else:
        raise NotImplementedError
    tB = torch.movedim(B.fill_masked(0.), dim2, 1).contiguous()      # (b, m, *rest, *dense)
    tM = torch.movedim(B.mask, dim2, 1).contiguous()                 # (b, m, *rest)
    m = tB.shape[1]
    rest = tuple(tM.shape[2:])
    dense = tuple(tB.shape[2 + len(rest):])
    rows = tB.reshape((b * m,) + rest + dense)
    src = (ind[0] * m + kk).contiguous()
    tarflat = (ind[0] * n + tar).contiguous()
    val = A.values
    if aggr in ("max", "min"):
        if len(rest) > 0:
            raise NotImplementedError("max/min spmamm with extra masked dims is not implemented")
        keep = torch.nonzero(tM.reshape(-1)[src]).flatten()          # drop messages from masked-out rows
        src, tarflat = src[keep].contiguous(), tarflat[keep].contiguous()
        val = None if val is None else _ops.gather_rows(val, keep)
    if val is not None and len(rest) > 0:
        val = val.reshape((val.shape[0],) + (1,) * len(rest) + tuple(val.shape[1:]))
    out = _ops.spmm_values(val, rows, src, tarflat, b * n, aggr)
    out = out.reshape((b, n) + rest + tuple(out.shape[1 + len(rest):]))
    out = torch.movedim(out, 1, dim2)
    if aggr in ("max", "min"):
        out = filterinf(out)
    return MaskedTensor(out, mask if mask is not None else B.mask)

"""
SparseTensor x dense Tensor (node-level message passing).  Mirror of ``pygho/backend/Spmm.py``
(reference Spmm.py:6-44): ``out[t] = (+)_e A.val[e] * X[src[e]]`` over the 2-D sparse ``A``,
``dim1`` being the contracted sparse dim.  The reference materialises ``val * X[srcind]`` and
scatters it; here gather, multiply and segment reduce are one HIP launch.
"""
from torch import Tensor

from .. import _ops
from .SpTensor import SparseTensor


def spmm(A: SparseTensor, dim1: int, X: Tensor, aggr: str = "sum") -> Tensor:
    assert A.sparse_dim == 2, "can only use 2-dim sparse tensor"
    if dim1 == 0:
        src, tar, n_tar = A._row(0), A._row(1), A.shape[1]
    else:
        src, tar, n_tar = A._row(1), A._row(0), A.shape[0]
    return _ops.spmm_values(A.values, X, src, tar, n_tar, aggr)

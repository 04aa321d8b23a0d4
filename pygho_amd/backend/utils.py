"""
Scatter / segment reduce primitive.  Mirror of ``pygho/backend/utils.py`` (reference
utils.py:6-56): same name, arguments and semantics -- output starts from zeros, segments that
receive nothing stay 0 for EVERY aggregation (``scatter_reduce_(include_self=False)`` into a
zero tensor), 'mean' divides by the segment count, 'max'/'min' are amax/amin.

The reference expands the index to an (M, d) int64 tensor and calls ATen; here the index is
turned once into an int32 CSR plan (cached on the index tensor) and ONE fused HIP kernel
(``pygho_seg_gather_mul_reduce``) reads every source row once and writes every output row once.
"""
from torch import LongTensor, Tensor

from .. import _ops


def torch_scatter_reduce(dim: int, src: Tensor, ind: LongTensor, dim_size: int, aggr: str) -> Tensor:
    assert dim == 0, "other dim not implemented"
    assert ind.ndim == 1, "indice must be 1-d"
    return _ops.scatter_reduce(src, ind, dim_size, aggr)

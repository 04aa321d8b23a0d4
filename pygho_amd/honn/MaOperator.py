"""
Graph operators on ``MaskedTensor`` (dense, padded representations).

Same public names, constructor arguments and ``forward`` signatures as ``pygho/honn/MaOperator.py``.  The
operators are thin, parameter-free shims over three backend calls, so instead of one hand-written class per
operator they are GENERATED from the table below (which masked dims are contracted / pooled / inserted and in
which order the operands enter the contraction) -- reference lines for the table entries:
contractions MaOperator.py:181,201 (OnSubg2D), :217 (OnSubg3D), :258,278 (CrossSubg2D), 2-FWL :138-160;
sparse-adjacency variants :281-372; pooling dims :390-460; unpooling dims :500-522.
"""
from typing import Dict, Iterable, Union

from torch import Tensor
from torch.nn import Module

from ..backend.Mamamm import mamamm
from ..backend.MaTensor import MaskedTensor
from ..backend.Spmamm import spmamm
from ..backend.SpTensor import SparseTensor


def _as_dims(dims: Union[int, Iterable[int]]):
    return sorted(set([dims] if isinstance(dims, int) else dims))


# ---------------------------------------------------------------------------------------------------------
# bases (the reference's generic operators)
# ---------------------------------------------------------------------------------------------------------
class OpMessagePassing(Module):
    """``mamamm(A, dim1, B, dim2, tarX.mask)``: contract masked dim `dim1` of A with `dim2` of B."""

    def __init__(self, dim1: int, dim2: int) -> None:
        super().__init__()
        self.dim1, self.dim2 = dim1, dim2

    def forward(self, A: MaskedTensor, B: MaskedTensor, tarX: MaskedTensor) -> MaskedTensor:
        return mamamm(A, self.dim1, B, self.dim2, tarX.mask, True)


class OpSpMessagePassing(Module):
    """sparse (b, n, n) adjacency against a masked representation: ``spmamm``."""

    def __init__(self, dim1: int, dim2: int, aggr: str = "sum") -> None:
        super().__init__()
        self.dim1, self.dim2, self.aggr = dim1, dim2, aggr

    def forward(self, A: SparseTensor, X: MaskedTensor, tarX: MaskedTensor) -> MaskedTensor:
        assert A.sparse_dim == 3, "A should be bxnxn adjacency matrix "
        return spmamm(A, self.dim1, X, self.dim2, tarX.mask, self.aggr)


class OpDiag(Module):
    def __init__(self, dims: Iterable[int]) -> None:
        super().__init__()
        self.dims = _as_dims(dims)

    def forward(self, A: MaskedTensor) -> MaskedTensor:
        return A.diag(self.dims)


class OpPooling(Module):
    def __init__(self, dims: Union[int, Iterable[int]], pool: str = "sum") -> None:
        super().__init__()
        self.dims, self.pool = _as_dims(dims), pool

    def forward(self, X: MaskedTensor) -> MaskedTensor:
        return getattr(X, self.pool)(dims=self.dims, keepdim=False)


class OpUnpooling(Module):
    def __init__(self, dims: Union[int, Iterable[int]]) -> None:
        super().__init__()
        self.dims = _as_dims(dims)

    def forward(self, X: MaskedTensor, tarX: MaskedTensor) -> MaskedTensor:
        return X.unpooling(self.dims, tarX)


class OpNodeMessagePassing(Module):
    """dense node-level message passing ``A x`` (adjacency (b, n, n) against node features (b, n, d))."""

    def forward(self, A: MaskedTensor, X: MaskedTensor, tarX: MaskedTensor) -> Tensor:
        return mamamm(A, 2, X, 1, tarX.mask)


class OpSpNodeMessagePassing(Module):
    def __init__(self, aggr: str = "sum") -> None:
        super().__init__()
        self.aggr = aggr

    def forward(self, A: SparseTensor, X: MaskedTensor, tarX: MaskedTensor) -> Tensor:
        return spmamm(A, 2, X, 1, tarX.mask, self.aggr)


# ---------------------------------------------------------------------------------------------------------
# generated operators
# ---------------------------------------------------------------------------------------------------------
_NAMES = {3: "bxnxn", 4: "bxnxnxn"}


def _dense_contraction(name: str, dim1: int, dim2: int, x_first: bool, a_md: int, x_md: int, doc: str):
    """forward(A, X, datadict, tarX): the tuple representation X and the adjacency A enter `mamamm` as
    (X, A) when `x_first` else (A, X)."""

    def __init__(self) -> None:
        OpMessagePassing.__init__(self, dim1, dim2)

    def forward(self, A: MaskedTensor, X: MaskedTensor, datadict: Dict, tarX: MaskedTensor) -> MaskedTensor:
        assert A.masked_dim == a_md, f"A should be {_NAMES[a_md]} adjacency matrix "
        assert X.masked_dim == x_md, f"X should be {_NAMES[x_md]} {x_md - 1}d representations"
        first, second = (X, A) if x_first else (A, X)
        return OpMessagePassing.forward(self, first, second, tarX)

    return type(name, (OpMessagePassing,), {"__init__": __init__, "forward": forward, "__doc__": doc})


def _sparse_contraction(name: str, dim1: int, dim2: int, x_md: int, doc: str):
    def __init__(self, aggr: str = "sum") -> None:
        OpSpMessagePassing.__init__(self, dim1, dim2, aggr)

    def forward(self, A: SparseTensor, X: MaskedTensor, datadict: Dict, tarX: MaskedTensor) -> MaskedTensor:
        assert X.masked_dim == x_md, f"X should be {_NAMES[x_md]} representation "
        return OpSpMessagePassing.forward(self, A, X, tarX)

    return type(name, (OpSpMessagePassing,), {"__init__": __init__, "forward": forward, "__doc__": doc})


def _pooling(name: str, dims, x_md: int, doc: str):
    def __init__(self, pool: str = "sum") -> None:
        OpPooling.__init__(self, dims, pool)

    def forward(self, X: MaskedTensor) -> MaskedTensor:
        assert X.masked_dim == x_md, f"X should be {_NAMES[x_md]} representations"
        return OpPooling.forward(self, X)

    return type(name, (OpPooling,), {"__init__": __init__, "forward": forward, "__doc__": doc})


def _fixed_dims(name: str, base, dims, doc: str, x_md=None):
    def __init__(self) -> None:
        base.__init__(self, dims)

    members = {"__init__": __init__, "__doc__": doc}
    if x_md is not None:
        def forward(self, X: MaskedTensor) -> MaskedTensor:
            assert X.masked_dim == x_md, f"X should be {_NAMES[x_md]} representations"
            return base.forward(self, X)
        members["forward"] = forward
    return type(name, (base,), members)


OpMessagePassingOnSubg2D = _dense_contraction(
    "OpMessagePassingOnSubg2D", 2, 1, True, 3, 3, "message passing inside every subgraph: X A (dims 2, 1)")
OpMessagePassingOnSubg3D = _dense_contraction(
    "OpMessagePassingOnSubg3D", 3, 1, True, 3, 4, "message passing inside every subgraph, 3-D representations (dims 3, 1)")
OpMessagePassingCrossSubg2D = _dense_contraction(
    "OpMessagePassingCrossSubg2D", 1, 1, False, 3, 3, "message passing across subgraphs: A X (dims 1, 1)")


class Op2FWL(OpMessagePassing):
    """2-FWL style product of two 2-D representations (dims 2, 1)."""

    def __init__(self) -> None:
        super().__init__(2, 1)

    def forward(self, X1: MaskedTensor, X2: MaskedTensor, datadict: Dict, tarX: MaskedTensor) -> MaskedTensor:
        assert X1.masked_dim == 3 and X2.masked_dim == 3, "X1, X2 should be bxnxn 2d representations"
        return super().forward(X1, X2, tarX)


OpSpMessagePassingOnSubg2D = _sparse_contraction("OpSpMessagePassingOnSubg2D", 1, 2, 3, "sparse adjacency, X A on 2-D X")
# the reference asserts masked_dim == 3 for this 3-D operator (MaOperator.py:319), which can never hold
OpSpMessagePassingOnSubg3D = _sparse_contraction("OpSpMessagePassingOnSubg3D", 1, 3, 4, "sparse adjacency, 3-D X")
OpSpMessagePassingCrossSubg2D = _sparse_contraction("OpSpMessagePassingCrossSubg2D", 1, 1, 3, "sparse adjacency, A X")

OpDiag2D = _fixed_dims("OpDiag2D", OpDiag, [1, 2], "diagonal X[b, i, i] of 2-D representations", x_md=3)
OpPoolingSubg2D = _pooling("OpPoolingSubg2D", [2], 3, "pool the nodes of each subgraph")
OpPoolingSubg3D = _pooling("OpPoolingSubg3D", [3], 4, "pool the last tuple dim of 3-D representations")
OpPoolingCrossSubg2D = _pooling("OpPoolingCrossSubg2D", [1], 3, "pool the same node over all subgraphs")
OpUnpoolingSubgNodes2D = _fixed_dims("OpUnpoolingSubgNodes2D", OpUnpooling, [2], "node representations -> every subgraph")
OpUnpoolingRootNodes2D = _fixed_dims("OpUnpoolingRootNodes2D", OpUnpooling, [1], "root-node representations -> their subgraph")

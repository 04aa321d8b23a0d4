"""
Graph operators on ``SparseTensor`` (parameter-free ``nn.Module`` shims).

Same public names, constructor arguments, ``forward`` signatures and ``datadict`` key convention as
``pygho/honn/SpOperator.py`` (KEYSEP :12, key format :135, lookups :165-183).  ``OpMessagePassing`` carries the
logic; its named specialisations only fix the operand roles, so they are generated from a table instead of being
written out one class at a time (reference: Op2FWL :185-227, OnSubg2D :230-277, OnSubg3D :280-327,
CrossSubg2D :330-372, pooling :470-545, unpooling :586-601).
"""
from typing import Callable, Dict, Iterable, List, Optional, Union

from torch import Tensor
from torch.nn import Module

from ..backend.SpTensor import SparseTensor
from ..backend.Spmm import spmm
from ..backend.Spspmm import spspmm, spspmpnn

KEYSEP = "___"


def parse_precomputekey(model: Module) -> List[str]:
    """sorted unique precompute keys of every message-passing operator inside ``model`` (reference :15-44)."""
    return sorted({m.precomputekey for m in model.modules() if isinstance(m, OpMessagePassing)})


def _as_dims(dims: Union[int, Iterable[int]]):
    return sorted(set([dims] if isinstance(dims, int) else dims))


class OpNodeMessagePassing(Module):
    """node-level message passing ``A X`` with a 2-D sparse adjacency."""

    def __init__(self, aggr: str = "sum") -> None:
        super().__init__()
        self.aggr = aggr

    def forward(self, A: SparseTensor, X: Tensor, tarX: Optional[Tensor] = None) -> Tensor:
        assert A.sparse_dim == 2, "A is adjacency matrix of the whole graph of shape nxn"
        return spmm(A, 1, X, self.aggr)


class OpMessagePassing(Module):
    """
    Generalised message passing ``out = A (x)_{dim1, dim2} B`` restricted to the pattern of ``tarX``.
    The plan is looked up in ``datadict`` under ``f"{op0}___{op1}___{dim1}___{op2}___{dim2}___acd"`` (also
    ``___bcd`` / ``___tarind``), exactly the keys the reference's pre-transform produces.
    """

    def __init__(self, op0: str = "X", op1: str = "X", dim1: int = 1, op2: str = "A", dim2: int = 0,
                 aggr: str = "sum", message_func: Optional[Callable] = None) -> None:
        super().__init__()
        self.dim1, self.dim2, self.aggr = dim1, dim2, aggr
        self.precomputekey = KEYSEP.join((op0, op1, str(dim1), op2, str(dim2)))
        self.message_func = message_func
        self.use_mpnn = message_func is not None

    def _plan(self, datadict: Dict, which: str):
        return datadict.get(self.precomputekey + KEYSEP + which, None)

    def forward(self, A: SparseTensor, B: SparseTensor, datadict: Dict, tarX: Optional[SparseTensor] = None) -> SparseTensor:
        if self.use_mpnn:
            assert tarX is not None, "target representation is a must when message func is not None"
            return spspmpnn(A, self.dim1, B, self.dim2, tarX, self._plan(datadict, "acd"), self.message_func, self.aggr)
        tar_ind = self._plan(datadict, "tarind") if tarX is None else tarX.indices
        return spspmm(A, self.dim1, B, self.dim2, self.aggr, acd=self._plan(datadict, "acd"),
                      bcd=self._plan(datadict, "bcd"), tar_ind=tar_ind)


def _variant(name: str, roles: str, dim1: int, x_sd: int, doc: str):
    """`roles`: which of (tuple feature X, adjacency A) is op1 / op2 of the product, e.g. "XA" = X (x) A."""

    def __init__(self, aggr: str = "sum", optuplefeat: str = "X", opadj: str = "A",
                 message_func: Optional[Callable] = None) -> None:
        ops = {"X": optuplefeat, "A": opadj}
        OpMessagePassing.__init__(self, optuplefeat, ops[roles[0]], dim1, ops[roles[1]], 0, aggr, message_func)

    def forward(self, A: SparseTensor, X: SparseTensor, datadict: Dict, tarX: Optional[SparseTensor] = None) -> SparseTensor:
        assert A.sparse_dim == 2, "A should be nxn adjacency matrix "
        assert X.sparse_dim == x_sd, f"X should be {x_sd}d representations"
        operands = {"X": X, "A": A}
        return OpMessagePassing.forward(self, operands[roles[0]], operands[roles[1]], datadict, tarX)

    return type(name, (OpMessagePassing,), {"__init__": __init__, "forward": forward, "__doc__": doc})


OpMessagePassingOnSubg2D = _variant("OpMessagePassingOnSubg2D", "XA", 1, 2,
                                    "message passing inside every subgraph of a 2-D representation: X A")
OpMessagePassingOnSubg3D = _variant("OpMessagePassingOnSubg3D", "XA", 2, 3,
                                    "message passing inside every subgraph of a 3-D representation: X A over the last dim")
OpMessagePassingCrossSubg2D = _variant("OpMessagePassingCrossSubg2D", "AX", 1, 2,
                                       "message passing across subgraphs: A X")


class Op2FWL(OpMessagePassing):
    """2-FWL style product of two 2-D representations, X <- X1 X2."""

    def __init__(self, aggr: str = "sum", optuplefeat: str = "X") -> None:
        super().__init__(optuplefeat, optuplefeat, 1, optuplefeat, 0, aggr)

    def forward(self, X1: SparseTensor, X2: SparseTensor, datadict: Dict, tarX: Optional[SparseTensor] = None) -> SparseTensor:
        assert X1.sparse_dim == 2 and X2.sparse_dim == 2, "X1, X2 should be 2d representations"
        return super().forward(X1, X2, datadict, tarX)


class OpDiag(Module):
    def __init__(self, dims: Iterable[int], return_sparse: bool = False) -> None:
        super().__init__()
        self.dims, self.return_sparse = _as_dims(dims), return_sparse

    def forward(self, A: SparseTensor) -> Union[Tensor, SparseTensor]:
        return A.diag(self.dims, return_sparse=self.return_sparse)


class OpDiag2D(OpDiag):
    def __init__(self) -> None:
        super().__init__([0, 1], False)

    def forward(self, X: SparseTensor) -> Tensor:
        assert X.sparse_dim == 2, "X should be 2d representations"
        return super().forward(X)


class OpPooling(Module):
    def __init__(self, dims: Union[int, Iterable[int]], pool: str = "sum", return_sparse: bool = False) -> None:
        super().__init__()
        self.dims, self.pool, self.return_sparse = _as_dims(dims), pool, return_sparse

    def forward(self, X: SparseTensor) -> Union[SparseTensor, Tensor]:
        return getattr(X, self.pool)(self.dims, return_sparse=self.return_sparse)


def _pooling(name: str, dim: int, sparse_out: bool, x_sd: int, doc: str):
    def __init__(self, pool) -> None:
        OpPooling.__init__(self, dim, pool, sparse_out)

    def forward(self, X: SparseTensor):
        assert X.sparse_dim == x_sd, f"X should be {x_sd}d representations"
        return OpPooling.forward(self, X)

    return type(name, (OpPooling,), {"__init__": __init__, "forward": forward, "__doc__": doc})


OpPoolingSubg2D = _pooling("OpPoolingSubg2D", 1, False, 2, "pool the nodes of each subgraph -> dense (n, d)")
OpPoolingSubg3D = _pooling("OpPoolingSubg3D", 2, True, 3, "pool the last dim of 3-D representations -> sparse 2-D")
OpPoolingCrossSubg2D = _pooling("OpPoolingCrossSubg2D", 0, False, 2, "pool the same node across subgraphs -> dense (n, d)")


class OpUnpooling(Module):
    """broadcast a dense per-node tensor or a lower-order sparse tensor to the pattern of ``tarX``."""

    def __init__(self, dims: Union[int, Iterable[int]], fromdense1dim: bool = True) -> None:
        super().__init__()
        self.dims, self.fromdense1dim = _as_dims(dims), fromdense1dim

    def forward(self, X: Union[Tensor, SparseTensor], tarX: SparseTensor) -> SparseTensor:
        if not isinstance(X, Tensor):
            return X.unpooling(self.dims, tarX)
        kept = [d for d in range(tarX.sparse_dim) if d not in self.dims]
        assert len(kept) == 1, "canonly pooling from 1 dim"
        return tarX.unpooling_fromdense1dim(kept[0], X)


class OpUnpoolingSubgNodes2D(OpUnpooling):
    def __init__(self) -> None:
        super().__init__(1, True)


class OpUnpoolingRootNodes2D(OpUnpooling):
    def __init__(self) -> None:
        super().__init__(0, True)

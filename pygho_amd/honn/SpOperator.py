"""
Graph operators on ``SparseTensor`` (parameter-free ``nn.Module`` shims).  Mirror of
``pygho/honn/SpOperator.py``: same class names, constructor / forward signatures and the ``datadict``
key convention (reference SpOperator.py:12, :135, :165-183); the work is done by the HIP-backed
``pygho_amd.backend`` functions.
"""
from typing import Callable, Dict, Iterable, List, Optional, Union

from torch import Tensor
from torch.nn import Module

from ..backend.SpTensor import SparseTensor
from ..backend.Spmm import spmm
from ..backend.Spspmm import spspmm, spspmpnn

KEYSEP = "___"


def parse_precomputekey(model: Module) -> List[str]:
    """sorted unique precompute keys of every message-passing operator inside ``model``
    (reference SpOperator.py:15-44)."""
    return sorted({m.precomputekey for m in model.modules() if isinstance(m, OpMessagePassing)})


class OpNodeMessagePassing(Module):
    """node-level message passing ``A X`` (reference SpOperator.py:47-85)."""

    def __init__(self, aggr: str = "sum") -> None:
        super().__init__()
        self.aggr = aggr

    def forward(self, A: SparseTensor, X: Tensor, tarX: Optional[Tensor] = None) -> Tensor:
        assert A.sparse_dim == 2, "A is adjacency matrix of the whole graph of shape nxn"
        return spmm(A, 1, X, self.aggr)


class OpMessagePassing(Module):
    """
    Generalised message passing ``out = A (x)_{dim1,dim2} B`` restricted to the pattern of ``tarX``.
    ``precomputekey = f"{op0}___{op1}___{dim1}___{op2}___{dim2}"`` names the plan
    (``datadict[key + "___acd"]``) produced off-line (reference SpOperator.py:88-183).
    """

    def __init__(self, op0: str = "X", op1: str = "X", dim1: int = 1, op2: str = "A", dim2: int = 0,
                 aggr: str = "sum", message_func: Optional[Callable] = None) -> None:
        super().__init__()
        self.dim1 = dim1
        self.dim2 = dim2
        self.precomputekey = KEYSEP.join((op0, op1, str(dim1), op2, str(dim2)))
        self.aggr = aggr
        self.message_func = message_func
        self.use_mpnn = message_func is not None

    def forward(self, A: SparseTensor, B: SparseTensor, datadict: Dict, tarX: Optional[SparseTensor] = None) -> SparseTensor:
        key = self.precomputekey + KEYSEP
        if self.use_mpnn:
            assert tarX is not None, "target representation is a must when message func is not None"
            return spspmpnn(A, self.dim1, B, self.dim2, tarX, datadict.get(key + "acd", None), self.message_func, self.aggr)
        return spspmm(A, self.dim1, B, self.dim2, self.aggr,
                      acd=datadict.get(key + "acd", None), bcd=datadict.get(key + "bcd", None),
                      tar_ind=datadict.get(key + "tarind", None) if tarX is None else tarX.indices)


class Op2FWL(OpMessagePassing):
    """2-FWL style product of two 2-D representations, X <- X1 X2 (reference SpOperator.py:185-227)."""

    def __init__(self, aggr: str = "sum", optuplefeat: str = "X") -> None:
        super().__init__(optuplefeat, optuplefeat, 1, optuplefeat, 0, aggr)

    def forward(self, X1: SparseTensor, X2: SparseTensor, datadict: Dict, tarX: Optional[SparseTensor] = None) -> SparseTensor:
        assert X1.sparse_dim == 2, "X1 should be 2d representations "
        assert X2.sparse_dim == 2, "X2 should be 2d representations"
        return super().forward(X1, X2, datadict, tarX)


class OpMessagePassingOnSubg2D(OpMessagePassing):
    """message passing inside every subgraph, 2-D representations: X A (reference SpOperator.py:230-277)."""

    def __init__(self, aggr: str = "sum", optuplefeat: str = "X", opadj: str = "A",
                 message_func: Optional[Callable] = None) -> None:
        super().__init__(optuplefeat, optuplefeat, 1, opadj, 0, aggr, message_func)

    def forward(self, A: SparseTensor, X: SparseTensor, datadict: Dict, tarX: Optional[SparseTensor] = None) -> SparseTensor:
        assert A.sparse_dim == 2, "A should be nxn adjacency matrix "
        assert X.sparse_dim == 2, "X should be 2d representations"
        return super().forward(X, A, datadict, tarX)


class OpMessagePassingOnSubg3D(OpMessagePassing):
    """message passing inside every subgraph, 3-D representations (reference SpOperator.py:280-327)."""

    def __init__(self, aggr: str = "sum", optuplefeat: str = "X", opadj: str = "A",
                 message_func: Optional[Callable] = None) -> None:
        super().__init__(optuplefeat, optuplefeat, 2, opadj, 0, aggr, message_func)

    def forward(self, A: SparseTensor, X: SparseTensor, datadict: Dict, tarX: Optional[SparseTensor] = None) -> SparseTensor:
        assert A.sparse_dim == 2, "A should be nxn adjacency matrix "
        assert X.sparse_dim == 3, "X should be 3d representations"
        return super().forward(X, A, datadict, tarX)


class OpMessagePassingCrossSubg2D(OpMessagePassing):
    """message passing across subgraphs: A X (reference SpOperator.py:330-372)."""

    def __init__(self, aggr: str = "sum", optuplefeat: str = "X", opadj: str = "A",
                 message_func: Optional[Callable] = None) -> None:
        super().__init__(optuplefeat, opadj, 1, optuplefeat, 0, aggr, message_func)

    def forward(self, A: SparseTensor, X: SparseTensor, datadict: Dict, tarX: Optional[SparseTensor] = None) -> SparseTensor:
        assert A.sparse_dim == 2, "A should be nxn adjacency matrix "
        assert X.sparse_dim == 2, "X should be 2d representations"
        return super().forward(A, X, datadict, tarX)


class OpDiag(Module):
    """diagonal extraction (reference SpOperator.py:375-403)."""

    def __init__(self, dims: Iterable[int], return_sparse: bool = False) -> None:
        super().__init__()
        self.dims = sorted(set(dims))
        self.return_sparse = return_sparse

    def forward(self, A: SparseTensor) -> Union[Tensor, SparseTensor]:
        return A.diag(self.dims, return_sparse=self.return_sparse)


class OpDiag2D(OpDiag):

    def __init__(self) -> None:
        super().__init__([0, 1], False)

    def forward(self, X: SparseTensor) -> Tensor:
        assert X.sparse_dim == 2, "X should be 2d representations"
        return X.diag(self.dims, return_sparse=self.return_sparse)


class OpPooling(Module):
    """pool tuple representations over sparse dims (reference SpOperator.py:427-467)."""

    def __init__(self, dims: Union[int, Iterable[int]], pool: str = "sum", return_sparse: bool = False) -> None:
        super().__init__()
        self.dims = sorted(set([dims] if isinstance(dims, int) else dims))
        self.pool = pool
        self.return_sparse = return_sparse

    def forward(self, X: SparseTensor) -> Union[SparseTensor, Tensor]:
        return getattr(X, self.pool)(self.dims, return_sparse=self.return_sparse)


class OpPoolingSubg2D(OpPooling):
    """pool the nodes of each subgraph -> dense (n, d) (reference SpOperator.py:470-493)."""

    def __init__(self, pool) -> None:
        super().__init__(1, pool, False)

    def forward(self, X: SparseTensor) -> Tensor:
        assert X.sparse_dim == 2, "X should be 2d representations"
        return super().forward(X)


class OpPoolingSubg3D(OpPooling):
    """pool the last dim of 3-D representations -> sparse 2-D (reference SpOperator.py:496-519)."""

    def __init__(self, pool) -> None:
        super().__init__(2, pool, True)

    def forward(self, X: SparseTensor) -> SparseTensor:
        assert X.sparse_dim == 3, "X should be 3d representations"
        return super().forward(X)


class OpPoolingCrossSubg2D(OpPooling):
    """pool the same node across subgraphs -> dense (reference SpOperator.py:522-545)."""

    def __init__(self, pool) -> None:
        super().__init__(0, pool, False)

    def forward(self, X: SparseTensor) -> Tensor:
        assert X.sparse_dim == 2, "X should be 2d representations"
        return super().forward(X)


class OpUnpooling(Module):
    """broadcast lower-order representations to a tuple pattern (reference SpOperator.py:548-583)."""

    def __init__(self, dims: Union[int, Iterable[int]], fromdense1dim: bool = True) -> None:
        super().__init__()
        self.dims = sorted(set([dims] if isinstance(dims, int) else dims))
        self.fromdense1dim = fromdense1dim

    def forward(self, X: Union[Tensor, SparseTensor], tarX: SparseTensor) -> SparseTensor:
        if isinstance(X, Tensor):
            leftdim = list(set(range(tarX.sparse_dim)) - set(self.dims))
            assert len(leftdim) == 1, "canonly pooling from 1 dim"
            return tarX.unpooling_fromdense1dim(leftdim[0], X)
        return X.unpooling(self.dims, tarX)


class OpUnpoolingSubgNodes2D(OpUnpooling):

    def __init__(self) -> None:
        super().__init__(1, True)


class OpUnpoolingRootNodes2D(OpUnpooling):

    def __init__(self) -> None:
        super().__init__(0, True)

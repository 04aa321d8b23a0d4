"""
Sparse / dense dispatch of the graph operators.  Mirror of ``pygho/honn/TensorOp.py``: every class picks
the ``SpOperator`` or ``MaOperator`` implementation from ``mode`` in {"SS", "SD", "DD"} (first letter =
adjacency, second = tuple representation; "S" sparse, "D" dense/masked) or {"S", "D"}.
"""
from typing import Callable, Dict, Literal, Optional, Union

from torch import Tensor
from torch.nn import Module

from . import MaOperator, SpOperator
from ..backend.MaTensor import MaskedTensor
from ..backend.SpTensor import SparseTensor

_DENSE_MSG = "general message passing with message_func is not implemented for Dense"
_DENSE_AGGR = "only sum aggragation implemented for Dense adjacency"


class OpNodeMessagePassing(Module):
    """node-level message passing (reference TensorOp.py:14-62)."""

    def __init__(self, mode: Literal["SS", "SD", "DD"] = "SS", aggr: str = "sum") -> None:
        super().__init__()
        if mode == "SS":
            self.mod = SpOperator.OpNodeMessagePassing(aggr)
        elif mode == "SD":
            self.mod = MaOperator.OpSpNodeMessagePassing(aggr)
        elif mode == "DD":
            assert aggr == "sum", f"aggr {aggr} is not implemented for DD"
            self.mod = MaOperator.OpNodeMessagePassing()

    def forward(self, A, X):
        return self.mod.forward(A, X, X)


class _Dispatch4(Module):
    """shared forward(A, X, datadict, tarX) of the tuple message-passing operators."""

    def forward(self, A: Union[SparseTensor, MaskedTensor], X: Union[SparseTensor, MaskedTensor],
                datadict: Optional[Dict] = None, tarX: Optional[Union[SparseTensor, MaskedTensor]] = None):
        return self.mod.forward(A, X, datadict, tarX)


class Op2FWL(_Dispatch4):
    """2-FWL product X1 X2 (reference TensorOp.py:65-113)."""

    def __init__(self, mode: Literal["SS", "DD"] = "SS", aggr: Literal["sum", "mean", "max"] = "sum",
                 optuplefeat: str = "X") -> None:
        super().__init__()
        if mode == "SS":
            self.mod = SpOperator.Op2FWL(aggr, optuplefeat)
        elif mode == "DD":
            assert aggr == "sum", _DENSE_AGGR
            self.mod = MaOperator.Op2FWL()
        else:
            raise NotImplementedError


class OpMessagePassingOnSubg2D(_Dispatch4):
    """message passing within each subgraph, 2-D (reference TensorOp.py:116-187)."""

    def __init__(self, mode: Literal["SD", "SS", "DD"] = "SS", aggr: Literal["sum", "mean", "max"] = "sum",
                 optuplefeat: str = "X", opadj: str = "A", message_func: Optional[Callable] = None) -> None:
        super().__init__()
        if mode == "SS":
            self.mod = SpOperator.OpMessagePassingOnSubg2D(aggr, optuplefeat, opadj, message_func)
        elif mode == "SD":
            assert message_func is None, _DENSE_MSG
            self.mod = MaOperator.OpSpMessagePassingOnSubg2D(aggr)
        elif mode == "DD":
            assert message_func is None, _DENSE_MSG
            assert aggr == "sum", _DENSE_AGGR
            self.mod = MaOperator.OpMessagePassingOnSubg2D()
        else:
            raise NotImplementedError


class OpMessagePassingOnSubg3D(_Dispatch4):
    """message passing within each subgraph, 3-D (reference TensorOp.py:190-256)."""

    def __init__(self, mode: Literal["SD", "SS", "DD"] = "SS", aggr: Literal["sum", "mean", "max"] = "sum",
                 optuplefeat: str = "X", opadj: str = "A", message_func: Optional[Callable] = None) -> None:
        super().__init__()
        if mode == "SS":
            # the reference drops message_func here (TensorOp.py:219-220); it is forwarded instead
            self.mod = SpOperator.OpMessagePassingOnSubg3D(aggr, optuplefeat, opadj, message_func)
        elif mode == "SD":
            assert message_func is None, _DENSE_MSG
            self.mod = MaOperator.OpSpMessagePassingOnSubg3D(aggr)
        elif mode == "DD":
            assert message_func is None, _DENSE_MSG
            assert aggr == "sum", _DENSE_AGGR
            self.mod = MaOperator.OpMessagePassingOnSubg3D()
        else:
            raise NotImplementedError


class OpMessagePassingCrossSubg2D(_Dispatch4):
    """message passing across subgraphs (reference TensorOp.py:259-329)."""

    def __init__(self, mode: Literal["SD", "SS", "DD"] = "SS", aggr: Literal["sum", "mean", "max"] = "sum",
                 optuplefeat: str = "X", opadj: str = "A", message_func: Optional[Callable] = None) -> None:
        super().__init__()
        if mode == "SS":
            self.mod = SpOperator.OpMessagePassingCrossSubg2D(aggr, optuplefeat, opadj, message_func)
        elif mode == "SD":
            assert message_func is None, _DENSE_MSG
            # the reference passes `aggr` to a 0-argument constructor here (TensorOp.py:287 -> TypeError);
            # the sparse-adjacency operator is what the mode means
            self.mod = MaOperator.OpSpMessagePassingCrossSubg2D(aggr)
        elif mode == "DD":
            assert message_func is None, _DENSE_MSG
            assert aggr == "sum", _DENSE_AGGR
            self.mod = MaOperator.OpMessagePassingCrossSubg2D()
        else:
            raise NotImplementedError


def _pick(mode: str, sp_cls, ma_cls, *args):
    if mode == "S":
        return sp_cls(*args)
    if mode == "D":
        return ma_cls(*args)
    raise NotImplementedError


class OpDiag2D(Module):
    """diagonal of 2-D representations (reference TensorOp.py:332-365)."""

    def __init__(self, mode: Literal["D", "S"] = "S") -> None:
        super().__init__()
        self.mod = _pick(mode, SpOperator.OpDiag2D, MaOperator.OpDiag2D)

    def forward(self, X: Union[MaskedTensor, SparseTensor]) -> Union[MaskedTensor, Tensor]:
        return self.mod.forward(X)


class OpPoolingSubg2D(Module):
    """pool nodes within each subgraph (reference TensorOp.py:368-395)."""

    def __init__(self, mode: Literal["S", "D"] = "S", pool: str = "sum") -> None:
        super().__init__()
        self.mod = _pick(mode, SpOperator.OpPoolingSubg2D, MaOperator.OpPoolingSubg2D, pool)

    def forward(self, X):
        return self.mod(X)


class OpPoolingSubg3D(Module):
    """pool the last tuple dim of 3-D representations (reference TensorOp.py:398-425)."""

    def __init__(self, mode: Literal["S", "D"] = "S", pool: str = "sum") -> None:
        super().__init__()
        self.mod = _pick(mode, SpOperator.OpPoolingSubg3D, MaOperator.OpPoolingSubg3D, pool)

    def forward(self, X):
        return self.mod(X)


class OpPoolingCrossSubg2D(Module):
    """pool the same node over all subgraphs (reference TensorOp.py:428-451)."""

    def __init__(self, mode: Literal["S", "D"] = "S", pool: str = "sum") -> None:
        super().__init__()
        self.mod = _pick(mode, SpOperator.OpPoolingCrossSubg2D, MaOperator.OpPoolingCrossSubg2D, pool)

    def forward(self, X):
        return self.mod(X)


class OpUnpoolingSubgNodes2D(Module):
    """node representations -> every subgraph (reference TensorOp.py:454-476)."""

    def __init__(self, mode: Literal["S", "D"] = "S") -> None:
        super().__init__()
        self.mod = _pick(mode, SpOperator.OpUnpoolingSubgNodes2D, MaOperator.OpUnpoolingSubgNodes2D)

    def forward(self, X, tarX):
        return self.mod.forward(X, tarX)


class OpUnpoolingRootNodes2D(Module):
    """root-node representations -> their subgraph (reference TensorOp.py:479-500)."""

    def __init__(self, mode: Literal["S", "D"] = "S") -> None:
        super().__init__()
        self.mod = _pick(mode, SpOperator.OpUnpoolingRootNodes2D, MaOperator.OpUnpoolingRootNodes2D)

    def forward(self, X, tarX):
        return self.mod.forward(X, tarX)

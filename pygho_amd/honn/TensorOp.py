"""
Sparse / dense dispatch of the graph operators.

Same class names, constructor arguments and ``forward`` signatures as ``pygho/honn/TensorOp.py``: each class
picks the ``SpOperator`` or ``MaOperator`` implementation from ``mode`` -- "SS" / "SD" / "DD" (first letter =
adjacency, second = tuple representation; S sparse, D dense) for the message-passing operators, "S" / "D" for
pooling, diagonal and unpooling -- and forwards to it as ``self.mod``.  The dispatch rules are one table here
(reference TensorOp.py:14-500).
"""
from typing import Callable, Dict, Optional, Union

from torch.nn import Module

from . import MaOperator, SpOperator
from ..backend.MaTensor import MaskedTensor
from ..backend.SpTensor import SparseTensor

Rep = Union[SparseTensor, MaskedTensor]
_DENSE_MSG = "general message passing with message_func is not implemented for Dense"
_DENSE_AGGR = "only sum aggragation implemented for Dense adjacency"


def _pick_tuple_operator(kind: str, mode: str, aggr: str, optuplefeat: str, opadj: str, message_func):
    """SS -> sparse operator (keeps aggr, names, message function); SD -> sparse adjacency on masked X (aggr
    only); DD -> masked contraction on the matrix cores (sum only)."""
    if mode == "SS":
        cls = getattr(SpOperator, kind)
        return cls(aggr, optuplefeat) if kind == "Op2FWL" else cls(aggr, optuplefeat, opadj, message_func)
    if mode == "SD" and kind != "Op2FWL":
        assert message_func is None, _DENSE_MSG
        # (the reference's CrossSubg2D "SD" branch passes aggr to a 0-argument constructor -> TypeError,
        #  TensorOp.py:287; the sparse-adjacency operator is what the mode means)
        return getattr(MaOperator, kind.replace("OpMessagePassing", "OpSpMessagePassing"))(aggr)
    if mode == "DD":
        assert message_func is None, _DENSE_MSG
        assert aggr == "sum", _DENSE_AGGR
        return getattr(MaOperator, kind)()
    raise NotImplementedError


class _TupleOperator(Module):
    """forward(A, X, datadict, tarX) -> self.mod.forward(...)"""
    _kind = ""

    def __init__(self, mode: str = "SS", aggr: str = "sum", optuplefeat: str = "X", opadj: str = "A",
                 message_func: Optional[Callable] = None) -> None:
        super().__init__()
        self.mod = _pick_tuple_operator(self._kind, mode, aggr, optuplefeat, opadj, message_func)

    def forward(self, A: Rep, X: Rep, datadict: Optional[Dict] = None, tarX: Optional[Rep] = None) -> Rep:
        return self.mod.forward(A, X, datadict, tarX)


class OpMessagePassingOnSubg2D(_TupleOperator):
    """message passing within each subgraph, 2-D representations"""
    _kind = "OpMessagePassingOnSubg2D"


class OpMessagePassingOnSubg3D(_TupleOperator):
    """message passing within each subgraph, 3-D representations (the reference drops `message_func` in SS mode,
    TensorOp.py:219-220; it is forwarded here)"""
    _kind = "OpMessagePassingOnSubg3D"


class OpMessagePassingCrossSubg2D(_TupleOperator):
    """message passing across subgraphs"""
    _kind = "OpMessagePassingCrossSubg2D"


class Op2FWL(Module):
    """2-FWL product X1 X2"""

    def __init__(self, mode: str = "SS", aggr: str = "sum", optuplefeat: str = "X") -> None:
        super().__init__()
        self.mod = _pick_tuple_operator("Op2FWL", mode, aggr, optuplefeat, "A", None)

    def forward(self, X1: Rep, X2: Rep, datadict: Optional[Dict] = None, tarX: Optional[Rep] = None) -> Rep:
        return self.mod.forward(X1, X2, datadict, tarX)


class OpNodeMessagePassing(Module):
    """node-level message passing A x"""

    def __init__(self, mode: str = "SS", aggr: str = "sum") -> None:
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


def _by_layout(kind: str, mode: str, *args):
    if mode not in ("S", "D"):
        raise NotImplementedError
    return getattr(SpOperator if mode == "S" else MaOperator, kind)(*args)


class _UnaryOperator(Module):
    _kind = ""

    def forward(self, X: Rep):
        return self.mod.forward(X)


class OpDiag2D(_UnaryOperator):
    """diagonal of 2-D representations"""
    _kind = "OpDiag2D"

    def __init__(self, mode: str = "S") -> None:
        super().__init__()
        self.mod = _by_layout(self._kind, mode)


class _Pooling(_UnaryOperator):
    def __init__(self, mode: str = "S", pool: str = "sum") -> None:
        super().__init__()
        self.mod = _by_layout(self._kind, mode, pool)


class OpPoolingSubg2D(_Pooling):
    """pool nodes within each subgraph"""
    _kind = "OpPoolingSubg2D"


class OpPoolingSubg3D(_Pooling):
    """pool the last tuple dim of 3-D representations"""
    _kind = "OpPoolingSubg3D"


class OpPoolingCrossSubg2D(_Pooling):
    """pool the same node over all subgraphs"""
    _kind = "OpPoolingCrossSubg2D"


class _Unpooling(Module):
    _kind = ""

    def __init__(self, mode: str = "S") -> None:
        super().__init__()
        self.mod = _by_layout(self._kind, mode)

    def forward(self, X, tarX: Rep) -> Rep:
        return self.mod.forward(X, tarX)


class OpUnpoolingSubgNodes2D(_Unpooling):
    """node representations -> every subgraph"""
    _kind = "OpUnpoolingSubgNodes2D"


class OpUnpoolingRootNodes2D(_Unpooling):
    """root-node representations -> their subgraph"""
    _kind = "OpUnpoolingRootNodes2D"

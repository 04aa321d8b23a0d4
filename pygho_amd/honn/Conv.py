"""
Model layers built on the operators.  Mirror of ``pygho/honn/Conv.py`` (constructor and
``forward(A, X, datadict)`` signatures, sub-module names, wiring; file:line per class).  The dense MLPs
stay ``torch.nn``; every aggregation runs on the HIP kernels behind ``TensorOp``.

``SUNConv`` needs ``torch_geometric.nn.HeteroLinear`` in the reference (Conv.py:15, :345); torch_geometric is
not a dependency here, so ``HeteroLinear`` is restated below (bias-free per-type linear).  No reference
test pins that layer: parity is UNPINNED at that boundary.
"""
import math
from typing import Callable, Literal, Optional, Union

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import Module

from . import TensorOp
from .utils import MLP
from ..backend.MaTensor import MaskedTensor
from ..backend.SpTensor import SparseTensor

Rep = Union[SparseTensor, MaskedTensor]


class HeteroLinear(Module):
    """``out[i] = x[i] @ W[type[i]] (+ b[type[i]])``: one linear map per integer type."""

    def __init__(self, in_channels: int, out_channels: int, num_types: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.num_types = in_channels, out_channels, num_types
        self.weight = nn.Parameter(torch.empty(num_types, in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(num_types, out_channels)) if bias else None
        bound = 1.0 / math.sqrt(in_channels)
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x: Tensor, type_vec: Tensor) -> Tensor:
        out = x.new_zeros((x.shape[0], self.out_channels))
        for t in range(self.num_types):
            sel = (type_vec == t).unsqueeze(-1).to(x.dtype)
            y = x @ self.weight[t].to(x.dtype)
            if self.bias is not None:
                y = y + self.bias[t].to(x.dtype)
            out = out + sel * y
        return out


class NGNNConv(Module):
    """nested GNN layer: MLP on every tuple, then message passing inside each subgraph
    (reference Conv.py:20-58)."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", mode: Literal["SD", "DD", "SS"] = "SS",
                 mlp: dict = {}, optuplefeat: str = "X", opadj: str = "A", message_func: Optional[Callable] = None):
        super().__init__()
        self.aggr = TensorOp.OpMessagePassingOnSubg2D(mode, aggr, optuplefeat, opadj, message_func)
        self.lin = MLP(indim, outdim, **mlp)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        tX = X.tuplewiseapply(self.lin)
        return self.aggr.forward(A, tX, datadict, tX)


class SSWLConv(Module):
    """subgraph-WL layer: in-subgraph and cross-subgraph aggregation, concatenated (reference Conv.py:62-103)."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", mode: Literal["SD", "DD", "SS"] = "SS",
                 mlp: dict = {}, optuplefeat: str = "X", opadj: str = "A"):
        super().__init__()
        self.aggr1 = TensorOp.OpMessagePassingOnSubg2D(mode, aggr, optuplefeat, opadj)
        self.aggr2 = TensorOp.OpMessagePassingCrossSubg2D(mode, aggr, optuplefeat, opadj)
        self.lin = MLP(3 * indim, outdim, **mlp)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        X1 = self.aggr1.forward(A, X, datadict, X)
        X2 = self.aggr2.forward(A, X, datadict, X)
        return X.catvalue([X1, X2], True).tuplewiseapply(self.lin)


class I2Conv(Module):
    """I2-GNN layer on 3-tuples (reference Conv.py:107-147)."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", mode: Literal["SD", "DD", "SS"] = "SS",
                 mlp: dict = {}, optuplefeat: str = "X", opadj: str = "A"):
        super().__init__()
        self.aggr = TensorOp.OpMessagePassingOnSubg3D(mode, aggr, optuplefeat, opadj)
        self.lin = MLP(indim, outdim, **mlp)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        tX = X.tuplewiseapply(self.lin)
        return self.aggr.forward(A, tX, datadict, tX)


class DSSGNNConv(Module):
    """DSS-GNN layer: subgraph aggregation plus a global (cross-subgraph pooled) aggregation
    (reference Conv.py:151-196)."""

    def __init__(self, indim: int, outdim: int, aggr_subg: str = "sum", aggr_global: str = "sum", pool: str = "mean",
                 mode: Literal["SD", "DD", "SS"] = "SS", mlp: dict = {}, optuplefeat: str = "X", opadj: str = "A"):
        super().__init__()
        self.aggr_subg = TensorOp.OpMessagePassingOnSubg2D(mode, aggr_subg, optuplefeat, opadj)
        self.pool2global = TensorOp.OpPoolingCrossSubg2D(mode[1], pool)
        self.aggr_global = TensorOp.OpNodeMessagePassing(mode, aggr_global)
        self.unpooling2subg = TensorOp.OpUnpoolingRootNodes2D(mode[1])
        self.lin = MLP(2 * indim, outdim, **mlp)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        X1 = self.unpooling2subg.forward(self.aggr_global.forward(A, self.pool2global.forward(X)), X)
        X2 = self.aggr_subg.forward(A, X, datadict, X)
        return X2.catvalue(X1, True).tuplewiseapply(self.lin)


class PPGNConv(Module):
    """PPGN layer: product of two transformed copies of X (reference Conv.py:200-232)."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", mode: Literal["DD", "SS"] = "SS", mlp: dict = {},
                 optuplefeat: str = "X"):
        super().__init__()
        self.op = TensorOp.Op2FWL(mode, aggr, optuplefeat)
        self.lin1 = MLP(indim, outdim, **mlp)
        self.lin2 = MLP(indim, outdim, **mlp)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        return self.op.forward(X.tuplewiseapply(self.lin1), X.tuplewiseapply(self.lin2), datadict, X)


class GNNAKConv(Module):
    """GNN-AK layer (reference Conv.py:236-297)."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", pool: str = "mean",
                 mode: Literal["SD", "DD", "SS"] = "SS", mlp0: dict = {}, mlp1: dict = {}, ctx: bool = True,
                 optuplefeat: str = "X", opadj: str = "A"):
        super().__init__()
        self.lin0 = MLP(indim, indim, **mlp0)
        self.aggr = TensorOp.OpMessagePassingOnSubg2D(mode, aggr, optuplefeat, opadj)
        self.diag = TensorOp.OpDiag2D(mode[1])
        self.pool2subg = TensorOp.OpPoolingSubg2D(mode[1], pool)
        self.unpool4subg = TensorOp.OpUnpoolingSubgNodes2D(mode[1])
        self.ctx = ctx
        if ctx:
            self.pool2node = TensorOp.OpPoolingCrossSubg2D(mode[1], pool)
            self.unpool4rootnode = TensorOp.OpUnpoolingRootNodes2D(mode[1])
        self.lin = MLP(3 * indim if ctx else 2 * indim, outdim, **mlp1)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        X = self.aggr.forward(A, X.tuplewiseapply(self.lin0), datadict, X)
        X1 = self.unpool4subg.forward(self.diag.forward(X), X)
        X2 = self.unpool4subg.forward(self.pool2subg.forward(X), X)
        if self.ctx:
            X3 = self.unpool4rootnode.forward(self.pool2node.forward(X), X)
            return X2.catvalue([X1, X3], True).tuplewiseapply(self.lin)
        return X2.catvalue(X1, True).tuplewiseapply(self.lin)


class SUNConv(Module):
    """SUN layer: seven tuple-wise views concatenated, a per-(diagonal / off-diagonal) linear map, an MLP
    (reference Conv.py:301-362)."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", pool: str = "mean",
                 mode: Literal["SD", "DD", "SS"] = "SS", mlp0: dict = {}, mlp1: dict = {}, optuplefeat: str = "X",
                 opadj: str = "A"):
        super().__init__()
        self.lin0 = MLP(indim, indim, **mlp0)
        self.aggr = TensorOp.OpMessagePassingOnSubg2D(mode, aggr, optuplefeat, opadj)
        self.diag = TensorOp.OpDiag2D(mode[1])
        self.pool2subg = TensorOp.OpPoolingSubg2D(mode[1], pool)
        self.unpool4subg = TensorOp.OpUnpoolingSubgNodes2D(mode[1])
        self.pool2node = TensorOp.OpPoolingCrossSubg2D(mode[1], pool)
        self.unpool4rootnode = TensorOp.OpUnpoolingRootNodes2D(mode[1])
        self.lin1_0 = HeteroLinear(7 * indim, indim, 2, False)
        self.lin1_1 = MLP(indim, outdim, **mlp1)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        X4 = self.aggr.forward(A, X.tuplewiseapply(self.lin0), datadict, X)
        Xdiag = self.diag.forward(X)
        X2 = self.unpool4subg.forward(Xdiag, X)
        X3 = self.unpool4rootnode.forward(Xdiag, X)
        X5 = self.unpool4rootnode.forward(self.pool2node(X), X)
        X6 = self.unpool4subg.forward(self.pool2subg(X), X)
        X7 = self.unpool4rootnode.forward(self.pool2node(X4), X)
        X = X.catvalue([X2, X3, X4, X5, X6, X7], True)
        X = X.diagonalapply(
            lambda val, ind: self.lin1_0(val.flatten(0, -2), ind.flatten()).unflatten(0, val.shape[0:-1]))
        return X.tuplewiseapply(self.lin1_1)

"""
Model layers built on the operators.  Mirror of ``pygho/honn/Conv.py`` (constructor and
``forward(A, X, datadict)`` signatures, sub-module names, wiring; file:line per class).  The dense MLPs
stay ``torch.nn``; every aggregation runs on the HIP kernels behind ``TensorOp``.

``SUNConv`` needs ``torch_geometric.nn.HeteroLinear`` in the reference (Conv.py:15, :345); torch_geometric is
not a dependency here, so ``HeteroLinear`` is restated below with torch_geometric 2.3.0's signature (per-type affine
map, bias on by default).  ``SUNConv`` is pinned to the reference's own forward wiring run around a labelled stand-in
for that one class (tests/golden/make_golden.py ``gen_sun`` -> sun.npz): pinned modulo the stand-in's arithmetic.
"""
import math
from typing import Callable, Literal, Optional, Union

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import Module

from . import TensorOp
from .SpOperator import KEYSEP, OpMessagePassing
from .utils import MLP, _SplitKLinearFn
from .. import _ops
from ..backend.MaTensor import MaskedTensor
from ..backend.SpTensor import SparseTensor, indicehash

Rep = Union[SparseTensor, MaskedTensor]


class HeteroLinear(Module):
    """``out[i] = x[i] @ W[type[i]] + b[type[i]]``: one affine map per integer type.  Constructor signature, parameter
    names and shapes follow ``torch_geometric.nn.HeteroLinear`` of torch_geometric 2.3.0 (the reference's pin,
    requirements.txt:7): ``(in_channels, out_channels, num_types, is_sorted=False, **kwargs)`` -- the reference's call
    ``HeteroLinear(7 * indim, indim, 2, False)`` (Conv.py:345) therefore sets ``is_sorted``; ``bias`` is a keyword
    (default True) -- with ``weight (num_types, in, out)`` and ``bias (num_types, out)``."""

    def __init__(self, in_channels: int, out_channels: int, num_types: int, is_sorted: bool = False, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.num_types = in_channels, out_channels, num_types
        self.is_sorted = is_sorted              # a hint that type_vec is sorted; the result does not depend on it
        self.weight = nn.Parameter(torch.empty(num_types, in_channels, out_channels))
        if kwargs.get("bias", True):
            self.bias = nn.Parameter(torch.empty(num_types, out_channels))
        else:
            self.register_parameter("bias", None)
        bound = 1.0 / math.sqrt(in_channels)
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x: Tensor, type_vec: Tensor) -> Tensor:
        """one GEMM with the most frequent type's weights over all rows, then the rows of every other type are gathered,
        multiplied with their own weights and written over the result (SUNConv: ~n of n^2 tuples are diagonal) -- instead of
        one full GEMM + mask-multiply + add per type."""
        w = self.weight.to(x.dtype)
        counts = torch.bincount(type_vec.reshape(-1), minlength=self.num_types)
        major = int(counts.argmax())
        out = x @ w[major]
        if self.bias is not None:
            out = out + self.bias[major].to(x.dtype)
        for t in range(self.num_types):
            if t == major or int(counts[t]) == 0:
                continue
            rows = torch.nonzero(type_vec.reshape(-1) == t).reshape(-1)
            y = x.index_select(0, rows) @ w[t]
            if self.bias is not None:
                y = y + self.bias[t].to(x.dtype)
            out = out.index_copy(0, rows, y)
        return out


def _views(mode: str, pool: str):
    """the parameter-free pooling / broadcast operators a layer needs for representation layout mode[1]"""
    layout = mode[1]
    return {
        "diag": TensorOp.OpDiag2D(layout),
        "pool_subg": TensorOp.OpPoolingSubg2D(layout, pool),          # over the nodes of a subgraph
        "pool_node": TensorOp.OpPoolingCrossSubg2D(layout, pool),     # over the subgraphs a node appears in
        "to_subg_nodes": TensorOp.OpUnpoolingSubgNodes2D(layout),
        "to_root": TensorOp.OpUnpoolingRootNodes2D(layout),
    }


GRAD_CHAIN_KEY = "_pygho_grad_chain"      # datadict key a model loop sets to {} for one forward pass (see _residual_update)


ADJ_LOOKUP_KEY = "A_lookup"      # datadict key: (table, flat index, the values tensor it describes, the table's master parameter or None)


def _residual_update(layer, A: Rep, X: Rep, datadict: dict, adj_lookup=None) -> Rep:
    """X + aggr(lin(X), A) for the layers whose update is `tuple-wise MLP, then X A inside the subgraph`.  Fused
    path (sparse X and A on the device, single-block MLP with square weight, sum / mean, precomputed acd):
    GEMM -> BatchNorm+act kernels -> aggregation kernel with the residual row added in its epilogue, one
    autograd node (``_ops.tuple_block``)."""
    op = getattr(layer.aggr, "mod", None)
    block = layer.lin.single_block() if isinstance(layer.lin, MLP) else None
    acd = None if op is None or datadict is None else datadict.get(getattr(op, "precomputekey", "") + KEYSEP + "acd")
    ok = (isinstance(X, SparseTensor) and isinstance(A, SparseTensor) and isinstance(op, OpMessagePassing)
          and not op.use_mpnn and op.aggr in ("sum", "mean") and block is not None and acd is not None
          and X.values is not None and A.values is not None and X.values.is_cuda and X.values.dim() == 2
          and A.values.dim() == 2 and A.values.shape[1] == X.values.shape[1] and A.values.dtype == X.values.dtype
          and block[0].in_features == block[0].out_features == X.values.shape[1]
          and _ops.bn_act_supported_shape(X.nnz, X.values.shape[1], X.values.dtype))
    if not ok:
        return X.add(layer.forward(A, X, datadict), True)
    lin, bn, act = block
    plan = _ops.message_plan(acd, X.nnz, X.nnz, A.nnz)
    # adjacency values that are an embedding lookup (a handful of distinct rows) are READ through the table.  The provenance is an
    # EXPLICIT argument -- `adj_lookup=(table, index)` with A.values == table[index] --, or the datadict entry the input encoder leaves
    # (ADJ_LOOKUP_KEY); the attribute `IndexEmbedding` hangs on its output is the legacy route for callers that pass neither
    lookup = None
    if _ops.USE_ADJ_TABLE:
        lookup = adj_lookup
        if lookup is None and datadict is not None:
            ent = datadict.get(ADJ_LOOKUP_KEY)           # (table, index, the values tensor it describes)
            if ent is not None and ent[2] is A.values:
                lookup = ent
        if lookup is None:
            lookup = getattr(A.values, "_pygho_lookup", None)
    # (table, index[, .., the table's master parameter]): with the master the block may return the TABLE's gradient directly
    master = None
    if lookup is not None:
        master = lookup[3] if len(lookup) > 3 else None
        lookup = (lookup[0], lookup[1])
    if lookup is not None and not (lookup[0].dim() == 2 and lookup[0].dtype == X.values.dtype and lookup[1].numel() == A.nnz
                                   and lookup[0].shape[1] == X.values.shape[1]):
        lookup = None
    # the layers of a model share A: each block hands A's values on to the next one as an autograd OUTPUT (same storage), so that
    # their gradient travels back through the blocks and is summed in the aggregation epilogues (`_ops._TupleBlock`).  Opt-in by
    # the model loop: it puts an empty dict under GRAD_CHAIN_KEY into the datadict of ONE forward pass (the links belong to that
    # pass's autograd graph and must not outlive it); layers called without it do not chain
    holder = datadict.get(GRAD_CHAIN_KEY) if _ops.USE_GRAD_CHAIN else None
    rhs, chain = A.values, holder is not None and torch.is_grad_enabled() and A.values.requires_grad
    if chain:
        link = holder.get(id(A.values))
        if link is not None and link[0] is A.values and link[1] == A.values._version:
            rhs = link[2]
    with torch.autocast("cuda", enabled=False):
        vals = _ops.tuple_block(X.values, lin, bn, act, rhs=rhs, plan=plan, aggr=op.aggr, residual=True, rhs_lookup=lookup, chain=chain,
                                rhs_master=master if lookup is not None else None)
    if chain:
        vals, nxt = vals
        holder[id(A.values)] = (A.values, A.values._version, nxt)
    return X.tuplewiseapply(lambda _: vals)


def _cat_apply(first: Rep, others, mlp, residual=False) -> Rep:
    """``first.catvalue(others, True).tuplewiseapply(mlp)`` (reference Conv.py:98-103, 190-196).  Sparse representations on
    the device whose MLP is one Linear -> BatchNorm -> act block over equally wide inputs skip the concatenation
    (``_ops.concat_block``: chained streaming GEMMs, one backward pass per input).  `residual`: True = the result + `first`,
    a representation R of the same pattern = R + the result (``R.add(.., True)``, the model loop's residual connection)."""
    block = mlp.single_block() if isinstance(mlp, MLP) else None
    reps = [first] + list(others)
    res_rep = residual if isinstance(residual, (SparseTensor, MaskedTensor)) else None
    if residual is True:
        res_rep, residual = first, first       # "+ first": the same rule as for a separate residual representation
    if res_rep is not None:
        # the sum is formed inside the block only when R already has the block's compute dtype: an f32 residual stream under
        # autocast stays f32 (R.add(out, True) promotes), exactly as in the model loop
        rows = res_rep.values if isinstance(res_rep, SparseTensor) else res_rep.raw
        lead = first.values if isinstance(first, SparseTensor) else first.raw
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else (None if lead is None else lead.dtype)
        if rows is None or lead is None or rows.dtype != cdt or type(res_rep) is not type(first):
            return res_rep.add(_cat_apply(first, others, mlp), True)
    if (block is not None and all(isinstance(r, SparseTensor) and r.values is not None for r in reps)
            and all(r.indices is first.indices or r.nnz == first.nnz for r in reps)):
        vals = [r.values for r in reps]
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else vals[0].dtype
        vals = [v if v.dtype == dt else v.to(dt) for v in vals]
        res_ok = res_rep is None or (isinstance(res_rep, SparseTensor) and res_rep.values is not None and res_rep.nnz == first.nnz
                                     and res_rep.values.shape == vals[0].shape)
        if res_ok and _ops.concat_block_supported(vals, block[0]):
            res_arg = residual if res_rep is None else (True if res_rep is first else res_rep.values)
            with torch.autocast("cuda", enabled=False):
                out = _ops.concat_block(vals, *block, residual=res_arg)      # residual: added inside the activation pass
            return (first if res_rep is None else res_rep).tuplewiseapply(lambda _: out)
    if block is not None and all(isinstance(r, MaskedTensor) for r in reps) and first.raw.is_cuda:
        # dense layout: the same block over the padded rows (catvalue + tuplewiseapply zero-fill the concatenation under
        # first's mask, reference MaTensor.py:264-270, 318-330; here each part is filled, parts a product already filled are not copied)
        parts = [(r if r.mask is first.mask else MaskedTensor(r.data, first.mask)).fill_masked(0.) for r in reps]
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else parts[0].dtype
        d = parts[0].shape[-1]
        vals = [(v if v.dtype == dt else v.to(dt)).reshape(-1, d) for v in parts]
        res_ok = res_rep is None or (isinstance(res_rep, MaskedTensor) and res_rep.raw.shape == parts[0].shape)
        if res_ok and all(v.shape == vals[0].shape for v in vals) and _ops.concat_block_supported(vals, block[0]):
            res_arg = residual
            if res_rep is not None:            # R.add(out, True) adds the raw data (MaTensor.py:251-262): masked entries stay don't-care
                res_arg = True if res_rep is first else res_rep.raw.reshape(-1, d)
            with torch.autocast("cuda", enabled=False):
                out = _ops.concat_block(vals, *block, residual=res_arg)
            return MaskedTensor(out.view(tuple(parts[0].shape[:-1]) + (out.shape[-1],)), first.mask if res_rep is None else res_rep.mask)
    out = first.catvalue(list(others), True).tuplewiseapply(mlp)
    if res_rep is not None:
        return res_rep.add(out, True)
    return first.add(out, True) if residual else out


class NGNNConv(Module):
    """nested GNN layer (reference Conv.py:20-58): tuple-wise MLP, then message passing inside each subgraph."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", mode: Literal["SD", "DD", "SS"] = "SS",
                 mlp: dict = {}, optuplefeat: str = "X", opadj: str = "A", message_func: Optional[Callable] = None):
        super().__init__()
        self.aggr = TensorOp.OpMessagePassingOnSubg2D(mode, aggr, optuplefeat, opadj, message_func)
        self.lin = MLP(indim, outdim, **mlp)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        H = X.tuplewiseapply(self.lin)
        return self.aggr.forward(A, H, datadict, H)

    def forward_residual(self, A: Rep, X: Rep, datadict: dict, adj_lookup=None) -> Rep:
        """``X.add(self.forward(A, X, datadict), True)`` (the model loop of example/minimal.py:76-79) as one fused
        block when the layer is sparse with a single Linear -> BatchNorm -> act update; otherwise exactly that.
        `adj_lookup=(table, index)`: A.values == table[index] (an embedding lookup of the edge feature) -- the block then reads the
        table instead of the gathered rows and, at width 128 in 16 bits, runs as the fused forward / fused backward kernels
        (`_ops.record_block_paths()` reports which path ran and why)."""
        return _residual_update(self, A, X, datadict, adj_lookup)


class SSWLConv(Module):
    """subgraph-WL layer (reference Conv.py:62-103): [X, X A (in-subgraph), A X (cross-subgraph)] -> MLP."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", mode: Literal["SD", "DD", "SS"] = "SS",
                 mlp: dict = {}, optuplefeat: str = "X", opadj: str = "A"):
        super().__init__()
        self.aggr1 = TensorOp.OpMessagePassingOnSubg2D(mode, aggr, optuplefeat, opadj)
        self.aggr2 = TensorOp.OpMessagePassingCrossSubg2D(mode, aggr, optuplefeat, opadj)
        self.lin = MLP(3 * indim, outdim, **mlp)

    def _fused(self, A: Rep, X: Rep, datadict: dict, residual: bool) -> Optional[Rep]:
        """the update as one autograd node (``_ops.sswl_block``) when the layer is sparse on the device with precomputed plans,
        sum / mean aggregation and a single Linear -> BatchNorm -> act block over equally wide operands; None otherwise."""
        block = self.lin.single_block() if isinstance(self.lin, MLP) else None
        ops = [getattr(op, "mod", op) for op in (self.aggr1, self.aggr2)]
        if not (_ops.USE_SSWL_BLOCK and block is not None and isinstance(X, SparseTensor) and isinstance(A, SparseTensor)
                and datadict is not None and all(isinstance(op, OpMessagePassing) and not op.use_mpnn for op in ops)
                and ops[0].aggr == ops[1].aggr and ops[0].aggr in ("sum", "mean")
                and X.values is not None and A.values is not None and X.values.is_cuda and X.values.dim() == 2
                and A.values.dim() == 2 and A.values.shape[1] == X.values.shape[1]):
            return None
        acds = [datadict.get(op.precomputekey + KEYSEP + "acd") for op in ops]
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else X.values.dtype
        if (any(acd is None for acd in acds) or X.values.dtype != dt or A.values.dtype != dt
                or not _ops.concat_block_supported([X.values] * 3, block[0])):
            return None
        plan1 = _ops.message_plan(acds[0], X.nnz, X.nnz, A.nnz)       # X A inside each subgraph
        plan2 = _ops.message_plan(acds[1], X.nnz, A.nnz, X.nnz)       # A X across subgraphs
        with torch.autocast("cuda", enabled=False):
            out = _ops.sswl_block(X.values, A.values, plan1, plan2, ops[0].aggr, *block, residual=residual)
        return X.tuplewiseapply(lambda _: out)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        fused = self._fused(A, X, datadict, False)
        if fused is not None:
            return fused
        neighbours = [op.forward(A, X, datadict, X) for op in (self.aggr1, self.aggr2)]
        return _cat_apply(X, neighbours, self.lin)

    def forward_residual(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        """``X.add(self.forward(A, X, datadict), True)`` (the model loop of example/zinc.py:287-290) with the residual row added
        inside the block's activation pass and its gradient inside the first backward GEMM's epilogue."""
        fused = self._fused(A, X, datadict, True)
        if fused is not None:
            return fused
        neighbours = [op.forward(A, X, datadict, X) for op in (self.aggr1, self.aggr2)]
        return _cat_apply(X, neighbours, self.lin, residual=True)


class I2Conv(Module):
    """I2-GNN layer on 3-tuples (reference Conv.py:107-147)."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", mode: Literal["SD", "DD", "SS"] = "SS",
                 mlp: dict = {}, optuplefeat: str = "X", opadj: str = "A"):
        super().__init__()
        self.aggr = TensorOp.OpMessagePassingOnSubg3D(mode, aggr, optuplefeat, opadj)
        self.lin = MLP(indim, outdim, **mlp)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        H = X.tuplewiseapply(self.lin)
        return self.aggr.forward(A, H, datadict, H)

    def forward_residual(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        """``X.add(self.forward(A, X, datadict), True)``, fused when possible (see NGNNConv.forward_residual)."""
        return _residual_update(self, A, X, datadict)


class DSSGNNConv(Module):
    """DSS-GNN layer (reference Conv.py:151-196): in-subgraph aggregation next to a global aggregation of the
    cross-subgraph pooled node features, broadcast back to the roots."""

    def __init__(self, indim: int, outdim: int, aggr_subg: str = "sum", aggr_global: str = "sum", pool: str = "mean",
                 mode: Literal["SD", "DD", "SS"] = "SS", mlp: dict = {}, optuplefeat: str = "X", opadj: str = "A"):
        super().__init__()
        self.aggr_subg = TensorOp.OpMessagePassingOnSubg2D(mode, aggr_subg, optuplefeat, opadj)
        self.pool2global = TensorOp.OpPoolingCrossSubg2D(mode[1], pool)
        self.aggr_global = TensorOp.OpNodeMessagePassing(mode, aggr_global)
        self.unpooling2subg = TensorOp.OpUnpoolingRootNodes2D(mode[1])
        self.lin = MLP(2 * indim, outdim, **mlp)

    def forward(self, A: Rep, X: Rep, datadict: dict, residual=False) -> Rep:
        node_level = self.aggr_global.forward(A, self.pool2global.forward(X))
        shared = self.unpooling2subg.forward(node_level, X)
        local = self.aggr_subg.forward(A, X, datadict, X)
        return _cat_apply(local, [shared], self.lin, residual=residual)

    def forward_residual(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        """``X.add(self.forward(A, X, datadict), True)`` (the model loop of example/zinc.py:287-290), X added inside the block's
        activation pass when the fused block applies."""
        return self.forward(A, X, datadict, residual=X)


class PPGNConv(Module):
    """PPGN layer (reference Conv.py:200-232): product of two transformed copies of X."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", mode: Literal["DD", "SS"] = "SS", mlp: dict = {},
                 optuplefeat: str = "X"):
        super().__init__()
        self.op = TensorOp.Op2FWL(mode, aggr, optuplefeat)
        self.lin1 = MLP(indim, outdim, **mlp)
        self.lin2 = MLP(indim, outdim, **mlp)

    def forward(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        left, right = (X.tuplewiseapply(f) for f in (self.lin1, self.lin2))
        return self.op.forward(left, right, datadict, X)

    def forward_residual(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        """``X.add(self.forward(A, X, datadict), True)`` (the model loop of example/zinc.py:287-290); on the sparse layout with a
        precomputed plan and sum / mean the residual row is added in the product kernel's epilogue."""
        op = getattr(self.op, "mod", None)
        acd = None if datadict is None or not isinstance(op, OpMessagePassing) else datadict.get(op.precomputekey + KEYSEP + "acd")
        if not (isinstance(X, SparseTensor) and acd is not None and not op.use_mpnn and op.aggr in ("sum", "mean")
                and X.values is not None and X.values.is_cuda):
            return X.add(self.forward(A, X, datadict), True)
        left, right = (X.tuplewiseapply(f) for f in (self.lin1, self.lin2))
        if not (left.values.dtype == right.values.dtype == X.values.dtype and left.values.shape == right.values.shape == X.values.shape):
            return X.add(self.op.forward(left, right, datadict, X), True)         # e.g. an f32 residual stream under autocast
        vals = _ops.message_reduce(left.values, right.values, acd, X.nnz, X.nnz, X.nnz, op.aggr, addend=X.values)
        return X.tuplewiseapply(lambda _: vals)


class GNNAKConv(Module):
    """GNN-AK layer (reference Conv.py:236-297): aggregate, then concatenate the subgraph-pooled view, the
    centroid (diagonal) view and optionally the context (cross-subgraph pooled) view."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", pool: str = "mean",
                 mode: Literal["SD", "DD", "SS"] = "SS", mlp0: dict = {}, mlp1: dict = {}, ctx: bool = True,
                 optuplefeat: str = "X", opadj: str = "A"):
        super().__init__()
        v = _views(mode, pool)
        self._pool = pool
        self.lin0 = MLP(indim, indim, **mlp0)
        self.aggr = TensorOp.OpMessagePassingOnSubg2D(mode, aggr, optuplefeat, opadj)
        self.diag, self.pool2subg, self.unpool4subg = v["diag"], v["pool_subg"], v["to_subg_nodes"]
        self.ctx = ctx
        if ctx:
            self.pool2node, self.unpool4rootnode = v["pool_node"], v["to_root"]
        self.lin = MLP((3 if ctx else 2) * indim, outdim, **mlp1)

    def _forward_node_level(self, H: SparseTensor, block, residual_rows: Optional[Tensor]) -> Rep:
        """the second half of the layer with the linear map pulled through the broadcasts (sparse layout).  All three concatenated
        views are NODE-level tensors unpooled onto the tuples -- pooled[i], centroid[i], context[j] -- and unpool(z) W = unpool(z W):
        the (3 d -> d) map runs on (n, d) tensors, the tuple level sees one broadcast-add pass, BatchNorm statistics, and the
        activation pass (with the layer's residual row).  The three (nnz, d) views, their (nnz, 3 d) concatenation (or the chained
        GEMMs that replace it) and the three tuple-level backward GEMMs never exist; the views' gradients return to H in one pass."""
        lin, bn, act = block
        hv = H.values
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else hv.dtype
        hv = hv if hv.dtype == cdt else hv.to(cdt)
        d, n = hv.shape[1], H.shape[0]
        ri, ci = H._row(0), H._row(1)
        cache = H._cache()
        if "sun_views" not in cache:
            diag_idx = torch.arange(n, device=ri.device)
            pos = _ops.sorted_match(H._hash(), _ops.hash_pack(diag_idx.reshape(1, -1).expand(2, -1).contiguous(), validate=n >= (1 << 31)))
            cnt = lambda r: torch.bincount(r, minlength=n).clamp_min(1).unsqueeze(-1)
            cache["sun_views"] = (pos, cnt(ri), cnt(ci))
        pos, cnt_r, cnt_c = cache["sun_views"]
        with torch.autocast("cuda", enabled=False):
            dg, s_r, s_c = _ops.sparse_pair_views(hv, ri, ci, pos, n, True)      # rows (i, i); sum over j -> [i]; sum over i -> [j]
            if self._pool == "mean":
                s_r, s_c = s_r / cnt_r.to(cdt), s_c / cnt_c.to(cdt)
            W = _ops.cast_param(lin.weight, cdt)                                  # (out, 3 d) or (out, 2 d): [pooled | centroid | context]
            b = None if lin.bias is None else _ops.cast_param(lin.bias, cdt)

            def node_lin(x, w, bias):            # x @ w^T (+ bias); tall inputs take the split-K weight gradient (honn/utils.py)
                if x.shape[0] >= 8192 or (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16)):
                    return _SplitKLinearFn.apply(x.contiguous(), w, bias, True)      # (16-bit rows: the tiled weight-gradient kernel at any height)
                return torch.nn.functional.linear(x, w, bias)
            u = node_lin(torch.cat((s_r, dg), dim=-1), W[:, :2 * d], b)           # indexed by i: pooled and centroid views
            v = node_lin(s_c, W[:, 2 * d:3 * d], None) if self.ctx else None      # indexed by j: context view
            pre = _ops.sparse_pair_broadcast(u, v, ri, ci, n)
            out = _ops.batch_norm_act(pre, bn, act, residual=residual_rows)
        return H.tuplewiseapply(lambda _: out)

    def forward(self, A: Rep, X: Rep, datadict: dict, residual=False) -> Rep:
        H = self.aggr.forward(A, X.tuplewiseapply(self.lin0), datadict, X)
        block = self.lin.single_block() if isinstance(self.lin, MLP) else None
        # the node-level path computes in the autocast dtype (u, v and the gathered rows of H are cast to it): every width check is
        # made on THAT element size (an f32 H under bf16 autocast has 2-byte rows)
        cdt = None
        if isinstance(H, SparseTensor) and H.values is not None:
            cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else H.values.dtype
        if (_ops.USE_NODE_LEVEL_LINEAR and block is not None and self._pool in ("sum", "mean") and isinstance(H, SparseTensor)
                and H.sparse_dim == 2 and H.values is not None and H.values.is_cuda and H.values.dim() == 2
                and H.values.is_floating_point() and H.shape[0] == H.shape[1] and _ops.pair_gather_supported(H.values, dtype=cdt)
                and block[0].in_features == (3 if self.ctx else 2) * H.values.shape[1]
                # the pair kernels run on rows of the block's OUTPUT width (u, v): those must be whole 16-byte pieces of at most 4 KB too
                and _ops.pair_gather_supported(H.values, width=block[0].out_features, dtype=cdt)
                and _ops.bn_act_supported_shape(H.nnz, block[0].out_features, cdt)):
            res_rows, outer = None, None
            if residual is not False and residual is not None:
                rep = X if residual is True else residual
                if (isinstance(rep, SparseTensor) and rep.values is not None and rep.values.dtype == cdt and rep.nnz == H.nnz
                        and rep.values.shape[1] == block[0].out_features):
                    res_rows = rep.values
                else:
                    outer = rep                                   # e.g. an f32 residual stream under autocast: added outside, as the model loop does
            out = self._forward_node_level(H, block, res_rows)
            return outer.add(out, True) if outer is not None else out
        centroid = self.unpool4subg.forward(self.diag.forward(H), H)
        pooled = self.unpool4subg.forward(self.pool2subg.forward(H), H)
        extra = [centroid]
        if self.ctx:
            extra.append(self.unpool4rootnode.forward(self.pool2node.forward(H), H))
        return _cat_apply(pooled, extra, self.lin, residual=residual)

    def forward_residual(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        """``X.add(self.forward(A, X, datadict), True)``, X added inside the last block's activation pass when it is fused."""
        return self.forward(A, X, datadict, residual=X)


class SUNConv(Module):
    """SUN layer (reference Conv.py:301-362): seven tuple-wise views of X concatenated, one linear map for
    diagonal and one for off-diagonal tuples (``HeteroLinear``), then an MLP."""

    def __init__(self, indim: int, outdim: int, aggr: str = "sum", pool: str = "mean",
                 mode: Literal["SD", "DD", "SS"] = "SS", mlp0: dict = {}, mlp1: dict = {}, optuplefeat: str = "X",
                 opadj: str = "A"):
        super().__init__()
        v = _views(mode, pool)
        self._pool = pool
        self.lin0 = MLP(indim, indim, **mlp0)
        self.aggr = TensorOp.OpMessagePassingOnSubg2D(mode, aggr, optuplefeat, opadj)
        self.diag, self.pool2subg, self.unpool4subg = v["diag"], v["pool_subg"], v["to_subg_nodes"]
        self.pool2node, self.unpool4rootnode = v["pool_node"], v["to_root"]
        self.lin1_0 = HeteroLinear(7 * indim, indim, 2, False)
        self.lin1_1 = MLP(indim, outdim, **mlp1)

    def forward_concat(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        """the reference's literal wiring (Conv.py:338-362): materialise the seven views, concatenate them to (.., 7 d) and
        apply the per-type linear map.  Kept as the parity reference of ``forward``."""
        to_nodes, to_root = self.unpool4subg.forward, self.unpool4rootnode.forward
        agg = self.aggr.forward(A, X.tuplewiseapply(self.lin0), datadict, X)
        centre = self.diag.forward(X)
        views = [to_nodes(centre, X), to_root(centre, X), agg, to_root(self.pool2node(X), X),
                 to_nodes(self.pool2subg(X), X), to_root(self.pool2node(agg), X)]     # X2 .. X7 of the reference
        stacked = X.catvalue(views, True)

        def per_type(val, is_diag):
            flat = self.lin1_0(val.flatten(0, -2), is_diag.flatten())
            return flat.unflatten(0, val.shape[0:-1])

        return stacked.diagonalapply(per_type).tuplewiseapply(self.lin1_1)

    def forward_residual(self, A: Rep, X: Rep, datadict: dict) -> Rep:
        """``X.add(self.forward(A, X, datadict), True)`` (the model loop of example/zinc.py:287-290); X is added inside the
        activation pass of the last MLP block when that block is fused."""
        return self.forward(A, X, datadict, residual=True)

    def forward(self, A: Rep, X: Rep, datadict: dict, residual: bool = False) -> Rep:
        """Same function with the linear map pulled through the broadcasts.  Five of the seven concatenated views are
        node-level tensors unpooled to tuple level, and ``unpool(v) W = unpool(v W)``: the (7 d -> d) map is applied block
        by block -- to X and agg at tuple level, to the five node-level tensors BEFORE they are broadcast -- and the
        diagonal tuples (their own weight set) are computed entirely at node level and selected in.  The (nnz, 7 d)
        concatenation (2.5 GB at b = 1024, n = 37, d = 128) and its 896-wide GEMMs never exist."""
        to_nodes, to_root = self.unpool4subg.forward, self.unpool4rootnode.forward
        d = self.lin1_0.in_channels // 7
        if self.lin1_0.num_types != 2:
            out = self.forward_concat(A, X, datadict)
            return X.add(out, True) if residual else out
        x_rows = X.values if isinstance(X, SparseTensor) else X.raw

        def tail(rep):                       # the last MLP, with the layer input as its residual row operand when asked for
            if not residual:
                return rep.tuplewiseapply(self.lin1_1)
            return rep.tuplewiseapply(lambda val: self.lin1_1(val, residual=x_rows))
        W = self.lin1_0.weight                                   # (2, 7 d, d): [off-diagonal, diagonal]
        blk = lambda t, v: W[t, v * d:(v + 1) * d]
        def mm(val, w):                      # val @ w; tall operands get the split-K weight gradient (honn/utils.py)
            if val.is_cuda and val.numel() // val.shape[-1] >= 8192:
                flat = val.reshape(-1, val.shape[-1])
                return _SplitKLinearFn.apply(flat, w.to(val.dtype).t(), None).reshape(val.shape[:-1] + (w.shape[1],))
            return val @ w.to(val.dtype)

        def lin(rep, w):                     # a bias-free linear map needs no fill of the masked entries (they stay don't-care)
            if isinstance(rep, MaskedTensor):
                return MaskedTensor(mm(rep.raw, w), rep.mask)
            return rep.tuplewiseapply(lambda val: mm(val, w)) if isinstance(rep, SparseTensor) else mm(rep, w)

        add = lambda a, b: a.add(b, True) if isinstance(a, (SparseTensor, MaskedTensor)) else a + b

        fused_ok = _ops.USE_PAIR_COMBINE and self._pool in ("sum", "mean")
        chained = {}
        if fused_ok and isinstance(self.lin0, MLP):
            # X feeds lin0 AND the recombination below: lin0's first block hands its input on as an autograd output, the recombination
            # reads that, and its gradient wrt the tuple rows is added inside lin0's input-gradient GEMM (no (nnz, d) accumulation)
            def lin0(v):
                y, chained["x"] = self.lin0(v, return_input=True)
                return y
        else:
            lin0 = self.lin0
        agg = self.aggr.forward(A, X.tuplewiseapply(lin0), datadict, X)
        if fused_ok:
            # compute dtype of the fused passes: the autocast dtype when autocast is on (operands are cast once), else X's
            cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
            as_c = lambda t: t if cdt is None or t.dtype == cdt or not t.is_floating_point() else t.to(cdt)
            xin = chained.get("x")
            if residual and xin is not None and xin.shape == x_rows.shape and xin.dtype == x_rows.dtype:
                x_rows = xin                 # the residual row operand of the last block too: its gradient joins lin0's GEMM epilogue
            if (isinstance(X, MaskedTensor) and isinstance(agg, MaskedTensor) and X.masked_dim == 3 and X.raw.is_floating_point()
                    and agg.raw.shape == X.raw.shape and (cdt is not None or agg.raw.dtype == X.raw.dtype)
                    and _ops.pair_combine_supported(as_c(X.raw))):
                Xc = MaskedTensor(xin, X.mask) if xin is not None and xin.shape == X.raw.shape else X
                return tail(self._recombine(Xc, agg, blk, True, as_c))
            if (isinstance(X, SparseTensor) and isinstance(agg, SparseTensor) and X.sparse_dim == 2 and X.values is not None
                    and agg.values is not None and X.values.is_floating_point() and agg.values.shape == X.values.shape
                    and (cdt is not None or agg.values.dtype == X.values.dtype) and agg.nnz == X.nnz and X.shape[0] == X.shape[1]
                    and _ops.pair_gather_supported(as_c(X.values))):
                Xc = X.tuplewiseapply(lambda _v: xin) if xin is not None and xin.shape == X.values.shape else X
                return tail(self._recombine(Xc, agg, blk, False, as_c))
        centre, n5, n6, n7 = self.diag.forward(X), self.pool2node(X), self.pool2subg(X), self.pool2node(agg)
        # concat order of the reference: [X, to_nodes(centre), to_root(centre), agg, to_root(n5), to_nodes(n6), to_root(n7)]
        off = add(lin(X, blk(0, 0)), lin(agg, blk(0, 3)))
        bias = self.lin1_0.bias                                  # (2, d) or None: [off-diagonal, diagonal]
        def nb(rep, t):                      # + b[t] on a node-level tensor (masked rows stay don't-care)
            if bias is None:
                return rep
            if isinstance(rep, MaskedTensor):
                return MaskedTensor(rep.raw + bias[t].to(rep.raw.dtype), rep.mask)
            return rep + bias[t].to(rep.dtype)
        off = add(off, to_nodes(nb(add(lin(centre, blk(0, 1)), lin(n6, blk(0, 5))), 0), X))
        off = add(off, to_root(add(add(lin(centre, blk(0, 2)), lin(n5, blk(0, 4))), lin(n7, blk(0, 6))), X))
        # diagonal tuples (i, i): every view reduces to a node-level tensor there
        dg = add(lin(centre, blk(1, 0) + blk(1, 1) + blk(1, 2)), lin(self.diag.forward(agg), blk(1, 3)))
        dg = nb(add(add(add(dg, lin(n5, blk(1, 4))), lin(n6, blk(1, 5))), lin(n7, blk(1, 6))), 1)
        dg_t = to_root(dg, X)
        if isinstance(off, MaskedTensor):
            eye = torch.eye(off.shape[1], off.shape[2], dtype=torch.bool, device=off.raw.device).reshape(1, off.shape[1], off.shape[2], 1)
            picked = MaskedTensor(torch.where(eye, dg_t.raw.to(off.raw.dtype), off.raw), off.mask)
        else:
            picked = off.diagonalapply(lambda val, is_diag: torch.where(is_diag.bool().unsqueeze(-1), dg_t.values.to(val.dtype), val))
        return tail(picked)

    def _recombine(self, X: Rep, agg: Rep, blk, dense: bool, as_c=lambda t: t) -> Rep:
        with torch.autocast("cuda", enabled=False):          # one compute dtype throughout (operands cast by `as_c`)
            return self._recombine_impl(X, agg, blk, dense, as_c)

    def _recombine_impl(self, X: Rep, agg: Rep, blk, dense: bool, as_c) -> Rep:
        """the same arithmetic with the tuple-level passes fused, on the padded (``dense``) or the sparse layout: the
        node-level views of X and agg (diagonal, pool over subgraphs, pool over nodes) come from one autograd node each
        (their gradients return to the tuple level in one pass: ``_ops.pair_views`` / ``_ops.sparse_pair_views``); the two
        tuple-level GEMMs, the three broadcasts, the three adds and the diagonal select are ``_ops.pair_linear_mix`` /
        ``_ops.sparse_pair_linear_mix`` (two GEMM launches + one pass)."""
        if dense:
            mask, amask = X.mask, agg.mask
            xv, av = as_c(X.raw), as_c(agg.raw)
            dt, d_ = xv.dtype, xv.shape[-1]

            def views(vals, m, subg=True):
                # `vals` comes back as an autograd output: the recombination below consumes THAT, so its gradient wrt the tuple rows
                # is the base of the views' one backward pass instead of a separate accumulation (a (b, n, n, d) add each)
                dg, s1, s2, vals = _ops.pair_views(vals, m, subg, True)   # rows (b,i,i); sum over dim 1 -> [b,j]; over dim 2 -> [b,i]
                if self._pool == "mean":
                    s1 = s1 / m.sum(1).clamp_min(1).unsqueeze(-1).to(dt)
                    s2 = s2 / m.sum(2).clamp_min(1).unsqueeze(-1).to(dt) if subg else None
                return dg, s1, s2, vals                             # (centre, pool2node, pool2subg)

            centre, n5, n6, xv = views(xv, mask)
            agg_dg, n7, _unused, av = views(av, amask, subg=False)   # pool2subg(agg) is not one of the seven views
        else:
            xv, av = as_c(X.values), as_c(agg.values)
            dt, d_ = xv.dtype, xv.shape[-1]
            n = X.shape[0]
            ri, ci = X._row(0), X._row(1)
            cache = X._cache()
            if "sun_views" not in cache:
                diag_idx = torch.arange(n, device=ri.device)
                pos = _ops.sorted_match(X._hash(), _ops.hash_pack(diag_idx.reshape(1, -1).expand(2, -1).contiguous(), validate=n >= (1 << 31)))
                cnt = lambda r: torch.bincount(r, minlength=n).clamp_min(1).unsqueeze(-1)
                cache["sun_views"] = (pos, cnt(ri), cnt(ci))
            pos, cnt_r, cnt_c = cache["sun_views"]

            def views(vals, subg=True):
                dg, s_r, s_c, vals = _ops.sparse_pair_views(vals, ri, ci, pos, n, subg, True)  # rows (i,i); sum over j -> [i]; over i -> [j]
                if self._pool == "mean":
                    s_r, s_c = (s_r / cnt_r.to(dt) if subg else None), s_c / cnt_c.to(dt)
                return dg, s_c, s_r, vals                           # (centre, pool2node, pool2subg; the values again, see above)

            centre, n5, n6, xv = views(xv)
            agg_dg, n7, _unused, av = views(av, subg=False)          # pool2subg(agg) is not one of the seven views
        w = lambda t, v: blk(t, v).to(dt)

        def node_lin(parts, blocks):
            """sum_k parts[k] @ blocks[k] as ONE product over the concatenated reduction dim: node-level rows are few, so
            the concatenation is cheap, and the weight gradient (reduction over all nodes into a d x d tile) runs on the
            split-K kernel once instead of k times on a single-tile library GEMM."""
            x = torch.cat(parts, dim=-1).reshape(-1, len(parts) * d_)
            wt = torch.cat(blocks, dim=0).to(dt)                    # (k d, d)
            if x.shape[0] >= 8192 or (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16)):
                y = _SplitKLinearFn.apply(x.contiguous(), wt.t(), None, True)
            else:
                y = x @ wt
            return y.reshape(parts[0].shape[:-1] + (wt.shape[1],))

        # unpooling along dim 1 repeats a node tensor over j (term indexed by i), along dim 0 over i (term indexed by j)
        # the per-type bias of HeteroLinear rides on the node-level terms: b[0] on the term broadcast to every off-diagonal
        # tuple, b[1] on the diagonal rows
        bias = self.lin1_0.bias
        u = node_lin([centre, n6], [blk(0, 1), blk(0, 5)])                                  # to_nodes(...)
        v = node_lin([centre, n5, n7], [blk(0, 2), blk(0, 4), blk(0, 6)])                   # to_root(...)
        dg = node_lin([centre, agg_dg, n5, n6, n7],
                      [blk(1, 0) + blk(1, 1) + blk(1, 2), blk(1, 3), blk(1, 4), blk(1, 5), blk(1, 6)])
        if bias is not None:
            u, dg = u + bias[0].to(dt), dg + bias[1].to(dt)
        if dense:
            out = _ops.pair_linear_mix(xv, av, w(0, 0), w(0, 3), u.contiguous(), v.contiguous(), dg.contiguous(), mask)
            return MaskedTensor(out, mask, 0.0, True)
        out = _ops.sparse_pair_linear_mix(xv, av, w(0, 0), w(0, 3), u.contiguous(), v.contiguous(), dg.contiguous(), ri, ci, pos, n)
        return X.tuplewiseapply(lambda _: out)

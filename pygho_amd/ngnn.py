"""
The end-to-end NGNN model of the reference's minimal example (example/minimal.py:22-85), restated on
the pygho_amd operators.  It defines the shapes the headline benchmark is quoted on (BASELINE.json
configs 1, 2, 4): InputEncoder (3 embeddings) -> tuple init -> 6 x [NGNNConv + residual] -> subgraph
mean pooling -> MLP -> per-graph sum -> prediction MLP.  175 489 parameters at hidden = 128.
"""
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import _ops
from .backend.SpTensor import SparseTensor
from .backend.utils import torch_scatter_reduce
from .honn.Conv import ADJ_LOOKUP_KEY, GRAD_CHAIN_KEY, NGNNConv
from .honn.TensorOp import OpPoolingSubg2D
from .honn.utils import MLP, Linear


_flat_index = _ops.flat_index      # the 1-D contiguous form of an integer feature tensor as a persistent object


class IndexEmbedding(nn.Embedding):
    """``nn.Embedding`` (same parameter / state_dict) whose lookup is the backend's row gather (K6) and whose
    backward is the hierarchical segment reduction instead of ATen's sort + index_put_(accumulate): with a
    handful of table rows and 10^6 lookups the stock backward was 35 % of the training step."""

    def __init__(self, num_embeddings: int, embedding_dim: int, out_dtype: Optional[torch.dtype] = None,
                 padding_idx: Optional[int] = None):
        super().__init__(num_embeddings, embedding_dim, padding_idx=padding_idx)
        self.out_dtype = out_dtype

    def forward(self, idx: Tensor) -> Tensor:
        if not idx.is_cuda:
            return super().forward(idx)
        if self.out_dtype is not None and self.out_dtype != self.weight.dtype and self.padding_idx is None and self.weight.requires_grad:
            # the table = the cast arena's copy of the parameter; the lookup's gradient returns to the PARAMETER in f32
            table = _ops.param_as(self.weight, self.out_dtype)
            out = _ops.gather_rows_master(self.weight, table, _flat_index(idx))
        else:
            table = self.weight if self.out_dtype is None else _ops.cast_param(self.weight, self.out_dtype)
            if self.padding_idx is not None:            # nn.Embedding semantics: that row reads as stored (zero) and gets no gradient
                keep = torch.ones((self.num_embeddings, 1), dtype=table.dtype, device=table.device)
                keep[self.padding_idx] = 0
                table = table * keep + (table * (1 - keep)).detach()
            out = _ops.gather_rows(table, _flat_index(idx))              # persistent index object: the gather plan is cached on it
        out = out.reshape(tuple(idx.shape) + (self.embedding_dim,))
        if idx.dim() == 1:
            # provenance for consumers that can index the (tiny, cache-resident) table themselves instead of streaming the
            # gathered rows: the fused layer block reads A's values this way (honn/Conv._residual_update)
            # (table, index, None, master): the master parameter lets a consumer return the TABLE's gradient directly
            out._pygho_lookup = (table, _flat_index(idx), None, self.weight if self.padding_idx is None else None)
        return out


class InputEncoderSp(nn.Module):
    """integer node / edge / tuple features -> hidden vectors (example/minimal.py:22-34)."""

    def __init__(self, hiddim: int, act_dtype: Optional[torch.dtype] = None) -> None:
        super().__init__()
        self.x_encoder = IndexEmbedding(32, hiddim, act_dtype)
        self.ea_encoder = IndexEmbedding(16, hiddim, act_dtype)
        self.tuplefeat_encoder = IndexEmbedding(16, hiddim, act_dtype)
        self.act_dtype = act_dtype

    def _cast(self, t: Tensor) -> Tensor:
        return t if self.act_dtype is None or t.dtype == self.act_dtype else t.to(self.act_dtype)

    def forward(self, datadict: dict, defer_tuplefeat: bool = False) -> dict:
        """`defer_tuplefeat`: leave X's integer feature in place and hand its embedding TABLE on as "X_table" -- the
        consumer (SpModel.tupleinit) performs the lookup inside its fused product kernel."""
        out = dict(datadict)
        out["x"] = self._cast(self.x_encoder(datadict["x"].flatten()))
        out["A"] = datadict["A"].tuplewiseapply(lambda v: self._cast(self.ea_encoder(v)))
        look = getattr(out["A"].values, "_pygho_lookup", None)
        if look is not None:
            # A's values are table[index]: said explicitly to the layers (honn.Conv.ADJ_LOOKUP_KEY) -- they read the table
            out[ADJ_LOOKUP_KEY] = (look[0], look[1], out["A"].values, look[3] if len(look) > 3 else None)
        if defer_tuplefeat:
            w = self.tuplefeat_encoder.weight                 # the table itself: its 16-bit copy comes from the cast arena
            if self.act_dtype is None or self.act_dtype == w.dtype:
                out["X_table"] = w
            else:
                # (table copy without an autograd node, the parameter beside it: the consumer returns the table's gradient to the
                # parameter in f32 -- `_ops.pair_product(val_master=)`)
                out["X_table"], out["X_table_master"] = _ops.param_as(w, self.act_dtype), w
        else:
            out["X"] = datadict["X"].tuplewiseapply(lambda v: self._cast(self.tuplefeat_encoder(v)))
        return out


class SpModel(nn.Module):
    """example/minimal.py:37-85 (sparse NGNN for graph regression)."""

    def __init__(self, num_tasks: int = 1, num_layer: int = 6, hiddim: int = 128, mlp: Optional[dict] = None,
                 act_dtype: Optional[torch.dtype] = None):
        super().__init__()
        mlp = dict(mlp or {"norm": "bn", "act": "silu", "dp": 0.0})
        self.lin_tupleinit0 = Linear(hiddim, hiddim)
        self.lin_tupleinit1 = Linear(hiddim, hiddim)
        self.npool = "sum"
        self.lpool = OpPoolingSubg2D("S", "mean")
        self.poolmlp = MLP(hiddim, hiddim, 1, tailact=True, **mlp)
        self.data_encoder = InputEncoderSp(hiddim, act_dtype)
        self.pred_lin = MLP(hiddim, num_tasks, 2, tailact=False, **mlp)
        conv_mlp = dict(mlp, numlayer=1, tailact=True)
        self.subggnns = nn.ModuleList([NGNNConv(hiddim, hiddim, "sum", "SS", conv_mlp) for _ in range(num_layer)])

    def tupleinit(self, X: SparseTensor, x: Tensor, table: Optional[Tensor] = None, table_master: Optional[Tensor] = None) -> SparseTensor:
        """X.values * lin0(x)[root] * lin1(x)[node] (example/minimal.py:62-67); with `table`, X.values are still the
        integer tuple features and the embedding lookup table[X.values] happens inside the same kernel."""
        left, right = self.lin_tupleinit0(x), self.lin_tupleinit1(x)
        if table is not None:
            feat = _flat_index(X.values)                                            # persistent object: plans are cached on it
            return X.tuplewiseapply(lambda _: _ops.pair_product(left, right, table, X._row(0), X._row(1), feat, val_master=table_master))
        val = X.values
        if val.is_cuda and val.dim() == 2 and left.dtype == right.dtype == val.dtype:
            return X.tuplewiseapply(lambda v: _ops.pair_product(left, right, v, X._row(0), X._row(1)))
        subgx0 = X.unpooling_fromdense1dim(0, left)
        subgx1 = X.unpooling_fromdense1dim(1, right)
        return X.tuplewiseapply(lambda v: subgx0.values * subgx1.values * v)

    def prepare(self, datadict: dict) -> None:
        """Build every index plan one training step on this batch will ask for (narrowed indices, transposed groupings,
        long-segment hierarchies, lookups), by running the model's index-consuming operators once, forward and backward, on
        width-8 dummy features: the plans depend on the index tensors only and are cached on them, so the real step finds
        them.  Meant to run on a side stream one batch ahead (``collate.BatchPrefetcher``): plan construction needs ~40 host
        synchronisations, which inside the step would each wait for the step's own launch queue."""
        X, A = datadict["X"], datadict["A"]
        dev = X.indices.device
        n, w = int(datadict["num_nodes"]), 8
        enc = self.data_encoder
        dt = enc.act_dtype or torch.float32
        z = lambda rows: torch.zeros((rows, w), dtype=dt, device=dev, requires_grad=True)
        outs = []
        flats = {}
        for name, idx, rows in (("x", datadict["x"].flatten(), enc.x_encoder.num_embeddings),
                                ("ea", A.values, enc.ea_encoder.num_embeddings),
                                ("tf", X.values, enc.tuplefeat_encoder.num_embeddings)):
            if idx is None or idx.dtype != torch.int64:
                continue
            flats[name] = _flat_index(idx)
            if name != "tf" or idx.numel() != X.nnz:
                outs.append(_ops.gather_rows(z(rows), flats[name]))
        if "tf" in flats and X.values.numel() == X.nnz:
            feat = _flat_index(X.values)
            outs.append(_ops.pair_product(z(n), z(n), z(enc.tuplefeat_encoder.num_embeddings), X._row(0), X._row(1), feat))
        else:
            outs.append(_ops.pair_product(z(n), z(n), z(X.nnz), X._row(0), X._row(1)))
        outs.append(_ops.scatter_reduce(z(X.nnz), X._row(0), n, "mean"))                      # lpool
        outs.append(_ops.scatter_reduce(z(n), datadict["batch"], int(datadict["num_graphs"]), self.npool))
        torch.autograd.backward([o.sum() for o in outs])
        for conv in self.subggnns:
            acd = datadict.get(conv.aggr.mod.precomputekey + "___acd")
            if acd is not None:
                plan = _ops.message_plan(acd, X.nnz, X.nnz, A.nnz)
                plan.by_c()
                plan.by_d()
                _ops.scatter_plan(plan)             # the by-edge gradient's scatter form: blocks / chunks / packed words (two host reads)
                if _ops.FUSED_FWD:
                    _ops.fused_plan(plan)           # the fused block forward's chunks (one host read)
                if "ea" in flats:
                    plan.lookup(flats["ea"])

    def forward(self, datadict: dict) -> Tensor:
        raw = datadict["X"]
        _ops.ensure_cast_arena(self, self.data_encoder.act_dtype)     # 16-bit parameter copies: one multi-tensor cast per step
        fuse = (_ops.USE_TABLE_PRODUCT and isinstance(raw, SparseTensor) and raw.values is not None and raw.values.is_cuda
                and raw.values.dtype == torch.int64 and raw.values.numel() == raw.nnz)
        datadict = self.data_encoder(datadict, defer_tuplefeat=fuse)        # a new dict per forward pass
        datadict[GRAD_CHAIN_KEY] = {}           # the layers share A: its gradient is summed inside their by-edge aggregations
        A, X, x = datadict["A"], datadict["X"], datadict["x"]
        with _ops.deferred_batch_counters():               # the BatchNorm step counters of the pass: one launch instead of one per layer
            X = self.tupleinit(X, x, datadict.get("X_table"), datadict.get("X_table_master"))
            for conv in self.subggnns:
                X = conv.forward_residual(A, X, datadict)  # == X.add(conv.forward(A, X, datadict), True), fused
            x = self.lpool(X)
            x = self.poolmlp(x)
        h_graph = torch_scatter_reduce(0, x, datadict["batch"], datadict["num_graphs"], self.npool)
        return self.pred_lin(h_graph)

"""
Padded-batch builders of the dense layout on the device: same names, arguments and results as the reference
(pygho/hodata/MaData.py:25-255), each ONE or two launches of ``pygho_pad_stack`` / ``pygho_dense_adj`` instead of
advanced-indexing gathers, ``cummin`` mask constructions and ``index_put_``.

The reference constructs the MaskedTensors of ``to_dense_x`` / ``to_dense_tuplefeat`` UNFILLED (padded slots hold the
clamped gather's neighbours); the raw arrays produced here are bit-identical to the reference's, padding included
(tests/golden/dense_collate.npz), and are wrapped unfilled as well -- the lazy fill of ``MaskedTensor`` zeroes them on
first use of ``.data``.
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional

import torch
from torch import BoolTensor, LongTensor, Tensor

from .. import _ops
from ..backend.MaTensor import MaskedTensor
from ..backend.SpTensor import SparseTensor


def to_dense_adj(edge_index: LongTensor, edge_batch: LongTensor, edge_attr: Optional[Tensor] = None,
                 max_num_nodes: Optional[int] = None, batch_size: Optional[int] = None,
                 filled_value: float = 0) -> MaskedTensor:
    """(b, n, n, *dense) adjacency with ``edge_attr`` (ones when absent) at the edges and ``filled_value`` elsewhere; mask =
    the edges.  ``edge_index`` holds graph-local, coalesced indices (reference MaData.py:25-72)."""
    if max_num_nodes is None:
        max_num_nodes = int(edge_index.max().item()) + 1
    if edge_attr is None:
        edge_attr = torch.ones(edge_batch.shape[0], device=edge_index.device)
    if batch_size is None:
        batch_size = int(torch.max(edge_batch).item()) + 1
    data, mask = _ops.dense_adj(edge_index, edge_batch, edge_attr, max_num_nodes, batch_size, filled_value)
    return MaskedTensor(data, mask, filled_value, True)


def to_sparse_adj(edge_index: LongTensor, edge_batch: LongTensor, edge_attr: Optional[Tensor] = None,
                  max_num_nodes: Optional[int] = None, batch_size: Optional[int] = None) -> SparseTensor:
    """(b, n, n, *dense) SparseTensor with indices [edge_batch; edge_index] (reference MaData.py:75-105)."""
    if max_num_nodes is None:
        max_num_nodes = int(edge_index.max().item()) + 1
    if batch_size is None:
        batch_size = int(torch.max(edge_batch).item()) + 1
    size = [batch_size, max_num_nodes, max_num_nodes] + list(edge_attr.size())[1:]
    return SparseTensor(torch.concatenate((edge_batch.unsqueeze(0), edge_index), dim=0), edge_attr, shape=size, is_coalesced=False)


def to_dense_x(nodeX: Tensor, Xptr: LongTensor, max_num_nodes: Optional[int] = None, batch_size: Optional[int] = None,
               filled_value: float = 0) -> MaskedTensor:
    """node features of a batch, (sum n_b, *dense) -> (b, n, *dense) with the node mask (reference MaData.py:108-147)."""
    if batch_size is None:
        batch_size = Xptr.shape[0] - 1
    counts = torch.diff(Xptr)
    if max_num_nodes is None:
        max_num_nodes = int(counts.max().item())
    data, mask = _ops.pad_stack(nodeX, Xptr, counts.reshape(-1, 1), [max_num_nodes])
    return MaskedTensor(data, mask, filled_value, False)


def to_dense_tuplefeat(tuplefeat: Tensor, tupleshape: LongTensor, tuplefeatptr: LongTensor,
                       max_tupleshape: Optional[LongTensor] = None, batch_size: Optional[int] = None,
                       feat2mask: Optional[Callable[[Tensor], BoolTensor]] = None) -> MaskedTensor:
    """tuple features of a batch, graph b a row-major grid of ``tupleshape[b]`` rows -> (b, n1, n2, .., *dense) with the mask of
    the real tuples, optionally intersected with ``feat2mask(padded features)`` (reference MaData.py:150-214)."""
    if batch_size is None:
        batch_size = tupleshape.shape[0]
    if max_tupleshape is None:
        max_tupleshape = torch.amax(tupleshape, dim=0)
    maxes = [int(v) for v in (max_tupleshape.tolist() if isinstance(max_tupleshape, Tensor) else max_tupleshape)]
    data, mask = _ops.pad_stack(tuplefeat, tuplefeatptr, tupleshape, maxes)
    if feat2mask is not None:
        mask = torch.logical_and(feat2mask(data), mask)
    return MaskedTensor(data, mask, 0, False)


def _get(batch: Any, key: str):
    return batch[key] if isinstance(batch, dict) else getattr(batch, key)


def _set(batch: Any, key: str, value) -> None:
    if isinstance(batch, dict):
        batch[key] = value
    else:
        setattr(batch, key, value)


def batch2dense(batch: Any, batch_size: Optional[int] = None, max_num_nodes: Optional[int] = None, denseadj: bool = False,
                keys: List[str] = [""]):
    """pad a collated batch (attribute object or dict with ``x, ptr, edge_index, edge_index_batch, edge_attr`` and, per key,
    ``tuplefeat{key}, tupleshape{key}, tuplefeat{key}_ptr``) into the dense layout: ``x``, ``A``, ``X{key}`` become
    Masked / Sparse tensors (reference MaData.py:217-255)."""
    x = to_dense_x(_get(batch, "x"), _get(batch, "ptr"), max_num_nodes, batch_size)
    _set(batch, "x", x)
    batch_size, max_num_nodes = x.shape[0], x.shape[1]
    build = to_dense_adj if denseadj else to_sparse_adj
    _set(batch, "A", build(_get(batch, "edge_index"), _get(batch, "edge_index_batch"), _get(batch, "edge_attr"), max_num_nodes,
                           batch_size))
    for key in keys:
        X = to_dense_tuplefeat(_get(batch, f"tuplefeat{key}"), _get(batch, f"tupleshape{key}"), _get(batch, f"tuplefeat{key}_ptr"),
                               None, batch_size, None)
        _set(batch, f"X{key}", X)
    return batch

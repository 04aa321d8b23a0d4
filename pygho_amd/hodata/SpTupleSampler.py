"""
Tuple samplers on the device (reference ``pygho/hodata/SpTupleSampler.py``).

``KhopSampler(data, hop)`` / ``I2Sampler(data, hop)`` keep the reference's names, arguments and result (a coalesced
``SparseTensor`` of tuple ids with the integer distance features as values: SpTupleSampler.py:91-126, :129-173); ``data`` is any
object with ``edge_index`` (2, E) int64 and ``num_nodes`` (torch_geometric's ``Data`` is not required).  The reference runs one
breadth-first search per root node / per edge on the host with a handful of torch calls each; ``khop_sample`` / ``i2_sample`` do a
whole block-diagonal BATCH of graphs in three launches (``csrc/sampler.hip``): all hop-distance matrices (one workgroup per graph,
LDS), tuple counts per root (+ a device scan), and the coalesced tuples written in sorted order.  Integer work, bit-exact against the
reference's samplers (``tests/golden/samplers.npz``).

Edges are walked from target to source like the reference's ``k_hop_subgraph`` (flow = 'source_to_target', :47-51).  ``I2Sampler``'s
features are shortest-path distances of the UNDIRECTED graph in the reference (scipy ``shortest_path(directed=False)``, :145-150);
here they come from the same search as the subsets, which is the same thing for the symmetric edge lists of every shipped dataset.
Graphs of any size (up to 255 nodes the search runs in LDS, larger graphs in global memory); hop distances are bytes, so a
search reaches at most 254 hops (`FULL`): in a graph of larger diameter the I2 features of farther nodes read "not reached".
"""
from __future__ import annotations

from typing import Any, Optional, Tuple

import torch
from torch import Tensor

from .. import _ops
from .._native import check, lib, ptr, require_device, stream_ptr
from ..backend.SpTensor import SparseTensor

FULL = 254          # hop limit that means "every reachable node" (distances are bytes, 255 = not reached)


class _GraphBatch:
    """int32 views of a block-diagonal batch the sampler kernels take: node_ptr, node_graph, predecessor CSR"""

    def __init__(self, edge_index: Tensor, num_nodes: int, node_graph: Optional[Tensor]):
        dev = require_device(edge_index, node_graph)
        assert edge_index.dim() == 2 and edge_index.shape[0] == 2 and edge_index.dtype == torch.int64
        self.dev, self.n = dev, int(num_nodes)
        if node_graph is None:
            node_graph = torch.zeros(self.n, dtype=torch.int64, device=dev)
        assert node_graph.numel() == self.n
        self.node_graph = node_graph.to(torch.int32).contiguous()
        self.n_graphs = int(node_graph.max().item()) + 1 if self.n else 0
        sizes = torch.bincount(node_graph, minlength=self.n_graphs)
        self.max_nodes = int(sizes.max().item()) if self.n_graphs else 0
        zero = torch.zeros(1, dtype=torch.int64, device=dev)
        self.node_ptr = torch.cat((zero, torch.cumsum(sizes, 0))).to(torch.int32)
        self.sq_ptr = torch.cat((zero, torch.cumsum(sizes * sizes, 0)))
        # predecessor lists: the sources of the edges that end in v, grouped by v (stable, so sorted when edge_index is)
        plan = _ops.plan_from_keys(edge_index[1].contiguous(), self.n)
        src32 = _ops.narrow_i32(edge_index[0].contiguous(), checked=True)
        self.rowptr, self.col = plan.seg_ptr, plan.take(src32)
        self.src32, self.dst32 = src32, _ops.narrow_i32(edge_index[1].contiguous(), checked=True)

    def distances(self, max_hop: int) -> Tensor:
        dist = torch.empty(int(self.sq_ptr[-1].item()), dtype=torch.uint8, device=self.dev)
        check(lib().pygho_graph_bfs_dist(ptr(dist), ptr(self.sq_ptr), ptr(self.node_ptr), ptr(self.rowptr), ptr(self.col), self.n_graphs,
                                         self.max_nodes, int(max_hop), stream_ptr(self.dev)), "graph_bfs_dist")
        return dist


def khop_sample(edge_index: Tensor, num_nodes: int, hop: int = 2, node_graph: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """k-hop tuples of a block-diagonal batch (``node_graph`` = the batch vector; one graph when omitted):
    tupleid (2, T) int64 sorted by (root, node), tuplefeat (T) int64 = hop distance."""
    gb = _GraphBatch(edge_index, num_nodes, node_graph)
    dev, hop = gb.dev, min(int(hop), FULL)
    dist = gb.distances(hop)
    count = torch.empty(gb.n, dtype=torch.int64, device=dev)
    check(lib().pygho_khop_count(ptr(count), ptr(dist), ptr(gb.sq_ptr), ptr(gb.node_ptr), ptr(gb.node_graph), gb.n, hop, stream_ptr(dev)),
          "khop_count")
    offset = _ops.exclusive_scan(count)
    total = int(offset[-1].item())
    tupleid = torch.empty((2, total), dtype=torch.int64, device=dev)
    feat = torch.empty(total, dtype=torch.int64, device=dev)
    check(lib().pygho_khop_emit(ptr(tupleid), ptr(feat), ptr(offset), total, ptr(dist), ptr(gb.sq_ptr), ptr(gb.node_ptr), ptr(gb.node_graph),
                                gb.n, hop, stream_ptr(dev)), "khop_emit")
    return tupleid, feat


def i2_sample(edge_index: Tensor, num_nodes: int, hop: int = 3, node_graph: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """pair-rooted tuples of a block-diagonal batch: for every directed edge (i, j) of the (coalesced, sorted) ``edge_index`` the
    nodes within ``hop`` of i or j: tupleid (3, T) int64 sorted, tuplefeat (T, 2) int64 = (distance to i, distance to j)."""
    gb = _GraphBatch(edge_index, num_nodes, node_graph)
    dev, hop = gb.dev, min(int(hop), FULL)
    dist = gb.distances(FULL)
    n_edges = edge_index.shape[1]
    count = torch.empty(n_edges, dtype=torch.int64, device=dev)
    check(lib().pygho_pair_count(ptr(count), ptr(gb.src32), ptr(gb.dst32), n_edges, ptr(dist), ptr(gb.sq_ptr), ptr(gb.node_ptr),
                                 ptr(gb.node_graph), hop, stream_ptr(dev)), "pair_count")
    offset = _ops.exclusive_scan(count)
    total = int(offset[-1].item())
    tupleid = torch.empty((3, total), dtype=torch.int64, device=dev)
    feat = torch.empty((total, 2), dtype=torch.int64, device=dev)
    check(lib().pygho_pair_emit(ptr(tupleid), ptr(feat), ptr(offset), total, ptr(gb.src32), ptr(gb.dst32), n_edges, ptr(dist), ptr(gb.sq_ptr),
                                ptr(gb.node_ptr), ptr(gb.node_graph), hop, stream_ptr(dev)), "pair_emit")
    return tupleid, feat


def _graph(data: Any) -> Tuple[Tensor, int]:
    ei = data["edge_index"] if isinstance(data, dict) else data.edge_index
    n = data["num_nodes"] if isinstance(data, dict) else data.num_nodes
    return ei, int(n)


def KhopSampler(data: Any, hop: int = 2) -> SparseTensor:
    """reference SpTupleSampler.py:91-126: k-hop subgraph around every node of ONE graph; values = hop distance."""
    ei, n = _graph(data)
    tupleid, feat = khop_sample(ei, n, hop)
    return SparseTensor(tupleid, feat, shape=2 * [n], is_coalesced=True)


def I2Sampler(data: Any, hop: int = 3) -> SparseTensor:
    """reference SpTupleSampler.py:129-173: subgraph around every directed edge of ONE graph; values = (distance to the
    first end, distance to the second end).  ``edge_index`` must be coalesced (sorted), as torch_geometric datasets are."""
    ei, n = _graph(data)
    tupleid, feat = i2_sample(ei, n, hop)
    # tuples come out in edge order: already coalesced when the edge list is sorted and duplicate-free (the reference coalesces
    # with reduce="min" regardless, :173)
    sorted_edges = ei.shape[1] < 2 or bool(torch.all(torch.diff(_ops.hash_pack(ei)) > 0))
    return SparseTensor(tupleid, feat, shape=3 * [n] + [2], is_coalesced=sorted_edges, reduce="min")

"""
Device-resident graph store and on-device mini-batch collation.

The reference collates on the host with PyG's ``Batch.from_data_list`` (``hodata/SpData.py:56-112``: concatenate the graphs
block-diagonally, ``__inc__`` adds the running node offset to ``edge_index`` / ``tupleid`` and the running tuple / edge
offsets to the rows of every ``*___acd`` plan) and ships int64 triples to the device for every batch.  Here the whole
dataset lives on the device once, graph-local and in **int32**; a mini-batch is a handful of integer kernels
(``pygho_collate_rows``: gather the selected graphs' columns, widen to the API's int64 and add the offsets) -- no H2D
traffic per batch (an 8192-graph batch is ~150 MB of int64 indices).  The output is the ``datadict`` of ``synth.to_datadict``.
"""
from typing import Dict, List, Sequence, Union

import numpy as np
import torch

from . import _ops
from ._native import check, lib, ptr, require_device, stream_ptr
from .backend.SpTensor import SparseTensor
from .synth import KEYSEP, GraphRecord, parse_key


import ctypes


class CollateDesc(ctypes.Structure):
    """`pygho_collate_desc` of include/pygho_hip.h: one output array of a batch"""
    _fields_ = [("out", ctypes.c_void_p), ("src", ctypes.c_void_p), ("src_start", ctypes.c_void_p), ("out_ptr", ctypes.c_void_p),
                ("inc", ctypes.c_void_p * 4), ("pad", ctypes.c_void_p), ("src_ld", ctypes.c_int64), ("out_ld", ctypes.c_int64),
                ("rows", ctypes.c_int32), ("out_i32", ctypes.c_int32), ("transposed", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def launch_collate(descs, n_sel: int, device) -> None:
    """every array of a batch in ONE launch (`pygho_collate_batch`): the descriptor table goes to the device in one small copy"""
    if not descs:
        return
    assert int(lib().pygho_collate_desc_bytes()) == ctypes.sizeof(CollateDesc)
    table = (CollateDesc * len(descs))(*descs)
    raw = torch.frombuffer(bytearray(memoryview(table).cast("B")), dtype=torch.uint8).to(device, non_blocking=True)
    check(lib().pygho_collate_batch(raw.data_ptr(), len(descs), n_sel, max(int(d.out_ld) for d in descs), stream_ptr(device)), "collate_batch")


class _BatchBuilder:
    """collects the output arrays of one exactly sized batch over a `_Layout`; `launch()` writes them all with one kernel"""

    def __init__(self, store: "DeviceGraphStore", lay: "_Layout"):
        self.store, self.lay, self.descs = store, lay, []

    def off(self, fam) -> int:
        """device address of the running offsets of family `fam` (an increment row: entry s = offset of graph s)"""
        return self.lay.dev[("optr", fam)].data_ptr()

    def total(self, fam) -> int:
        """device address of the batch's total of family `fam` (a pad value: the closing entry of a pointer array)"""
        return self.lay.dev[("optr", fam)].data_ptr() + 8 * self.lay.g

    def rows_of(self, inc: torch.Tensor):
        """row addresses of a (rows, g) increment piece of the layout"""
        return tuple(inc.data_ptr() + 8 * self.lay.g * r for r in range(inc.shape[0]))

    def add(self, src: torch.Tensor, fam, incs=(), pad=None, i32: bool = False, transposed: bool = False, extra: int = 0) -> torch.Tensor:
        lay = self.lay
        rows, cols = src.shape[0], lay.total[fam] + extra
        out = torch.empty((cols, rows) if transposed else (rows, cols), dtype=torch.int32 if i32 else torch.int64, device=self.store.device)
        if cols == 0 or rows == 0:
            return out
        d = CollateDesc()
        d.out, d.src = out.data_ptr(), src.data_ptr()
        d.src_start, d.out_ptr = lay.dev[("start", fam)].data_ptr(), lay.dev[("optr", fam)].data_ptr()
        for r, a in enumerate(incs):
            d.inc[r] = a
        d.pad = pad
        d.src_ld, d.out_ld, d.rows = src.shape[1], cols, rows
        d.out_i32, d.transposed = int(i32), int(transposed)
        self.descs.append(d)
        return out

    def launch(self) -> None:
        launch_collate(self.descs, self.lay.g, self.store.device)
        self.descs = []


def _cat32(arrs: List[np.ndarray], axis: int, device) -> torch.Tensor:
    a = np.concatenate(arrs, axis=axis)
    assert a.size == 0 or (a.min() >= 0 and a.max() < 2 ** 31)
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(device)


def _mirror_of(r: GraphRecord, n_types: int):
    """(positions of the transposed tuples, verdict) of one graph, as `segment.pair_mirror` decides for a batch"""
    row, col = r.tupleid[0].astype(np.int64), r.tupleid[1].astype(np.int64)
    t = row.size
    if t == 0:
        return np.zeros(0, dtype=np.int64), True
    n, f = int(r.num_nodes), np.asarray(r.tuplefeat).reshape(-1)
    key, key_t = row * n + col, col * n + row
    pos = np.minimum(np.searchsorted(key, key_t), t - 1)
    ok = (np.all(key[1:] > key[:-1]) and np.all(key[pos] == key_t) and np.all(f[pos] == f) and f.min() >= 0 and f.max() < n_types
          and col.max() < n)
    return (pos if ok else np.zeros(t, dtype=np.int64)), bool(ok)


def _validate_records(records: List[GraphRecord], keys) -> None:
    """Every batch collated from the store is marked range-checked (`_pygho_value_bound`, `_pygho_hash_ok`, plans installed without
    the planners' range flags), so the checks the reference makes per batch -- its gathers raise IndexError, its hash asserts
    refuse negative indices (SpTensor.py:32,37, Spspmm.py:309-311) -- happen HERE, once per dataset: integer features are
    non-negative, every node id addresses a node of its graph, every triple addresses rows of its operands.  A node id >= num_nodes
    would also lengthen a `bincount(..., minlength=num_nodes)` row and silently misalign every later graph's plans."""
    for gi, r in enumerate(records):
        n = int(r.num_nodes)
        where = f"graph {gi}"
        if n < 0 or np.shape(r.x)[0] != n:
            raise ValueError(f"pygho_amd: {where}: x has {np.shape(r.x)[0]} rows for {n} nodes")
        for name, a in (("x", r.x), ("edge_attr", r.edge_attr), ("tuplefeat", r.tuplefeat)):
            if np.size(a) and int(np.min(a)) < 0:
                raise ValueError(f"pygho_amd: {where}: negative integer feature in {name} (features index embedding tables)")
        for name, ind in (("edge_index", r.edge_index), ("tupleid", r.tupleid)):
            if np.size(ind) and (int(np.min(ind)) < 0 or int(np.max(ind)) >= n):
                raise ValueError(f"pygho_amd: {where}: {name} addresses a node outside [0, {n})")
        if np.shape(r.edge_attr)[0] != r.edge_index.shape[1] or np.shape(r.tuplefeat)[0] != r.tupleid.shape[1]:
            raise ValueError(f"pygho_amd: {where}: feature rows do not match the index columns")
        for k in keys:
            roles = parse_key(k)
            acd = r.acd[k]
            if acd.shape[0] != 3:
                raise ValueError(f"pygho_amd: {where}: acd of {k} must be (3, M)")
            for row, role in zip(acd, (roles[0], roles[1], roles[3])):
                rows = r.tupleid.shape[1] if role[0] == "X" else r.edge_index.shape[1]
                if row.size and (int(row.min()) < 0 or int(row.max()) >= rows):
                    raise ValueError(f"pygho_amd: {where}: acd of {k} addresses a row outside [0, {rows}) of operand {role}")


def _ptr64(lengths: Sequence[int], device) -> torch.Tensor:
    return torch.from_numpy(np.concatenate(([0], np.cumsum(np.asarray(lengths, dtype=np.int64))))).to(device)


class DeviceGraphStore:
    """all graphs of a dataset on the device: graph-local int32 indices + int64 segment pointers per graph."""

    def __init__(self, records: List[GraphRecord], device):
        self.device = torch.device(device)
        self.num_graphs = len(records)
        self.keys = list(records[0].acd.keys())
        self.sd = records[0].tupleid.shape[0]
        _validate_records(records, self.keys)       # ONCE, on the host: what the per-batch range flags used to catch (see below)
        d = self.device
        self.node_ptr = _ptr64([r.num_nodes for r in records], d)
        self.edge_ptr = _ptr64([r.edge_index.shape[1] for r in records], d)
        self.tup_ptr = _ptr64([r.tupleid.shape[1] for r in records], d)
        self.x = _cat32([r.x.reshape(1, -1) for r in records], 1, d)
        self._zeros_node = torch.zeros((1, max(int(self.node_ptr[-1]), 1)), dtype=torch.int32, device=d)     # + a per-graph increment = the batch vector
        self.edge_index = _cat32([r.edge_index for r in records], 1, d)
        self.edge_attr = _cat32([r.edge_attr.reshape(1, -1) for r in records], 1, d)
        self.tupleid = _cat32([r.tupleid for r in records], 1, d)
        tf = [r.tuplefeat.reshape(r.tuplefeat.shape[0], -1).T for r in records]          # (f, t)
        self.tuplefeat = _cat32(tf, 1, d)
        self.feat_shape = tuple(records[0].tuplefeat.shape[1:])
        self.acd = {k: _cat32([r.acd[k] for r in records], 1, d) for k in self.keys}
        self.acd_ptr = {k: _ptr64([r.acd[k].shape[1] for r in records], d) for k in self.keys}
        # the transposed groupings of every graph's message triples, computed ONCE here: a block-diagonal batch's grouping by c
        # (or d) is the concatenation of its graphs' groupings with the message / row offsets added, so a batch's message plan
        # is collated like everything else instead of being re-sorted (two 3.5 M-key radix sorts per 8192-graph batch)
        self.plan_parts = {}
        for k in self.keys:
            roles = parse_key(k)
            rows_of = lambda r, role: r.tupleid.shape[1] if role[0] == "X" else r.edge_index.shape[1]
            if not all(r.acd[k].shape[1] == 0 or np.all(np.diff(r.acd[k][0]) >= 0) for r in records):
                continue                                         # triples not sorted by target (not from filterind): plan by sorting
            perm_c, perm_d, cnt_a, cnt_c, cnt_d = [], [], [], [], []
            # for the fixed-capacity batch slots (`slots.BatchSlot`): graph-LOCAL CSR pointers (a batch's pointers are these plus the
            # graph's message offset -- collated like any index array, no scan per batch), the triples' other two coordinates already
            # in by-c / by-d order, and the edge feature each message looks up (forward and by-c order) when the second operand is A
            ptr_a, ptr_c, ptr_d, by_c, by_d, look, max_a = [], [], [], [], [], [], []
            want_look = roles[3][0] != "X" and all(np.ndim(r.edge_attr) == 1 for r in records)
            excl = lambda cnt: (np.cumsum(cnt) - cnt).reshape(1, -1)
            for r in records:
                a, c, dd_ = r.acd[k]
                pc, pd = np.argsort(c, kind="stable"), np.argsort(dd_, kind="stable")
                perm_c.append(pc.reshape(1, -1))
                perm_d.append(pd.reshape(1, -1))
                cnt_a.append(np.bincount(a, minlength=rows_of(r, roles[0])).reshape(1, -1))
                max_a.append(int(cnt_a[-1].max()) if cnt_a[-1].size else 0)
                cnt_c.append(np.bincount(c, minlength=rows_of(r, roles[1])).reshape(1, -1))
                cnt_d.append(np.bincount(dd_, minlength=rows_of(r, roles[3])).reshape(1, -1))
                ptr_a.append(excl(cnt_a[-1][0]))
                ptr_c.append(excl(cnt_c[-1][0]))
                ptr_d.append(excl(cnt_d[-1][0]))
                by_c.append(np.stack((a[pc], dd_[pc])))
                by_d.append(np.stack((a[pd], c[pd])))
                if want_look:
                    look.append(np.stack((r.edge_attr[dd_], r.edge_attr[dd_[pc]])))
            self.plan_parts[k] = {"perm_c": _cat32(perm_c, 1, d), "perm_d": _cat32(perm_d, 1, d), "cnt_a": _cat32(cnt_a, 1, d),
                                  "cnt_c": _cat32(cnt_c, 1, d), "cnt_d": _cat32(cnt_d, 1, d),
                                  "ptr_a": _cat32(ptr_a, 1, d), "ptr_c": _cat32(ptr_c, 1, d), "ptr_d": _cat32(ptr_d, 1, d),
                                  "by_c": _cat32(by_c, 1, d), "by_d": _cat32(by_d, 1, d)}
            if want_look:
                self.plan_parts[k]["look"] = _cat32(look, 1, d)
            # longest forward segment per graph: a batch's `plan.fwd.max_len` (asked by the max / min forward: do the tie counts fit the
            # value dtype?) is a maximum over the selected graphs -- answered on the host, no device read
            self.plan_parts[k]["h_max_a"] = np.asarray(max_a, dtype=np.int64)
        # the by-edge gradient's scatter plans (csrc/seg_scatter.hip), likewise ONCE for the whole store: the device planner over the
        # graph-local triples with one block per graph.  A batch's chunks are its graphs' chunks with the message / row offsets added
        # and its packed words are its graphs' words unchanged (they are relative to chunk windows and block edge ranges)
        self.scatter_parts = {}
        for k in self.plan_parts:
            acd = self.acd[k]
            if acd.shape[1] == 0 or acd.shape[1] >= (1 << 31):
                continue
            block_m = self.acd_ptr[k].to(torch.int32)
            roles = parse_key(k)
            parts = None
            if _ops.DUAL_BWD and roles[0][0] == roles[1][0]:
                # ALIGNED chunks (every message of a chunk's first-operand rows inside the chunk) also serve the fused backward
                # (csrc/seg_dual.hip); a graph with a group of messages outside the chunk limits leaves the store with the plain chunks
                # (the planner wants the first-operand rows of all blocks in ONE ascending numbering: store-wide rows here, the
                # chunk records' window starts made graph-local again below)
                rows1 = (self.tup_ptr if roles[1][0] == "X" else self.edge_ptr)
                if int(rows1[-1]) < (1 << 31):
                    msg_len = self.acd_ptr[k][1:] - self.acd_ptr[k][:-1]
                    cut1 = rows1.to(torch.int32)
                    c_store = (acd[1] + torch.repeat_interleave(cut1[:-1], msg_len)).contiguous()
                    parts = _ops.scatter_plan_parts_aligned(acd[0].contiguous(), c_store, acd[2].contiguous(), block_m, cut1)
                    if parts is not None:
                        g_of = torch.repeat_interleave(torch.arange(self.num_graphs, device=d), parts[0].long())     # graph of every chunk
                        parts[3][:, 2] -= cut1[g_of]
            aligned = parts is not None
            if parts is None:
                parts = _ops.scatter_plan_parts(acd[0].contiguous(), acd[1].contiguous(), acd[2].contiguous(), block_m)
            if parts is None:
                continue
            n_chunks, chunk0, blk_e, chunks, words, max_edges = parts[:6]
            # a chunk's first message, graph-LOCAL like its row fields: a batch (or a batch slot) adds the graph's message offset
            chunks[:, 0] -= self.acd_ptr[k].to(torch.int32)[torch.repeat_interleave(torch.arange(self.num_graphs, device=d), n_chunks.long())]
            rows3 = (self.tup_ptr if roles[3][0] == "X" else self.edge_ptr)
            n_rows3 = (rows3[1:] - rows3[:-1]).to(torch.int32)
            covers = bool(((blk_e[:, 0] == 0) & (blk_e[:, 1] == n_rows3)).all())        # every graph's triples reach all of its rows
            self.scatter_parts[k] = {"chunk_ptr": chunk0.to(torch.int64), "chunks_t": chunks.t().contiguous(), "words": words.reshape(1, -1),
                                     "blk_e": blk_e.t().contiguous(), "max_edges": max_edges, "covers": covers}
            if aligned:
                # rows without messages are written by the chunk that owns them; a graph with rows but NO message has no chunk: a batch
                # that selects one pre-fills the by-tuple gradient
                has_rows = (rows1[1:] - rows1[:-1]) > 0
                has_msgs = (self.acd_ptr[k][1:] - self.acd_ptr[k][:-1]) > 0
                self.scatter_parts[k]["cgap"] = parts[7].reshape(1, -1)
                self.scatter_parts[k]["h_cok"] = (has_msgs | ~has_rows).cpu().numpy()
        # the fused block forward's chunk plans (csrc/seg_fused.hip: Linear -> BatchNorm -> act inside the aggregation), ONCE for the whole
        # store: the device planner over the store's rows with one block per graph.  A batch's chunks are its graphs' chunks with the
        # message / row offsets added; the ownership masks travel unchanged (a chunk's window never leaves its graph)
        self.fused_parts = {}
        for k, parts in self.plan_parts.items():
            roles = parse_key(k)
            acd = self.acd[k]
            if not (roles[0][0] == "X" and roles[1][0] == "X" and "look" in parts) or acd.shape[1] == 0 or acd.shape[1] >= (1 << 31):
                continue
            tup32, msg_len = self.tup_ptr.to(torch.int32), self.acd_ptr[k][1:] - self.acd_ptr[k][:-1]
            seg_ptr = torch.zeros(parts["cnt_a"].shape[1] + 1, dtype=torch.int32, device=d)
            torch.cumsum(parts["cnt_a"].reshape(-1), 0, out=seg_ptr[1:])
            c_rows = (acd[1] + torch.repeat_interleave(tup32[:-1], msg_len)).contiguous()             # store-wide row of every message's c
            fu = _ops.fused_plan_parts(seg_ptr, c_rows, tup32, int(self.tup_ptr[-1]))
            if fu is None:
                continue
            n_chunks, chunk0, chunks, own = fu
            g_of = torch.repeat_interleave(torch.arange(self.num_graphs, device=d), n_chunks.long())      # graph of every chunk
            base = tup32[g_of]
            chunks = chunks.clone()
            chunks[:, 0] -= self.acd_ptr[k][g_of].to(torch.int32)  # first message and rows graph-local: a batch adds the graph's offsets
            chunks[:, 1] -= base
            chunks[:, 2] = torch.where((chunks[:, 3] >> 16) > 0, chunks[:, 2] - base, torch.zeros_like(base))
            self.fused_parts[k] = {"chunk_ptr": chunk0.to(torch.int64), "chunks_t": chunks.t().contiguous(), "own": own.reshape(1, -1)}
        # 3-tuple stores (I2GNN): the MERGED (i, j) pattern that pooling the last coordinate away produces (reference
        # SpTensor.py:368-380 via OpPoolingSubg3D, SpOperator.py:496-522), per graph and ONCE: the tuples are sorted by (i, j, k), so
        # the distinct (i, j) pairs are runs -- a batch's pooled pattern, its CSR pointers over the tuples, the tuple -> pair map and
        # the pairs' grouping by root are the graphs' parts with offsets added (round 5: a sort + unique with host reads per fresh batch)
        self.pair_parts = None
        if self.sd == 3 and all(r.tupleid.shape[1] > 0 for r in records):
            idx, ptr, inv, rptr, n_pairs, h_max, h_rmax, ok = [], [], [], [], [], [], [], True
            for r in records:
                t = r.tupleid.astype(np.int64)
                key = t[0] * r.num_nodes + t[1]
                if np.any(np.diff(key) < 0):
                    ok = False
                    break
                start = np.concatenate(([0], np.nonzero(np.diff(key))[0] + 1))
                cnt = np.diff(np.concatenate((start, [t.shape[1]])))
                idx.append(t[:2, start])
                ptr.append((np.cumsum(cnt) - cnt).reshape(1, -1))
                inv.append(np.repeat(np.arange(start.size), cnt).reshape(1, -1))
                rc = np.bincount(t[0, start], minlength=r.num_nodes)
                rptr.append((np.cumsum(rc) - rc).reshape(1, -1))
                n_pairs.append(start.size)
                h_max.append(int(cnt.max()))
                h_rmax.append(int(rc.max()))
            if ok:
                self.pair_parts = {"index": _cat32(idx, 1, d), "ptr": _cat32(ptr, 1, d), "inv": _cat32(inv, 1, d), "root_ptr": _cat32(rptr, 1, d),
                                   "h_len": np.asarray(n_pairs, dtype=np.int64), "h_max": np.asarray(h_max, dtype=np.int64),
                                   "h_root_max": np.asarray(h_rmax, dtype=np.int64)}
        self.y = torch.tensor([r.y for r in records], dtype=torch.float32, device=d)
        # largest integer feature of the store per array: a lookup into a table with more rows than that needs no range flag
        vmax = lambda f: max((int(np.max(f(r))) for r in records if np.size(f(r))), default=-1)
        self.h_vmax = {"x": vmax(lambda r: r.x), "ea": vmax(lambda r: r.edge_attr), "tf": vmax(lambda r: r.tuplefeat)}
        # per-graph lengths on the HOST as well: a batch's output sizes (and the longest segments its plans ask about) are sums /
        # maxima over the selected graphs, so collation needs no device-to-host read
        ln = lambda f: np.asarray([f(r) for r in records], dtype=np.int64)
        self.h_len = {"node": ln(lambda r: r.num_nodes), "edge": ln(lambda r: r.edge_index.shape[1]), "tup": ln(lambda r: r.tupleid.shape[1])}
        for k in self.keys:
            self.h_len[("acd", k)] = ln(lambda r: r.acd[k].shape[1])
        if self.pair_parts is not None:
            self.h_len["pair"] = self.pair_parts["h_len"]
        for k, v in self.scatter_parts.items():
            self.h_len[("sc", k)] = np.diff(v["chunk_ptr"].cpu().numpy())
        for k, v in self.fused_parts.items():
            self.h_len[("fu", k)] = np.diff(v["chunk_ptr"].cpu().numpy())
        self.h_ptr = {f: np.concatenate(([0], np.cumsum(v))).astype(np.int64) for f, v in self.h_len.items()}
        # tuples per root node (the pattern is sorted by root, SpTupleSampler.py:91-126): a batch's grouping of its tuples by root --
        # subgraph pooling, the tuple initialisation's by-row plan -- is the scan of the collated counts
        self.root_parts = None
        if all(r.tupleid.shape[1] == 0 or np.all(np.diff(r.tupleid[0]) >= 0) for r in records):
            cnt = [np.bincount(r.tupleid[0], minlength=r.num_nodes) for r in records]
            self.root_parts = {"cnt": _cat32([c.reshape(1, -1) for c in cnt], 1, d),
                               "ptr": _cat32([(np.cumsum(c) - c).reshape(1, -1) for c in cnt], 1, d),
                               "h_max": np.asarray([c.max() if c.size else 0 for c in cnt], dtype=np.int64)}
        # position of every node's diagonal tuple (i, i) inside its graph's tuple list (GNNAKConv / SUNConv look it up by hash search per
        # batch, honn/Conv.py "sun_views"): collated with the graph's tuple offset when every node of every graph has one
        self.diag_parts = None
        if self.sd == 2:
            pos, ok = [], True
            for r in records:
                row, col = r.tupleid
                where = np.nonzero(row == col)[0]
                if where.size != r.num_nodes or not np.array_equal(row[where], np.arange(r.num_nodes)):
                    ok = False
                    break
                pos.append(where.reshape(1, -1))
            if ok:
                self.diag_parts = {"pos": _cat32(pos, 1, d)}
        # groupings of the tuples by their OTHER coordinates and of the edges by either endpoint (cross-subgraph pooling, unpooling
        # gradients, spmm): per-graph counts per node and, where the coordinate is not sorted, the stable order -- assembled per batch
        # only when an operator asks (`plans.cached_plan` -> `_pygho_plan_factory`)
        self.group_parts = {}
        for name, rows_of in [(("X", dim), (lambda r, dim=dim: r.tupleid[dim])) for dim in range(1, self.sd)] + \
                             [(("A", dim), (lambda r, dim=dim: r.edge_index[dim])) for dim in (0, 1)]:
            arrs = [rows_of(r) for r in records]
            cnt = [np.bincount(a, minlength=r.num_nodes) for a, r in zip(arrs, records)]
            part = {"cnt": _cat32([c.reshape(1, -1) for c in cnt], 1, d), "h_max": np.asarray([c.max() if c.size else 0 for c in cnt], dtype=np.int64),
                    "ptr": _cat32([(np.cumsum(c) - c).reshape(1, -1) for c in cnt], 1, d)}
            if not all(a.size == 0 or np.all(np.diff(a) >= 0) for a in arrs):
                part["perm"] = _cat32([np.argsort(a, kind="stable").reshape(1, -1) for a in arrs], 1, d)
            self.group_parts[name] = part
        # mirror positions of symmetric 2-tuple sets (segment.pair_mirror: the one-pass backward of the tuple initialisation):
        # graph-local here, plus the tuple offset of the graph in a batch; the verdict of a batch is the AND over its graphs
        self.mirror_parts = None
        if self.sd == 2 and not self.feat_shape:
            nt = int(lib().pygho_pair_bwd_types())
            pos, ok = zip(*(_mirror_of(r, nt) for r in records))
            self.mirror_parts = {"pos": _cat32([p.reshape(1, -1) for p in pos], 1, d), "h_ok": np.asarray(ok, dtype=bool)}

    # ------------------------------------------------------------------
    def _rows(self, src: torch.Tensor, lay: "_Layout", fam, inc: Union[torch.Tensor, None] = None, i32: bool = False,
              transposed: bool = False) -> torch.Tensor:
        """the columns of `src` (rows, store length of family `fam`) that belong to the selected graphs, concatenated, with
        `inc[r, s]` added to row r of graph s: int64 (the API's arrays) or int32 (plan arrays), (rows, total) or (total, rows)"""
        rows, total = src.shape[0], lay.total[fam]
        out = torch.empty((total, rows) if transposed else (rows, total), dtype=torch.int32 if i32 else torch.int64, device=self.device)
        args = (ptr(out), ptr(src), rows, src.shape[1], total, ptr(lay.dev[("start", fam)]), ptr(lay.dev[("optr", fam)]),
                ptr(inc), lay.g, total)
        if i32:
            check(lib().pygho_collate_rows_i32(*args, int(transposed), stream_ptr(self.device)), "collate_rows_i32")
        else:
            check(lib().pygho_collate_rows(*args, stream_ptr(self.device)), "collate_rows")
        return out

    def _install_node_plans(self, dd, lay: "_Layout", B: "_BatchBuilder"):
        """the groupings a model step asks for beyond the message plans, installed where the operators look for them: nodes by
        graph (graph pooling), tuples by root (subgraph pooling, `pair_product`'s by-row plan) with their longest segments, and the
        mirror verdict of the tuple set (`segment.pair_mirror`).  With these a step on the batch builds no plan and reads nothing back.
        The arrays are only REQUESTED from the builder here; the returned closure installs what needs the narrowed index rows, after
        the batch's one kernel has been launched."""
        g, n, X, ids_h = lay.g, lay.total["node"], dd["X"], lay.ids_h
        if g == 0:
            return lambda: None
        _ops.install_plan(dd["batch"], _ops.SegPlan(lay.dev[("optr32", "node")], None, g, n), ("scatter",),
                          max_len=self.h_len["node"][ids_h].max())
        for (which, dim), part in self.group_parts.items():
            sp = X if which == "X" else dd["A"]
            fam = "tup" if which == "X" else "edge"
            keys = sp._row(dim)
            keys._pygho_plan_factory = (keys._version, self._group_factory(part, lay, fam))
        if self.root_parts is None:
            return lambda: None
        row = X._row(0)
        # tuples by root: the graph-local CSR pointers + the graph's tuple offset, closed by the batch's tuple total (no scan)
        root_ptr = B.add(self.root_parts["ptr"], "node", incs=(B.off("tup"),), pad=B.total("tup"), i32=True, extra=1).reshape(-1)
        _ops.install_plan(row, _ops.SegPlan(root_ptr, None, n, lay.total["tup"]), ("scatter", "pair-row"),
                          max_len=self.root_parts["h_max"][ids_h].max())
        if self.mirror_parts is None or lay.total["tup"] == 0 or lay.total["tup"] >= (1 << 31):
            return lambda: None
        res = None
        if bool(self.mirror_parts["h_ok"][ids_h].all()):
            res = B.add(self.mirror_parts["pos"], "tup", incs=(B.off("tup"),), i32=True).reshape(-1)

        def finish():
            row32, col32, vidx32 = _ops.narrow_i32(row), _ops.narrow_i32(X._row(1)), _ops.narrow_i32(_ops.flat_index(X.values))
            row32._pygho_mirror = (col32, vidx32, n, res, (row32._version, col32._version, vidx32._version))
        return finish

    def _install_pair_plans(self, dd, lay: "_Layout", B: "_BatchBuilder"):
        """3-tuple batches: the merged (i, j) pattern of `X.sum / max / mean(dims=[2], return_sparse=True)` (OpPoolingSubg3D) with its
        plan, and the pooled pattern's grouping by root (the OpPoolingSubg2D that follows), where `SparseTensor._reduce_to_sparse` /
        `_reduce_to_dense` look for them: no sort, no unique, no host read per fresh batch"""
        pp = self.pair_parts
        if pp is None or lay.g == 0 or lay.total["pair"] == 0 or lay.total["tup"] >= (1 << 31):
            return lambda: None
        X, n, ids_h = dd["X"], lay.total["node"], lay.ids_h
        n_pairs, nnz = lay.total["pair"], lay.total["tup"]
        new_ind = B.add(pp["index"], "pair", incs=(B.off("node"), B.off("node")))
        pair_ptr = B.add(pp["ptr"], "pair", incs=(B.off("tup"),), pad=B.total("tup"), i32=True, extra=1).reshape(-1)
        inv32 = B.add(pp["inv"], "tup", incs=(B.off("pair"),), i32=True).reshape(-1)
        root_ptr = B.add(pp["root_ptr"], "node", incs=(B.off("pair"),), pad=B.total("pair"), i32=True, extra=1).reshape(-1)

        def finish():
            plan = _ops.SegPlan(pair_ptr, None, n_pairs, nnz)
            plan._memo = {"max_len": int(pp["h_max"][ids_h].max())}
            if n < (1 << 31):
                new_ind._pygho_hash_ok = new_ind._version
            X._cache()[("pool_sparse", (0, 1))] = (new_ind, plan, inv32)
            row0 = _ops.unbased(new_ind[0])
            new_ind._pygho_cache = {"_v": new_ind._version, ("row", 0): row0}
            _ops.install_plan(row0, _ops.SegPlan(root_ptr, None, n, n_pairs), ("scatter",), max_len=int(pp["h_root_max"][ids_h].max()))
        return finish

    def _group_factory(self, part, lay: "_Layout", fam):
        def build(n_seg: int):
            if n_seg != lay.total["node"] or lay.total[fam] >= (1 << 31):
                return None
            cnt = self._rows(part["cnt"], lay, "node").reshape(-1)
            perm = None
            if "perm" in part:
                perm = self._rows(part["perm"], lay, fam, lay.dev[("off", fam)], i32=True).reshape(-1)
            plan = _ops.SegPlan(_ops.exclusive_scan(cnt).to(torch.int32), perm, n_seg, lay.total[fam])
            plan._memo = {"max_len": int(part["h_max"][lay.ids_h].max()) if lay.g else 0}
            return plan
        return build

    def collate(self, graph_ids: Union[Sequence[int], torch.Tensor]) -> Dict:
        """datadict of the block-diagonal batch of ``graph_ids`` (any order, repeats allowed; a host sequence / CPU tensor: a
        device tensor of ids is read back first).  Output sizes, running offsets and per-graph increments are computed on the
        host from the store's per-graph lengths and reach the device in ONE upload; every array of the batch -- the API's int64 index
        arrays and all plan arrays -- is then written by ONE kernel from a descriptor table (`pygho_collate_batch`; round 4: one
        kernel per array and a scan per CSR pointer array, ~45 launches per batch)."""
        ids_h = (graph_ids.detach().cpu().numpy() if isinstance(graph_ids, torch.Tensor) else np.asarray(graph_ids)).astype(np.int64).reshape(-1)
        assert ids_h.size == 0 or (ids_h.min() >= 0 and ids_h.max() < self.num_graphs), "graph id out of range"
        lay = _Layout(self, ids_h)
        B = _BatchBuilder(self, lay)
        g, n, total = lay.g, lay.total["node"], lay.total
        fam_of = lambda role: "tup" if role[0] == "X" else "edge"
        ei = B.add(self.edge_index, "edge", incs=B.rows_of(lay.dev[("inc", "ei")]))
        ea = B.add(self.edge_attr, "edge").reshape(-1)
        tid = B.add(self.tupleid, "tup", incs=B.rows_of(lay.dev[("inc", "tid")]))
        if self.feat_shape:
            tf = B.add(self.tuplefeat, "tup", transposed=True).reshape((total["tup"],) + self.feat_shape)
        else:
            tf = B.add(self.tuplefeat, "tup").reshape(-1)
        x = B.add(self.x, "node").reshape(-1)
        batch = B.add(self._zeros_node, "node", incs=(B.off("graph"),)).reshape(-1)         # 0 + the graph's position in the batch
        for name, t in (("x", x), ("ea", ea), ("tf", tf)):      # values below this bound (the store checked them when it was built)
            t._pygho_value_bound = (t._version, self.h_vmax[name] + 1)
        for ind in (ei, tid):                       # non-negative and below n by construction: the hash asserts need no read-back
            if n < (1 << (63 // ind.shape[0])):
                ind._pygho_hash_ok = ind._version
        dd = {
            "x": x, "batch": batch, "num_graphs": g, "y": self.y[lay.dev[("ids",)]], "num_nodes": n,
            "A": SparseTensor(ei, ea, [n, n], is_coalesced=True),
            "X": SparseTensor(tid, tf, [n] * self.sd + list(self.feat_shape), is_coalesced=True),
        }
        # the int32 copies the kernels read (a fresh batch used to narrow them inside the step: ~8 launches), from the same launch
        seeds = [(x, B.add(self.x, "node", i32=True).reshape(-1)), (ea, B.add(self.edge_attr, "edge", i32=True).reshape(-1)),
                 (batch, B.add(self._zeros_node, "node", incs=(B.off("graph"),), i32=True).reshape(-1))]
        if not self.feat_shape:
            seeds.append((tf, B.add(self.tuplefeat, "tup", i32=True).reshape(-1)))
        tid32 = B.add(self.tupleid, "tup", incs=B.rows_of(lay.dev[("inc", "tid")]), i32=True)
        ei32 = B.add(self.edge_index, "edge", incs=B.rows_of(lay.dev[("inc", "ei")]), i32=True)
        seeds += [(dd["X"]._row(dim), tid32[dim]) for dim in range(self.sd)] + [(dd["A"]._row(dim), ei32[dim]) for dim in range(2)]
        for t64, t32 in seeds:
            if t64.numel():
                t64._pygho_i32 = (t64._version, t32)
        finish = [self._install_node_plans(dd, lay, B), self._install_pair_plans(dd, lay, B)]
        for k in self.keys:
            roles = parse_key(k)
            fm = ("acd", k)
            acd = B.add(self.acd[k], fm, incs=B.rows_of(lay.dev[("inc", "acd", k)]))
            dd[k + KEYSEP + "acd"] = acd
            # message plan from the stored per-graph groupings (no sort, no scan, no host sync): graph-local CSR pointers and
            # permutations get the graph's message offset, the triples' coordinates in by-c / by-d order their rows' offsets
            if k not in self.plan_parts:
                continue
            parts = self.plan_parts[k]
            fa, fc, fd = fam_of(roles[0]), fam_of(roles[1]), fam_of(roles[3])
            csr = lambda which, fam: B.add(parts[which], fam, incs=(B.off(fm),), pad=B.total(fm), i32=True, extra=1).reshape(-1)
            perm = lambda which: B.add(parts[which], fm, incs=(B.off(fm),), i32=True).reshape(-1)
            arrs = dict(acd32=B.add(self.acd[k], fm, incs=B.rows_of(lay.dev[("inc", "acd", k)]), i32=True),
                        ptr_a=csr("ptr_a", fa), ptr_c=csr("ptr_c", fc), ptr_d=csr("ptr_d", fd), perm_c=perm("perm_c"), perm_d=perm("perm_d"),
                        by_c=B.add(parts["by_c"], fm, incs=(B.off(fa), B.off(fd)), i32=True),
                        by_d=B.add(parts["by_d"], fm, incs=(B.off(fa), B.off(fc)), i32=True))
            if "look" in parts and fd == "edge":                # the edge feature every message looks up (forward and by-c order)
                arrs["look"] = B.add(parts["look"], fm, i32=True)
            sc = self.scatter_parts.get(k)
            sc_arrs = None
            if sc is not None and total[("sc", k)] > 0:
                # chunk records {first message, first a row, first c row, packed}: message offset of the graph inside the batch instead
                # of inside the store, row offsets of the two operands' graphs; the packed field and the words travel unchanged
                sc_arrs = (B.add(sc["chunks_t"], ("sc", k), incs=B.rows_of(lay.dev[("inc", "sc", k)]), i32=True, transposed=True),
                           B.add(sc["words"], fm, i32=True).reshape(-1),
                           B.add(sc["blk_e"], "graph", incs=B.rows_of(lay.dev[("inc", "blk", k)]), i32=True, transposed=True),
                           B.add(sc["cgap"], ("sc", k), i32=True).reshape(-1) if "cgap" in sc else None)

            fu = self.fused_parts.get(k)
            fu_arrs = None
            if fu is not None and total[("fu", k)] > 0:
                fu_arrs = (B.add(fu["chunks_t"], ("fu", k), incs=(B.off(fm), B.off("tup"), B.off("tup")), i32=True, transposed=True),
                           B.add(fu["own"], ("fu", k), i32=True).reshape(-1))

            def install(k=k, acd=acd, arrs=arrs, sc=sc, sc_arrs=sc_arrs, fu_arrs=fu_arrs, sizes=(total[fa], total[fc], total[fd]), parts=parts):
                plan = _ops.MessagePlan.from_arrays(acd, *sizes, arrs["acd32"], arrs["ptr_a"], arrs["ptr_c"], arrs["perm_c"], arrs["by_c"],
                                                    arrs["ptr_d"], arrs["perm_d"], arrs["by_d"], volatile=False)
                plan.fwd._memo = {"max_len": int(parts["h_max_a"][lay.ids_h].max()) if lay.g else 0}
                if "look" in arrs and arrs["look"].shape[1]:
                    plan._lookup = (ea, (arrs["look"][0], arrs["look"][1]))     # A's values as a lookup of the edge feature
                _ops.install_message_plan(acd, plan)
                if sc_arrs is not None:
                    ch, words, blk_e, cgap = sc_arrs
                    _ops.install_scatter_plan(plan, lay.dev[("optr32", ("sc", k))], blk_e, ch, words, sc["max_edges"], sc["covers"], cgap,
                                              bool(sc["h_cok"][lay.ids_h].all()) if cgap is not None else False)
                if fu_arrs is not None:
                    _ops.install_fused_plan(plan, *fu_arrs)
            finish.append(install)
        B.launch()                                  # every array requested above: ONE kernel
        for f in finish:
            f()
        return dd


class _Layout:
    """where everything of one batch goes, computed on the host from the store's per-graph lengths: per length family (nodes, edges,
    tuples, the triples and scatter chunks of every key, and "graph" = one item per graph) the total, the running offsets of the
    selected graphs in the batch (`optr`) and their first columns in the store (`start`), plus the per-graph increments of every
    index array.  All of it reaches the device in one upload (`dev[...]` are views of that one tensor)."""

    def __init__(self, store: "DeviceGraphStore", ids_h: np.ndarray):
        self.ids_h, self.g = ids_h, int(ids_h.size)
        g = self.g
        fams = list(store.h_len.keys())
        optr = {f: np.concatenate(([0], np.cumsum(store.h_len[f][ids_h]))).astype(np.int64) for f in fams}
        optr["graph"] = np.arange(g + 1, dtype=np.int64)
        self.total = {f: int(v[-1]) for f, v in optr.items()}
        off = {f: v[:-1] for f, v in optr.items()}
        pieces = {("ids",): ids_h, ("len", "node"): store.h_len["node"][ids_h], ("start", "graph"): ids_h}
        for f in fams:
            pieces[("start", f)] = store.h_ptr[f][ids_h]
        for f, v in optr.items():
            pieces[("optr", f)] = v
        fam_of = lambda role: "tup" if role[0] == "X" else "edge"
        pieces[("inc", "ei")] = np.tile(off["node"], (2, 1))
        pieces[("inc", "tid")] = np.tile(off["node"], (store.sd, 1))
        for k in store.keys:
            roles = parse_key(k)
            pieces[("inc", "acd", k)] = np.stack([off[fam_of(roles[i])] for i in (0, 1, 3)])
            if k in store.scatter_parts:
                pieces[("inc", "sc", k)] = np.stack([off[("acd", k)], off[fam_of(roles[0])], off[fam_of(roles[1])], np.zeros(g, dtype=np.int64)])
                pieces[("inc", "blk", k)] = np.stack([off[fam_of(roles[3])], np.zeros(g, dtype=np.int64)])
        wide = [np.ascontiguousarray(p, dtype=np.int64).reshape(-1) for p in pieces.values()]
        # int32 copies of the pointers that ARE plan arrays (nodes by graph; the scatter plans' chunk pointers), behind the int64 part
        narrow = {("optr32", f): optr[f].astype(np.int32) for f in ["node"] + [("sc", k) for k in store.scatter_parts]}
        tail = np.concatenate([v for v in narrow.values()]) if narrow else np.zeros(0, dtype=np.int32)
        if tail.size % 2:
            tail = np.concatenate((tail, np.zeros(1, dtype=np.int32)))
        buf = np.concatenate(wide + [tail.view(np.int64)])
        dev = torch.from_numpy(buf).to(store.device)
        self.dev, o = {}, 0
        for name, p in pieces.items():
            self.dev[name] = dev[o:o + p.size].view(p.shape)
            o += p.size
        for f, v in optr.items():
            self.dev[("off", f)] = self.dev[("optr", f)][:-1].reshape(1, -1)          # running offsets as a (1, g) increment
        t32 = dev[o:].view(torch.int32)
        o = 0
        for name, v in narrow.items():
            self.dev[name] = t32[o:o + v.size]
            o += v.size


class BatchPrefetcher:
    """Iterate device-collated batches with their index plans ALREADY BUILT, one batch ahead of the consumer.

    Every kernel of the backend works from int32 / CSR plans derived from a batch's index tensors (narrowed copies, transposed
    groupings, long-segment hierarchies ...), built on first use and cached on those tensors.  First use costs ~40 host
    synchronisations (sizes of sorted outputs, error flags); inside a training step they stall the launch queue: a step on a
    fresh 8192-graph batch took 17.8 ms against 13.4 ms on a batch whose plans exist.  Here batch k + 1 is collated on a SIDE
    stream while the consumer trains on batch k and its plans are built there, so the synchronisations wait for the side stream's
    own few kernels instead of the training step's queue.

    Since round 4 a collated batch carries the plans of the shipped sparse layers' steps itself (message / scatter plans per key,
    tuples by root, nodes by graph, the tuple set's mirror; `DeviceGraphStore.collate`), and collation reads nothing back, so for
    those models `prepare` is not needed.  ``prepare(datadict)`` remains for anything else that groups by a batch's index tensors
    (``SpModel.prepare``: the model's index-consuming operators run once on width-8 dummy features; the plans depend on the index
    tensors only, so they land in the caches the real step will hit) -- it runs on the side stream as well.  Batches are kept
    alive until the consumer's stream has passed the point where the next batch was requested (their memory belongs to the side
    stream's allocator pool).
    """

    def __init__(self, store: DeviceGraphStore, id_batches, prepare=None, gated: bool = False):
        """`gated`: the consumer marks, once per step, the point of ITS stream from which the NEXT batch's device work (the offsets'
        upload and the collate kernel: ~0.4 ms of index traffic at 8192 graphs) may run -- `gate()`, e.g. from a forward pre-hook on
        the model's first graph-level module: the stretch of small launches between the last tuple-level forward kernel and the first
        tuple-level backward kernel leaves most of the chip idle.  Ungated, the collate kernel starts the moment it is queued, i.e.
        against whatever tuple-level kernel the previous step is in, and costs the step what it takes."""
        self.store, self.id_batches, self.prepare, self.gated = store, id_batches, prepare, bool(gated)
        self._gate = None
        # high priority: the side stream issues a few dozen tiny kernels with host synchronisations in between; behind the training
        # stream's saturating kernels each of them would wait for a scheduling slot (measured: 11.5 ms per batch against 4.8 ms idle)
        self.side = torch.cuda.Stream(device=store.device, priority=-1)
        self._pending = []          # (datadict, event on the consumer stream after which it may be freed)

    def _produce(self, ids):
        # no dependency on the consumer's stream: the store's arrays are static and the ids come from the host, so the side
        # stream's synchronisations wait for its own few kernels only (waiting for the consumer stream here would put every one
        # of them behind the training step that is still executing)
        if self._gate is not None:
            self.side.wait_event(self._gate)            # not before the consumer's stream has reached the marked point
            self._gate = None
        with torch.cuda.stream(self.side), _ops.deferred_index_checks():
            dd = self.store.collate(ids)
            if self.prepare is not None:
                self.prepare(dd)
        with torch.cuda.stream(self.side):
            ev = self.side.record_event()
        return dd, ev

    def gate(self) -> None:
        """mark the current point of the consumer's stream: the next batch's device work starts no earlier (see `gated`)"""
        if self.gated:
            self._gate = torch.cuda.current_stream(self.store.device).record_event()

    def __iter__(self):
        it = iter(self.id_batches)
        try:
            nxt = self._produce(next(it))
        except StopIteration:
            return
        while nxt is not None:
            dd, ev = nxt
            main = torch.cuda.current_stream(self.store.device)
            main.wait_event(ev)
            if not self.gated:
                try:
                    nxt = self._produce(next(it))       # overlaps with the consumer's work on `dd`
                except StopIteration:
                    nxt = None
                yield dd
            else:
                # the consumer queues its step on `dd` first (and calls gate() somewhere inside it); the next batch is queued behind
                # that mark.  The host runs about a step ahead of the device, so the side stream still has the batch ready in time
                yield dd
                try:
                    nxt = self._produce(next(it))
                except StopIteration:
                    nxt = None
            self._pending.append((dd, main.record_event()))
            self._pending = [(d, e) for d, e in self._pending if not e.query()]
        torch.cuda.current_stream(self.store.device).synchronize()
        self._pending.clear()

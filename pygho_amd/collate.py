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


def _cat32(arrs: List[np.ndarray], axis: int, device) -> torch.Tensor:
    a = np.concatenate(arrs, axis=axis)
    assert a.size == 0 or (a.min() >= 0 and a.max() < 2 ** 31)
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(device)


def _ptr64(lengths: Sequence[int], device) -> torch.Tensor:
    return torch.from_numpy(np.concatenate(([0], np.cumsum(np.asarray(lengths, dtype=np.int64))))).to(device)


class DeviceGraphStore:
    """all graphs of a dataset on the device: graph-local int32 indices + int64 segment pointers per graph."""

    def __init__(self, records: List[GraphRecord], device):
        self.device = torch.device(device)
        self.num_graphs = len(records)
        self.keys = list(records[0].acd.keys())
        self.sd = records[0].tupleid.shape[0]
        d = self.device
        self.node_ptr = _ptr64([r.num_nodes for r in records], d)
        self.edge_ptr = _ptr64([r.edge_index.shape[1] for r in records], d)
        self.tup_ptr = _ptr64([r.tupleid.shape[1] for r in records], d)
        self.x = _cat32([r.x.reshape(1, -1) for r in records], 1, d)
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
            for r in records:
                a, c, dd_ = r.acd[k]
                perm_c.append(np.argsort(c, kind="stable").reshape(1, -1))
                perm_d.append(np.argsort(dd_, kind="stable").reshape(1, -1))
                cnt_a.append(np.bincount(a, minlength=rows_of(r, roles[0])).reshape(1, -1))
                cnt_c.append(np.bincount(c, minlength=rows_of(r, roles[1])).reshape(1, -1))
                cnt_d.append(np.bincount(dd_, minlength=rows_of(r, roles[3])).reshape(1, -1))
            self.plan_parts[k] = {"perm_c": _cat32(perm_c, 1, d), "perm_d": _cat32(perm_d, 1, d), "cnt_a": _cat32(cnt_a, 1, d),
                                  "cnt_c": _cat32(cnt_c, 1, d), "cnt_d": _cat32(cnt_d, 1, d)}
        # the by-edge gradient's scatter plans (csrc/seg_scatter.hip), likewise ONCE for the whole store: the device planner over the
        # graph-local triples with one block per graph.  A batch's chunks are its graphs' chunks with the message / row offsets added
        # and its packed words are its graphs' words unchanged (they are relative to chunk windows and block edge ranges)
        self.scatter_parts = {}
        for k in self.plan_parts:
            acd = self.acd[k]
            if acd.shape[1] == 0 or acd.shape[1] >= (1 << 31):
                continue
            block_m = self.acd_ptr[k].to(torch.int32)
            parts = _ops.scatter_plan_parts(acd[0].contiguous(), acd[1].contiguous(), acd[2].contiguous(), block_m)
            if parts is None:
                continue
            n_chunks, chunk0, blk_e, chunks, words, max_edges = parts
            roles = parse_key(k)
            rows3 = (self.tup_ptr if roles[3][0] == "X" else self.edge_ptr)
            n_rows3 = (rows3[1:] - rows3[:-1]).to(torch.int32)
            covers = bool(((blk_e[:, 0] == 0) & (blk_e[:, 1] == n_rows3)).all())        # every graph's triples reach all of its rows
            self.scatter_parts[k] = {"chunk_ptr": chunk0.to(torch.int64), "chunks_t": chunks.t().contiguous(), "words": words.reshape(1, -1),
                                     "blk_e": blk_e.t().contiguous(), "max_edges": max_edges, "covers": covers}
        self.y = torch.tensor([r.y for r in records], dtype=torch.float32, device=d)

    # ------------------------------------------------------------------
    def _rows(self, src: torch.Tensor, seg_ptr: torch.Tensor, ids: torch.Tensor, out_ptr: torch.Tensor, total: int,
              inc: Union[torch.Tensor, None]) -> torch.Tensor:
        rows = src.shape[0]
        out = torch.empty((rows, total), dtype=torch.int64, device=self.device)
        start = _ops.gather_cols(seg_ptr, ids)
        check(lib().pygho_collate_rows(ptr(out), ptr(src), rows, src.shape[1], total, ptr(start), ptr(out_ptr),
                                       ptr(None if inc is None else inc.contiguous()), ids.numel(), total,
                                       stream_ptr(self.device)), "collate_rows")
        return out

    def collate(self, graph_ids: Union[Sequence[int], torch.Tensor]) -> Dict:
        """datadict of the block-diagonal batch of ``graph_ids`` (any order, repeats allowed)."""
        ids = torch.as_tensor(graph_ids, dtype=torch.int64).to(self.device).contiguous()
        require_device(ids)
        g = ids.numel()
        lens = {"node": self.node_ptr, "edge": self.edge_ptr, "tup": self.tup_ptr, **{("acd", k): v for k, v in self.acd_ptr.items()},
                **{("sc", k): v["chunk_ptr"] for k, v in self.scatter_parts.items()}}
        ptrs = {}
        for name, sp in lens.items():
            ptrs[name] = _ops.exclusive_scan(_ops.gather_cols(sp, ids + 1) - _ops.gather_cols(sp, ids))
        totals = torch.stack([p[-1] for p in ptrs.values()]).tolist()                   # ONE host sync sizes every output
        total = dict(zip(ptrs.keys(), (int(t) for t in totals)))
        off = {name: ptrs[name][:-1] for name in ("node", "edge", "tup")}                # running offsets per selected graph
        n = total["node"]
        ei = self._rows(self.edge_index, self.edge_ptr, ids, ptrs["edge"], total["edge"], off["node"].repeat(2, 1))
        ea = self._rows(self.edge_attr, self.edge_ptr, ids, ptrs["edge"], total["edge"], None).reshape(-1)
        tid = self._rows(self.tupleid, self.tup_ptr, ids, ptrs["tup"], total["tup"], off["node"].repeat(self.sd, 1))
        tf = self._rows(self.tuplefeat, self.tup_ptr, ids, ptrs["tup"], total["tup"], None)
        tf = tf.reshape(-1) if not self.feat_shape else tf.t().contiguous().reshape((total["tup"],) + self.feat_shape)
        x = self._rows(self.x, self.node_ptr, ids, ptrs["node"], n, None).reshape(-1)
        counts = ptrs["node"][1:] - ptrs["node"][:-1]
        batch, _ = _ops.expand_pairs(torch.zeros_like(counts), counts)
        dd = {
            "x": x, "batch": batch, "num_graphs": g, "y": self.y[ids], "num_nodes": n,
            "A": SparseTensor(ei, ea, [n, n], is_coalesced=True),
            "X": SparseTensor(tid, tf, [n] * self.sd + list(self.feat_shape), is_coalesced=True),
        }
        for k in self.keys:
            roles = parse_key(k)
            inc = torch.stack([off["tup"] if roles[i][0] == "X" else off["edge"] for i in (0, 1, 3)])
            acd = self._rows(self.acd[k], self.acd_ptr[k], ids, ptrs[("acd", k)], total[("acd", k)], inc)
            dd[k + KEYSEP + "acd"] = acd
            # message plan from the stored per-graph groupings (no sort, no host sync): permutations get the message offset of
            # their graph, the per-row message counts are collated by the operand's rows and scanned into CSR pointers
            if k not in self.plan_parts:
                continue
            parts, m_off = self.plan_parts[k], ptrs[("acd", k)][:-1].reshape(1, -1)
            name = lambda role: "tup" if role[0] == "X" else "edge"
            sp = {"tup": self.tup_ptr, "edge": self.edge_ptr}

            def csr(cnt, role):
                c = self._rows(cnt, sp[name(role)], ids, ptrs[name(role)], total[name(role)], None).reshape(-1)
                return _ops.exclusive_scan(c).to(torch.int32)
            perm = lambda which: self._rows(parts[which], self.acd_ptr[k], ids, ptrs[("acd", k)], total[("acd", k)], m_off).reshape(-1).to(torch.int32)
            plan = _ops.MessagePlan.from_parts(acd, total[name(roles[0])], total[name(roles[1])], total[name(roles[3])],
                                               csr(parts["cnt_a"], roles[0]), csr(parts["cnt_c"], roles[1]), perm("perm_c"),
                                               csr(parts["cnt_d"], roles[3]), perm("perm_d"))
            _ops.install_message_plan(acd, plan)
            sc = self.scatter_parts.get(k)
            if sc is not None and total[("sc", k)] > 0:
                # chunk records {first message, first a row, first c row, packed}: message offset of the graph inside the batch instead
                # of inside the store, row offsets of the two operands' graphs; the packed field and the words travel unchanged
                first_m = ptrs[("acd", k)][:-1] - _ops.gather_cols(self.acd_ptr[k], ids)
                inc = torch.stack([first_m, off[name(roles[0])], off[name(roles[1])], torch.zeros_like(first_m)])
                ch = self._rows(sc["chunks_t"], sc["chunk_ptr"], ids, ptrs[("sc", k)], total[("sc", k)], inc)
                words = self._rows(sc["words"], self.acd_ptr[k], ids, ptrs[("acd", k)], total[("acd", k)], None).reshape(-1)
                e = sc["blk_e"][:, ids].to(torch.int64)
                blk_e = torch.stack([e[0] + off[name(roles[3])], e[1]], dim=1).to(torch.int32).contiguous()
                _ops.install_scatter_plan(plan, ptrs[("sc", k)].to(torch.int32), blk_e, ch.t().to(torch.int32).contiguous(),
                                          words.to(torch.int32), sc["max_edges"], sc["covers"])
        return dd


class BatchPrefetcher:
    """Iterate device-collated batches with their index plans ALREADY BUILT, one batch ahead of the consumer.

    Every kernel of the backend works from int32 / CSR plans derived from a batch's index tensors (narrowed copies, transposed
    groupings, long-segment hierarchies ...), built on first use and cached on those tensors.  First use costs ~40 host
    synchronisations (sizes of sorted outputs, error flags); inside a training step they stall the launch queue: a step on a
    fresh 8192-graph batch took 17.8 ms against 13.4 ms on a batch whose plans exist.  Here batch k + 1 is collated on a SIDE
    stream while the consumer trains on batch k and its plans are built there, so the synchronisations wait for the side stream's
    own few kernels instead of the training step's queue.

    ``prepare(datadict)`` builds the plans (``SpModel.prepare``: the model's index-consuming operators run once on width-8 dummy
    features; the plans depend on the index tensors only, so they land in the caches the real step will hit).  Batches are kept
    alive until the consumer's stream has passed the point where the next batch was requested (their memory belongs to the side
    stream's allocator pool).
    """

    def __init__(self, store: DeviceGraphStore, id_batches, prepare=None):
        self.store, self.id_batches, self.prepare = store, id_batches, prepare
        # high priority: the side stream issues a few dozen tiny kernels with host synchronisations in between; behind the training
        # stream's saturating kernels each of them would wait for a scheduling slot (measured: 11.5 ms per batch against 4.8 ms idle)
        self.side = torch.cuda.Stream(device=store.device, priority=-1)
        self._pending = []          # (datadict, event on the consumer stream after which it may be freed)

    def _produce(self, ids):
        # no dependency on the consumer's stream: the store's arrays are static and the ids come from the host, so the side
        # stream's synchronisations wait for its own few kernels only (waiting for the consumer stream here would put every one
        # of them behind the training step that is still executing)
        with torch.cuda.stream(self.side), _ops.deferred_index_checks():
            dd = self.store.collate(ids)
            if self.prepare is not None:
                self.prepare(dd)
        with torch.cuda.stream(self.side):
            ev = self.side.record_event()
        return dd, ev

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
            try:
                nxt = self._produce(next(it))           # overlaps with the consumer's work on `dd`
            except StopIteration:
                nxt = None
            yield dd
            self._pending.append((dd, main.record_event()))
            self._pending = [(d, e) for d, e in self._pending if not e.query()]
        torch.cuda.current_stream(self.store.device).synchronize()
        self._pending.clear()

"""
Fixed-capacity batch slots: ONE captured training step that serves every mini-batch.

The reference's loop draws a fresh shuffled 128-graph batch every step (example/minimal.py:119, :141-149; collated by
hodata/SpData.py:60-77).  At that size the GPU work of a step is ~1.2 ms and an eager step is bound by ~135 launches and the
Python / autograd work around them (3.4-3.7 ms).  A HIP graph removes that, but a captured launch has its sizes and addresses
baked in.  A `BatchSlot` makes them batch-independent:

* every array of the batch (the API's int64 index arrays, their int32 copies, all plan arrays: CSR pointers, permutations, the
  triples' coordinates in by-c / by-d order, lookup rows, mirror positions) lives in a STATIC buffer with a fixed CAPACITY per row
  family (nodes, edges, tuples, the messages of every precompute key), chosen from the dataset's per-graph sizes so that a random
  batch fits with overwhelming probability (`capacity_sigmas`; a batch that does not fit takes the eager path, `SlotStep`);
* a batch is written into the slot by ONE kernel (`pygho_collate_batch`: a device-resident descriptor table, one row per array)
  from ONE small upload (the selected graphs' offsets); columns past the batch's true sizes are padded -- index arrays with 0 (a
  valid row), CSR pointer arrays with the batch's message total, so that every pad row is an EMPTY segment;
* the true sizes stay on the device: kernels that reduce over rows (BatchNorm statistics and backward, weight / bias gradients,
  embedding gradients) read them there (`plans.row_families` -> the "_dyn" entry points); row-wise kernels and ATen elementwise
  ops run over the capacity, their pad rows are don't-care values that nothing reads; segment kernels never see a pad message.

`graphs.SlotStep` captures [collate kernel + training step] once over a slot and replays it per batch.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence, Union

import numpy as np
import torch

from . import _ops
from ._native import check, lib, stream_ptr
from .backend.SpTensor import SparseTensor
from .collate import CollateDesc as _Desc, DeviceGraphStore
from .synth import KEYSEP, parse_key

_I32, _I64 = torch.int32, torch.int64


def _slotted_family(store: DeviceGraphStore, fam) -> bool:
    """row families a slot holds: all but the by-edge scatter chunk lists, of which only ALIGNED ones are kept (they serve the fused
    backward's table-gradient form, csrc/seg_dual.hip; the scatter kernel itself needs per-block arrays a slot does not collate)"""
    if isinstance(fam, tuple) and fam[0] == "sc":
        return "cgap" in getattr(store, "scatter_parts", {}).get(fam[1], {})
    return True


def slot_capacities(store: DeviceGraphStore, n_graphs: int, sigmas: float = 4.5, align: int = 64) -> Dict:
    """capacity per row family for batches of `n_graphs` graphs drawn at random from the store: mean + `sigmas` standard deviations
    of the batch total (a sum of n_graphs draws), never more than the n_graphs largest graphs together, rounded up to `align`, and
    made pairwise distinct and distinct from n_graphs (the extent of dim 0 names the family, `plans.row_families`)."""
    caps, used = {}, {n_graphs, n_graphs + 1}
    for fam, lens in store.h_len.items():
        if not _slotted_family(store, fam):
            continue
        lens = np.asarray(lens, dtype=np.float64)
        worst = float(np.sort(lens)[::-1][:n_graphs].sum()) if lens.size else 0.0
        est = n_graphs * lens.mean() + sigmas * np.sqrt(n_graphs) * lens.std() if lens.size else 0.0
        cap = int(np.ceil(min(worst, est) / align) * align) if min(worst, est) > 0 else align
        while cap in used or cap + 1 in used:
            cap += align
        used.update((cap, cap + 1))
        caps[fam] = cap
    return caps


def batch_layout(h_len: np.ndarray, h_ptr: np.ndarray, cap_vec: np.ndarray, ids: np.ndarray) -> Optional[np.ndarray]:
    """the per-batch upload of a slot, computed on the host from per-graph lengths: rows [ids | 0..G | running offsets of every family
    (G + 1 entries, the last one = the batch's true total) | first store column of every selected graph per family], each row G + 1
    int64 -- or None when a family's total exceeds its capacity.  `h_len` / `h_ptr` are (families, store graphs): lengths and first
    columns.  Everything the collate kernel and the "_dyn" kernels need about a batch is in here (hodata/SpData.py:60-77's increments
    are the running offsets)."""
    g, nf = ids.shape[0], h_len.shape[0]
    optr = np.zeros((nf, g + 1), dtype=np.int64)
    np.cumsum(h_len[:, ids], axis=1, out=optr[:, 1:])
    if np.any(optr[:, -1] > cap_vec):
        return None
    lay = np.zeros((2 + 2 * nf, g + 1), dtype=np.int64)
    lay[0, :g] = ids
    lay[1] = np.arange(g + 1)
    lay[2:2 + nf] = optr
    lay[2 + nf:, :g] = h_ptr[:, ids]
    return lay


class BatchSlot:
    """static buffers for one mini-batch of exactly `n_graphs` graphs of `store` + the datadict over them (`.datadict`).

    ``fits(ids)``      -- do these graphs fit the capacities?
    ``upload(ids)``    -- host: the batch's offsets into a pinned staging buffer and an asynchronous copy to the device (current stream)
    ``launch()``       -- device: the collate kernel + the target gather, on the current stream; capturable
    ``collate(ids)``   -- upload + launch + `reset_caches()`: the eager form
    ``rows()``         -- context manager: kernels read the true row counts of this slot's families from the device
    """

    def __init__(self, store: DeviceGraphStore, n_graphs: int, capacities: Optional[Dict] = None, capacity_sigmas: float = 4.5):
        assert n_graphs >= 1
        self.store, self.g = store, int(n_graphs)
        dev = self.device = store.device
        self.caps = dict(capacities) if capacities is not None else slot_capacities(store, n_graphs, capacity_sigmas)
        g, sd = self.g, store.sd
        # the extent of dim 0 NAMES a row family (`plans.row_families`): capacities (and capacity + 1, the CSR pointer arrays) must be
        # pairwise distinct and distinct from the graph count -- two families of one extent would silently share a row count
        fams_ = [f for f in store.h_len if _slotted_family(store, f) and not (isinstance(f, tuple) and f[0] == "sc" and f not in self.caps)]
        missing = [f for f in fams_ if f not in self.caps]
        if missing:
            raise ValueError(f"BatchSlot: no capacity for the row families {missing}")
        extents = [g, g + 1] + [v for f in fams_ for v in (int(self.caps[f]), int(self.caps[f]) + 1)]
        if min(int(self.caps[f]) for f in fams_) < 1 or len(set(extents)) != len(extents):
            raise ValueError("BatchSlot: capacities must be >= 1, pairwise distinct (also capacity + 1) and different from the number of "
                             f"graphs and that + 1; got {self.caps} for {g} graphs (slot_capacities() builds a valid set)")
        fam_of = lambda role: "tup" if role[0] == "X" else "edge"
        # ---- the per-batch upload: rows of G + 1 int64 ------------------------------------------------------------------
        self.fams = list(fams_)
        names = ["ids", "arange"] + [("optr", f) for f in self.fams] + [("start", f) for f in self.fams]
        self.row_of = {n: i for i, n in enumerate(names)}
        self.lay_dev = torch.zeros((len(names), g + 1), dtype=_I64, device=dev)
        self._pinned = [torch.zeros((len(names), g + 1), dtype=_I64).pin_memory() for _ in range(4)]
        self._pin_events = [None] * len(self._pinned)
        self._turn = 0
        for t in self._pinned:
            t[self.row_of["arange"]] = torch.arange(g + 1)
        self._h_len = np.stack([np.asarray(store.h_len[f], dtype=np.int64) for f in self.fams])            # (F, num_graphs)
        self._h_ptr = np.stack([np.asarray(store.h_ptr[f], dtype=np.int64)[:-1] for f in self.fams])       # first store column per graph
        self._cap_vec = np.asarray([self.caps[f] for f in self.fams], dtype=np.int64)
        base, pitch = self.lay_dev.data_ptr(), (g + 1) * 8
        row_ptr = lambda name: base + self.row_of[name] * pitch
        off = lambda f: row_ptr(("optr", f))                           # running offsets of family f (first G entries)
        total = lambda f: row_ptr(("optr", f)) + g * 8                 # the batch's true total of family f (entry G)
        # one-element int32 views of the totals (little endian: the low half of the int64): what the "_dyn" kernels read
        self.counts = {f: self.lay_dev[self.row_of[("optr", f)], g:].view(_I32)[:1] for f in self.fams}
        self.ids_dev = self.lay_dev[self.row_of["ids"], :g]
        # ---- static buffers + descriptors -----------------------------------------------------------------------------------
        self._descs, self._keep = [], []
        zeros_graph = torch.zeros((1, store.num_graphs), dtype=_I32, device=dev)       # "one item per graph" source of constants
        zeros_node = torch.zeros((1, max(int(store.node_ptr[-1]), 1)), dtype=_I32, device=dev)
        self._keep += [zeros_graph, zeros_node]

        def out(rows, fam, i32, extra=0, transposed=False):
            cols = (g if fam == "graph" else self.caps[fam]) + extra
            return torch.zeros((cols, rows) if transposed else (rows, cols), dtype=_I32 if i32 else _I64, device=dev)

        def desc(dst, src, fam, incs=(), pad=None, transposed=False):
            """one array of the batch: `src` (rows, store length of `fam`) -> `dst` (rows, capacity [+ 1])"""
            d = _Desc()
            d.out, d.src = dst.data_ptr(), src.data_ptr()
            if fam == "graph":                     # one column per selected graph: column s comes from store column ids[s]
                d.src_start, d.out_ptr = row_ptr("ids"), row_ptr("arange")
            else:
                d.src_start, d.out_ptr = row_ptr(("start", fam)), row_ptr(("optr", fam))
            for r, inc in enumerate(incs):
                d.inc[r] = inc
            d.pad = pad
            d.src_ld = src.shape[1]
            d.rows = src.shape[0]
            d.out_ld = dst.shape[0] if transposed else dst.shape[1]
            d.out_i32, d.transposed = int(dst.dtype == _I32), int(transposed)
            assert (dst.shape[1] if transposed else dst.shape[0]) == d.rows and d.rows <= 64
            self._descs.append(d)
            return dst

        node_off, arange = off("node"), row_ptr("arange")
        # the API's arrays (int64) and their int32 copies
        self.x = desc(out(1, "node", False), store.x, "node").reshape(-1)
        self.x32 = desc(out(1, "node", True), store.x, "node").reshape(-1)
        self.batch = desc(out(1, "node", False), zeros_node, "node", incs=(arange,)).reshape(-1)
        self.batch32 = desc(out(1, "node", True), zeros_node, "node", incs=(arange,)).reshape(-1)
        self.ei = desc(out(2, "edge", False), store.edge_index, "edge", incs=(node_off, node_off))
        self.ei32 = desc(out(2, "edge", True), store.edge_index, "edge", incs=(node_off, node_off))
        self.ea = desc(out(1, "edge", False), store.edge_attr, "edge").reshape(-1)
        self.ea32 = desc(out(1, "edge", True), store.edge_attr, "edge").reshape(-1)
        self.tid = desc(out(sd, "tup", False), store.tupleid, "tup", incs=(node_off,) * sd)
        self.tid32 = desc(out(sd, "tup", True), store.tupleid, "tup", incs=(node_off,) * sd)
        f_rows = store.tuplefeat.shape[0]
        if store.feat_shape:
            self._tf_t = desc(out(f_rows, "tup", False, transposed=True), store.tuplefeat, "tup", transposed=True)
            self.tf = self._tf_t.reshape((self.caps["tup"],) + store.feat_shape)
            self.tf32 = None
        else:
            self.tf = desc(out(1, "tup", False), store.tuplefeat, "tup").reshape(-1)
            self.tf32 = desc(out(1, "tup", True), store.tuplefeat, "tup").reshape(-1)
        self.y = torch.zeros(g, dtype=store.y.dtype, device=dev)
        # nodes by graph: the running node offsets themselves, as int32 (entry G = the node total)
        self.graph_ptr = desc(out(1, "graph", True, extra=1), zeros_graph, "graph", incs=(node_off,), pad=total("node")).reshape(-1)
        # tuples by root, tuples by their other coordinates, edges by either endpoint: graph-local pointers + the graph's row offset
        self.root_ptr = None
        if store.root_parts is not None:
            self.root_ptr = desc(out(1, "node", True, extra=1), store.root_parts["ptr"], "node", incs=(off("tup"),), pad=total("tup")).reshape(-1)
        self.group = {}
        for (which, dim), part in store.group_parts.items():
            fam = "tup" if which == "X" else "edge"
            gp = desc(out(1, "node", True, extra=1), part["ptr"], "node", incs=(off(fam),), pad=total(fam)).reshape(-1)
            perm = desc(out(1, fam, True), part["perm"], fam, incs=(off(fam),)).reshape(-1) if "perm" in part else None
            self.group[(which, dim)] = (gp, perm)
        # diagonal positions + tuples per root / per second coordinate: the "sun_views" GNNAKConv / SUNConv cache on the tuple pattern
        # (honn/Conv.py; there a hash search and two bincounts over the index rows -- which a padded pattern cannot answer)
        self.diag_pos = self.cnt_r = self.cnt_c = None
        if getattr(store, "diag_parts", None) is not None and self.root_ptr is not None and ("X", 1) in self.group:
            # (pad nodes have NO diagonal tuple: position -1, so their diagonal rows read as zeros and receive a zero gradient -- the
            # node-level products of SUNConv are plain GEMMs over all capacity rows, which zero gradient rows leave alone)
            self._neg1 = torch.full((1,), -1, dtype=_I64, device=dev)
            self.diag_pos = desc(out(1, "node", False), store.diag_parts["pos"], "node", incs=(off("tup"),), pad=self._neg1.data_ptr()).reshape(-1)
            # tuples per root / per second coordinate, clamped like honn/Conv.py's bincounts: the store's per-node counts (clamped once,
            # here) travel with the batch in the collate kernel; pad nodes count 1  (round 5 derived them from the collated pointers
            # with six small launches per batch)
            self._one = torch.ones((1,), dtype=_I64, device=dev)
            self._cnt1 = [store.root_parts["cnt"].clamp_min(1), store.group_parts[("X", 1)]["cnt"].clamp_min(1)]
            self.cnt_r = desc(out(1, "node", False), self._cnt1[0], "node", pad=self._one.data_ptr()).reshape(-1, 1)
            self.cnt_c = desc(out(1, "node", False), self._cnt1[1], "node", pad=self._one.data_ptr()).reshape(-1, 1)
        # 3-tuple stores: the merged (i, j) pattern of pooling the last coordinate away, its CSR pointers over the tuples, the tuple ->
        # pair map and the pairs' grouping by root (collate.DeviceGraphStore.pair_parts); pad pairs are (0, 0) with empty segments
        self.pair = None
        pp = getattr(store, "pair_parts", None)
        if pp is not None:
            self.pair = {"index": desc(out(2, "pair", False), pp["index"], "pair", incs=(node_off, node_off)),
                         "ptr": desc(out(1, "pair", True, extra=1), pp["ptr"], "pair", incs=(off("tup"),), pad=total("tup")).reshape(-1),
                         "inv": desc(out(1, "tup", True), pp["inv"], "tup", incs=(off("pair"),)).reshape(-1),
                         "root_ptr": desc(out(1, "node", True, extra=1), pp["root_ptr"], "node", incs=(off("pair"),), pad=total("pair")).reshape(-1)}
        self.mirror = None
        if store.mirror_parts is not None and bool(store.mirror_parts["h_ok"].all()):
            self.mirror = desc(out(1, "tup", True), store.mirror_parts["pos"], "tup", incs=(off("tup"),)).reshape(-1)
        self.msg = {}
        for k in store.keys:
            roles = parse_key(k)
            fa, fc, fd, fm = fam_of(roles[0]), fam_of(roles[1]), fam_of(roles[3]), ("acd", k)
            ent = {"acd": desc(out(3, fm, False), store.acd[k], fm, incs=(off(fa), off(fc), off(fd)))}
            parts = store.plan_parts.get(k)
            if parts is not None:
                ent["acd32"] = desc(out(3, fm, True), store.acd[k], fm, incs=(off(fa), off(fc), off(fd)))
                ent["ptr_a"] = desc(out(1, fa, True, extra=1), parts["ptr_a"], fa, incs=(off(fm),), pad=total(fm)).reshape(-1)
                ent["ptr_c"] = desc(out(1, fc, True, extra=1), parts["ptr_c"], fc, incs=(off(fm),), pad=total(fm)).reshape(-1)
                ent["ptr_d"] = desc(out(1, fd, True, extra=1), parts["ptr_d"], fd, incs=(off(fm),), pad=total(fm)).reshape(-1)
                ent["perm_c"] = desc(out(1, fm, True), parts["perm_c"], fm, incs=(off(fm),)).reshape(-1)
                ent["perm_d"] = desc(out(1, fm, True), parts["perm_d"], fm, incs=(off(fm),)).reshape(-1)
                ent["by_c"] = desc(out(2, fm, True), parts["by_c"], fm, incs=(off(fa), off(fd)))
                ent["by_d"] = desc(out(2, fm, True), parts["by_d"], fm, incs=(off(fa), off(fc)))
                if "look" in parts:
                    ent["look"] = desc(out(2, fm, True), parts["look"], fm)
                fu = store.fused_parts.get(k)
                if fu is not None:
                    # the fused block forward's chunk list (csrc/seg_fused.hip): graph-local records + the graph's message / tuple offsets;
                    # the columns beyond the batch's chunk count stay all-zero records, which end a workgroup's share of the list
                    ff = ("fu", k)
                    ent["fu_chunks"] = desc(out(4, ff, True, transposed=True), fu["chunks_t"], ff, incs=(off(fm), off("tup"), off("tup")),
                                            transposed=True)
                    ent["fu_own"] = desc(out(1, ff, True), fu["own"], ff).reshape(-1)
                sc = store.scatter_parts.get(k)
                fs = ("sc", k)
                if sc is not None and "cgap" in sc and fs in self.caps and "look" in parts and fa == fc == "tup":
                    # the fused backward's chunk list (csrc/seg_dual.hip, table-gradient form): records {first message, first output row,
                    # first first-operand row, packed} + the graph's offsets, one packed word per message, the rows without messages a
                    # chunk owns; records beyond the batch's chunk count stay all-zero (they end a workgroup's share), the true count
                    # stays on the device (`counts`)
                    ent["sc_chunks"] = desc(out(4, fs, True, transposed=True), sc["chunks_t"], fs, incs=(off(fm), off(fa), off(fc)), transposed=True)
                    ent["sc_words"] = desc(out(1, fm, True), sc["words"], fm).reshape(-1)
                    ent["sc_cgap"] = desc(out(1, fs, True), sc["cgap"], fs).reshape(-1)
            self.msg[k] = ent
        assert int(lib().pygho_collate_desc_bytes()) == ctypes.sizeof(_Desc)
        table = (_Desc * len(self._descs))(*self._descs)
        raw = np.frombuffer(memoryview(table).cast("B"), dtype=np.uint8).copy()
        self.desc_dev = torch.from_numpy(raw).to(dev)
        self.max_cols = max(int(d.out_ld) for d in self._descs)
        self._static = self._all_tensors()
        self.datadict = self._datadict()
        self.n_uploads = 0

    # ------------------------------------------------------------------
    def _all_tensors(self):
        ts = [self.x, self.x32, self.batch, self.batch32, self.ei, self.ei32, self.ea, self.ea32, self.tid, self.tid32, self.tf, self.tf32,
              self.y, self.graph_ptr, self.root_ptr, self.mirror, self.diag_pos, self.cnt_r, self.cnt_c]
        for gp, perm in self.group.values():
            ts += [gp, perm]
        if self.pair is not None:
            ts += list(self.pair.values())
        for ent in self.msg.values():
            ts += list(ent.values())
        return [t for t in ts if t is not None]

    def _datadict(self) -> Dict:
        g, n, sd = self.g, self.caps["node"], self.store.sd
        dd = {
            "x": self.x, "batch": self.batch, "num_graphs": g, "y": self.y, "num_nodes": n,
            "A": SparseTensor(self.ei, self.ea, [n, n], is_coalesced=True),
            "X": SparseTensor(self.tid, self.tf, [n] * sd + list(self.store.feat_shape), is_coalesced=True),
        }
        for k, ent in self.msg.items():
            dd[k + KEYSEP + "acd"] = ent["acd"]
        self._A, self._X = dd["A"], dd["X"]
        self._install()
        return dd

    def _install(self) -> None:
        """put the slot's plan arrays where the operators look for them (the caches that hang on the index tensors)"""
        st, n, g = self.store, self.caps["node"], self.g
        fam_of = lambda role: "tup" if role[0] == "X" else "edge"
        seed32 = lambda t64, t32: setattr(t64, "_pygho_i32", (t64._version, t32))
        for name, t in (("x", self.x), ("ea", self.ea), ("tf", self.tf)):
            t._pygho_value_bound = (t._version, st.h_vmax[name] + 1)
        for ind in (self.ei, self.tid):
            if n < (1 << (63 // ind.shape[0])):
                ind._pygho_hash_ok = ind._version
            ind._pygho_slot = True                      # hash-searching operators refuse padded index arrays (backend/SpTensor._hash)
        seed32(self.x, self.x32)
        seed32(self.ea, self.ea32)
        seed32(self.batch, self.batch32)
        if self.tf32 is not None:
            seed32(self.tf, self.tf32)
        rows_x = [self._X._row(dim) for dim in range(st.sd)]
        rows_a = [self._A._row(dim) for dim in range(2)]
        self._rows32_x = getattr(self, "_rows32_x", None) or [self.tid32[dim] for dim in range(st.sd)]      # the SAME view objects every time
        self._rows32_a = getattr(self, "_rows32_a", None) or [self.ei32[dim] for dim in range(2)]
        for r64, r32 in zip(rows_x + rows_a, self._rows32_x + self._rows32_a):
            seed32(r64, r32)

        def seg(ptr, perm, n_seg, m, max_len):
            p = _ops.SegPlan(ptr, perm, n_seg, m)
            p.volatile = True
            p._memo = {"max_len": int(max_len)}
            return p

        def serve(keys, plan):
            """the slot's grouping of `keys` under WHATEVER tag an operator asks for (`plans.cached_plan`: "scatter", "pair-row",
            "pair-col" ...): a plan built from the padded keys instead would count the pad columns into row 0's segment, and
            building one reads back from the device (not possible under capture)"""
            def factory(n_seg, plan=plan):
                if n_seg != plan.n_seg:
                    raise RuntimeError(f"pygho_amd: a grouping of a batch slot's index row into {n_seg} segments was asked for; the slot "
                                       f"holds its grouping into {plan.n_seg} (the node capacity)")
                return plan
            keys._pygho_plan_factory = (keys._version, factory)
        h_max = lambda lens: int(np.max(lens)) if np.size(lens) else 0
        serve(self.batch, seg(self.graph_ptr, None, g, n, h_max(st.h_len["node"])))
        for (which, dim), (gp, perm) in self.group.items():
            keys = (rows_x if which == "X" else rows_a)[dim]
            m = self.caps["tup" if which == "X" else "edge"]
            serve(keys, seg(gp, perm, n, m, h_max(st.group_parts[(which, dim)]["h_max"])))
        if self.pair is not None:
            pr, pp = self.pair, st.pair_parts
            pr["index"]._pygho_slot = True
            if n < (1 << 31):
                pr["index"]._pygho_hash_ok = pr["index"]._version
            self._X._cache()[("pool_sparse", (0, 1))] = (pr["index"], seg(pr["ptr"], None, self.caps["pair"], self.caps["tup"], h_max(pp["h_max"])),
                                                        pr["inv"])
            self._pair_row0 = getattr(self, "_pair_row0", None)
            if self._pair_row0 is None:
                self._pair_row0 = _ops.unbased(pr["index"][0])            # the SAME row object every time (plans hang on it)
            pr["index"]._pygho_cache = {"_v": pr["index"]._version, ("row", 0): self._pair_row0}
            serve(self._pair_row0, seg(pr["root_ptr"], None, n, self.caps["pair"], h_max(pp["h_root_max"])))
        if self.diag_pos is not None:
            self._X._cache()["sun_views"] = (self.diag_pos, self.cnt_r, self.cnt_c)
        if self.root_ptr is not None:
            serve(rows_x[0], seg(self.root_ptr, None, n, self.caps["tup"], h_max(st.root_parts["h_max"])))
            if self.mirror is not None and self.tf32 is not None and st.sd == 2:
                row32, col32, vidx32 = self._rows32_x[0], self._rows32_x[1], self.tf32
                row32._pygho_mirror = (col32, vidx32, n, self.mirror, (row32._version, col32._version, vidx32._version))
        for k, ent in self.msg.items():
            if "acd32" not in ent:
                continue
            roles = parse_key(k)
            na, nc, nd = (self.caps[fam_of(roles[i])] for i in (0, 1, 3))
            plan = _ops.MessagePlan.from_arrays(ent["acd"], na, nc, nd, ent["acd32"], ent["ptr_a"], ent["ptr_c"], ent["perm_c"], ent["by_c"],
                                                ent["ptr_d"], ent["perm_d"], ent["by_d"])
            plan.fwd._memo = {"max_len": h_max(st.plan_parts[k]["h_max_a"])}     # the store's longest forward segment (a bound for any batch)
            if "look" in ent and fam_of(roles[3]) == "edge":
                plan._lookup = (self.ea, (ent["look"][0], ent["look"][1]))      # A's values as a lookup of the edge feature
            if "fu_chunks" in ent:
                _ops.install_fused_plan(plan, ent["fu_chunks"], ent["fu_own"])
            if "sc_chunks" in ent:
                # covers_c: every first-operand row of every graph of the STORE belongs to a chunk (then the by-tuple gradient needs no
                # pre-fill; the slot's pad rows are don't-care rows like those of every row-wise result)
                sc = st.scatter_parts[k]
                _ops.install_scatter_plan(plan, None, None, ent["sc_chunks"], ent["sc_words"], int(sc["max_edges"]), False, ent["sc_cgap"],
                                          bool(np.all(sc["h_cok"])), n_dyn=self.counts[("sc", k)])
            _ops.install_message_plan(ent["acd"], plan)

    def reset_caches(self) -> None:
        """forget everything DERIVED from the slot's arrays (they were just rewritten in place, behind the version counters) and
        re-install the slot's own plans.  Needed after every eager `collate`; before a capture it makes the captured step recompute
        whatever it derives INSIDE the graph, so a replay recomputes it for the new batch."""
        keep = set()
        objs = list(self._static) + list(self._rows32_x) + list(self._rows32_a) + ([self._pair_row0] if getattr(self, "_pair_row0", None) is not None else [])
        for sp in (self._A, self._X):
            c = sp._cache()
            objs += [v for v in c.values() if isinstance(v, torch.Tensor)]
        for t in objs:
            for name in [a for a in vars(t) if a.startswith("_pygho")] if hasattr(t, "__dict__") else []:
                if name not in keep:
                    delattr(t, name)
        for sp in (self._A, self._X):
            c = sp._cache()
            for key in [key for key in c if not (isinstance(key, tuple) and key[0] == "row")]:
                del c[key]
        self._install()

    # ------------------------------------------------------------------
    def layout(self, ids: np.ndarray) -> Optional[np.ndarray]:
        """(rows, G + 1) int64 upload of the batch `ids`, or None when it does not fit the capacities"""
        assert ids.shape == (self.g,), f"a slot of {self.g} graphs was given {ids.shape[0]}"
        return batch_layout(self._h_len, self._h_ptr, self._cap_vec, ids)

    def fits(self, graph_ids) -> bool:
        ids = _as_ids(graph_ids, self.store.num_graphs)
        return ids.shape[0] == self.g and bool(np.all(self._h_len[:, ids].sum(axis=1) <= self._cap_vec))

    def upload(self, graph_ids) -> bool:
        """host side of a batch: False (nothing uploaded) when the graphs do not fit"""
        ids = _as_ids(graph_ids, self.store.num_graphs)
        lay = self.layout(ids)
        if lay is None:
            return False
        i = self._turn = (self._turn + 1) % len(self._pinned)
        if self._pin_events[i] is not None:
            self._pin_events[i].synchronize()                           # the copy that last read this staging buffer has run
        self._pinned[i].numpy()[...] = lay
        self.lay_dev.copy_(self._pinned[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._pin_events[i] = ev
        self.n_uploads += 1
        return True

    def launch(self) -> None:
        """device side of a batch: every array in one kernel, then the targets (capturable; reads the uploaded layout)"""
        check(lib().pygho_collate_batch(self.desc_dev.data_ptr(), len(self._descs), self.g, self.max_cols, stream_ptr(self.device)),
              "collate_batch")
        torch.index_select(self.store.y, 0, self.ids_dev, out=self.y)

    def collate(self, graph_ids) -> Dict:
        """eager use of the slot: write the batch, drop what earlier batches left in the caches; returns `.datadict`"""
        if not self.upload(graph_ids):
            raise ValueError("pygho_amd: this batch does not fit the slot's capacities (use DeviceGraphStore.collate)")
        self.launch()
        self.reset_caches()
        return self.datadict

    def rows(self):
        """`with slot.rows(): step()` -- row counts of the slot's families are read on the device"""
        return _ops.row_families({self.caps[f]: self.counts[f] for f in self.fams})

    def true_sizes(self) -> Dict:
        """the uploaded batch's true totals per family (a host read: tests / debugging)"""
        row = self.lay_dev[:, self.g].tolist()
        return {f: int(row[self.row_of[("optr", f)]]) for f in self.fams}


def _as_ids(graph_ids: Union[Sequence[int], torch.Tensor, np.ndarray], num_graphs: int) -> np.ndarray:
    ids = (graph_ids.detach().cpu().numpy() if isinstance(graph_ids, torch.Tensor) else np.asarray(graph_ids)).astype(np.int64).reshape(-1)
    assert ids.size == 0 or (ids.min() >= 0 and ids.max() < num_graphs), "graph id out of range"
    return ids

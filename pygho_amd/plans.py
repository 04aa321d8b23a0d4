"""
Index plans of the HIP path: int32 / CSR views of the int64 index arrays of the Python API (built once per index tensor and cached
on it), the deferred range checks that ride on the next host fetch, and the integer planner primitives (hash pack / unpack,
sorted match, radix sort, run ids, scans) the device planner of backend/Spspmm.py is written with.  ROCm device memory only.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from ._native import AGGR_CODE, DTYPE_CODE, check, dtype_code, lib, ptr, require_device, stream_ptr

_I32 = torch.int32
ACT_CODE = {"none": 0, "relu": 1, "silu": 2}      # activation codes of the C ABI (bn_act / rowblock_linear / seg_*_act)


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def _flag(dev) -> Tensor:
    return torch.zeros(1, dtype=_I32, device=dev)


# --------------------------------------------------------------------------
# row families: row counts that live on the device (fixed-capacity batch slots, `collate.BatchSlot` / `graphs.SlotStep`)
# --------------------------------------------------------------------------
# A HIP graph captured once serves every mini-batch when its tensors have a fixed CAPACITY of rows per family (nodes, edges, tuples,
# messages of a key) and the kernels that reduce over rows read the TRUE count from the device (the "_dyn" entry points of
# include/pygho_hip.h).  Inside `with row_families({capacity: count}):` a launch wrapper that is handed `capacity` rows passes the
# family's device-side count along; capacities are made pairwise distinct (and distinct from the batch's graph count) by the slot,
# so the extent of dim 0 names the family.  Rows past the count hold don't-care values: row-wise kernels compute them, nothing
# that sums over rows reads them.
_ROW_FAMILIES = {}      # capacity (int) -> one-element int32 device tensor with the true row count


class row_families:
    def __init__(self, families):
        self.families = dict(families)

    def __enter__(self):
        self.prev = dict(_ROW_FAMILIES)
        for cap, cnt in self.families.items():
            assert cnt.dtype == _I32 and cnt.numel() >= 1 and cnt.is_cuda
            _ROW_FAMILIES[int(cap)] = cnt
        return self

    def __exit__(self, *exc):
        _ROW_FAMILIES.clear()
        _ROW_FAMILIES.update(self.prev)
        return False


def dyn_rows(m: int) -> Optional[Tensor]:
    """the device-side row count behind a dim-0 extent of `m` (None: `m` is the true count)"""
    return _ROW_FAMILIES.get(m) if _ROW_FAMILIES else None


def require_static_rows(m: int, what: str) -> None:
    """a path that sums over rows WITHOUT a device-count form was reached with a capacity-sized tensor: fail loudly (the pad rows
    would enter the sum)"""
    if _ROW_FAMILIES and m in _ROW_FAMILIES:
        raise RuntimeError(f"pygho_amd: {what} has no device-side row-count form, but its input has the capacity ({m} rows) of a "
                           "batch slot's row family; run this step eagerly on `DeviceGraphStore.collate` batches instead")


def narrow_i32(x: Tensor, checked: bool = False, bound: Optional[int] = None, bound_msg: str = "") -> Tensor:
    """int64 -> int32 copy on the device (cached on the source tensor object).  `bound`: the values must lie in [0, bound) -- checked
    by the narrowing pass itself, reported as a DEFERRED error (`defer_error`: the flag rides on the next host fetch)."""
    if x.dtype == _I32:
        if bound is not None and x.numel():
            lo, hi = torch.aminmax(x)
            defer_error(((lo < 0) | (hi >= bound)).to(_I32).reshape(1), bound_msg or f"pygho_amd: index outside [0, {bound})")
        return x.contiguous()
    cache = getattr(x, "_pygho_i32", None)
    if cache is not None and cache[0] == x._version:
        # the smallest bound this copy was already checked against rides in the cache entry: a repeated request with the same or a
        # larger bound launches nothing and defers nothing (every MessagePlan over the same triples re-ran aminmax + three
        # elementwise ops and grew the deferred-error list)
        checked_bound = cache[2] if len(cache) > 2 else None
        if bound is not None and x.numel() and (checked_bound is None or bound < checked_bound):
            lo, hi = torch.aminmax(cache[1])
            defer_error(((lo < 0) | (hi >= bound)).to(_I32).reshape(1), bound_msg or f"pygho_amd: index outside [0, {bound})")
            try:
                x._pygho_i32 = (cache[0], cache[1], bound)
            except Exception:
                pass
        return cache[1]
    dev = require_device(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=_I32, device=dev)
    if bound is not None and 0 <= bound <= (1 << 31):
        err = _flag(dev)
        check(lib().pygho_narrow_i64_i32_bounded(ptr(out), ptr(x), x.numel(), int(bound), ptr(err), stream_ptr(dev)), "narrow_i64_i32_bounded")
        if x.numel():
            defer_error(err, bound_msg or f"pygho_amd: index outside [0, {bound})")
        try:
            x._pygho_i32 = (x._version, out, int(bound))
        except Exception:
            pass
        return out
    err = _flag(dev) if checked else None
    check(lib().pygho_narrow_i64_i32(ptr(out), ptr(x), x.numel(), ptr(err), stream_ptr(dev)), "narrow_i64_i32")
    if checked and int(err.item()) != 0:
        raise ValueError("pygho_amd: index does not fit int32 or is negative")
    try:
        x._pygho_i32 = (x._version, out)
    except Exception:
        pass
    return out


def unbased(t: Tensor) -> Tensor:
    """`t` without the link to the tensor it is a view of (same storage, same version counter).  A view that is CACHED ON ITS BASE
    (``base._pygho_x = base[0]``) closes a reference cycle through the view's C++-side `_base` pointer that Python's collector
    cannot see: the base, and everything cached on it, is never freed -- every training batch leaked its index arrays and plans
    that way (4.6 MB per 128-graph batch, 330 MB per 8192-graph batch) until round 4's soak run."""
    return t.detach() if t._base is not None else t


def flat_index(idx: Tensor) -> Tensor:
    """the 1-D contiguous form of an integer feature tensor as a PERSISTENT object (gather plans are cached on it)"""
    if idx.dim() == 1 and idx.is_contiguous():
        return idx                                           # it is its own flat form (and must not be cached on itself)
    if not hasattr(idx, "_pygho_flat") or idx._pygho_flat[0] != idx._version:
        idx._pygho_flat = (idx._version, unbased(idx.reshape(-1).contiguous()))
    return idx._pygho_flat[1]


def gather_i32(table: Tensor, idx: Tensor) -> Tensor:
    dev = require_device(table, idx)
    out = torch.empty(idx.shape, dtype=_I32, device=dev)
    check(lib().pygho_gather_i32(ptr(out), ptr(table), ptr(idx), idx.numel(), stream_ptr(dev)), "gather_i32")
    return out


class SegPlan:
    """CSR grouping of `m` messages into `n_seg` segments: ``seg_ptr`` (n_seg+1) int32 and
    ``perm`` (m) int32 = message ids in grouped order (None when the key array was already
    sorted, i.e. grouped order == message order)."""
    __slots__ = ("seg_ptr", "perm", "n_seg", "m", "_inv_cnt", "_memo", "_partner", "volatile")

    def __init__(self, seg_ptr: Tensor, perm: Optional[Tensor], n_seg: int, m: int):
        self.seg_ptr, self.perm, self.n_seg, self.m = seg_ptr, perm, n_seg, m
        self._inv_cnt = None
        self._memo = None
        self._partner = None         # (key, index arrays in grouped order) of the last three-operand user
        self.volatile = False        # the arrays are REWRITTEN IN PLACE per batch (a batch slot): nothing derived from them is kept

    @property
    def inv_count(self) -> Tensor:
        """1 / max(segment length, 1) as f32 (mean backward)."""
        if self._inv_cnt is None or self.volatile:
            cnt = (self.seg_ptr[1:] - self.seg_ptr[:-1]).clamp_min(1)
            inv = cnt.to(torch.float32).reciprocal()
            if self.volatile:
                return inv
            self._inv_cnt = inv
        return self._inv_cnt

    def take(self, idx32: Tensor) -> Tensor:
        """idx32 re-ordered into grouped order."""
        return idx32 if self.perm is None else gather_i32(idx32, self.perm)

    _HOST_LENS_LIMIT = 8192      # plans with at most this many segments fetch their lengths once and plan hierarchies on the host

    def _host_lens(self):
        """segment lengths on the host for small plans (embedding tables, feature types: a handful of segments, possibly very
        long): ONE synchronisation serves `max_len` and every level of `levels`."""
        if self._memo is None:
            self._memo = {}
        if "host_lens" not in self._memo:
            import numpy as _np
            self._memo["host_lens"] = _np.diff(_np.asarray(_fetch(self.seg_ptr), dtype=_np.int64))
        return self._memo["host_lens"]

    @property
    def max_len(self) -> int:
        """longest segment (one host sync, cached)."""
        if self._memo is None:
            self._memo = {}
        if "max_len" not in self._memo:
            if self.n_seg == 0:
                self._memo["max_len"] = 0
            elif self.n_seg <= self._HOST_LENS_LIMIT:
                self._memo["max_len"] = int(self._host_lens().max())
            else:
                self._memo["max_len"] = int(_fetch((self.seg_ptr[1:] - self.seg_ptr[:-1]).max().reshape(1))[0])
        return self._memo["max_len"]

    def levels(self, limit: int):
        """CSR pointers of a hierarchical reduction whose segments never exceed `limit` items: level 0
        groups the messages into bounded chunks, every further level groups the previous level's partial
        rows, the last one into the n_seg output segments.  A lane group walks its segment sequentially, so
        an unbounded segment (a 4-row embedding table receiving 10^6 gradient rows) would serialise."""
        if self._memo is None:
            self._memo = {}
        key = ("levels", limit)
        if key not in self._memo and 0 < self.n_seg <= self._HOST_LENS_LIMIT:
            # small plan: the whole hierarchy is computed on the host from the fetched lengths and uploaded (no further sync)
            import numpy as _np
            lens = self._host_lens()
            cur = _np.concatenate(([0], _np.cumsum(lens)))
            out = []
            while True:
                if lens.max() <= limit:
                    out.append(cur)
                    break
                nch = (lens + (limit - 1)) // limit
                ends = _np.cumsum(nch)
                first = ends - nch
                seg_of_sub = _np.repeat(_np.arange(lens.shape[0]), nch)
                q = _np.arange(int(ends[-1])) - first[seg_of_sub]
                out.append(_np.concatenate((cur[seg_of_sub] + q * limit, cur[-1:])))
                lens = nch
                cur = _np.concatenate(([0], ends))
            dev = self.seg_ptr.device
            self._memo[key] = [torch.from_numpy(a.astype(_np.int32)).to(dev, non_blocking=True) for a in out]
        if key not in self._memo:
            out = []
            cur = self.seg_ptr.to(torch.int64)
            n_seg = self.n_seg
            dev = cur.device
            while True:
                lens = cur[1:] - cur[:-1]
                nch = (lens + (limit - 1)) // limit
                ends = torch.cumsum(nch, 0)
                # ONE host sync per level: longest segment and number of chunks together
                longest, n_sub = (0, 0) if n_seg == 0 else (int(v) for v in _fetch(torch.stack((lens.max(), ends[-1]))))
                if n_seg == 0 or longest <= limit:
                    out.append(cur.to(_I32))
                    break
                first = ends - nch
                seg_of_sub = torch.repeat_interleave(torch.arange(n_seg, device=dev), nch, output_size=n_sub)
                q = torch.arange(n_sub, device=dev) - first[seg_of_sub]
                start = cur[seg_of_sub] + q * limit
                out.append(torch.cat((start, cur[-1:])).to(_I32))
                cur = torch.cat((torch.zeros(1, dtype=torch.int64, device=dev), ends))
            self._memo[key] = out
        return self._memo[key]


def unit_ptr(m: int, dev) -> Tensor:
    """seg_ptr of the trivial plan (one message per segment)."""
    return torch.arange(m + 1, dtype=_I32, device=dev)


_PENDING_ERRORS = []     # (flag tensor, message, stream): checks that ride on the next host fetch instead of costing their own sync

FETCHES = [0]            # host synchronisations made by the planners so far (tests assert that a collated batch's step makes none)


def _stream_id(dev) -> int:
    return int(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else 0


def defer_error(flag: Tensor, msg: str) -> None:
    """`flag` (one int32, non-zero = error) was written by a launch on the CURRENT stream; report `msg` when it is found set.  It is
    read by the next planner fetch made on the same stream (whose synchronisation orders the read behind the write -- a fetch on
    another stream would read a flag that may not even be zero-filled yet), or by `check_deferred_errors()`."""
    _PENDING_ERRORS.append((flag, msg, _stream_id(flag.device)))
    if len(_PENDING_ERRORS) > 64 and not (flag.is_cuda and torch.cuda.is_current_stream_capturing()):
        check_deferred_errors()              # (a device synchronisation: not legal while a stream is capturing -- the list waits)


def _fetch(t: Tensor):
    """host copy of a small device tensor (ONE synchronisation of the current stream) that also carries every deferred error flag
    this stream produced on that device"""
    FETCHES[0] += 1
    sid = _stream_id(t.device)
    mine = [e for e in _PENDING_ERRORS if e[0].device == t.device and e[2] == sid]
    if not mine:
        return t.tolist()
    for e in mine:
        _PENDING_ERRORS.remove(e)
    vals = torch.cat([t.reshape(-1).to(torch.int64)] + [e[0].reshape(-1).to(torch.int64) for e in mine]).tolist()
    n = t.numel()
    for e, v in zip(mine, vals[n:]):
        if v != 0:
            raise ValueError(e[1])
    out = vals[:n]
    return out if t.dim() > 0 else out[0]


_DEFER_CHECKS = [False]


class deferred_index_checks:
    """``with deferred_index_checks(): ...``: index-range checks of the plans built inside do not synchronise on their own; they
    are verified by the next host fetch or, at the latest, when the block ends (`collate.BatchPrefetcher` builds a batch's plans
    this way).  Outside such a block a bad index raises at the call, as the reference's asserts do."""

    def __enter__(self):
        self.prev = _DEFER_CHECKS[0]
        _DEFER_CHECKS[0] = True
        return self

    def __exit__(self, *exc):
        _DEFER_CHECKS[0] = self.prev
        if exc[0] is None:
            for dev in {e[0].device for e in _PENDING_ERRORS if e[2] == _stream_id(e[0].device)}:    # this stream's flags only:
                _fetch(torch.zeros(1, dtype=torch.int64, device=dev))                            # no wait for other streams' work
        return False


def check_deferred_errors() -> None:
    """verify EVERY index-range check that was deferred, whichever stream produced it (one device synchronisation per device with
    pending flags)"""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("pygho_amd: check_deferred_errors() synchronises the device; call it outside the stream capture")
    pending = list(_PENDING_ERRORS)
    _PENDING_ERRORS.clear()
    for dev in {e[0].device for e in pending}:
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        FETCHES[0] += 1
    if pending:
        vals = [int(v) for v in torch.cat([e[0].reshape(-1).to(torch.int64).cpu() for e in pending]).tolist()]
        bad = []
        for e, v in zip(pending, vals):
            if v != 0 and e[1] not in bad:
                bad.append(e[1])
        if bad:                                  # EVERY failed check is reported (the list was cleared above)
            raise ValueError("; ".join(bad))


def plan_from_keys(keys: Tensor, n_seg: int, assume_sorted: Optional[bool] = None) -> SegPlan:
    """Group messages by ``keys`` (int64, values in [0, n_seg)).  Sorted keys give a permutation-free
    plan (one kernel); otherwise a stable radix sort builds the permutation."""
    dev = require_device(keys)
    keys = keys.contiguous()
    m = keys.numel()
    st = stream_ptr(dev)
    seg_ptr = torch.empty(n_seg + 1, dtype=_I32, device=dev)
    if assume_sorted is not False:
        err = _flag(dev)
        check(lib().pygho_csr_from_sorted(ptr(seg_ptr), ptr(keys), m, n_seg, ptr(err), st), "csr_from_sorted")
        if int(_fetch(err)[0]) == 0:             # the probe's synchronisation also carries every pending range flag
            return SegPlan(seg_ptr, None, n_seg, m)
        if assume_sorted:
            raise ValueError("pygho_amd: keys are not sorted / out of range")
    err = _flag(dev)
    perm = torch.empty(m, dtype=_I32, device=dev)
    seg_ptr.zero_()              # entries the CSR kernel skips for out-of-range keys must not be garbage (see below)
    nbytes = int(lib().pygho_group_by_key_workspace(m, n_seg))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().pygho_group_by_key(ptr(seg_ptr), ptr(perm), ptr(keys), m, n_seg, ptr(ws), nbytes, ptr(err), st),
          "group_by_key")
    # the range check rides on the next host fetch (`_fetch`) or on `check_deferred_errors()`.  Until then a plan built from bad
    # keys is wrong but harmless: `perm` is a permutation of [0, m) whatever the keys were, and the pointers the CSR kernel
    # skips stay 0, so every consumer still reads messages inside [0, m)
    if not _DEFER_CHECKS[0]:
        if int(_fetch(err)[0]) != 0:
            raise ValueError("pygho_amd: scatter index out of range [0, dim_size)")
        return SegPlan(seg_ptr, perm, n_seg, m)
    defer_error(err, "pygho_amd: scatter index out of range [0, dim_size)")
    return SegPlan(seg_ptr, perm, n_seg, m)


def install_plan(keys: Tensor, plan: SegPlan, tags: Tuple[str, ...] = ("scatter",), max_len: Optional[int] = None) -> None:
    """put a ready grouping of `keys` where `cached_plan` will look for it (`collate.DeviceGraphStore`: a block-diagonal batch's
    sorted groupings are its graphs' row counts, scanned); `max_len` pre-answers the plan's only host read"""
    if max_len is not None:
        plan._memo = dict(plan._memo or {}, max_len=int(max_len))
    cache = getattr(keys, "_pygho_plans", None)
    if cache is None:
        cache = {}
        keys._pygho_plans = cache
    for tag in tags:
        cache[(tag, plan.n_seg, keys._version)] = plan


def cached_plan(keys: Tensor, n_seg: int, tag: str = "", assume_sorted: Optional[bool] = None) -> SegPlan:
    """plan cache keyed on the index tensor OBJECT (index tensors are shared by reference between
    results, SpTensor.py:493) and its in-place version counter.  `assume_sorted=False`: the caller knows the keys are not
    sorted (second coordinates of a pattern, feature ids): skips the sortedness probe and its host sync."""
    cache = getattr(keys, "_pygho_plans", None)
    if cache is None:
        cache = {}
        try:
            keys._pygho_plans = cache
        except Exception:
            pass
    k = (tag, n_seg, keys._version)
    plan = cache.get(k)
    if plan is None:
        # a batch collated from `collate.DeviceGraphStore` knows how to assemble the groupings of its index rows from per-graph parts
        # (no sort, no host read): asked only when somebody needs one
        factory = getattr(keys, "_pygho_plan_factory", None)
        if factory is not None and factory[0] == keys._version:
            plan = factory[1](n_seg)
        if plan is None:
            plan = plan_from_keys(keys, n_seg, assume_sorted)
        cache[k] = plan
    return plan



# --------------------------------------------------------------------------
# integer planner primitives
# --------------------------------------------------------------------------
def row_gather(src: Tensor, idx32: Tensor, valid: Optional[Tensor] = None) -> Tensor:
    dev = require_device(src, idx32, valid)
    src = src.contiguous()
    n = idx32.numel()
    tail = src.shape[1:]
    d = 1
    for s in tail:
        d *= s
    out = torch.empty((n,) + tuple(tail), dtype=src.dtype, device=dev)
    check(lib().pygho_row_gather(ptr(out), ptr(src), ptr(idx32), ptr(valid), n, d, dtype_code(src), stream_ptr(dev)),
          "row_gather")
    return out


def row_gather_mean_ok(src: Tensor) -> bool:
    return (src.is_cuda and src.dim() == 2 and src.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and (src.shape[1] * src.element_size()) % 16 == 0 and src.shape[1] > 0)


def row_gather_mean(src: Tensor, idx32: Tensor, seg_ptr: Tensor) -> Tensor:
    """src[idx] * (1 / max(segment length, 1)).to(src.dtype)[idx]: the gradient of a segment mean w.r.t. its rows in one pass -- the bits
    of `row_gather(src * plan.inv_count.to(src.dtype).unsqueeze(-1), idx)`"""
    dev = require_device(src, idx32, seg_ptr)
    src = src.contiguous()
    n, d = idx32.numel(), src.shape[1]
    out = torch.empty((n, d), dtype=src.dtype, device=dev)
    check(lib().pygho_row_gather_mean(ptr(out), ptr(src), ptr(idx32), ptr(seg_ptr), n, d, dtype_code(src), stream_ptr(dev)), "row_gather_mean")
    return out


def hash_pack(ind: Tensor, validate: bool = True) -> Tensor:
    """indicehash (SpTensor.py:10-44) on the device."""
    dev = require_device(ind)
    assert ind.dim() == 2
    sd, nnz = ind.shape
    if sd == 1:
        return ind[0]
    ind = ind.contiguous()
    out = torch.empty(nnz, dtype=torch.int64, device=dev)
    err = _flag(dev) if validate else None
    check(lib().pygho_hash_pack(ptr(out), ptr(ind), sd, nnz, nnz, ptr(err), stream_ptr(dev)), "hash_pack")
    if validate:
        code = int(err.item())
        assert code != 1, "indice cannot be negative"
        assert code != 2, "too large indice, hash is not injective"
    return out


def hash_unpack(h: Tensor, sparse_dim: int) -> Tensor:
    dev = require_device(h)
    if sparse_dim == 1:
        return h.unsqueeze(0)
    h = h.contiguous()
    out = torch.empty((sparse_dim, h.numel()), dtype=torch.int64, device=dev)
    check(lib().pygho_hash_unpack(ptr(out), ptr(h), sparse_dim, h.numel(), stream_ptr(dev)), "hash_unpack")
    return out


def sorted_match(table: Tensor, query: Tensor) -> Tensor:
    """position of every query in the strictly increasing table, -1 when absent."""
    dev = require_device(table, query)
    table, query = table.contiguous(), query.contiguous()
    pos = torch.empty(query.shape, dtype=torch.int64, device=dev)
    check(lib().pygho_sorted_match(ptr(pos), ptr(table), table.numel(), ptr(query), query.numel(), stream_ptr(dev)),
          "sorted_match")
    return pos


def search_bounds(table: Tensor, query: Tensor) -> Tuple[Tensor, Tensor]:
    dev = require_device(table, query)
    table, query = table.contiguous(), query.contiguous()
    lo = torch.empty(query.shape, dtype=torch.int64, device=dev)
    hi = torch.empty(query.shape, dtype=torch.int64, device=dev)
    check(lib().pygho_search_bounds(ptr(lo), ptr(hi), ptr(table), table.numel(), ptr(query), query.numel(),
                                    stream_ptr(dev)), "search_bounds")
    return lo, hi


def sort_with_perm(keys: Tensor, end_bit: int = 63) -> Tuple[Tensor, Tensor]:
    """stable ascending sort of non-negative int64 keys; returns (sorted keys, int32 permutation)."""
    dev = require_device(keys)
    keys = keys.contiguous()
    n = keys.numel()
    out = torch.empty(n, dtype=torch.int64, device=dev)
    perm = torch.empty(n, dtype=_I32, device=dev)
    nbytes = int(lib().pygho_sort_pairs_i64_workspace(n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().pygho_sort_pairs_i64(ptr(out), ptr(perm), ptr(keys), n, end_bit, ptr(ws), nbytes, stream_ptr(dev)),
          "sort_pairs_i64")
    return out, perm


def unique_sorted(sorted_keys: Tensor) -> Tuple[Tensor, Tensor, int]:
    """run ids of a sorted array: (unique keys, run id per position (int32), number of runs)."""
    dev = require_device(sorted_keys)
    n = sorted_keys.numel()
    run = torch.empty(n, dtype=_I32, device=dev)
    cnt = torch.zeros(1, dtype=_I32, device=dev)
    nbytes = int(lib().pygho_run_ids_workspace(n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().pygho_run_ids(ptr(run), ptr(cnt), ptr(sorted_keys), n, ptr(ws), nbytes, stream_ptr(dev)), "run_ids")
    n_runs = int(cnt.item())
    if n:
        plan = plan_from_keys(run.to(torch.int64), n_runs, assume_sorted=True)
        uniq = row_gather(sorted_keys.reshape(-1, 1), plan.seg_ptr[:-1].contiguous()).reshape(-1)
    else:
        uniq = sorted_keys
    return uniq, run, n_runs


def unique_plan(keys: Tensor) -> Tuple[Tensor, SegPlan, Tensor]:
    """torch.unique(keys, sorted=True, return_inverse=True) as (unique keys, plan grouping the original
    positions by unique slot, inverse (int32))."""
    dev = require_device(keys)
    m = keys.numel()
    skeys, perm = sort_with_perm(keys)
    run = torch.empty(m, dtype=_I32, device=dev)
    cnt = torch.zeros(1, dtype=_I32, device=dev)
    nbytes = int(lib().pygho_run_ids_workspace(m))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = stream_ptr(dev)
    check(lib().pygho_run_ids(ptr(run), ptr(cnt), ptr(skeys), m, ptr(ws), nbytes, st), "run_ids")
    n_runs = int(cnt.item())
    seg_ptr = torch.empty(n_runs + 1, dtype=_I32, device=dev)
    # run ids are sorted int32: widen once for the CSR builder
    check(lib().pygho_csr_from_sorted(ptr(seg_ptr), ptr(run.to(torch.int64)), m, n_runs, None, st), "csr_from_sorted")
    inv = torch.empty(m, dtype=_I32, device=dev)
    check(lib().pygho_scatter_i32(ptr(inv), ptr(perm), ptr(run), m, st), "scatter_i32")
    uniq = row_gather(skeys.reshape(-1, 1), seg_ptr[:-1].contiguous()).reshape(-1) if m else skeys
    return uniq, SegPlan(seg_ptr, perm, n_runs, m), inv


def exclusive_scan(counts: Tensor) -> Tensor:
    """(n + 1) int64 offsets with offsets[0] = 0 (Spspmm.py:119-123)."""
    dev = require_device(counts)
    counts = counts.contiguous()
    n = counts.numel()
    out = torch.empty(n + 1, dtype=torch.int64, device=dev)
    nbytes = int(lib().pygho_exclusive_scan_i64_workspace(n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().pygho_exclusive_scan_i64(ptr(out), ptr(counts), n, ptr(ws), nbytes, stream_ptr(dev)), "exclusive_scan_i64")
    return out


def expand_pairs(lower: Tensor, counts: Tensor) -> Tuple[Tensor, Tensor]:
    """(c, d) pair enumeration of the product planner (Spspmm.py:119-129)."""
    dev = require_device(lower, counts)
    nnz1 = counts.numel()
    offsets = exclusive_scan(counts)
    total = int(offsets[-1].item())
    c = torch.empty(total, dtype=torch.int64, device=dev)
    d = torch.empty(total, dtype=torch.int64, device=dev)
    check(lib().pygho_expand_pairs(ptr(c), ptr(d), ptr(lower.contiguous()), ptr(offsets), nnz1, total, stream_ptr(dev)),
          "expand_pairs")
    return c, d


def product_hash(ind1: Tensor, dim1: int, ind2: Tensor, dim2: int, c: Tensor, d: Tensor) -> Tensor:
    """hash of (ind1 rows != dim1 at c, ind2 rows != dim2 at d), Spspmm.py:132-135, without the (sd, M)
    concatenated coordinate temporaries."""
    dev = require_device(ind1, ind2, c, d)
    ind1, ind2, c, d = ind1.contiguous(), ind2.contiguous(), c.contiguous(), d.contiguous()
    total = c.numel()
    out = torch.empty(total, dtype=torch.int64, device=dev)
    err = _flag(dev)
    check(lib().pygho_product_hash(ptr(out), ptr(ind1), ind1.shape[0], ind1.shape[1], dim1, ptr(ind2), ind2.shape[0],
                                   ind2.shape[1], dim2, ptr(c), ptr(d), total, ptr(err), stream_ptr(dev)), "product_hash")
    code = int(err.item())
    assert code != 1, "indice cannot be negative"
    assert code != 2, "too large indice, hash is not injective"
    return out


def gather_cols(src: Tensor, idx: Tensor) -> Tensor:
    """``src[:, idx]`` (or ``src[idx]`` for a vector) of an int64 array, idx int64 or int32."""
    dev = require_device(src, idx)
    assert src.dtype == torch.int64 and idx.dtype in (torch.int64, _I32)
    vec = src.dim() == 1
    src2 = src.reshape(1, -1) if vec else src
    src2, idx = src2.contiguous(), idx.contiguous()
    rows, ld = src2.shape
    m = idx.numel()
    out = torch.empty((rows, m), dtype=torch.int64, device=dev)
    check(lib().pygho_gather_cols_i64(ptr(out), ptr(src2), rows, ld, ptr(idx), int(idx.dtype == _I32), m, stream_ptr(dev)),
          "gather_cols_i64")
    return out.reshape(-1) if vec else out


def widen_gather(table: Tensor, idx: Tensor) -> Tensor:
    """``table[idx]`` for an int32 table and int64 positions, int64 result (Spspmm.py:104)."""
    dev = require_device(table, idx)
    assert table.dtype == _I32 and idx.dtype == torch.int64
    table, idx = table.contiguous(), idx.contiguous()
    out = torch.empty(idx.numel(), dtype=torch.int64, device=dev)
    check(lib().pygho_gather_i32_to_i64(ptr(out), ptr(table), ptr(idx), idx.numel(), stream_ptr(dev)), "gather_i32_to_i64")
    return out


def plan_triples(slot: Tensor, c: Tensor, d: Tensor, perm: Tensor) -> Tensor:
    """(3, M) int64 plan ``(slot[perm], c[perm], d[perm])`` in one pass (Spspmm.py:136-143)."""
    dev = require_device(slot, c, d, perm)
    assert slot.dtype == _I32 and perm.dtype == _I32
    m = perm.numel()
    out = torch.empty((3, m), dtype=torch.int64, device=dev)
    check(lib().pygho_plan_triples(ptr(out), ptr(slot.contiguous()), ptr(c.contiguous()), ptr(d.contiguous()),
                                   ptr(perm.contiguous()), m, stream_ptr(dev)), "plan_triples")
    return out


def nonneg_positions(vals: Tensor, via: Optional[Tensor] = None) -> Tensor:
    """ordered positions i with ``(vals[via[i]] if via is given else vals[i]) >= 0``: the boolean-mask
    compaction of Spspmm.py:219-221 / :256-263 as flag -> scan -> scatter."""
    dev = require_device(vals, via)
    vals = vals.contiguous()
    via = None if via is None else via.contiguous()
    n = vals.numel() if via is None else via.numel()
    offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    nbytes = int(lib().pygho_exclusive_scan_i64_workspace(n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = stream_ptr(dev)
    check(lib().pygho_flag_scan_nonneg(ptr(offsets), ptr(vals), ptr(via), n, ptr(ws), nbytes, st), "flag_scan_nonneg")
    kept = int(offsets[-1].item())
    pos = torch.empty(kept, dtype=torch.int64, device=dev)
    check(lib().pygho_compact_positions(ptr(pos), ptr(offsets), n, st), "compact_positions")
    return pos

"""
Torch-level wrappers over the C ABI: plan objects (int32 CSR views of the int64 index
arrays of the Python API, built once per batch and cached) and the autograd Functions
that launch the HIP kernels.  Everything here runs on ROCm device memory; there is no
CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple

from ctypes import c_void_p

import os
import torch
from torch import Tensor

from ._native import AGGR_CODE, DTYPE_CODE, check, dtype_code, lib, ptr, require_device, stream_ptr

_I32 = torch.int32


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def _flag(dev) -> Tensor:
    return torch.zeros(1, dtype=_I32, device=dev)


def narrow_i32(x: Tensor, checked: bool = False) -> Tensor:
    """int64 -> int32 copy on the device (cached on the source tensor object)."""
    if x.dtype == _I32:
        return x.contiguous()
    cache = getattr(x, "_pygho_i32", None)
    if cache is not None and cache[0] == x._version:
        return cache[1]
    dev = require_device(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=_I32, device=dev)
    err = _flag(dev) if checked else None
    check(lib().pygho_narrow_i64_i32(ptr(out), ptr(x), x.numel(), ptr(err), stream_ptr(dev)), "narrow_i64_i32")
    if checked and int(err.item()) != 0:
        raise ValueError("pygho_amd: index does not fit int32 or is negative")
    try:
        x._pygho_i32 = (x._version, out)
    except Exception:
        pass
    return out


def gather_i32(table: Tensor, idx: Tensor) -> Tensor:
    dev = require_device(table, idx)
    out = torch.empty(idx.shape, dtype=_I32, device=dev)
    check(lib().pygho_gather_i32(ptr(out), ptr(table), ptr(idx), idx.numel(), stream_ptr(dev)), "gather_i32")
    return out


class SegPlan:
    """CSR grouping of `m` messages into `n_seg` segments: ``seg_ptr`` (n_seg+1) int32 and
    ``perm`` (m) int32 = message ids in grouped order (None when the key array was already
    sorted, i.e. grouped order == message order)."""
    __slots__ = ("seg_ptr", "perm", "n_seg", "m", "_inv_cnt", "_memo", "_partner")

    def __init__(self, seg_ptr: Tensor, perm: Optional[Tensor], n_seg: int, m: int):
        self.seg_ptr, self.perm, self.n_seg, self.m = seg_ptr, perm, n_seg, m
        self._inv_cnt = None
        self._memo = None
        self._partner = None         # (key, index arrays in grouped order) of the last three-operand user

    @property
    def inv_count(self) -> Tensor:
        """1 / max(segment length, 1) as f32 (mean backward)."""
        if self._inv_cnt is None:
            cnt = (self.seg_ptr[1:] - self.seg_ptr[:-1]).clamp_min(1)
            self._inv_cnt = cnt.to(torch.float32).reciprocal()
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


_PENDING_ERRORS = []     # (flag tensor, message): checks that ride on the next host fetch instead of costing their own sync


def _fetch(t: Tensor):
    """host copy of a small device tensor (ONE synchronisation) that also carries every deferred error flag of that device"""
    mine = [(f, m) for f, m in _PENDING_ERRORS if f.device == t.device]
    if not mine:
        return t.tolist()
    for e in mine:
        _PENDING_ERRORS.remove(e)
    vals = torch.cat([t.reshape(-1).to(torch.int64)] + [f.reshape(-1).to(torch.int64) for f, _ in mine]).tolist()
    n = t.numel()
    for (f, msg), v in zip(mine, vals[n:]):
        if v != 0:
            raise ValueError(msg)
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
            check_deferred_errors()
        return False


def check_deferred_errors() -> None:
    """verify the index-range checks that were deferred (one synchronisation per device with pending flags)"""
    for dev in {f.device for f, _ in _PENDING_ERRORS}:
        _fetch(torch.zeros(1, dtype=torch.int64, device=dev))


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
    _PENDING_ERRORS.append((err, "pygho_amd: scatter index out of range [0, dim_size)"))
    if len(_PENDING_ERRORS) > 64:
        check_deferred_errors()
    return SegPlan(seg_ptr, perm, n_seg, m)


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
        plan = plan_from_keys(keys, n_seg, assume_sorted)
        cache[k] = plan
    return plan


# --------------------------------------------------------------------------
# raw launches
# --------------------------------------------------------------------------
def _as2d(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    t = t.contiguous()
    return t.reshape(t.shape[0], -1) if t.dim() != 2 else t


class LaunchTimer:
    """Opt-in per-launch timing of the fused segment kernel with HIP events recorded on the stream the
    kernel is launched on (torch's current stream is the one handed to the C ABI).  Used by bench.py for the
    roofline figure; `records` holds (kernel variant, algorithmic bytes, start event, end event)."""
    active: Optional["LaunchTimer"] = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        LaunchTimer.active = self
        return self

    def __exit__(self, *exc):
        LaunchTimer.active = None

    def summary(self):
        """{variant: (launches, mean ms, mean algorithmic bytes)} -- call after a device synchronize."""
        agg = {}
        for name, nbytes, e0, e1 in self.records:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            a[2] += nbytes
        return {k: (v[0], v[1] / v[0], v[2] / v[0]) for k, v in agg.items()}


USE_UNIT_TRIPLE = True        # forward of the three-operand tuple initialisation (unit segments) on its own elementwise kernel
DEBUG_INDICES = os.environ.get("PYGHO_DEBUG", "0") not in ("", "0")   # validate every index array of a segment launch (host sync per call)
USE_SEG_WINDOW = os.environ.get("PYGHO_SEG_WINDOW", "1") != "0"


def _debug_check_segments(out_rows: int, seg_ptr: Tensor, idx_rows) -> None:
    """PYGHO_DEBUG=1: the kernels trust their index arrays (the reference's gathers raise IndexError); this checks, before a launch,
    that the CSR pointers are monotone from 0 and that every index addresses a row of its operand."""
    ptr_ok = seg_ptr.numel() == out_rows + 1 and int(seg_ptr[0]) == 0 and bool((seg_ptr[1:] >= seg_ptr[:-1]).all())
    if not ptr_ok:
        raise IndexError("pygho_amd (PYGHO_DEBUG): segment pointers are not a monotone CSR array starting at 0")
    m = int(seg_ptr[-1])
    for name, idx, rows in idx_rows:
        if idx is None:
            if rows is not None and rows < m:
                raise IndexError(f"pygho_amd (PYGHO_DEBUG): {name} has {rows} rows for {m} messages")
            continue
        if idx.numel() < m or (m and (int(idx[:m].min()) < 0 or int(idx[:m].max()) >= rows)):
            raise IndexError(f"pygho_amd (PYGHO_DEBUG): {name} index out of range [0, {rows})")
USE_SEG_WINDOW_BY_EDGE = os.environ.get("PYGHO_SEG_WINDOW_BY_EDGE", "1") != "0"
SEG_WINDOW_MIN_ROW_BYTES = int(os.environ.get("PYGHO_SEG_WINDOW_MIN_ROW_BYTES", "512"))


def _window_eligible(out_rows: int, lhs: Optional[Tensor], rhs: Optional[Tensor], rhs_idx: Optional[Tensor], aggr: str) -> bool:
    """two-operand sum / mean whose rhs is the small operand (edge rows, an embedding table): its rows are served from LDS
    (`pygho_seg_gather_mul_reduce_window`).  Rows of 512 B and more: that is where the L2 -> L1 gather path binds."""
    if not USE_SEG_WINDOW or lhs is None or rhs is None or rhs_idx is None or aggr not in ("sum", "mean"):
        return False
    rb = rhs.shape[1] * rhs.element_size()
    if (USE_SEG_WINDOW_BY_EDGE and rhs.shape[0] > 2 * out_rows and lhs.shape[0] == rhs.shape[0] and rb % 16 == 0 and 128 <= rb <= 256
            and rhs.dtype in (torch.bfloat16, torch.float16) and out_rows >= 4096 and rhs.shape[0] * rb < (1 << 32)):
        # the by-edge backward plan of spspmm (gradient of the adjacency values): both operands are tuple-level rows spread over
        # their whole graph (a 110-KB working set per graph against 32 KB of L1), few segments per pass so that one operand's
        # row range fits the window: 288 -> 258 us per launch in the ZINC step
        return True
    return (rhs.dtype in (torch.float32, torch.bfloat16, torch.float16) and rb % 16 == 0 and SEG_WINDOW_MIN_ROW_BYTES <= rb <= 1024
            and 2 * rhs.shape[0] <= out_rows and out_rows >= 4096 and max(out_rows, lhs.shape[0]) * rb < (1 << 32))


SEG_TILE = os.environ.get("PYGHO_SEG_TILE", "0")            # "0": never, "1": whenever the shape allows, "auto": by plan shape
SEG_TILE_WIN_ROWS = int(os.environ.get("PYGHO_SEG_TILE_WIN_ROWS", "32"))


def tile_plan(seg_ptr: Tensor, lhs_idx: Tensor, n_seg: int, win_rows: int):
    """(tile_cnt, tiles) of `pygho_seg_tile_plan` for one (CSR pointers, lhs index) pair: consecutive segments whose lhs rows lie in
    a window of `win_rows` consecutive rows.  A pure function of the index arrays; cached on the index tensor object."""
    cache = getattr(lhs_idx, "_pygho_tiles", None)
    if cache is None:
        cache = {}
        try:
            lhs_idx._pygho_tiles = cache
        except Exception:
            pass
    k = (seg_ptr.data_ptr(), n_seg, win_rows, lhs_idx._version, seg_ptr._version)
    hit = cache.get(k)
    if hit is None:
        dev = require_device(seg_ptr, lhs_idx)
        chunk = int(lib().pygho_seg_tile_chunk())
        n_chunks = (n_seg + chunk - 1) // chunk
        tile_cnt = torch.empty(n_chunks, dtype=_I32, device=dev)
        tiles = torch.empty((n_chunks, chunk, 4), dtype=_I32, device=dev)
        check(lib().pygho_seg_tile_plan(ptr(tile_cnt), ptr(tiles), ptr(seg_ptr), ptr(lhs_idx), n_seg, win_rows, stream_ptr(dev)),
              "seg_tile_plan")
        hit = (tile_cnt, tiles, seg_ptr)            # the pointers are kept alive with the entry that is keyed on their address
        cache[k] = hit
    return hit[0], hit[1]


def _tile_eligible(out_rows: int, lhs: Optional[Tensor], rhs: Optional[Tensor], lhs_idx: Optional[Tensor], rhs_idx: Optional[Tensor],
                   aggr: str) -> bool:
    """two-operand sum / mean with both index arrays and rows of 256 / 512 / 1024 bytes (`pygho_seg_gather_mul_reduce_tiled`)."""
    if SEG_TILE == "0" or lhs is None or rhs is None or lhs_idx is None or rhs_idx is None or aggr not in ("sum", "mean"):
        return False
    rb = lhs.shape[1] * lhs.element_size()
    if rb not in (256, 512, 1024) or lhs.dtype not in (torch.float32, torch.bfloat16, torch.float16) or out_rows < 4096:
        return False
    if SEG_TILE == "1":
        return True
    # auto: the plans whose segments gather lhs rows of one narrow block (forward and by-tuple backward of the tuple products);
    # the by-edge backward plan (few long segments over rows spread across a graph) stays on the window kernel
    return rb >= 512 and 2 * rhs.shape[0] <= out_rows


def seg_gmr(out_rows: int, lhs: Optional[Tensor], rhs: Optional[Tensor], seg_ptr: Tensor,
            lhs_idx: Optional[Tensor], rhs_idx: Optional[Tensor], aggr: str,
            lhs_rowscale: Optional[Tensor] = None, addend: Optional[Tensor] = None,
            act: Optional[Tuple[Tensor, Tensor, str, int]] = None) -> Tensor:
    """out[s] = [addend[s] +] (+)_{m in seg s} scale * lhs[lhs_idx[m]] * rhs[rhs_idx[m]]  (2-D operands).
    `act` = (scale, shift, name, side): operand `side` (1 lhs, 2 rhs) holds pre-activations and
    act(x * scale + shift) is applied to its rows as they are loaded."""
    ref = lhs if lhs is not None else rhs
    dev = require_device(lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, addend)
    if DEBUG_INDICES:
        _debug_check_segments(out_rows, seg_ptr, (("lhs", lhs_idx, None if lhs is None else lhs.shape[0]),
                                                  ("rhs", rhs_idx, None if rhs is None else rhs.shape[0])))
    d = ref.shape[1]
    out = torch.empty((out_rows, d), dtype=ref.dtype, device=dev)
    timer = LaunchTimer.active
    windowed = tiled = False
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
    dims = (out_rows, d, d if lhs is not None else 0, d if rhs is not None else 0,
            lhs.shape[0] if lhs is not None else 0, rhs.shape[0] if rhs is not None else 0,
            dtype_code(ref), AGGR_CODE[aggr], stream_ptr(dev))
    if act is not None:
        a_scale, a_shift, a_name, a_side = act
        assert lhs is not None and rhs is not None and a_scale.dtype == torch.float32 and a_shift.dtype == torch.float32
        if addend is not None:
            assert addend.shape == out.shape and addend.dtype == out.dtype and addend.is_contiguous()
        check(lib().pygho_seg_gather_mul_reduce_act(
            ptr(out), ptr(addend), ptr(lhs), ptr(rhs), ptr(seg_ptr), ptr(lhs_idx), ptr(rhs_idx), ptr(lhs_rowscale),
            ptr(a_scale.contiguous()), ptr(a_shift.contiguous()), ACT_CODE[a_name], a_side, out_rows, d, lhs.shape[0], rhs.shape[0],
            dtype_code(ref), AGGR_CODE[aggr], stream_ptr(dev)), "seg_gather_mul_reduce_act")
    elif _tile_eligible(out_rows, lhs, rhs, lhs_idx, rhs_idx, aggr):
        tiled = True
        if addend is not None:
            assert addend.shape == out.shape and addend.dtype == out.dtype and addend.is_contiguous()
        tile_cnt, tiles = tile_plan(seg_ptr, lhs_idx, out_rows, SEG_TILE_WIN_ROWS)
        check(lib().pygho_seg_gather_mul_reduce_tiled(
            ptr(out), ptr(addend), ptr(lhs), ptr(rhs), ptr(seg_ptr), ptr(lhs_idx), ptr(rhs_idx), ptr(lhs_rowscale), ptr(tile_cnt),
            ptr(tiles), out_rows, d, lhs.shape[0], rhs.shape[0], SEG_TILE_WIN_ROWS, dtype_code(ref), AGGR_CODE[aggr],
            stream_ptr(dev)), "seg_gather_mul_reduce_tiled")
    elif _window_eligible(out_rows, lhs, rhs, rhs_idx, aggr):
        windowed = True
        if addend is not None:
            assert addend.shape == out.shape and addend.dtype == out.dtype and addend.is_contiguous()
        check(lib().pygho_seg_gather_mul_reduce_window(
            ptr(out), ptr(addend), ptr(lhs), ptr(rhs), ptr(seg_ptr), ptr(lhs_idx), ptr(rhs_idx), ptr(lhs_rowscale), out_rows, d,
            lhs.shape[0], rhs.shape[0], dtype_code(ref), AGGR_CODE[aggr], stream_ptr(dev)), "seg_gather_mul_reduce_window")
    elif addend is None:
        check(lib().pygho_seg_gather_mul_reduce(
            ptr(out), ptr(lhs), ptr(rhs), ptr(seg_ptr), ptr(lhs_idx), ptr(rhs_idx), ptr(lhs_rowscale), *dims),
            "seg_gather_mul_reduce")
    else:
        assert addend.shape == out.shape and addend.dtype == out.dtype and addend.is_contiguous()
        check(lib().pygho_seg_gather_mul_reduce_add(
            ptr(out), ptr(addend), ptr(lhs), ptr(rhs), ptr(seg_ptr), ptr(lhs_idx), ptr(rhs_idx), ptr(lhs_rowscale), *dims),
            "seg_gather_mul_reduce_add")
    if timer is not None:
        e1.record(torch.cuda.current_stream(dev))
        # algorithmic bytes: every operand row once, every output row once, int32 indices once
        es = ref.element_size()
        m = lhs_idx.numel() if lhs_idx is not None else (rhs_idx.numel() if rhs_idx is not None else ref.shape[0])
        rows = (lhs.shape[0] if lhs is not None else 0) + (rhs.shape[0] if rhs is not None else 0) + out_rows
        nbytes = es * d * rows + 4 * m * ((lhs_idx is not None) + (rhs_idx is not None)) + 4 * (out_rows + 1)
        if lhs_rowscale is not None:
            nbytes += 4 * lhs_rowscale.numel()
        if addend is not None:
            nbytes += es * d * out_rows
        mode = "both" if (lhs is not None and rhs is not None) else ("lhs" if lhs is not None else "rhs")
        timer.records.append((f"seg_gmr[{str(ref.dtype).split('.')[-1]},{aggr},{mode}{',scaled' if lhs_rowscale is not None else ''}{',res' if addend is not None else ''}{',act' if act is not None else ''}{',window' if windowed else ''}{',tiled' if tiled else ''}]",
                              nbytes, e0, e1))
    return out


def seg_triple(out_rows: int, a: Tensor, b: Tensor, c: Tensor, seg_ptr: Optional[Tensor], a_idx: Optional[Tensor],
               b_idx: Optional[Tensor], c_idx: Optional[Tensor], out_f32: bool = False) -> Tensor:
    """out[s] = sum_{m in seg s} a[a_idx[m]] * b[b_idx[m]] * c[c_idx[m]]  (2-D operands of one dtype and width);
    `out_f32`: f32 result for 16-bit operands (first level of a long-segment hierarchy).  `seg_ptr = None`: unit segments
    (message s belongs to output row s): a plain three-row gather-multiply kernel without the segment machinery."""
    dev = require_device(a, b, c, seg_ptr, a_idx, b_idx, c_idx)
    assert a.dim() == b.dim() == c.dim() == 2 and a.shape[1] == b.shape[1] == c.shape[1] and a.dtype == b.dtype == c.dtype
    a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
    d = a.shape[1]
    out = torch.empty((out_rows, d), dtype=torch.float32 if out_f32 else a.dtype, device=dev)
    timer = LaunchTimer.active
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
    check(lib().pygho_seg_triple_product(ptr(out), ptr(a), ptr(b), ptr(c), ptr(seg_ptr), ptr(a_idx), ptr(b_idx), ptr(c_idx),
                                         out_rows, d, a.shape[0], b.shape[0], c.shape[0], dtype_code(a),
                                         1 if out_f32 and a.dtype != torch.float32 else 0, stream_ptr(dev)),
          "seg_triple_product")
    if timer is not None:
        e1.record(torch.cuda.current_stream(dev))
        m = next((i.numel() for i in (a_idx, b_idx, c_idx) if i is not None), a.shape[0])
        nbytes = (a.element_size() * d * (a.shape[0] + b.shape[0] + c.shape[0] + out_rows)
                  + 4 * m * sum(i is not None for i in (a_idx, b_idx, c_idx)) + 4 * (out_rows + 1))
        timer.records.append((f"seg_triple[{str(a.dtype).split('.')[-1]}]", nbytes, e0, e1))
    return out


LONG_SEGMENT = 256      # segments longer than this are reduced hierarchically ...
LONG_CHUNK = 32         # ... in chunks of this many rows (a lane group walks its chunk sequentially)


def seg_sum_f32out(src: Tensor, seg_ptr: Tensor, idx: Optional[Tensor], n_seg: int) -> Tensor:
    dev = require_device(src, seg_ptr, idx)
    d = src.shape[1]
    out = torch.empty((n_seg, d), dtype=torch.float32, device=dev)
    check(lib().pygho_seg_sum_f32out(ptr(out), ptr(src), ptr(seg_ptr), ptr(idx), n_seg, d, src.shape[0], dtype_code(src),
                                     stream_ptr(dev)), "seg_sum_f32out")
    return out


def seg_reduce_rows(src: Tensor, plan: SegPlan, aggr: str) -> Tensor:
    """out[s] = (+)_{m in segment s} src[perm[m]] for a 2-D `src`; long segments go through a hierarchy of
    bounded chunks (f32 partial sums for 16-bit inputs)."""
    if plan.m == 0 or plan.max_len <= LONG_SEGMENT:
        return seg_gmr(plan.n_seg, src, None, plan.seg_ptr, plan.perm, None, aggr)
    levels = plan.levels(LONG_CHUNK)
    red = "sum" if aggr == "mean" else aggr
    sixteen = src.dtype in (torch.bfloat16, torch.float16) and red == "sum" and (src.shape[1] * 2) % 16 == 0
    n0 = levels[0].numel() - 1
    if sixteen:
        cur = seg_sum_f32out(src, levels[0], plan.perm, n0)
    else:
        cur = seg_gmr(n0, src, None, levels[0], plan.perm, None, red)
    for lv in levels[1:]:
        cur = seg_gmr(lv.numel() - 1, cur, None, lv, None, None, red)
    if aggr == "mean":
        cur = cur * plan.inv_count.to(cur.dtype).unsqueeze(-1)
    return cur.to(src.dtype)


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


def _ties(fwd: Tensor, lhs, rhs, seg_ptr, lhs_idx, rhs_idx) -> Tensor:
    dev = fwd.device
    ties = torch.empty(fwd.shape, dtype=torch.float32, device=dev)
    check(lib().pygho_seg_extremum_ties(ptr(ties), ptr(fwd), ptr(lhs), ptr(rhs), ptr(seg_ptr), ptr(lhs_idx),
                                        ptr(rhs_idx), fwd.shape[0], fwd.shape[1], dtype_code(fwd), stream_ptr(dev)),
          "seg_extremum_ties")
    return ties


def _extremum_bwd(n_rows, gin, fwd, ties, self_vals, other, seg_ptr, out_idx, other_idx) -> Tensor:
    dev = gin.device
    d = gin.shape[1]
    gout = torch.empty((n_rows, d), dtype=gin.dtype, device=dev)
    check(lib().pygho_seg_extremum_bwd(ptr(gout), ptr(gin), ptr(fwd), ptr(ties), ptr(self_vals), ptr(other),
                                       ptr(seg_ptr), ptr(out_idx), ptr(other_idx), n_rows, d, dtype_code(gin),
                                       stream_ptr(dev)), "seg_extremum_bwd")
    return gout


# --------------------------------------------------------------------------
# the message-passing plan of one (acd, n_out) pair
# --------------------------------------------------------------------------
class MessagePlan:
    """int32 / CSR view of an ``acd`` triple array (Spspmm.py:186-222): forward grouping by the
    output slot `a`, and (lazily, for backward) the transposed groupings by `c` and by `d`."""

    def __init__(self, acd: Tensor, n_out: int, n_lhs: int, n_rhs: int):
        require_device(acd)
        assert acd.dim() == 2 and acd.shape[0] == 3, "acd must be (3, M)"
        self.m = acd.shape[1]
        self.n_out, self.n_lhs, self.n_rhs = n_out, n_lhs, n_rhs
        self._a64, self._c64, self._d64 = acd[0], acd[1], acd[2]
        if self.m > 0:
            # operand indices must address rows of the operands (the reference's gathers raise IndexError,
            # Spspmm.py:309-311); the flag rides on the synchronisation of the forward plan's sortedness probe below
            lo, hi = torch.aminmax(acd[1:3], dim=1)
            bad = ((lo < 0).any() | (hi[0] >= n_lhs) | (hi[1] >= n_rhs)).to(torch.int32).reshape(1)
            _PENDING_ERRORS.append((bad, f"pygho_amd: acd operand index out of range (acd[1] must lie in [0, {n_lhs}), "
                                         f"acd[2] in [0, {n_rhs}))"))
        self.fwd = plan_from_keys(acd[0], n_out)
        a32, c32, d32 = narrow_i32(acd[0]), narrow_i32(acd[1]), narrow_i32(acd[2])
        self.a32, self.c32, self.d32 = a32, c32, d32                 # message order
        self.c_fwd, self.d_fwd = self.fwd.take(c32), self.fwd.take(d32)   # grouped-by-a order
        self._by_c = None
        self._by_d = None
        self._lookup = None

    @classmethod
    def from_parts(cls, acd: Tensor, n_out: int, n_lhs: int, n_rhs: int, fwd_ptr: Tensor, ptr_c: Tensor, perm_c: Tensor,
                   ptr_d: Tensor, perm_d: Tensor) -> "MessagePlan":
        """the same plan from groupings that already exist (int32 CSR pointers and permutations): a block-diagonal batch's
        groupings are the concatenation of its graphs' precomputed ones (`collate.DeviceGraphStore`), so no sort and no host
        synchronisation is needed per batch."""
        require_device(acd, fwd_ptr, ptr_c, perm_c, ptr_d, perm_d)
        self = cls.__new__(cls)
        self.m = acd.shape[1]
        self.n_out, self.n_lhs, self.n_rhs = n_out, n_lhs, n_rhs
        self._a64, self._c64, self._d64 = acd[0], acd[1], acd[2]
        self.fwd = SegPlan(fwd_ptr, None, n_out, self.m)
        self.a32, self.c32, self.d32 = narrow_i32(acd[0]), narrow_i32(acd[1]), narrow_i32(acd[2])
        self.c_fwd, self.d_fwd = self.c32, self.d32
        pc, pd = SegPlan(ptr_c, perm_c, n_lhs, self.m), SegPlan(ptr_d, perm_d, n_rhs, self.m)
        self._by_c = (pc, pc.take(self.a32), pc.take(self.d32))
        self._by_d = (pd, pd.take(self.a32), pd.take(self.c32))
        self._lookup = None
        return self

    def by_c(self):
        """(plan, a-in-grouped-order, d-in-grouped-order) for the gradient wrt the first operand."""
        if self._by_c is None:
            p = plan_from_keys(self._c64, self.n_lhs, False)         # the acd triples are sorted by a, not by c / d
            self._by_c = (p, p.take(self.a32), p.take(self.d32))
        return self._by_c

    def by_d(self):
        if self._by_d is None:
            p = plan_from_keys(self._d64, self.n_rhs, False)
            self._by_d = (p, p.take(self.a32), p.take(self.c32))
        return self._by_d

    def lookup(self, row_of: Tensor):
        """for a second operand that equals table[row_of]: the per-message table rows in forward and in by-c order
        (cached per index tensor object)."""
        memo = self._lookup
        if memo is None or memo[0] is not row_of:
            r32 = narrow_i32(row_of)
            memo = (row_of, (gather_i32(r32, self.d_fwd), gather_i32(r32, self.by_c()[2])))
            self._lookup = memo
        return memo[1]


def install_message_plan(acd: Tensor, plan: MessagePlan) -> None:
    """put a ready plan where `message_plan` will look for it"""
    cache = getattr(acd, "_pygho_plans", None)
    if cache is None:
        cache = {}
        acd._pygho_plans = cache
    cache[("msg", plan.n_out, plan.n_lhs, plan.n_rhs, acd._version)] = plan


def message_plan(acd: Tensor, n_out: int, n_lhs: int, n_rhs: int) -> MessagePlan:
    cache = getattr(acd, "_pygho_plans", None)
    if cache is None:
        cache = {}
        try:
            acd._pygho_plans = cache
        except Exception:
            pass
    k = ("msg", n_out, n_lhs, n_rhs, acd._version)
    plan = cache.get(k)
    if plan is None:
        plan = MessagePlan(acd, n_out, n_lhs, n_rhs)
        cache[k] = plan
    return plan


class _MessageReduce(torch.autograd.Function):
    """out[a] = (+) lhs[c] * rhs[d] over the plan; either operand may be None (pattern only)."""

    @staticmethod
    def forward(ctx, lhs: Optional[Tensor], rhs: Optional[Tensor], plan: MessagePlan, aggr: str, addend: Optional[Tensor] = None):
        out = seg_gmr(plan.n_out, lhs, rhs, plan.fwd.seg_ptr, plan.c_fwd if lhs is not None else None,
                      plan.d_fwd if rhs is not None else None, aggr, addend=None if addend is None else addend.contiguous())
        ctx.plan, ctx.aggr = plan, aggr
        ctx.has = (lhs is not None, rhs is not None)
        ctx.save_for_backward(lhs, rhs, out if aggr in ("max", "min") else None)
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        lhs, rhs, fwd = ctx.saved_tensors
        plan, aggr = ctx.plan, ctx.aggr
        gout = gout.contiguous()
        g_lhs = g_rhs = None
        scale = plan.fwd.inv_count if aggr == "mean" else None
        ties = None
        if aggr in ("max", "min"):
            ties = _ties(fwd, lhs, rhs, plan.fwd.seg_ptr, plan.c_fwd if lhs is not None else None,
                         plan.d_fwd if rhs is not None else None)
        if lhs is not None and ctx.needs_input_grad[0]:
            p, a_g, d_g = plan.by_c()
            if ties is None:
                g_lhs = seg_gmr(plan.n_lhs, gout, rhs, p.seg_ptr, a_g, d_g if rhs is not None else None, "sum", scale)
            else:
                g_lhs = _extremum_bwd(plan.n_lhs, gout, fwd, ties, lhs, rhs, p.seg_ptr, a_g, d_g)
        if rhs is not None and ctx.needs_input_grad[1]:
            p, a_g, c_g = plan.by_d()
            if ties is None:
                g_rhs = seg_gmr(plan.n_rhs, gout, lhs, p.seg_ptr, a_g, c_g if lhs is not None else None, "sum", scale)
            else:
                g_rhs = _extremum_bwd(plan.n_rhs, gout, fwd, ties, rhs, lhs, p.seg_ptr, a_g, c_g)
        g_add = gout if len(ctx.needs_input_grad) > 4 and ctx.needs_input_grad[4] else None      # out = addend + reduction
        return g_lhs, g_rhs, None, None, g_add


def _broadcast_dense(a: Optional[Tensor], b: Optional[Tensor]) -> Tuple[Optional[Tensor], Optional[Tensor], Tuple[int, ...]]:
    """broadcast the dense (trailing) shapes of two value tensors and flatten them to 2-D."""
    if a is None or b is None:
        t = a if a is not None else b
        return (None if a is None else _as2d(a)), (None if b is None else _as2d(b)), tuple(t.shape[1:])
    if a.dtype != b.dtype:
        dt = torch.promote_types(a.dtype, b.dtype)
        a, b = a.to(dt), b.to(dt)
    if a.shape[1:] == b.shape[1:]:
        return _as2d(a), _as2d(b), tuple(a.shape[1:])
    nd = max(a.dim(), b.dim()) - 1
    sa = (1,) * (nd - (a.dim() - 1)) + tuple(a.shape[1:])
    sb = (1,) * (nd - (b.dim() - 1)) + tuple(b.shape[1:])
    dense = torch.broadcast_shapes(sa, sb)
    a = a.reshape((a.shape[0],) + sa).expand((a.shape[0],) + dense)
    b = b.reshape((b.shape[0],) + sb).expand((b.shape[0],) + dense)
    return _as2d(a), _as2d(b), tuple(dense)


def message_reduce(lhs: Optional[Tensor], rhs: Optional[Tensor], acd: Tensor, n_out: int, n_lhs: int, n_rhs: int,
                   aggr: str, addend: Optional[Tensor] = None) -> Tensor:
    """spspmm value computation (Spspmm.py:307-315) on the HIP path; `addend` (sum / mean, shape of the result): + addend in the
    kernel's epilogue (a residual connection around the product)."""
    if lhs is None and rhs is None:
        raise ValueError("pygho_amd: both operands are pattern-only; nothing to multiply")
    if aggr not in AGGR_CODE:
        raise NotImplementedError(f"aggr {aggr!r} is not supported (sum, mean, max, min)")
    l2, r2, dense = _broadcast_dense(lhs, rhs)
    plan = message_plan(acd, n_out, n_lhs, n_rhs)
    if addend is not None:
        assert aggr in ("sum", "mean") and tuple(addend.shape) == (n_out,) + dense and addend.dtype == (l2 if l2 is not None else r2).dtype
        out = _MessageReduce.apply(l2, r2, plan, aggr, _as2d(addend))
    else:
        out = _MessageReduce.apply(l2, r2, plan, aggr)
    return out.reshape((n_out,) + dense)


# --------------------------------------------------------------------------
# scatter / segment reduce and gather
# --------------------------------------------------------------------------
class _ScatterReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src: Tensor, plan: SegPlan, ind32: Tensor, aggr: str):
        out = seg_reduce_rows(src, plan, aggr)
        ctx.plan, ctx.aggr, ctx.ind32 = plan, aggr, ind32
        ctx.save_for_backward(*((src, out) if aggr in ("max", "min") else ()))
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        plan, aggr, ind32 = ctx.plan, ctx.aggr, ctx.ind32
        gout = gout.contiguous()
        if aggr == "sum":
            return row_gather(gout, ind32), None, None, None
        if aggr == "mean":
            scaled = gout * plan.inv_count.to(gout.dtype).unsqueeze(-1)
            return row_gather(scaled, ind32), None, None, None
        src, fwd = ctx.saved_tensors
        ties = _ties(fwd, src, None, plan.seg_ptr, plan.perm, None)
        up = unit_ptr(src.shape[0], src.device)
        return _extremum_bwd(src.shape[0], gout, fwd, ties, src, None, up, ind32, None), None, None, None


def scatter_reduce(src: Tensor, ind: Tensor, dim_size: int, aggr: str) -> Tensor:
    """torch_scatter_reduce(dim=0) (utils.py:44-56) on the HIP path."""
    require_device(src, ind)
    if aggr not in AGGR_CODE:
        raise NotImplementedError(f"aggr {aggr!r} is not supported (sum, mean, max, min)")
    assert ind.dim() == 1, "indice must be 1-d"
    assert src.shape[0] == ind.shape[0], "src and index length differ"
    plan = cached_plan(ind, dim_size, "scatter")
    tail = tuple(src.shape[1:])
    src2 = _as2d(src) if src.dim() > 1 else src.contiguous().reshape(-1, 1)
    out = _ScatterReduce.apply(src2, plan, narrow_i32(ind), aggr)
    return out.reshape((dim_size,) + tail)


def scatter_reduce_planned(src: Tensor, plan: SegPlan, ind32: Tensor, aggr: str) -> Tensor:
    """scatter-reduce along a prebuilt plan (coalesce / sparse pooling)."""
    if aggr not in AGGR_CODE:
        raise NotImplementedError(f"aggr {aggr!r} is not supported (sum, mean, max, min)")
    tail = tuple(src.shape[1:])
    src2 = _as2d(src) if src.dim() > 1 else src.contiguous().reshape(-1, 1)
    out = _ScatterReduce.apply(src2, plan, ind32, aggr)
    return out.reshape((plan.n_seg,) + tail)


class _RowGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src: Tensor, ind: Tensor):
        ctx.ind, ctx.n = ind, src.shape[0]
        return row_gather(src, narrow_i32(ind))

    @staticmethod
    def backward(ctx, gout: Tensor):
        ind = ctx.ind
        plan = cached_plan(ind, ctx.n, "scatter")
        g2 = _as2d(gout) if gout.dim() > 1 else gout.contiguous().reshape(-1, 1)
        g = seg_reduce_rows(g2, plan, "sum")
        return g.reshape((ctx.n,) + tuple(gout.shape[1:])), None


def gather_rows(src: Tensor, ind: Tensor) -> Tensor:
    """src[ind] along dim 0 (SpTensor.py:476) with a segment-reduce backward."""
    require_device(src, ind)
    return _RowGather.apply(src, ind)


class _MaskedRowGather(torch.autograd.Function):
    """out[r] = pos[r] >= 0 ? src[pos[r]] : 0   (diag / sparse unpooling)."""

    @staticmethod
    def forward(ctx, src: Tensor, pos: Tensor):
        valid = (pos >= 0).to(_I32)
        idx = narrow_i32(pos.clamp_min(0))
        ctx.pos, ctx.n = pos, src.shape[0]
        return row_gather(src, idx, valid)

    @staticmethod
    def backward(ctx, gout: Tensor):
        pos, n = ctx.pos, ctx.n
        cache = getattr(pos, "_pygho_plans", None)
        if cache is None:
            cache = {}
            try:
                pos._pygho_plans = cache
            except Exception:
                pass
        k = ("spill", n, pos._version)
        if k not in cache:
            keys = torch.where(pos >= 0, pos, torch.full_like(pos, n))      # misses go to a spill segment
            cache[k] = plan_from_keys(keys, n + 1)
        plan = cache[k]
        g2 = _as2d(gout) if gout.dim() > 1 else gout.contiguous().reshape(-1, 1)
        g = seg_reduce_rows(g2, plan, "sum")[:n]
        return g.reshape((n,) + tuple(gout.shape[1:])), None


def gather_rows_matched(src: Tensor, pos: Tensor) -> Tensor:
    require_device(src, pos)
    return _MaskedRowGather.apply(src, pos)


# --------------------------------------------------------------------------
# node-level sparse x dense
# --------------------------------------------------------------------------
class _Spmm(torch.autograd.Function):
    """out[t] = (+)_e val[e] * X[src[e]] grouped by tar[e]   (Spmm.py:31-44)."""

    @staticmethod
    def forward(ctx, val: Optional[Tensor], X: Tensor, src: Tensor, tar: Tensor, n_tar: int, aggr: str):
        plan = cached_plan(tar, n_tar, "scatter")
        src32, tar32 = narrow_i32(src), narrow_i32(tar)
        src_g = plan.take(src32)
        out = seg_gmr(n_tar, val, X, plan.seg_ptr, plan.perm if val is not None else None, src_g, aggr)
        ctx.meta = (plan, src, src32, tar32, src_g, n_tar, aggr)
        ctx.save_for_backward(val, X, out if aggr in ("max", "min") else None)
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        val, X, fwd = ctx.saved_tensors
        plan, src, src32, tar32, src_g, n_tar, aggr = ctx.meta
        gout = gout.contiguous()
        e = src32.numel()
        g_val = g_x = None
        scale = plan.inv_count if aggr == "mean" else None
        ties = None
        if aggr in ("max", "min"):
            ties = _ties(fwd, val, X, plan.seg_ptr, plan.perm if val is not None else None, src_g)
        if val is not None and ctx.needs_input_grad[0]:
            up = unit_ptr(e, gout.device)
            if ties is None:
                g_val = seg_gmr(e, gout, X, up, tar32, src32, "sum", scale)
            else:
                g_val = _extremum_bwd(e, gout, fwd, ties, val, X, up, tar32, src32)
        if ctx.needs_input_grad[1]:
            p = cached_plan(src, X.shape[0], "scatter")
            tar_g = p.take(tar32)
            if ties is None:
                g_x = seg_gmr(X.shape[0], gout, val, p.seg_ptr, tar_g, p.perm if val is not None else None, "sum", scale)
            else:
                eid = p.perm if p.perm is not None else torch.arange(e, dtype=_I32, device=gout.device)
                g_x = _extremum_bwd(X.shape[0], gout, fwd, ties, X, val, p.seg_ptr, tar_g, eid)
        return g_val, g_x, None, None, None, None


def spmm_values(val: Optional[Tensor], X: Tensor, src: Tensor, tar: Tensor, n_tar: int, aggr: str) -> Tensor:
    require_device(val, X, src, tar)
    if aggr not in AGGR_CODE:
        raise NotImplementedError(f"aggr {aggr!r} is not supported (sum, mean, max, min)")
    if val is not None:
        v2, x2, dense = _broadcast_dense(val, X)
    else:
        v2, x2, dense = None, _as2d(X) if X.dim() > 1 else X.reshape(-1, 1), tuple(X.shape[1:])
    out = _Spmm.apply(v2, x2, src, tar, n_tar, aggr)
    return out.reshape((n_tar,) + dense)


class _PairProduct(torch.autograd.Function):
    """out[t] = (left[row[t]] * right[col[t]]) * val[vidx[t]] (vidx None = t): the tuple initialisation of
    example/minimal.py:62-67 (two unpoolings of node features onto the tuple pattern and two elementwise products; with
    `vidx` also the embedding lookup of the tuple feature, example/minimal.py:30-33) as ONE pass; the three operand
    gradients are the same three-operand kernel over the unit / by-row / by-col / by-feature groupings of the tuples."""

    @staticmethod
    def forward(ctx, left, right, val, row32, col32, vidx32, by_row, by_col, by_val):
        n = row32.numel()
        unit_ok = USE_UNIT_TRIPLE and (left.shape[1] * left.element_size()) % 16 == 0 and left.shape[1] * left.element_size() <= 1024 \
            and left.dtype in (torch.float32, torch.bfloat16, torch.float16)
        out = seg_triple(n, left, right, val, None if unit_ok else unit_ptr(n, val.device), row32, col32, vidx32)
        ctx.save_for_backward(left, right, val)
        ctx.idx = (row32, col32, vidx32, by_row, by_col, by_val)
        return out

    @staticmethod
    def backward(ctx, g):
        left, right, val = ctx.saved_tensors
        row32, col32, vidx32, by_row, by_col, by_val = ctx.idx
        g = g.contiguous()
        n = row32.numel()
        g_left = g_right = g_val = None
        if ctx.needs_input_grad[0]:
            p, col_p, v_p = by_row
            g_left = seg_triple(p.n_seg, g, val, right, p.seg_ptr, p.perm, v_p if vidx32 is not None else p.perm, col_p)
        if ctx.needs_input_grad[1]:
            p, row_p, v_p = by_col
            g_right = seg_triple(p.n_seg, g, val, left, p.seg_ptr, p.perm, v_p if vidx32 is not None else p.perm, row_p)
        if ctx.needs_input_grad[2]:
            if vidx32 is None:
                g_val = seg_triple(n, g, left, right, unit_ptr(n, g.device), None, row32, col32)
            else:
                # gradient of the (small) table: a handful of very long segments -> chunked f32 partial sums, then a tree
                p, row_p, col_p = by_val
                levels = p.levels(LONG_CHUNK) if p.max_len > LONG_SEGMENT else [p.seg_ptr]
                cur = seg_triple(levels[0].numel() - 1, g, left, right, levels[0], p.perm, row_p, col_p, out_f32=len(levels) > 1)
                for lv in levels[1:]:
                    cur = seg_gmr(lv.numel() - 1, cur, None, lv, None, None, "sum")
                g_val = cur.to(val.dtype)
        return g_left, g_right, g_val, None, None, None, None, None, None


def _grouped(plan: SegPlan, key, *idx32):
    """index arrays re-ordered into the plan's grouped order, memoised on the plan object."""
    memo = plan._partner                     # `key`: the index tensor OBJECTS (kept alive by the memo, compared by identity)
    if memo is None or len(memo[0]) != len(key) or any(a is not b for a, b in zip(memo[0], key)):
        memo = (key, tuple(None if i is None else plan.take(i) for i in idx32))
        plan._partner = memo
    return memo[1]


def pair_product(left: Tensor, right: Tensor, val: Tensor, row: Tensor, col: Tensor, val_index: Optional[Tensor] = None) -> Tensor:
    """``left[row] * right[col] * val`` for (n_rows, d) / (n_cols, d) node features and (nnz, d) tuple values -- or, with
    `val_index`, ``... * val[val_index]`` for a small (n_types, d) table.  `row` / `col` / `val_index` are persistent
    int64 index arrays of the tuple pattern (plans are cached on them)."""
    require_device(left, right, val, row, col, val_index)
    assert left.dim() == right.dim() == val.dim() == 2
    row32, col32 = narrow_i32(row), narrow_i32(col)
    vidx32 = None if val_index is None else narrow_i32(val_index)
    key = (row32, col32, vidx32)
    p_row = cached_plan(row, left.shape[0], "pair-row")
    p_col = cached_plan(col, right.shape[0], "pair-col", assume_sorted=False)
    by_row = (p_row,) + _grouped(p_row, key, col32, vidx32)
    by_col = (p_col,) + _grouped(p_col, key, row32, vidx32)
    by_val = None
    if val_index is not None:
        p_val = cached_plan(val_index, val.shape[0], "pair-val", assume_sorted=False)
        by_val = (p_val,) + _grouped(p_val, key, row32, col32)
    return _PairProduct.apply(left, right, val, row32, col32, vidx32, by_row, by_col, by_val)


# --------------------------------------------------------------------------
# integer planner primitives
# --------------------------------------------------------------------------
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


# --------------------------------------------------------------------------
# masked (dense) path
# --------------------------------------------------------------------------
def _mask_u8(mask: Tensor) -> Tensor:
    """bool mask as a uint8 view (no copy), cached on the mask tensor object."""
    c = getattr(mask, "_pygho_u8", None)
    if c is None:
        c = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous().to(torch.uint8)
        try:
            mask._pygho_u8 = c
        except Exception:
            pass
    return c


def _rows_d(data: Tensor, masked_dim: int) -> Tuple[int, int]:
    rows = 1
    for s in data.shape[:masked_dim]:
        rows *= s
    d = 1
    for s in data.shape[masked_dim:]:
        d *= s
    return rows, d


class _MaskedFill(torch.autograd.Function):
    """out = mask ? data : value; gradient flows through the unmasked entries only."""

    @staticmethod
    def forward(ctx, data: Tensor, mask: Tensor, value: float):
        dev = require_device(data, mask)
        data = data.contiguous()
        rows, d = _rows_d(data, mask.dim())
        out = torch.empty_like(data)
        m8 = _mask_u8(mask)
        check(lib().pygho_masked_fill(ptr(out), ptr(data), ptr(m8), float(value), rows, d, dtype_code(data),
                                      stream_ptr(dev)), "masked_fill")
        ctx.mask = mask
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        return _MaskedFill.apply(g, ctx.mask, 0.0), None, None


def masked_fill(data: Tensor, mask: Tensor, value: float) -> Tensor:
    return _MaskedFill.apply(data, mask, value)


class _MaskedReduce(torch.autograd.Function):
    """reduce ONE masked dim `dim` of data (masked dims first, dense dims last)."""

    @staticmethod
    def forward(ctx, data: Tensor, mask: Tensor, dim: int, aggr: str):
        dev = require_device(data, mask)
        data = data.contiguous()
        md = mask.dim()
        outer = 1
        for s in data.shape[:dim]:
            outer *= s
        r = data.shape[dim]
        inner = 1
        for s in data.shape[dim + 1:md]:
            inner *= s
        d = 1
        for s in data.shape[md:]:
            d *= s
        oshape = tuple(data.shape[:dim]) + tuple(data.shape[dim + 1:])
        mshape = tuple(mask.shape[:dim]) + tuple(mask.shape[dim + 1:])
        out = torch.empty(oshape, dtype=data.dtype, device=dev)
        omask = torch.empty(mshape, dtype=torch.uint8, device=dev)
        m8 = _mask_u8(mask)
        check(lib().pygho_masked_reduce(ptr(out), ptr(omask), ptr(data), ptr(m8), outer, r, inner, d, dtype_code(data),
                                        AGGR_CODE[aggr], stream_ptr(dev)), "masked_reduce")
        ctx.meta = (m8, outer, r, inner, d, aggr, tuple(data.shape))
        ctx.save_for_backward(*((data, out) if aggr in ("max", "min") else ()))
        ctx.mark_non_differentiable(omask)
        return out, omask

    @staticmethod
    def backward(ctx, g: Tensor, _gm):
        m8, outer, r, inner, d, aggr, shape = ctx.meta
        g = g.contiguous()
        data = fwd = None
        if aggr in ("max", "min"):
            data, fwd = ctx.saved_tensors
        gdata = torch.empty(shape, dtype=g.dtype, device=g.device)
        check(lib().pygho_masked_reduce_bwd(ptr(gdata), ptr(g), ptr(data), ptr(fwd), ptr(m8), outer, r, inner, d,
                                            dtype_code(g), AGGR_CODE[aggr], stream_ptr(g.device)), "masked_reduce_bwd")
        return gdata, None, None, None


def masked_reduce(data: Tensor, mask: Tensor, dim: int, aggr: str) -> Tuple[Tensor, Tensor]:
    """(reduced data, reduced mask as bool) over one masked dim."""
    out, om = _MaskedReduce.apply(data, mask, dim, aggr)
    return out, om.view(torch.bool)


class _MaskedBroadcast(torch.autograd.Function):
    """out[o, k, i] = mask[o, k, i] ? src[o, i] : value   (unpooling along one new masked dim)."""

    @staticmethod
    def forward(ctx, src: Tensor, mask: Tensor, dim: int, value: float, src_masked_dim: int):
        dev = require_device(src, mask)
        src = src.contiguous()
        outer = 1
        for s in src.shape[:dim]:
            outer *= s
        inner = 1
        for s in src.shape[dim:src_masked_dim]:
            inner *= s
        d = 1
        for s in src.shape[src_masked_dim:]:
            d *= s
        r = mask.shape[dim]
        oshape = tuple(src.shape[:dim]) + (r,) + tuple(src.shape[dim:])
        out = torch.empty(oshape, dtype=src.dtype, device=dev)
        m8 = _mask_u8(mask)
        check(lib().pygho_masked_broadcast(ptr(out), ptr(src), ptr(m8), float(value), outer, r, inner, d,
                                           dtype_code(src), stream_ptr(dev)), "masked_broadcast")
        ctx.meta = (mask, dim)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        mask, dim = ctx.meta
        gs, _ = _MaskedReduce.apply(g, mask, dim, "sum")
        return gs, None, None, None, None


def masked_broadcast(src: Tensor, mask: Tensor, dim: int, value: float, src_masked_dim: int) -> Tensor:
    return _MaskedBroadcast.apply(src, mask, dim, value, src_masked_dim)


def pad_stack(src: Tensor, start: Tensor, shape: Tensor, max_shape) -> Tuple[Tensor, Tensor]:
    """ragged per-graph grids -> padded (nb, *max_shape, *dense) + bool mask (`pygho_pad_stack`; hodata/MaData.py:108-214).
    graph b owns the rows [start[b], start[b] + prod(shape[b])) of src as a row-major grid of shape[b]."""
    dev = require_device(src, start, shape)
    src = src.contiguous()
    start = start.to(torch.int64).contiguous()
    shape = shape.to(torch.int64).contiguous()
    nb, nd = shape.shape
    assert len(max_shape) == nd and 1 <= nd <= 3, "1 to 3 grid dims"
    assert start.numel() == nb + 1
    m = [1] * (3 - nd) + [int(v) for v in max_shape]
    tail = tuple(src.shape[1:])
    row_bytes = src.element_size()
    for t in tail:
        row_bytes *= t
    out = torch.empty((nb,) + tuple(int(v) for v in max_shape) + tail, dtype=src.dtype, device=dev)
    mask = torch.empty((nb,) + tuple(int(v) for v in max_shape), dtype=torch.uint8, device=dev)
    check(lib().pygho_pad_stack(ptr(out), ptr(mask), ptr(src), ptr(start), ptr(shape), nb, nd, m[0], m[1], m[2], row_bytes,
                                src.shape[0], stream_ptr(dev)), "pad_stack")
    return out, mask.view(torch.bool)


def dense_adj(edge_index: Tensor, edge_batch: Tensor, edge_attr: Tensor, n: int, nb: int, filled_value=0) -> Tuple[Tensor, Tensor]:
    """(nb, n, n, *dense) filled with `filled_value`, edge_attr scattered at (edge_batch, edge_index[0], edge_index[1]), + mask
    (`pygho_dense_adj`; hodata/MaData.py:25-72)."""
    dev = require_device(edge_index, edge_batch, edge_attr)
    edge_attr = edge_attr.contiguous()
    eb = edge_batch.to(torch.int64).contiguous()
    er, ec = edge_index[0].to(torch.int64).contiguous(), edge_index[1].to(torch.int64).contiguous()
    tail = tuple(edge_attr.shape[1:])
    es = edge_attr.element_size()
    row_bytes = es
    for t in tail:
        row_bytes *= t
    out = torch.empty((nb, n, n) + tail, dtype=edge_attr.dtype, device=dev)
    mask = torch.empty((nb, n, n), dtype=torch.uint8, device=dev)
    view = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[es]
    bits = int(torch.tensor([filled_value], dtype=edge_attr.dtype).view(view).item()) & ((1 << (8 * es)) - 1)
    check(lib().pygho_dense_adj(ptr(out), ptr(mask), ptr(edge_attr), ptr(eb), ptr(er), ptr(ec), eb.numel(), nb, n, row_bytes,
                                bits, es, stream_ptr(dev)), "dense_adj")
    return out, mask.view(torch.bool)


def pair_combine_supported(data: Tensor) -> bool:
    return (data.is_cuda and data.dim() == 4 and data.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and (data.shape[-1] * data.element_size()) % 16 == 0 and data.shape[-1] * data.element_size() <= 4096)


def masked_pair_combine(base: Optional[Tensor], row_term: Optional[Tensor], col_term: Optional[Tensor],
                        diag_term: Optional[Tensor], replace_diag: bool, mask: Optional[Tensor], shape, dtype, device) -> Tensor:
    """out[b,i,j] = mask ? ((base[b,i,j] + row_term[b,i]) + col_term[b,j]) : 0; diag_term[b,i] is added on / replaces the
    diagonal (no autograd; `pygho_masked_pair_combine`)."""
    nb, n1, n2, d = shape
    ops = [None if t is None else t.contiguous() for t in (base, row_term, col_term, diag_term)]
    for t in ops:
        assert t is None or (t.dtype == dtype and t.device == device)
    out = torch.empty(shape, dtype=dtype, device=device)
    m8 = None if mask is None else _mask_u8(mask)
    check(lib().pygho_masked_pair_combine(ptr(out), ptr(ops[0]), ptr(ops[1]), ptr(ops[2]), ptr(ops[3]), 1 if replace_diag else 0,
                                          ptr(m8), nb, n1, n2, d, DTYPE_CODE[dtype], stream_ptr(device)), "masked_pair_combine")
    return out


def _diag_rows(data: Tensor) -> Tensor:
    """(b, n1, n2, d) -> (b, min(n1, n2), d): rows (b, i, i)."""
    return torch.diagonal(data, 0, 1, 2).movedim(-1, 1)


class _PairViews(torch.autograd.Function):
    """(data (b, n1, n2, d), mask) -> (diagonal rows, sum over dim 1, sum over dim 2): the three node-level views a
    subgraph layer takes of a 2-D representation; their gradients return to the tuple level in ONE pass."""

    @staticmethod
    def forward(ctx, data: Tensor, mask: Tensor, want_dim2: bool = True, chain: bool = False):
        require_device(data, mask)
        data = data.contiguous()
        dmask = torch.diagonal(mask, 0, 1, 2)
        dg = torch.where(dmask.unsqueeze(-1), _diag_rows(data), torch.zeros((), dtype=data.dtype, device=data.device))
        s1, _ = _MaskedReduce.apply(data, mask, 1, "sum")
        s2 = _MaskedReduce.apply(data, mask, 2, "sum")[0] if want_dim2 else data.new_empty((0,) + tuple(data.shape[2:]))
        ctx.mask = mask
        ctx.meta = (tuple(data.shape), data.dtype, want_dim2)
        ctx.set_materialize_grads(False)
        if chain:       # `data` again as an output: a later consumer's gradient arrives here and rides in the combine pass as its base
            return dg.contiguous(), s1, s2, data.view_as(data)
        return dg.contiguous(), s1, s2

    @staticmethod
    def backward(ctx, g_dg, g_s1, g_s2, g_chain=None):
        shape, dtype, want_dim2 = ctx.meta
        dev = ctx.mask.device
        cast = lambda t: None if t is None else t.to(dtype).contiguous()
        if not want_dim2:
            g_s2 = None
        if g_dg is None and g_s1 is None and g_s2 is None:
            return g_chain, None, None, None
        return masked_pair_combine(cast(g_chain), cast(g_s2), cast(g_s1), cast(g_dg), False, ctx.mask, shape, dtype, dev), None, None, None


def pair_views(data: Tensor, mask: Tensor, want_dim2: bool = True, chain: bool = False):
    """`chain`: also returns `data` as an autograd output (same storage) for the consumer that comes after the views, so that its
    gradient is summed with the views' gradients inside their one combine pass instead of by a separate accumulation."""
    return _PairViews.apply(data, mask, want_dim2, chain)


def _dense_linear(flat: Tensor, w_in_out: Tensor, addend: Optional[Tensor] = None) -> Tensor:
    """flat @ w (+ addend) with w stored (in, out): the streaming MFMA kernel when its shape is supported, else the library."""
    if rowblock_linear_supported(flat, w_in_out.shape[1]) and w_in_out.shape[0] == w_in_out.shape[1]:
        return rowblock_linear(flat, w_in_out.t().contiguous(), None, addend)[0]
    out = flat @ w_in_out
    return out if addend is None else out + addend


class _PairLinearMix(torch.autograd.Function):
    """out[b,i,j] = mask ? (i == j ? dg[b,i] : ((x[b,i,j] @ w_x + y[b,i,j] @ w_y) + u[b,i]) + v[b,j]) : 0

    The recombination step of SUNConv (reference Conv.py:338-362) after the linear map has been pulled through the
    broadcasts: two tuple-level GEMMs (the second with the first in its epilogue), then one pass adding the node-level
    terms and selecting the diagonal.  Backward: one masked copy (off-diagonal part of g), two masked reductions, two
    input-gradient GEMMs, two weight gradients."""

    @staticmethod
    def forward(ctx, x, y, w_x, w_y, u, v, dg, mask):
        require_device(x, y, mask)
        x, y = x.contiguous(), y.contiguous()
        shape = tuple(x.shape)
        d = shape[-1]
        a = _dense_linear(x.reshape(-1, d), w_x)
        ab = _dense_linear(y.reshape(-1, d), w_y, a)
        out = masked_pair_combine(ab.reshape(shape), u, v, dg, True, mask, shape[:3] + (w_x.shape[1],), x.dtype, x.device)
        ctx.save_for_backward(x, y, w_x, w_y)
        ctx.mask = mask
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, w_x, w_y = ctx.saved_tensors
        mask = ctx.mask
        g = g.contiguous()
        d_in, d_out = w_x.shape
        n1, n2 = mask.shape[1], mask.shape[2]
        off = getattr(mask, "_pygho_offdiag", None)
        if off is None:
            eye = torch.eye(n1, n2, dtype=torch.bool, device=mask.device)
            off = mask & ~eye
            try:
                mask._pygho_offdiag = off
            except Exception:
                pass
        goff = _MaskedFill.apply(g, off, 0.0)
        gu, _ = _MaskedReduce.apply(goff, off, 2, "sum")
        gv, _ = _MaskedReduce.apply(goff, off, 1, "sum")
        dmask = torch.diagonal(mask, 0, 1, 2)
        gdg = torch.where(dmask.unsqueeze(-1), _diag_rows(g), torch.zeros((), dtype=g.dtype, device=g.device))
        gf = goff.reshape(-1, d_out)
        xf, yf = x.reshape(-1, d_in), y.reshape(-1, d_in)
        gx = _dense_linear(gf, w_x.t()).reshape(x.shape) if ctx.needs_input_grad[0] else None
        gy = _dense_linear(gf, w_y.t()).reshape(y.shape) if ctx.needs_input_grad[1] else None
        gwx = weight_grad_splitk(gf, xf, w_x.dtype).t() if ctx.needs_input_grad[2] else None
        gwy = weight_grad_splitk(gf, yf, w_y.dtype).t() if ctx.needs_input_grad[3] else None
        return gx, gy, gwx, gwy, gu, gv, gdg, None


def pair_linear_mix(x, y, w_x, w_y, u, v, dg, mask):
    return _PairLinearMix.apply(x, y, w_x, w_y, u, v, dg, mask)


def pair_gather_combine(base: Optional[Tensor], row_term: Optional[Tensor], col_term: Optional[Tensor],
                        diag_term: Optional[Tensor], replace_diag: bool, ri32: Tensor, ci32: Tensor, d: int, dtype, device) -> Tensor:
    """sparse twin of masked_pair_combine: out[t] = (base[t] + row_term[ri[t]]) + col_term[ci[t]], diag_term[ri[t]] added on /
    replacing the tuples with ri == ci (no autograd; `pygho_pair_gather_combine`)."""
    ops = [None if t is None else t.contiguous() for t in (base, row_term, col_term, diag_term)]
    for t in ops:
        assert t is None or (t.dtype == dtype and t.device == device)
    n_rows = ri32.numel()
    out = torch.empty((n_rows, d), dtype=dtype, device=device)
    check(lib().pygho_pair_gather_combine(ptr(out), ptr(ops[0]), ptr(ops[1]), ptr(ops[2]), ptr(ops[3]), 1 if replace_diag else 0,
                                          ptr(ri32), ptr(ci32), n_rows, d, DTYPE_CODE[dtype], stream_ptr(device)),
          "pair_gather_combine")
    return out


def pair_gather_supported(values: Tensor) -> bool:
    return (values.is_cuda and values.dim() == 2 and values.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and (values.shape[1] * values.element_size()) % 16 == 0 and values.shape[1] * values.element_size() <= 4096)


def _matched_rows(src: Tensor, pos: Tensor) -> Tensor:
    """out[r] = pos[r] >= 0 ? src[pos[r]] : 0 (no autograd)."""
    return row_gather(src, narrow_i32(pos.clamp_min(0)), (pos >= 0).to(_I32))


class _SparsePairViews(torch.autograd.Function):
    """(values of a sparse 2-D representation) -> (diagonal rows (n, d), sum over tuples sharing index 0, sum over tuples
    sharing index 1); the three gradients return to the tuples in ONE gather pass."""

    @staticmethod
    def forward(ctx, values: Tensor, ri: Tensor, ci: Tensor, diag_pos: Tensor, n: int, want_rows: bool = True, chain: bool = False):
        require_device(values, ri, ci, diag_pos)
        values = values.contiguous()
        dg = _matched_rows(values, diag_pos)
        # want_rows = False: the per-i sums are not needed by the caller (SUNConv takes only the diagonal and the per-j sums of
        # the aggregated representation): one pooling pass less, and an empty placeholder in its place
        s_r = (_ScatterReduce.apply(values, cached_plan(ri, n, "scatter"), narrow_i32(ri), "sum") if want_rows
               else values.new_empty((0, values.shape[1])))
        s_c = _ScatterReduce.apply(values, cached_plan(ci, n, "scatter"), narrow_i32(ci), "sum")
        ctx.idx = (narrow_i32(ri), narrow_i32(ci))
        ctx.meta = (values.shape[1], values.dtype, want_rows)
        ctx.set_materialize_grads(False)
        if chain:       # see _PairViews
            return dg, s_r, s_c, values.view_as(values)
        return dg, s_r, s_c

    @staticmethod
    def backward(ctx, g_dg, g_r, g_c, g_chain=None):
        d, dtype, want_rows = ctx.meta
        ri32, ci32 = ctx.idx
        cast = lambda t: None if t is None else t.to(dtype).contiguous()
        if not want_rows:
            g_r = None
        if g_dg is None and g_r is None and g_c is None:
            return g_chain, None, None, None, None, None, None
        return (pair_gather_combine(cast(g_chain), cast(g_r), cast(g_c), cast(g_dg), False, ri32, ci32, d, dtype, ri32.device),
                None, None, None, None, None, None)


def sparse_pair_views(values: Tensor, ri: Tensor, ci: Tensor, diag_pos: Tensor, n: int, want_rows: bool = True, chain: bool = False):
    return _SparsePairViews.apply(values, ri, ci, diag_pos, n, want_rows, chain)


class _SparsePairBroadcast(torch.autograd.Function):
    """out[t] = u[i] (+ v[j]) for the tuple t = (i, j): two node-level tensors broadcast onto a sparse 2-D pattern and added in one
    pass (`pygho_pair_gather_combine`); the gradients are the two segment sums of the output gradient."""

    @staticmethod
    def forward(ctx, u, v, ri, ci, n):
        require_device(u, v, ri, ci)
        ri32, ci32 = narrow_i32(ri), narrow_i32(ci)
        ctx.idx = (ri, ci, n, v is not None)
        return pair_gather_combine(None, u, v, None, False, ri32, ci32, u.shape[1], u.dtype, u.device)

    @staticmethod
    def backward(ctx, g):
        ri, ci, n, has_v = ctx.idx
        g = g.contiguous()
        gu = seg_reduce_rows(g, cached_plan(ri, n, "scatter"), "sum") if ctx.needs_input_grad[0] else None
        gv = seg_reduce_rows(g, cached_plan(ci, n, "scatter"), "sum") if has_v and ctx.needs_input_grad[1] else None
        return gu, gv, None, None, None


def sparse_pair_broadcast(u: Tensor, v: Optional[Tensor], ri: Tensor, ci: Tensor, n: int) -> Tensor:
    return _SparsePairBroadcast.apply(u.contiguous(), None if v is None else v.contiguous(), ri, ci, n)


class _SparsePairLinearMix(torch.autograd.Function):
    """out[t] = (i == j) ? dg[i] : ((x[t] @ w_x + y[t] @ w_y) + u[i]) + v[j] for the tuple t = (i, j): `_PairLinearMix` on the
    sparse layout (SUNConv mode "SS")."""

    @staticmethod
    def forward(ctx, x, y, w_x, w_y, u, v, dg, ri, ci, diag_pos, n):
        require_device(x, y, ri, ci)
        x, y = x.contiguous(), y.contiguous()
        ri32, ci32 = narrow_i32(ri), narrow_i32(ci)
        ab = _dense_linear(y, w_y, _dense_linear(x, w_x))
        out = pair_gather_combine(ab, u, v, dg, True, ri32, ci32, w_x.shape[1], x.dtype, x.device)
        ctx.save_for_backward(x, y, w_x, w_y)
        ctx.idx = (ri, ci, diag_pos, n)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, w_x, w_y = ctx.saved_tensors
        ri, ci, diag_pos, n = ctx.idx
        ri32, ci32 = narrow_i32(ri), narrow_i32(ci)
        g = g.contiguous()
        d_out = w_x.shape[1]
        zeros = torch.zeros((n, d_out), dtype=g.dtype, device=g.device)
        goff = pair_gather_combine(g, None, None, zeros, True, ri32, ci32, d_out, g.dtype, g.device)   # diagonal tuples zeroed
        gu = _ScatterReduce.apply(goff, cached_plan(ri, n, "scatter"), ri32, "sum")
        gv = _ScatterReduce.apply(goff, cached_plan(ci, n, "scatter"), ci32, "sum")
        gdg = _matched_rows(g, diag_pos)
        gx = _dense_linear(goff, w_x.t()) if ctx.needs_input_grad[0] else None
        gy = _dense_linear(goff, w_y.t()) if ctx.needs_input_grad[1] else None
        gwx = weight_grad_splitk(goff, x, w_x.dtype).t() if ctx.needs_input_grad[2] else None
        gwy = weight_grad_splitk(goff, y, w_y.dtype).t() if ctx.needs_input_grad[3] else None
        return gx, gy, gwx, gwy, gu, gv, gdg, None, None, None, None


def sparse_pair_linear_mix(x, y, w_x, w_y, u, v, dg, ri, ci, diag_pos, n):
    return _SparsePairLinearMix.apply(x, y, w_x, w_y, u, v, dg, ri, ci, diag_pos, n)


USE_BMM_BLOCKS = True         # rows of whole 256-B multiples, k <= 64: the multi-block matrix-core kernel serves every mask pattern
USE_BMM_LISTS = True          # (other shapes) masked contraction with a sparse-masked operand: neighbour-list kernel instead of the dense MFMA one
BMM_LIST_DENSITY = 0.15       # ... when at most this fraction of that operand's positions is unmasked


USE_BMM_EXTENTS = True        # matrix-core contraction: stage / multiply only up to the last unmasked row, k and column of each batch element


def _mask_extents(amask, bmask, omask, nb, ni, nk, nj, a_kfirst: bool, b_kfirst: bool) -> Optional[Tensor]:
    """(nb, 3) int32 (ei, ek, ej) per batch element (`pygho_mask_extents`), cached on the first mask of the triple (the cache
    entry keeps the masks alive, so their identities cannot be recycled); None when no mask is given."""
    holder = amask if amask is not None else (bmask if bmask is not None else omask)
    if holder is None:
        return None
    cache = getattr(holder, "_pygho_extents", None)
    if cache is None:
        cache = {}
        try:
            holder._pygho_extents = cache
        except Exception:
            pass
    ver = lambda m: None if m is None else (id(m), m._version)
    key = (ver(amask), ver(bmask), ver(omask), ni, nk, nj, a_kfirst, b_kfirst)
    hit = cache.get(key)
    if hit is None:
        dev = holder.device
        ext = torch.empty((nb, 3), dtype=torch.int32, device=dev)
        check(lib().pygho_mask_extents(ptr(ext), ptr(amask), ptr(bmask), ptr(omask), nb, ni, nk, nj, 1 if a_kfirst else 0,
                                       1 if b_kfirst else 0, stream_ptr(dev)), "mask_extents")
        if len(cache) > 8:
            cache.clear()
        hit = cache[key] = (ext, amask, bmask, omask)
    return hit[0]


def _mask_density(m8: Optional[Tensor]) -> float:
    """unmasked fraction of a uint8 mask, computed once per mask tensor object (one small reduction + one sync per batch)."""
    if m8 is None:
        return 1.0
    c = getattr(m8, "_pygho_density", None)
    if c is None or c[0] != m8._version:
        c = (m8._version, float(m8.sum(dtype=torch.int64).item()) / max(1, m8.numel()))
        try:
            m8._pygho_density = c
        except Exception:
            pass
    return c[1]


def _mask_lists(m8: Tensor, nb: int, nk: int, nc: int, k_first: bool):
    """(list (nb, nc, roundup4(nk)) int16, -1 terminated; count (nb, nc) int32) of the unmasked k per (b, c), cached on the mask."""
    cache = getattr(m8, "_pygho_lists", None)
    if cache is None:
        cache = {}
        try:
            m8._pygho_lists = cache
        except Exception:
            pass
    key = (m8._version, nk, nc, k_first)
    if key not in cache:
        dev = m8.device
        lst = torch.empty((nb, nc, (nk + 3) & ~3), dtype=torch.int16, device=dev)       # -1 terminated rows, 8-byte groups
        cnt = torch.empty((nb, nc), dtype=torch.int32, device=dev)
        check(lib().pygho_mask_lists(ptr(lst), ptr(cnt), ptr(m8), nb, nk, nc, 1 if k_first else 0, stream_ptr(dev)), "mask_lists")
        cache[key] = (lst, cnt)                                  # column and row lists of one mask coexist (forward / backward)
    return cache[key]


def _bmm_launch(A: Tensor, B: Tensor, amask, bmask, omask, nb, ni, nk, nj, d, a_kfirst: bool, b_kfirst: bool) -> Tensor:
    dev = require_device(A, B, amask, bmask, omask)
    out = torch.empty((nb, ni, nj, d), dtype=A.dtype, device=dev)
    # the multi-block matrix-core kernel (csrc/masked_bmm_blocks.h) serves every contraction whose rows are whole 256-B
    # multiples and whose contracted dim fits its 64-bit row bitmasks -- including a sparse operand or output mask, where it
    # beats the neighbour-list kernels below (forward 156 vs 185 us, forward + both gradients 0.47 vs 0.55 ms at
    # (1024, 37, 37, 128) bf16) and needs no density probe (a reduction + a host synchronisation per new mask)
    blocks_ok = USE_BMM_BLOCKS and (d * A.element_size()) % 256 == 0 and nk <= 64
    if (USE_BMM_LISTS and not blocks_ok and 0 < nk <= 32767 and nb * ni * nj > 0 and (d * A.element_size()) % 16 == 0
            and d * A.element_size() <= 4096 and A.dtype in (torch.float32, torch.bfloat16, torch.float16)):
        da, db = _mask_density(amask), _mask_density(bmask)
        if min(da, db) <= BMM_LIST_DENSITY:
            on_j = db <= da                                      # the sparser operand supplies the lists
            if on_j:
                lst, cnt = _mask_lists(bmask, nb, nk, nj, b_kfirst)
                dense_mask = amask
            else:
                lst, cnt = _mask_lists(amask, nb, nk, ni, a_kfirst)
                dense_mask = bmask
            check(lib().pygho_masked_bmm_lists(ptr(out), ptr(A), ptr(B), ptr(dense_mask), ptr(omask), ptr(lst), ptr(cnt),
                                               1 if on_j else 0, nb, ni, nk, nj, d, 1 if a_kfirst else 0, 1 if b_kfirst else 0,
                                               dtype_code(A), stream_ptr(dev)), "masked_bmm_lists")
            return out
        # output-sparse: two dense operands, few outputs wanted (the gradient of an adjacency's values)
        if (omask is not None and nb * ni * nk * d * A.element_size() < 2 ** 31 - 1 and nb * nk * nj * d * A.element_size() < 2 ** 31 - 1
                and ni <= 32767 and _mask_density(omask) <= BMM_LIST_DENSITY):
            lst, cnt = _mask_lists(omask, nb, ni, nj, True)
            maxc = getattr(cnt, "_pygho_max", None)
            if maxc is None:
                maxc = int(cnt.max().item()) if cnt.numel() else 0
                cnt._pygho_max = maxc
            out.zero_()
            check(lib().pygho_masked_bmm_outlists(ptr(out), ptr(A), ptr(B), ptr(amask), ptr(bmask), ptr(lst), maxc, nb, ni, nk, nj, d,
                                                  1 if a_kfirst else 0, 1 if b_kfirst else 0, dtype_code(A), stream_ptr(dev)),
                  "masked_bmm_outlists")
            return out
    ext = _mask_extents(amask, bmask, omask, nb, ni, nk, nj, a_kfirst, b_kfirst) if USE_BMM_EXTENTS else None
    if ext is not None:
        check(lib().pygho_masked_bmm_clipped(ptr(out), ptr(A), ptr(B), ptr(amask), ptr(bmask), ptr(omask), ptr(ext), nb, ni, nk, nj, d,
                                             1 if a_kfirst else 0, 1 if b_kfirst else 0, dtype_code(A), stream_ptr(dev)),
              "masked_bmm_clipped")
        return out
    check(lib().pygho_masked_bmm(ptr(out), ptr(A), ptr(B), ptr(amask), ptr(bmask), ptr(omask), nb, ni, nk, nj, d,
                                 1 if a_kfirst else 0, 1 if b_kfirst else 0, dtype_code(A), stream_ptr(dev)), "masked_bmm")
    return out


class _MaskedBmm(torch.autograd.Function):
    """out[b,i,j,:] = omask ? sum_k A[b,i,k,:] * B[b,k,j,:] : 0 on the matrix cores; A stored (b,i,k,d) or
    k-first (b,k,i,d), B stored (b,k,j,d) (k-first) or (b,j,k,d).  Masks are uint8 or None (= all valid)."""

    @staticmethod
    def forward(ctx, A, B, amask, bmask, omask, dims, a_kfirst, b_kfirst):
        nb, ni, nk, nj, d = dims
        out = _bmm_launch(A, B, amask, bmask, omask, nb, ni, nk, nj, d, a_kfirst, b_kfirst)
        ctx.save_for_backward(A, B)
        ctx.meta = (amask, bmask, omask, dims, a_kfirst, b_kfirst)
        return out

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        amask, bmask, omask, (nb, ni, nk, nj, d), akf, bkf = ctx.meta
        g = g.contiguous()
        gA = gB = None
        if ctx.needs_input_grad[0]:
            if not akf:   # gA[b,i,k] = sum_j g[b,i,j] * B[k,j]
                gA = _bmm_launch(g, B, omask, bmask, amask, nb, ni, nj, nk, d, False, not bkf)
            else:         # gA[b,k,i] = sum_j B[k,j] * g[b,i,j]
                gA = _bmm_launch(B, g, bmask, omask, amask, nb, nk, nj, ni, d, not bkf, False)
        if ctx.needs_input_grad[1]:
            if bkf:       # gB[b,k,j] = sum_i A[i,k] * g[b,i,j]
                gB = _bmm_launch(A, g, amask, omask, bmask, nb, nk, ni, nj, d, not akf, True)
            else:         # gB[b,j,k] = sum_i g[b,i,j] * A[i,k]
                gB = _bmm_launch(g, A, omask, amask, bmask, nb, nj, ni, nk, d, True, not akf)
        return gA, gB, None, None, None, None, None, None


def masked_bmm(A, B, amask, bmask, omask, nb, ni, nk, nj, d, a_kfirst, b_kfirst) -> Tensor:
    """channel-innermost batched contraction; pads d up to the kernel's channel granule when needed."""
    gran = 8 if A.dtype in (torch.bfloat16, torch.float16) else 4
    if A.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise TypeError(f"pygho_amd: masked_bmm supports bf16 / f16 / f32, got {A.dtype}")
    pad = (-d) % gran
    if pad:
        A = torch.nn.functional.pad(A, (0, pad))
        B = torch.nn.functional.pad(B, (0, pad))
    out = _MaskedBmm.apply(A.contiguous(), B.contiguous(), amask, bmask, omask, (nb, ni, nk, nj, d + pad), a_kfirst, b_kfirst)
    return out[..., :d] if pad else out


# --------------------------------------------------------------------------
# fused BatchNorm + activation (dense neighbour of the aggregation)
# --------------------------------------------------------------------------
ACT_CODE = {"none": 0, "relu": 1, "silu": 2}


def bn_act_supported(x: Tensor) -> bool:
    return (x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and x.shape[0] > 1
            and int(lib().pygho_bn_workspace(x.shape[0], x.shape[1], dtype_code(x))) > 0)


def _bn_forward(x: Optional[Tensor], weight, bias, running_mean, running_var, training: bool, eps: float, act: str,
                fold_momentum: Optional[float] = None, partial: Optional[Tuple[Tensor, Tensor]] = None,
                apply: bool = True, addend: Optional[Tensor] = None, producer=None):
    """(y, mean, var, saved) of act(batch_norm(x)) (+ addend: a residual row added inside the activation pass) for a contiguous 2-D x.  Statistics, 1/sqrt(var + eps), the fused
    scale / shift and (with `fold_momentum`) the running-average update all come out of ONE finalisation kernel;
    `partial` = (per-block shifted sums, their shift) when the producer of x already took the sums (rowblock_linear).
    `apply=False`: y is not formed; (scale, shift) are returned in its place for a consumer that applies them on load.
    `producer=(x_in, wl, lin_bias)` with x = None: the BatchNorm input x_in @ wl^T + lin_bias is NOT in memory; the statistics
    came from `partial` (training) or are the running ones, and y is produced by recomputing the product inside the
    activation pass (`rowblock_linear_bn_act`)."""
    src = x if x is not None else producer[0]
    dev = src.device
    m, c = src.shape
    dt = dtype_code(src)
    st = stream_ptr(dev)
    ws = None
    if x is not None:
        nbytes = int(lib().pygho_bn_workspace(m, c, dt))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    else:
        assert apply and (partial is not None or not training)
    w32 = None if weight is None else weight.detach().float().contiguous()
    b32 = None if bias is None else bias.detach().float().contiguous()
    if training:
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        var = torch.empty(c, dtype=torch.float32, device=dev)
    else:
        mean, var = running_mean.float().clone(), running_var.float().clone()
    invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(3))
    fold = training and fold_momentum is not None
    if training and partial is not None:
        sums, sum_shift = partial
        check(lib().pygho_bn_finalize(ptr(mean), ptr(var), ptr(invstd), ptr(scale), ptr(shift), ptr(sums), sums.shape[0],
                                      ptr(sum_shift), m, c, ptr(w32), ptr(b32), float(eps), ptr(running_mean) if fold else None,
                                      ptr(running_var) if fold else None, float(fold_momentum or 0.0), st), "bn_finalize")
    else:
        check(lib().pygho_bn_prepare(ptr(mean), ptr(var), ptr(invstd), ptr(scale), ptr(shift), ptr(x) if training else None, m, c,
                                     ptr(w32), ptr(b32), float(eps), ptr(running_mean) if fold else None,
                                     ptr(running_var) if fold else None, float(fold_momentum or 0.0), ptr(ws), dt, st),
              "bn_prepare")
    if not apply:
        return (scale, shift), mean, var, (mean, invstd, w32, b32, ws)
    if x is None:
        y = rowblock_linear_bn_act(producer[0], producer[1], producer[2], scale, shift, act, addend)
        return y, mean, var, (mean, invstd, w32, b32, ws)
    y = torch.empty_like(x)
    if addend is not None:
        check(lib().pygho_bn_act_fwd_add(ptr(y), ptr(x), ptr(addend.contiguous()), ptr(scale), ptr(shift), m, c, ACT_CODE[act], dt, st),
              "bn_act_fwd_add")
    else:
        check(lib().pygho_bn_act_fwd(ptr(y), ptr(x), ptr(scale), ptr(shift), m, c, ACT_CODE[act], dt, st), "bn_act_fwd")
    return y, mean, var, (mean, invstd, w32, b32, ws)


def _bn_backward(x: Tensor, gy: Tensor, saved, training: bool, act: str, want_colsum: bool = False):
    """(dx, d bn.bias, d bn.weight, column sums of dx or None)."""
    mean, invstd, w32, b32, ws = saved
    m, c = x.shape
    dev = x.device
    dx = torch.empty_like(x)
    s1 = torch.empty(c, dtype=torch.float32, device=dev)
    s2 = torch.empty(c, dtype=torch.float32, device=dev)
    sdx = torch.empty(c, dtype=torch.float32, device=dev) if want_colsum else None
    check(lib().pygho_bn_act_bwd(ptr(dx), ptr(s1), ptr(s2), ptr(x), ptr(gy), ptr(mean), ptr(invstd), ptr(w32), ptr(b32),
                                 m, c, ACT_CODE[act], 1 if training else 0, ptr(ws), dtype_code(x), ptr(sdx),
                                 stream_ptr(dev)), "bn_act_bwd")
    return dx, s1, s2, sdx


def bn_act_supported_shape(m: int, c: int, dtype: torch.dtype) -> bool:
    return (dtype in (torch.float32, torch.bfloat16, torch.float16) and m > 1
            and int(lib().pygho_bn_workspace(m, c, DTYPE_CODE[dtype])) > 0)


class _BNAct(torch.autograd.Function):
    """y = act(batch_norm(x)); training uses batch statistics (and returns them for the running averages)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, eps, act, fold_momentum=None, addend=None):
        require_device(x, addend)
        x = x.contiguous()
        y, mean, var, saved = _bn_forward(x, weight, bias, running_mean, running_var, training, eps, act, fold_momentum,
                                          addend=None if addend is None else addend.contiguous())
        ctx.save_for_backward(x, *saved)
        ctx.meta = (training, act, weight is not None, bias is not None)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)            # no zero tensors for the statistics' (never used) gradients
        return y, mean, var

    @staticmethod
    def backward(ctx, gy, _gm, _gv):
        if gy is None:
            return (None,) * len(ctx.needs_input_grad)
        x, *saved = ctx.saved_tensors
        training, act, has_w, has_b = ctx.meta
        dx, s1, s2, _ = _bn_backward(x, gy.contiguous(), saved, training, act)
        g_add = gy if len(ctx.needs_input_grad) > 9 and ctx.needs_input_grad[9] else None     # y = act(bn(x)) + addend
        return dx, (s2 if has_w else None), (s1 if has_b else None), None, None, None, None, None, None, g_add


def _fold_momentum(bn) -> Optional[float]:
    """momentum when the running-average update can run inside the statistics kernel (f32 contiguous buffers and a
    fixed momentum; the cumulative-average mode needs the batch counter on the host and takes the torch path)."""
    if (bn.training and bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None
            and bn.running_mean.dtype == torch.float32 and bn.running_var.dtype == torch.float32
            and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous()):
        return float(bn.momentum)
    return None


def _update_running(bn, mean: Tensor, var: Tensor, n: int, folded: bool = False) -> None:
    if bn.training and bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            bn.num_batches_tracked += 1
            if folded:
                return
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var.to(bn.running_var.dtype), alpha=mom * n / max(n - 1, 1))


def batch_norm_act(x: Tensor, bn: "torch.nn.BatchNorm1d", act: str, residual: Optional[Tensor] = None) -> Tensor:
    """BatchNorm1d(x) followed by `act` (+ `residual`, added inside the activation pass), with torch's semantics (batch statistics
    + running-average update in training mode, running statistics in eval mode)."""
    training = bn.training or bn.running_mean is None
    fold = _fold_momentum(bn)
    y, mean, var = _BNAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, act, fold, residual)
    _update_running(bn, mean, var, x.shape[0], folded=fold is not None)
    return y


# --------------------------------------------------------------------------
# low-precision copies of the f32 master parameters: ONE multi-tensor copy per optimizer step instead of a cast kernel per use
# --------------------------------------------------------------------------
USE_CAST_ARENA = os.environ.get("PYGHO_CAST_ARENA", "1") != "0"
_ARENA_OF = {}          # id(parameter) -> (weakref to its arena, position); validated by identity on lookup
_ARENA_EPOCH = [0]      # bumped by whoever changes parameters behind the version counters' back (a HIP graph replay)


def invalidate_cast_arenas() -> None:
    """every arena copy is out of date (parameters were updated without their version counters moving: a replayed HIP graph
    contains the optimizer's in-place update, and replaying it does not touch Python-side versions)."""
    _ARENA_EPOCH[0] += 1


class ParamCastArena:
    """16-bit copies of a module's f32 parameters in one flat buffer (16-byte aligned views).  `refresh()` re-casts every
    parameter whose version changed since the last refresh with ONE `torch._foreach_copy_` (the training step had ~25 separate
    cast launches of 1-16 k elements, 4.7 us each); a lookup is valid only while the parameter's version is the refreshed one,
    so an in-place update that nobody told the arena about simply falls back to a direct cast."""

    def __init__(self, params, dtype: torch.dtype):
        import weakref
        self.dtype = dtype
        self.params = [p for p in params if p.is_cuda and p.dtype == torch.float32]
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8
        dev = self.params[0].device if self.params else None
        self.flat = torch.empty(total, dtype=dtype, device=dev) if self.params else None
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, self.params)]
        self.versions = [-1] * len(self.params)
        self.ptrs = [0] * len(self.params)          # storage address at the last refresh: `module.to()` / `p.data = ...` swap it
        self.epoch = -1
        ref = weakref.ref(self)
        for k in [k for k, (r, _i) in _ARENA_OF.items() if r() is None]:      # entries of arenas that are gone
            del _ARENA_OF[k]
        for i, p in enumerate(self.params):
            _ARENA_OF[id(p)] = (ref, i)

    def refresh(self) -> None:
        """Freshness is decided by the parameter's version counter and storage address.  Covered update paths: in-place ops on the
        parameter (optimizers, `p.copy_`, `load_state_dict`), `module.to()` / `p.data = t` (new storage), a replayed HIP graph
        (`GraphedStep.replay` bumps the epoch).  NOT visible from here: writes through a `.data` alias (`p.data.mul_(...)` has its
        own version counter) -- call `invalidate_cast_arenas()` after those.  Under stream capture every copy is re-cast INSIDE
        the graph: a graph that captured only forward + backward must not bake in the views of a cast that happened before it."""
        if self.epoch != _ARENA_EPOCH[0] or (self.flat is not None and torch.cuda.is_current_stream_capturing()):
            stale = list(range(len(self.params)))
        else:
            stale = [i for i, p in enumerate(self.params) if self.versions[i] != p._version or self.ptrs[i] != p.data_ptr()]
        self.epoch = _ARENA_EPOCH[0]
        if not stale:
            return
        with torch.no_grad():
            torch._foreach_copy_([self.views[i] for i in stale], [self.params[i].detach() for i in stale])
        for i in stale:
            self.versions[i] = self.params[i]._version
            self.ptrs[i] = self.params[i].data_ptr()


def ensure_cast_arena(module, dtype: Optional[torch.dtype]) -> None:
    """create (once) and refresh the cast arena of `module`'s parameters for the 16-bit compute dtype; call at the top of forward."""
    if not USE_CAST_ARENA or dtype not in (torch.bfloat16, torch.float16):
        return
    arena = module.__dict__.get("_pygho_cast_arena")
    params = list(module.parameters())
    if (arena is None or arena.dtype != dtype or len(arena.params) != sum(1 for p in params if p.is_cuda and p.dtype == torch.float32)
            or any(a is not b for a, b in zip(arena.params, (p for p in params if p.is_cuda and p.dtype == torch.float32)))):
        arena = ParamCastArena(params, dtype)
        module.__dict__["_pygho_cast_arena"] = arena
        if not module.__dict__.get("_pygho_cast_hook"):
            # belt and braces next to the version / address checks: a loaded state dict invalidates every copy
            module.register_load_state_dict_post_hook(lambda _m, _keys: invalidate_cast_arenas())
            module.__dict__["_pygho_cast_hook"] = True
    arena.refresh()


def param_as(p: Tensor, dtype: torch.dtype) -> Tensor:
    """`p` in `dtype`, without autograd: the arena's copy when it is current, a direct cast otherwise."""
    if p.dtype == dtype:
        return p
    ent = _ARENA_OF.get(id(p))
    if ent is not None:
        arena = ent[0]()
        if arena is None:
            del _ARENA_OF[id(p)]
        elif (arena.dtype == dtype and arena.epoch == _ARENA_EPOCH[0] and arena.params[ent[1]] is p
              and arena.versions[ent[1]] == p._version and arena.ptrs[ent[1]] == p.data_ptr()):
            return arena.views[ent[1]]
    return p.detach().to(dtype)


class _CastParam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, dtype):
        ctx.src_dtype = p.dtype
        # a NEW tensor object every call: autograd writes this node into the returned object, and the arena's own view object
        # would carry it -- and the AccumulateGrad node behind it, with the stream it was created on -- into the next iteration
        # (a HIP graph capture after eager steps then pulled the eager stream into the capture and crashed in hipStreamEndCapture)
        return param_as(p, dtype).detach()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.src_dtype), None


def cast_param(p: Tensor, dtype: torch.dtype) -> Tensor:
    """differentiable `p.to(dtype)` that reads the cast arena (the gradient returns in p's dtype)."""
    if p.dtype == dtype:
        return p
    return _CastParam.apply(p, dtype)


# --------------------------------------------------------------------------
# one tuple-wise block: Linear -> BatchNorm -> act [-> message passing [+ residual]]   
# --------------------------------------------------------------------------
def weight_grad_splitk(g: Tensor, x: Tensor, out_dtype: torch.dtype, want_colsum: bool = False):
    """dW = g^T x for tall (nnz ~ 10^5..10^6) operands; with `want_colsum` returns (dW, g.sum(0)).
    Square 16-bit Linears of width 64 / 128 run on the transpose-read MFMA kernel (`pygho_weight_grad`); the rest falls back
    to the library: a plain GEMM below 2^19 rows, a batched split-K product above (the BLAS heuristics pick no split-K for a
    128 x 128 output, 2.9 ms at 1.8 M rows, but the batched call costs ~3.7 ms of host time, so it only pays for huge m)."""
    m, n, k = g.shape[0], g.shape[1], x.shape[1]
    cs = None
    if (g.is_cuda and g.dtype in (torch.bfloat16, torch.float16) and x.dtype == g.dtype and n in (64, 128) and k % n == 0
            and k // n <= 8 and m >= 8192):
        g, x = g.contiguous(), x.contiguous()
        dev = g.device
        nblk = int(lib().pygho_bn_bwd_linear_dw_blocks(m))
        parts = []
        for j in range(k // n):                                       # one launch per n-wide column block of x (in_features = j n)
            cs_here = want_colsum and j == 0
            width = n * n + (2 * n if cs_here else 0)                 # one interleaved workspace, one folding launch
            ws = torch.empty((nblk, width), dtype=torch.float32, device=dev)
            cws_ptr = c_void_p(ws.data_ptr() + 4 * n * n) if cs_here else None
            check(lib().pygho_weight_grad(ptr(ws), cws_ptr, ptr(g), c_void_p(x.data_ptr() + j * n * x.element_size()), k, m, n,
                                          dtype_code(g), width, stream_ptr(dev)), "weight_grad")
            tot = sum_blocks(ws)
            parts.append(tot[:n * n].reshape(n, n))
            if cs_here:
                cs = tot[n * n:n * n + n]
        gw = (parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)).to(out_dtype)
    else:
        slabs = min(256, m // 2048)
        if slabs < 4 or m < (1 << 19):
            gw = (g.t() @ x).to(out_dtype)
        else:
            rows = m // slabs
            main = rows * slabs
            part = torch.bmm(g[:main].view(slabs, rows, n).transpose(1, 2), x[:main].view(slabs, rows, k))
            gw = part.float().sum(0)
            if main < m:
                gw = gw + (g[main:].t() @ x[main:]).float()
            gw = gw.to(out_dtype)
        if want_colsum:
            cs = g.sum(0, dtype=torch.float32)
    return (gw, cs) if want_colsum else gw


USE_FUSED_DW = True      # weight gradient inside the backward kernel (gpre never reaches HBM)
USE_TABLE_PRODUCT = True
USE_GRAD_CHAIN = os.environ.get("PYGHO_GRAD_CHAIN", "1") != "0"   # layers sharing A: A's gradient is summed inside the aggregation epilogues
USE_ADJ_TABLE = True     # adjacency values that are an embedding lookup are read through the table inside the fused block
USE_ACT_ON_LOAD = True   # f32 blocks: BatchNorm + activation applied inside the aggregation kernel's loads
USE_BN_BWD_LINEAR = True
USE_RECOMPUTE_PRE = os.environ.get("PYGHO_RECOMPUTE_PRE", "1") != "0"   # training blocks: the pre-activation is never stored (3 + 2 + 4 streams
                                                                        # per block instead of 4 + 2 + 5; every pass recomputes it from x)
USE_ROWBLOCK_LINEAR = True      # module switch for A/B measurements (the library GEMM + separate passes otherwise)
USE_CONCAT_BLOCK = True   # SSWLConv / DSSGNNConv: Linear-BN-act over concatenated inputs without the concatenation
USE_PAIR_COMBINE = True   # SUNConv on the padded layout: fused node-view / recombination passes
USE_NODE_LEVEL_LINEAR = os.environ.get("PYGHO_NODE_LEVEL_LINEAR", "1") != "0"   # GNNAKConv (sparse): the 3 d -> d map applied before the broadcasts


def rowblock_linear_supported(x: Tensor, out_features: int) -> bool:
    return (USE_ROWBLOCK_LINEAR and x.is_cuda and x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16) and x.shape[1] == out_features
            and out_features in (64, 128) and x.shape[0] >= 8192)


def rowblock_linear(x: Tensor, wl: Tensor, bias: Optional[Tensor] = None, addend: Optional[Tensor] = None,
                    stats_shift=None, store: bool = True):
    """out = x @ wl^T (+ bias) (+ addend) on the skinny-GEMM kernel; with `stats_shift` also returns the per-block partial
    sums of (out - shift), (out - shift)^2 for pygho_bn_finalize: (out, partial_sums or None).  `stats_shift` is an (f32, d)
    tensor, or True: the kernel takes row 0 of its own output as the shift and the result is (out, (partial_sums, shift))."""
    dev = require_device(x, wl, bias, addend, stats_shift if isinstance(stats_shift, Tensor) else None)
    x, wl = x.contiguous(), wl.contiguous()
    m, d = x.shape
    assert wl.shape == (d, d) and wl.dtype == x.dtype
    assert store or stats_shift is not None        # store=False: statistics of the (never stored) product only
    out = torch.empty_like(x) if store else None
    ws = None
    if stats_shift is not None:
        nblk = int(lib().pygho_rowblock_linear_blocks(m))
        ws = torch.empty((nblk, 2, d), dtype=torch.float32, device=dev)
    if addend is not None:
        addend = addend.contiguous()
    if stats_shift is True:
        shift = torch.empty(d, dtype=torch.float32, device=dev)
        check(lib().pygho_rowblock_linear_autoshift(ptr(out), ptr(x), ptr(wl), ptr(bias), ptr(addend), ptr(ws), ptr(shift), m, d,
                                                    dtype_code(x), stream_ptr(dev)), "rowblock_linear")
        return out, (ws, shift)
    check(lib().pygho_rowblock_linear(ptr(out), ptr(x), ptr(wl), ptr(bias), ptr(addend), ptr(ws), ptr(stats_shift), m, d,
                                      dtype_code(x), stream_ptr(dev)), "rowblock_linear")
    return out, ws


def rowblock_linear_bn_act(x: Tensor, wl: Tensor, bias: Optional[Tensor], scale: Tensor, shift: Tensor, act: str,
                           addend: Optional[Tensor] = None) -> Tensor:
    """act((x @ wl^T + bias) * scale + shift) (+ addend) in ONE pass over x: the product is rounded to the storage type exactly as
    `rowblock_linear` stores it, then normalised / activated in the epilogue (= bn_act_fwd on the stored product, bit for bit)."""
    dev = require_device(x, wl, bias, scale, shift, addend)
    x, wl = x.contiguous(), wl.contiguous()
    m, d = x.shape
    assert wl.shape == (d, d) and wl.dtype == x.dtype and scale.dtype == torch.float32 and shift.dtype == torch.float32
    out = torch.empty_like(x)
    if addend is not None:
        addend = addend.contiguous()
    check(lib().pygho_rowblock_linear_bn_act(ptr(out), ptr(x), ptr(wl), ptr(bias), ptr(scale), ptr(shift), ptr(addend), m, d,
                                             ACT_CODE[act], dtype_code(x), stream_ptr(dev)), "rowblock_linear_bn_act")
    return out


def rowblock_linear_bwd_sums(x: Tensor, wl: Tensor, bias: Optional[Tensor], gh: Tensor, saved, act: str):
    """the two channel sums of the BatchNorm + activation backward for pre = x @ wl^T + bias, which is recomputed (same bits as
    the forward's) instead of read: one pass over (x, gh)."""
    mean, invstd, w32, b32, _ws = saved
    dev = require_device(x, wl, bias, gh)
    m, d = x.shape
    s1 = torch.empty(d, dtype=torch.float32, device=dev)
    s2 = torch.empty(d, dtype=torch.float32, device=dev)
    nblk = int(lib().pygho_rowblock_linear_blocks(m))
    ws = torch.empty((nblk, 2, d), dtype=torch.float32, device=dev)
    check(lib().pygho_rowblock_linear_bwd_sums(ptr(s1), ptr(s2), ptr(x), ptr(wl.contiguous()), ptr(bias), ptr(gh.contiguous()), ptr(mean),
                                               ptr(invstd), ptr(w32), ptr(b32), m, d, ACT_CODE[act], ptr(ws), dtype_code(x),
                                               stream_ptr(dev)), "rowblock_linear_bwd_sums")
    return s1, s2


def sum_blocks(partials: Tensor) -> Tensor:
    """(n_blocks, ...) f32 per-workgroup partial results -> their sum over the first dim (one deterministic kernel)."""
    dev = require_device(partials)
    assert partials.dtype == torch.float32 and partials.is_contiguous()
    out = torch.empty(partials.shape[1:], dtype=torch.float32, device=dev)
    check(lib().pygho_sum_blocks(ptr(out), ptr(partials), partials.shape[0], out.numel(), stream_ptr(dev)), "sum_blocks")
    return out


def bn_bwd_sums(pre: Tensor, gh: Tensor, saved, act: str):
    """the two channel sums of the BatchNorm + activation backward (sum dy, sum dy * xhat), two-stage and deterministic."""
    mean, invstd, w32, b32, ws = saved
    m, c = pre.shape
    dev = pre.device
    s1 = torch.empty(c, dtype=torch.float32, device=dev)
    s2 = torch.empty(c, dtype=torch.float32, device=dev)
    check(lib().pygho_bn_act_bwd_sums(ptr(s1), ptr(s2), ptr(pre), ptr(gh), ptr(mean), ptr(invstd), ptr(w32), ptr(b32), m, c,
                                      ACT_CODE[act], ptr(ws), dtype_code(pre), stream_ptr(dev)), "bn_act_bwd_sums")
    return s1, s2


def bn_bwd_linear(pre: Optional[Tensor], gh: Tensor, saved, training: bool, act: str, w: Tensor, addend: Optional[Tensor],
                  want_colsum: bool, x: Optional[Tensor] = None, sums=None, lin_bias: Optional[Tensor] = None):
    """(gx, gpre or dW, d bn.bias, d bn.weight, column sums of gpre or None): BatchNorm/act backward and the
    input-gradient GEMM gx = gpre @ w (+ addend) in one streaming kernel after the two-stage channel reduction.
    With `x` (the Linear's input) the weight gradient gpre^T @ x (f32) is accumulated in the same pass and returned in
    place of gpre, which then never reaches HBM."""
    mean, invstd, w32, b32, ws = saved
    if pre is None:
        # the pre-activation was not kept: both passes recompute it from x (`w` here is the Linear's weight in x's dtype)
        assert x is not None
        m, c = x.shape
        dev = x.device
        if sums is None:
            sums = rowblock_linear_bwd_sums(x, w, lin_bias, gh, saved, act)
        s1, s2 = sums
        gx = torch.empty_like(x)
        addend = None if addend is None else addend.contiguous()
        nblk = int(lib().pygho_bn_bwd_linear_dw_blocks(m))
        width = c * c + (2 * c if want_colsum else 0)
        ws = torch.empty((nblk, width), dtype=torch.float32, device=dev)
        cws_ptr = c_void_p(ws.data_ptr() + 4 * c * c) if want_colsum else None
        check(lib().pygho_bn_bwd_linear_dw_recompute(ptr(gx), ptr(ws), ptr(gh), ptr(x), ptr(w.contiguous()), ptr(lin_bias), ptr(addend), cws_ptr,
                                                     ptr(mean), ptr(invstd), ptr(w32), ptr(b32), ptr(s1), ptr(s2), m, c, ACT_CODE[act],
                                                     1 if training else 0, dtype_code(x), width, stream_ptr(dev)),
              "bn_bwd_linear_dw_recompute")
        tot = sum_blocks(ws)
        return gx, tot[:c * c].reshape(c, c), s1, s2, (tot[c * c:c * c + c] if want_colsum else None)
    m, c = pre.shape
    dev = pre.device
    st = stream_ptr(dev)
    dt = dtype_code(pre)
    s1, s2 = sums if sums is not None else bn_bwd_sums(pre, gh, saved, act)
    gx = torch.empty_like(pre)
    wl = w.t().contiguous()
    if addend is not None:
        addend = addend.contiguous()
    nblk = int(lib().pygho_bn_bwd_linear_dw_blocks(m) if x is not None else lib().pygho_rowblock_linear_blocks(m))
    if x is not None:
        width = c * c + (2 * c if want_colsum else 0)              # dW slabs and column sums interleaved: one folding launch
        ws = torch.empty((nblk, width), dtype=torch.float32, device=dev)
        cws_ptr = c_void_p(ws.data_ptr() + 4 * c * c) if want_colsum else None
        check(lib().pygho_bn_bwd_linear_dw(ptr(gx), ptr(ws), ptr(pre), ptr(gh), ptr(x), ptr(wl), ptr(addend), cws_ptr, ptr(mean),
                                           ptr(invstd), ptr(w32), ptr(b32), ptr(s1), ptr(s2), m, c, ACT_CODE[act],
                                           1 if training else 0, dt, width, st), "bn_bwd_linear_dw")
        tot = sum_blocks(ws)
        return gx, tot[:c * c].reshape(c, c), s1, s2, (tot[c * c:c * c + c] if want_colsum else None)
    cws = torch.empty((nblk, 2, c), dtype=torch.float32, device=dev) if want_colsum else None
    second = torch.empty_like(pre)
    check(lib().pygho_bn_bwd_linear(ptr(gx), ptr(second), ptr(pre), ptr(gh), ptr(wl), ptr(addend), ptr(cws), ptr(mean), ptr(invstd),
                                    ptr(w32), ptr(b32), ptr(s1), ptr(s2), m, c, ACT_CODE[act], 1 if training else 0, dt, st),
          "bn_bwd_linear")
    sdx = sum_blocks(cws)[0] if cws is not None else None
    return gx, second, s1, s2, sdx


class _TupleBlock(torch.autograd.Function):
    """H = act(bn(x W^T + b));  out = H                                  (plan is None)
                                  out = [x +] (+)_{(a,c,d)} H[c] * rhs[d]   (plan given; `residual` adds x)
    One autograd node for the whole block so that (i) the Linear's bias gradient comes out of the BatchNorm
    backward pass, (ii) the residual add runs in the aggregation epilogue and (iii) the residual gradient is
    added in place into the fresh input-gradient GEMM output."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, training, eps, act, rhs, plan, aggr, residual,
                fold_momentum=None, rhs_lookup=None, chain=False, chain_x=False):
        require_device(x, w, rhs)
        x = x.contiguous()
        # master weights (usually f32) are cast to the activation dtype here, outside the autograd graph; their
        # gradients are returned in the master dtype straight from the f32 split-K / column sums
        wc = param_as(w, x.dtype)
        bc = None if b is None else param_as(b, x.dtype)
        skinny = rowblock_linear_supported(x, w.shape[0]) and w.shape[0] == w.shape[1]
        partial = None
        # the pre-activation is kept only when a backward pass will read it: with the weight gradient folded into the backward
        # kernel every pass recomputes it from x (same bits), and without a backward nobody needs it
        needs = ctx.needs_input_grad
        recompute = (USE_RECOMPUTE_PRE and skinny and USE_BN_BWD_LINEAR and USE_FUSED_DW
                     and (needs[1] or not any(needs[i] for i in (0, 2, 3, 4, 10))))
        if recompute:
            pre = None
            if training:
                _none, partial = rowblock_linear(x, wc, bc, stats_shift=True, store=False)
        elif skinny:
            # hand-written streaming GEMM: the BatchNorm statistics of its output ride in the epilogue
            pre, partial = rowblock_linear(x, wc, bc, stats_shift=True if training else None)     # partial = (sums, shift)
        else:
            pre = torch.nn.functional.linear(x, wc, bc)
        # f32 rows carry half the elements per byte: there the BatchNorm + activation can ride on the aggregation's loads
        # (act-on-load: 0.61 vs 0.35 + 0.62 ms) and the activated tensor is never formed; with 16-bit rows the two
        # transcendentals per element make that kernel VALU-bound (0.52 vs 0.50 ms forward, 0.41 vs 0.30 ms backward)
        on_load = (not recompute and USE_ACT_ON_LOAD and plan is not None and rhs is not None and x.dtype == torch.float32 and aggr in ("sum", "mean")
                   and (x.shape[1] * 4) % 16 == 0 and rhs.dtype == x.dtype and rhs.shape[1] == x.shape[1])
        # without a plan `rhs` is a residual row operand: out = H + rhs, added inside the activation pass
        row_res = rhs.contiguous() if (plan is None and rhs is not None) else None
        h, mean, var, saved = _bn_forward(pre, gamma, beta, running_mean, running_var, training, eps, act, fold_momentum, partial,
                                          apply=not on_load, addend=row_res, producer=(x, wc, bc) if recompute else None)
        affine = look = None
        rhs_read, d_idx = rhs, (plan.d_fwd if plan is not None and rhs is not None else None)
        if plan is not None and rhs is not None and rhs_lookup is not None:
            # rhs == table[row_of] for a small table: index the table per message (its rows stay in L1) instead of streaming
            # the (n_edges, d) gathered array (0.339 -> 0.310 ms forward, 0.259 -> 0.244 ms backward at B = 8192).  The
            # gradient still goes to `rhs` per edge, and from there through the lookup's own backward.
            look = (rhs_lookup[0].detach(),) + plan.lookup(rhs_lookup[1])
            rhs_read, d_idx = look[0], look[1]
        if plan is None:
            out = h
        elif on_load:
            affine, h = h, None
            out = seg_gmr(plan.n_out, pre, rhs_read, plan.fwd.seg_ptr, plan.c_fwd, d_idx, aggr, addend=x if residual else None,
                          act=(affine[0], affine[1], act, 1))
        else:
            out = seg_gmr(plan.n_out, h, rhs_read, plan.fwd.seg_ptr, plan.c_fwd, d_idx, aggr, addend=x if residual else None)
        ctx.affine, ctx.look = affine, look
        ctx.save_for_backward(x, wc, pre, h if plan is not None else None, rhs, bc, *saved)
        ctx.meta = (training, act, None if b is None else b.dtype, gamma is not None, beta is not None, plan, aggr, residual, w.dtype,
                    skinny)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)            # no zero tensors for the statistics' (never used) gradients
        ctx.chain = (bool(chain), bool(chain_x))
        extra = ()
        if chain:
            # `rhs` again as an OUTPUT: the next block that shares this operand takes it from here, so the operand's gradient
            # arrives in this block's backward already summed over the later blocks and is extended in the aggregation's epilogue
            # (out = addend + ...), instead of autograd adding one (n_rhs, d) tensor per consumer
            extra += (rhs.view_as(rhs),)
        if chain_x:
            # the same for the block's INPUT: whoever else reads x takes it from here; that gradient arrives below and is added in the
            # epilogue of the input-gradient GEMM (where the residual gradient goes), not by a separate (m, d) accumulation
            extra += (x.view_as(x),)
        return (out, mean, var) + extra

    @staticmethod
    def backward(ctx, g, _gm, _gv, *g_extra):
        g_extra = list(g_extra)
        g_chain = g_extra.pop(0) if ctx.chain[0] else None
        g_x = g_extra.pop(0) if ctx.chain[1] else None
        if g is None:
            n_in = len(ctx.needs_input_grad)
            return (g_x,) + (None,) * 9 + (g_chain,) + (None,) * (n_in - 11)
        x, w, pre, h, rhs, bc, *saved = ctx.saved_tensors
        training, act, b_dtype, has_gamma, has_beta, plan, aggr, residual, w_dtype, skinny = ctx.meta
        g = g.contiguous()
        # what is added to the input gradient in the GEMM epilogue: the residual gradient and / or the gradient of x's other readers
        res_g = g if residual else None
        if g_x is not None:
            res_g = g_x.contiguous() if res_g is None else res_g + g_x
        g_rhs = None
        gh = g
        if plan is not None:
            scale = plan.fwd.inv_count if aggr == "mean" else None
            p, a_g, d_g = plan.by_c()
            rhs_read = rhs
            if ctx.look is not None:
                rhs_read, d_g = ctx.look[0], ctx.look[2]
            gh = seg_gmr(plan.n_lhs, g, rhs_read, p.seg_ptr, a_g, d_g if rhs is not None else None, "sum", scale)
            if rhs is not None and ctx.needs_input_grad[10]:
                p, a_g, c_g = plan.by_d()
                if ctx.affine is not None:
                    g_rhs = seg_gmr(plan.n_rhs, g, pre, p.seg_ptr, a_g, c_g, "sum", scale,
                                    act=(ctx.affine[0], ctx.affine[1], act, 2))
                    if g_chain is not None:
                        g_rhs = g_rhs + g_chain
                else:
                    g_rhs = seg_gmr(plan.n_rhs, g, h, p.seg_ptr, a_g, c_g, "sum", scale,
                                    addend=None if g_chain is None else g_chain.contiguous())
            elif g_chain is not None:
                g_rhs = g_chain
        elif rhs is not None and ctx.needs_input_grad[10]:
            g_rhs = g if g_chain is None else g + g_chain      # residual row operand: receives the output gradient as it is
        want_cs = b_dtype is not None and ctx.needs_input_grad[2]
        gx = gw = gb = None
        if pre is None or (skinny and USE_BN_BWD_LINEAR and USE_FUSED_DW and ctx.needs_input_grad[1]):
            gx, gw32, s1, s2, sdx = bn_bwd_linear(pre, gh.contiguous(), saved, training, act, w, res_g, want_cs, x=x,
                                                  lin_bias=bc)
            gw = gw32.to(w_dtype)
        elif skinny and USE_BN_BWD_LINEAR:
            gx, gpre, s1, s2, sdx = bn_bwd_linear(pre, gh.contiguous(), saved, training, act, w, res_g, want_cs)
        else:
            gpre, s1, s2, sdx = _bn_backward(pre, gh, saved, training, act, want_colsum=want_cs)
        if gx is None and ctx.needs_input_grad[0]:
            if skinny:
                # dX = gpre . W (+ g): the residual gradient is added in the GEMM epilogue
                gx, _ = rowblock_linear(gpre, w.t().contiguous(), None, addend=res_g)
            else:
                # (addmm(g, gpre, w) copies g first and then runs a slower beta = 1 GEMM: product + add is faster)
                gx = gpre @ w
                if res_g is not None:
                    gx = gx.add_(res_g)
        if gw is None and ctx.needs_input_grad[1]:
            gw = weight_grad_splitk(gpre, x, w_dtype)
        if sdx is not None:
            gb = sdx.to(b_dtype)
        if gx is None and g_x is not None:
            gx = g_x
        return (gx, gw, gb, (s2 if has_gamma else None), (s1 if has_beta else None), None, None, None, None, None,
                g_rhs, None, None, None, None, None, None, None)


def tuple_block(x: Tensor, lin: "torch.nn.Linear", bn: "torch.nn.BatchNorm1d", act: str, rhs: Optional[Tensor] = None,
                plan: Optional[MessagePlan] = None, aggr: str = "sum", residual: bool = False,
                rhs_lookup: Optional[Tuple[Tensor, Tensor]] = None, chain: bool = False, chain_x: bool = False):
    """fused Linear -> BatchNorm1d -> act (-> aggregation over `plan` with `rhs` (-> + x)); parameters are read
    from the stock modules (f32 master weights are cast to the activation dtype like autocast would)."""
    training = bn.training or bn.running_mean is None
    if residual:
        assert plan is not None and plan.n_out == x.shape[0] and lin.out_features == x.shape[1]
    if plan is None and rhs is not None:                # residual row operand (see _TupleBlock.forward)
        assert rhs.shape == (x.shape[0], lin.out_features) and rhs.dtype == x.dtype and rhs_lookup is None
    fold = _fold_momentum(bn)
    res = _TupleBlock.apply(x, lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                            bn.eps, act, rhs, plan, aggr, residual, fold, rhs_lookup, chain, chain_x)
    out, mean, var = res[:3]
    _update_running(bn, mean, var, x.shape[0], folded=fold is not None)
    # (out[, rhs again when `chain`][, x again when `chain_x`]): see _TupleBlock.forward
    return (out,) + tuple(res[3:]) if (chain or chain_x) else out


class _ConcatBlock(torch.autograd.Function):
    """act(bn(concat(x_0 .. x_{K-1}) W^T + b)) without the concatenation: W = [W_0 | .. | W_{K-1}] column blocks,
    pre = (..((x_0 W_0^T + b) + x_1 W_1^T) ..) as a chain of streaming GEMMs, each with the previous result in its epilogue and
    the last with the BatchNorm statistics; backward = the channel reduction once, then per input ONE pass producing its
    input gradient and its weight-gradient block.  (SSWLConv, reference Conv.py:98-103: the (nnz, 3 d) concatenation is
    1.4 GB at B = 8192 and was a quarter of the layer.)"""

    @staticmethod
    def forward(ctx, w, b, gamma, beta, running_mean, running_var, training, eps, act, fold_momentum, *xs):
        # fold_momentum may arrive as (momentum, residual): residual = the block's output gets xs[0] added (the layer's residual
        # connection, in the activation pass) and xs[0]'s gradient gets the output gradient added (in its backward GEMM's epilogue)
        residual = False
        if isinstance(fold_momentum, tuple):
            fold_momentum, residual = fold_momentum
        require_device(w, *xs)
        xs = [x.contiguous() for x in xs]
        res_row = None
        if residual == "last":                 # the residual operand is a tensor of its own, passed after the block's inputs
            res_row, xs = xs[-1], xs[:-1]
        d = xs[0].shape[1]
        dt = xs[0].dtype
        wc = param_as(w, dt)
        bc = None if b is None else param_as(b, dt)
        blocks = [wc[:, k * d:(k + 1) * d].contiguous() for k in range(len(xs))]
        pre, partial = None, None
        for k, (x, wk) in enumerate(zip(xs, blocks)):
            last = k == len(xs) - 1
            pre, partial = rowblock_linear(x, wk, bc if k == 0 else None, addend=pre, stats_shift=True if (last and training) else None)
        h, mean, var, saved = _bn_forward(pre, gamma, beta, running_mean, running_var, training, eps, act, fold_momentum,
                                          partial if training else None, apply=True,
                                          addend=res_row if res_row is not None else (xs[0] if residual else None))
        ctx.save_for_backward(pre, *xs, *blocks, *saved)
        ctx.residual = residual
        ctx.meta = (len(xs), training, act, None if b is None else b.dtype, gamma is not None, beta is not None, w.dtype)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)            # no zero tensors for the statistics' (never used) gradients
        return h, mean, var

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        if g is None:
            return (None,) * len(ctx.needs_input_grad)
        k_in, training, act, b_dtype, has_gamma, has_beta, w_dtype = ctx.meta
        pre = ctx.saved_tensors[0]
        xs = ctx.saved_tensors[1:1 + k_in]
        blocks = ctx.saved_tensors[1 + k_in:1 + 2 * k_in]
        saved = ctx.saved_tensors[1 + 2 * k_in:]
        g = g.contiguous()
        sums = bn_bwd_sums(pre, g, saved, act)
        want_cs = b_dtype is not None and ctx.needs_input_grad[1]
        gxs, gws, gb = [], [], None
        for k in range(k_in):
            gx, gw32, _s1, _s2, sdx = bn_bwd_linear(pre, g, saved, training, act, blocks[k], g if (ctx.residual is True and k == 0) else None,
                                                    want_cs and k == 0, x=xs[k], sums=sums)
            gxs.append(gx if ctx.needs_input_grad[10 + k] else None)
            gws.append(gw32)
            if sdx is not None:
                gb = sdx.to(b_dtype)
        gw = torch.cat(gws, dim=1).to(w_dtype) if ctx.needs_input_grad[0] else None
        s1, s2 = sums
        if ctx.residual == "last":             # the separate residual operand receives the output gradient as it is
            gxs.append(g if ctx.needs_input_grad[10 + k_in] else None)
        return (gw, gb, (s2 if has_gamma else None), (s1 if has_beta else None), None, None, None, None, None, None, *gxs)


class _ProxyCtx:
    """stands in for the autograd context when a fused Function runs another Function's forward / backward as one of its steps"""

    def __init__(self, needs_input_grad=()):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


USE_SSWL_BLOCK = True


class _SSWLBlock(torch.autograd.Function):
    """the whole SSWLConv update (reference Conv.py:98-103) as one autograd node: x1 = X A inside the subgraphs, x2 = A X across
    them, h = act(bn([x | x1 | x2] W^T + b)) [+ x].  What one node buys over three: the gradient of X has three contributions
    (the block's first input, the two products) and the gradient of A two; here each aggregation launch takes the running sum in
    its epilogue instead of autograd adding (nnz, d) tensors afterwards (two read-read-write passes per layer)."""

    @staticmethod
    def forward(ctx, x, a, plan1, plan2, aggr, residual, w, b, gamma, beta, running_mean, running_var, training, eps, act, fold_momentum):
        x, a = x.contiguous(), a.contiguous()
        x1 = seg_gmr(plan1.n_out, x, a, plan1.fwd.seg_ptr, plan1.c_fwd, plan1.d_fwd, aggr)
        x2 = seg_gmr(plan2.n_out, a, x, plan2.fwd.seg_ptr, plan2.c_fwd, plan2.d_fwd, aggr)
        sub = _ProxyCtx()
        h, mean, var = _ConcatBlock.forward(sub, w, b, gamma, beta, running_mean, running_var, training, eps, act,
                                            (fold_momentum, residual), x, x1, x2)
        ctx.save_for_backward(a, *sub.saved_tensors)
        ctx.sub = (sub.residual, sub.meta)
        ctx.plans, ctx.aggr = (plan1, plan2), aggr
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)            # no zero tensors for the statistics' (never used) gradients
        return h, mean, var

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        if g is None:
            return (None,) * len(ctx.needs_input_grad)
        a = ctx.saved_tensors[0]
        sub = _ProxyCtx((ctx.needs_input_grad[6], ctx.needs_input_grad[7]) + (False,) * 8 + (True, True, True))
        sub.saved_tensors = ctx.saved_tensors[1:]
        sub.residual, sub.meta = ctx.sub
        res = _ConcatBlock.backward(sub, g, None, None)
        gw, gb, ggamma, gbeta = res[:4]
        g0, g1, g2 = res[10:13]
        x = sub.saved_tensors[1]
        plan1, plan2 = ctx.plans
        sc1 = plan1.fwd.inv_count if ctx.aggr == "mean" else None
        sc2 = plan2.fwd.inv_count if ctx.aggr == "mean" else None
        gx = ga = None
        if ctx.needs_input_grad[0]:
            p, a_g, d_g = plan1.by_c()                      # x is the left operand of X A ...
            gx = seg_gmr(plan1.n_lhs, g1, a, p.seg_ptr, a_g, d_g, "sum", sc1, addend=g0)
            p, a_g, c_g = plan2.by_d()                      # ... and the right operand of A X
            gx = seg_gmr(plan2.n_rhs, g2, a, p.seg_ptr, a_g, c_g, "sum", sc2, addend=gx)
        if ctx.needs_input_grad[1]:
            p, a_g, c_g = plan1.by_d()
            ga = seg_gmr(plan1.n_rhs, g1, x, p.seg_ptr, a_g, c_g, "sum", sc1)
            p, a_g, d_g = plan2.by_c()
            ga = seg_gmr(plan2.n_lhs, g2, x, p.seg_ptr, a_g, d_g, "sum", sc2, addend=ga)
        return (gx, ga, None, None, None, None, gw, gb, ggamma, gbeta) + (None,) * 6


def sswl_block(x: Tensor, a: Tensor, plan1: "MessagePlan", plan2: "MessagePlan", aggr: str, lin: "torch.nn.Linear",
               bn: "torch.nn.BatchNorm1d", act: str, residual: bool) -> Tensor:
    training = bn.training or bn.running_mean is None
    fold = _fold_momentum(bn)
    out, mean, var = _SSWLBlock.apply(x, a, plan1, plan2, aggr, residual, lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean,
                                      bn.running_var, training, bn.eps, act, fold)
    _update_running(bn, mean, var, x.shape[0], folded=fold is not None)
    return out


def concat_block_supported(xs, lin: "torch.nn.Linear") -> bool:
    d = xs[0].shape[1] if xs and xs[0].dim() == 2 else -1
    return (USE_CONCAT_BLOCK and USE_ROWBLOCK_LINEAR and USE_BN_BWD_LINEAR and len(xs) >= 2
            and all(x.dim() == 2 and x.shape == xs[0].shape and x.dtype == xs[0].dtype and x.is_cuda for x in xs)
            and lin.in_features == len(xs) * d and lin.out_features == d and rowblock_linear_supported(xs[0], d)
            and bn_act_supported_shape(xs[0].shape[0], d, xs[0].dtype))


def concat_block(xs, lin: "torch.nn.Linear", bn: "torch.nn.BatchNorm1d", act: str, residual=False) -> Tensor:
    """fused Linear -> BatchNorm1d -> act applied to concat(xs, dim=1), the concatenation never formed; `residual`: True adds
    xs[0] to the result, a tensor (same shape and dtype as the result) adds that tensor — both inside the activation pass."""
    training = bn.training or bn.running_mean is None
    fold = _fold_momentum(bn)
    if isinstance(residual, Tensor):
        assert residual.shape == xs[0].shape and residual.dtype == xs[0].dtype
        mode, extra = (fold, "last"), (residual,)
    else:
        mode, extra = ((fold, True) if residual else fold), ()
    out, mean, var = _ConcatBlock.apply(lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                                        bn.eps, act, mode, *xs, *extra)
    _update_running(bn, mean, var, xs[0].shape[0], folded=fold is not None)
    return out

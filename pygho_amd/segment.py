"""
Segment kernels behind the sparse operators: the raw launches of the fused gather * gather -> segment reduce family (fast /
LDS-window / tiled), the message plan of an `acd` triple array with its transposed groupings, and the autograd Functions of
spspmm, scatter-reduce, row gather, spmm and the three-operand tuple initialisation.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from ._native import AGGR_CODE, DTYPE_CODE, check, dtype_code, lib, ptr, require_device, stream_ptr

from . import plans as _plans
from .plans import *          # noqa: F401,F403
from .plans import _I32, _PENDING_ERRORS, _DEFER_CHECKS, _fetch, _flag


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


SEG_TILE = os.environ.get("PYGHO_SEG_TILE", "auto")            # "0": never, "1": whenever the shape allows, "auto": by plan shape
SEG_TILE_WIN_ROWS = int(os.environ.get("PYGHO_SEG_TILE_WIN_ROWS", "24"))


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
        tiles = torch.empty((n_chunks, chunk, 8), dtype=_I32, device=dev)
        check(lib().pygho_seg_tile_plan(ptr(tile_cnt), ptr(tiles), ptr(seg_ptr), ptr(lhs_idx), n_seg, win_rows, stream_ptr(dev)),
              "seg_tile_plan")
        hit = (tile_cnt, tiles, seg_ptr)            # the pointers are kept alive with the entry that is keyed on their address
        cache[k] = hit
    return hit[0], hit[1]


def _tile_eligible(out_rows: int, lhs: Optional[Tensor], rhs: Optional[Tensor], lhs_idx: Optional[Tensor], rhs_idx: Optional[Tensor],
                   aggr: str, scaled: bool = False, residual: bool = False) -> bool:
    """two-operand sum / mean with both index arrays and rows of 256 / 512 / 1024 bytes (`pygho_seg_gather_mul_reduce_tiled`)."""
    if SEG_TILE == "0" or lhs is None or rhs is None or lhs_idx is None or rhs_idx is None or aggr not in ("sum", "mean"):
        return False
    rb = lhs.shape[1] * lhs.element_size()
    if rb not in (256, 512, 1024) or lhs.dtype not in (torch.float32, torch.bfloat16, torch.float16) or out_rows < 4096:
        return False
    if (scaled and (residual or aggr != "sum")) or max(out_rows, lhs.shape[0], rhs.shape[0]) * rb >= (1 << 32) or lhs_idx.numel() == 0:
        return False
    if SEG_TILE == "1":
        return True
    # auto: plans whose segments re-use the lhs rows of one narrow block several times -- forward and by-tuple backward of the 3-tuple
    # product (3.4 messages per output row, 512-B rows: 0.72-0.75 ms against 0.78-0.80 ms on the window kernel at 2048 I2 graphs).
    # Two messages per row (the 2-tuple plans) or 256-B rows gain nothing over the fast kernel; the by-edge backward plan (few long
    # segments over rows spread across a graph) stays on the window kernel
    return rb >= 512 and 2 * rhs.shape[0] <= out_rows and 2 * lhs_idx.numel() >= 5 * out_rows


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
    elif _tile_eligible(out_rows, lhs, rhs, lhs_idx, rhs_idx, aggr, lhs_rowscale is not None, addend is not None):
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


USE_EXTREMUM_VEC = os.environ.get("PYGHO_EXTREMUM_VEC", "1") != "0"     # 16-byte-per-lane max / min backward (A/B switch)


def _extremum_vec_ok(*tensors) -> bool:
    ts = [t for t in tensors if t is not None]
    ref = ts[0]
    rb = ref.shape[1] * ref.element_size() if ref.dim() == 2 else 0
    return (USE_EXTREMUM_VEC and ref.dtype in (torch.float32, torch.bfloat16, torch.float16) and rb > 0 and rb % 16 == 0 and rb <= 1024
            and all(t.dim() == 2 and t.dtype == ref.dtype and t.shape[1] == ref.shape[1] and t.is_contiguous()
                    and t.data_ptr() % 16 == 0 and t.shape[0] * rb < (1 << 32) for t in ts))


def _timed(name: str, nbytes: int, dev, launch) -> None:
    timer = LaunchTimer.active
    if timer is None:
        launch()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    launch()
    e1.record(torch.cuda.current_stream(dev))
    timer.records.append((name, nbytes, e0, e1))


USE_FORWARD_TIES = os.environ.get("PYGHO_FORWARD_TIES", "1") != "0"    # max / min: tie counts from the forward launch (A/B switch)


def seg_gmr_ties(out_rows: int, lhs: Optional[Tensor], rhs: Optional[Tensor], seg_ptr: Tensor, lhs_idx, rhs_idx, aggr: str):
    """(out, ties) of a max / min reduction on the fast kernel: `ties[s]` = number of messages of segment s attaining the stored
    extremum, + 1 where the extremum is 0 (torch's N_to_distribute, see `_ties`), in the value dtype."""
    ref = lhs if lhs is not None else rhs
    dev = require_device(lhs, rhs, seg_ptr, lhs_idx, rhs_idx)
    d = ref.shape[1]
    out = torch.empty((out_rows, d), dtype=ref.dtype, device=dev)
    ties = torch.empty((out_rows, d), dtype=ref.dtype, device=dev)
    es = ref.element_size()
    m = lhs_idx.numel() if lhs_idx is not None else (rhs_idx.numel() if rhs_idx is not None else ref.shape[0])
    rows = (lhs.shape[0] if lhs is not None else 0) + (rhs.shape[0] if rhs is not None else 0) + 2 * out_rows
    nbytes = es * d * rows + 4 * m * ((lhs_idx is not None) + (rhs_idx is not None)) + 4 * (out_rows + 1)
    mode = "both" if (lhs is not None and rhs is not None) else ("lhs" if lhs is not None else "rhs")
    _timed(f"seg_gmr[{str(ref.dtype).split('.')[-1]},{aggr},{mode},ties]", nbytes, dev, lambda: check(lib().pygho_seg_gather_mul_reduce_ties(
        ptr(out), ptr(ties), ptr(lhs), ptr(rhs), ptr(seg_ptr), ptr(lhs_idx), ptr(rhs_idx), out_rows, d,
        lhs.shape[0] if lhs is not None else 0, rhs.shape[0] if rhs is not None else 0, dtype_code(ref), AGGR_CODE[aggr], stream_ptr(dev)),
        "seg_gather_mul_reduce_ties"))
    return out, ties


def _ties(fwd: Tensor, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, gin: Optional[Tensor] = None):
    """tie bookkeeping of the max / min backward.  With `gin` and 16-byte-vector shapes: ("share", gin / ties rounded to the value
    dtype) from ONE 16-byte-per-lane pass; otherwise the f32 tie counts of the scalar kernel."""
    dev = fwd.device
    if gin is not None and _extremum_vec_ok(fwd, gin, lhs, rhs):
        share = torch.empty_like(fwd)
        es, d = fwd.element_size(), fwd.shape[1]
        m = int(lhs_idx.numel() if lhs_idx is not None else (rhs_idx.numel() if rhs_idx is not None else fwd.shape[0]))
        nbytes = es * d * (3 * fwd.shape[0] + (lhs.shape[0] if lhs is not None else 0) + (rhs.shape[0] if rhs is not None else 0)) \
            + 4 * m * ((lhs_idx is not None) + (rhs_idx is not None)) + 4 * (fwd.shape[0] + 1)
        _timed(f"seg_ext_share[{str(fwd.dtype).split('.')[-1]}]", nbytes, dev, lambda: check(lib().pygho_seg_extremum_share(
            ptr(share), ptr(gin), ptr(fwd), ptr(lhs), ptr(rhs), ptr(seg_ptr), ptr(lhs_idx), ptr(rhs_idx), fwd.shape[0], m, d,
            lhs.shape[0] if lhs is not None else 0, rhs.shape[0] if rhs is not None else 0, dtype_code(fwd), stream_ptr(dev)),
            "seg_extremum_share"))
        return ("share", share)
    ties = torch.empty(fwd.shape, dtype=torch.float32, device=dev)
    check(lib().pygho_seg_extremum_ties(ptr(ties), ptr(fwd), ptr(lhs), ptr(rhs), ptr(seg_ptr), ptr(lhs_idx),
                                        ptr(rhs_idx), fwd.shape[0], fwd.shape[1], dtype_code(fwd), stream_ptr(dev)),
          "seg_extremum_ties")
    return ties


def _extremum_bwd(n_rows, gin, fwd, ties, self_vals, other, seg_ptr, out_idx, other_idx) -> Tensor:
    dev = gin.device
    d = gin.shape[1]
    gout = torch.empty((n_rows, d), dtype=gin.dtype, device=dev)
    if isinstance(ties, tuple):
        share = ties[1]
        if _extremum_vec_ok(gout, share, fwd, self_vals, other):
            es = gin.element_size()
            nbytes = es * d * (2 * fwd.shape[0] + n_rows * (2 if self_vals is not None else 1) + (other.shape[0] if other is not None else 0)) \
                + 4 * out_idx.numel() * (2 if other_idx is not None else 1) + 4 * (n_rows + 1)
            _timed(f"seg_ext_bwd[{str(gin.dtype).split('.')[-1]}]", nbytes, dev, lambda: check(lib().pygho_seg_extremum_bwd_shared(
                ptr(gout), ptr(share), ptr(fwd), ptr(self_vals), ptr(other), ptr(seg_ptr), ptr(out_idx), ptr(other_idx), n_rows, out_idx.numel(), d,
                fwd.shape[0], other.shape[0] if other is not None else 0, dtype_code(gin), stream_ptr(dev)), "seg_extremum_bwd_shared"))
            return gout
        # an operand of this plan falls outside the vector kernel's domain: the scalar kernel with unit tie counts on the shares
        ones = torch.ones(fwd.shape, dtype=torch.float32, device=dev)
        gin, ties = share, ones
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
        self._a64, self._c64, self._d64 = unbased(acd[0]), unbased(acd[1]), unbased(acd[2])   # (cached on acd: no view links)
        # operand indices must address rows of the operands (the reference's gathers raise IndexError, Spspmm.py:309-311): checked
        # by the narrowing pass itself (round 4: a separate aminmax over the int64 rows, 1.0-1.4 ms at BASELINE shapes); the flags
        # ride on the synchronisation of the forward plan's sortedness probe below
        msg = f"pygho_amd: acd operand index out of range (acd[1] must lie in [0, {n_lhs}), acd[2] in [0, {n_rhs}))"
        c32, d32 = narrow_i32(acd[1], bound=n_lhs, bound_msg=msg), narrow_i32(acd[2], bound=n_rhs, bound_msg=msg)
        self.fwd = plan_from_keys(acd[0], n_out)
        a32 = narrow_i32(acd[0])
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
        self._a64, self._c64, self._d64 = unbased(acd[0]), unbased(acd[1]), unbased(acd[2])   # (cached on acd: no view links)
        self.fwd = SegPlan(fwd_ptr, None, n_out, self.m)
        self.a32, self.c32, self.d32 = narrow_i32(acd[0]), narrow_i32(acd[1]), narrow_i32(acd[2])
        self.c_fwd, self.d_fwd = self.c32, self.d32
        pc, pd = SegPlan(ptr_c, perm_c, n_lhs, self.m), SegPlan(ptr_d, perm_d, n_rhs, self.m)
        self._by_c = (pc, pc.take(self.a32), pc.take(self.d32))
        self._by_d = (pd, pd.take(self.a32), pd.take(self.c32))
        self._lookup = None
        return self

    @classmethod
    def from_arrays(cls, acd: Tensor, n_out: int, n_lhs: int, n_rhs: int, acd32: Tensor, fwd_ptr: Tensor, ptr_c: Tensor, perm_c: Tensor,
                    by_c: Tensor, ptr_d: Tensor, perm_d: Tensor, by_d: Tensor, volatile: bool = True) -> "MessagePlan":
        """the plan over arrays that are ALL given (`collate.DeviceGraphStore.collate`, `slots.BatchSlot`): `acd32` (3, M) the narrowed
        triples, `by_c` (2, M) = (a, d) in by-c order, `by_d` (2, M) = (a, c) in by-d order.  Nothing is computed here.  `volatile`:
        the arrays are rewritten in place per batch (a slot) -- nothing derived from them is cached, and the by-edge gradient runs the
        gather form (its chunk list has no fixed length)."""
        self = cls.__new__(cls)
        self.m = acd.shape[1]
        self.n_out, self.n_lhs, self.n_rhs = n_out, n_lhs, n_rhs
        self._a64, self._c64, self._d64 = unbased(acd[0]), unbased(acd[1]), unbased(acd[2])
        self.fwd = SegPlan(fwd_ptr, None, n_out, self.m)
        self.a32, self.c32, self.d32 = acd32[0], acd32[1], acd32[2]
        self.c_fwd, self.d_fwd = self.c32, self.d32
        pc, pd = SegPlan(ptr_c, perm_c, n_lhs, self.m), SegPlan(ptr_d, perm_d, n_rhs, self.m)
        for p in (self.fwd, pc, pd):
            p.volatile = volatile
        self._by_c = (pc, by_c[0], by_c[1])
        self._by_d = (pd, by_d[0], by_d[1])
        self._lookup = None
        if volatile:
            self._scatter = False
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


# --------------------------------------------------------------------------
# the by-edge gradient as a scatter over the forward message order, every operand row fetched once (csrc/seg_scatter.hip)
# --------------------------------------------------------------------------
SEG_SCATTER = os.environ.get("PYGHO_SEG_SCATTER", "auto")      # "0": never, "1" / "auto": whenever a plan's blocks allow
# below ~10^6 messages the launch does not fill the chip's resident set of (blocks x slices) workgroups and the gather form on the
# window kernel is as fast (128 / 1024-graph ZINC-shape batches: 56 k / 440 k messages)
SEG_SCATTER_MIN_MESSAGES = int(os.environ.get("PYGHO_SEG_SCATTER_MIN_MESSAGES", str(1 << 20)))


class ScatterPlan:
    """blocks of consecutive messages whose second-operand rows (edges d) form pairwise disjoint contiguous ranges -- the graphs of a
    block-diagonal batch -- cut into chunks of at most 64 messages over two windows of at most 32 rows, with one packed word per
    message (`pygho_seg_scatter_count` / `_write`).  Integer work on the device; two host reads per plan (block count; chunk count +
    eligibility), cached with the MessagePlan."""
    __slots__ = ("n_blocks", "n_chunks", "chunk0", "blk_e", "chunks", "words", "max_edges", "covers", "cgap", "covers_c", "n_dyn")


DUAL_BWD = os.environ.get("PYGHO_DUAL_BWD", "1") != "0"         # A/B switch: both gradients of a layer's aggregation in one pass (csrc/seg_dual.hip)


def scatter_plan_parts_aligned(a32: Tensor, c32: Tensor, d32: Tensor, block_m: Tensor, row_cut: Tensor, n_rhs: Optional[int] = None):
    """the ALIGNED planner (csrc/seg_scatter.hip, for csrc/seg_dual.hip): chunks that hold every message of the c rows they touch, over
    blocks whose c rows lie in [row_cut[b], row_cut[b + 1]).  Returns what `scatter_plan_parts` returns + (cgap (total), the number of
    blocks that have rows but no message), or None when a group of messages is outside the chunk limits -- ONE host read"""
    dev = d32.device
    nb = block_m.numel() - 1
    n_chunks = torch.empty(nb, dtype=_I32, device=dev)
    blk_e = torch.empty((nb, 2), dtype=_I32, device=dev)
    flags = torch.zeros(3, dtype=_I32, device=dev)
    sufmin = torch.empty(max(d32.numel(), 1), dtype=_I32, device=dev)
    row_cut = row_cut.contiguous()
    check(lib().pygho_seg_scatter_count_aligned(ptr(n_chunks), ptr(blk_e), ptr(flags), ptr(sufmin), ptr(a32), ptr(c32), ptr(d32), ptr(block_m),
                                                ptr(row_cut), nb, stream_ptr(dev)), "seg_scatter_count_aligned")
    chunk0 = torch.zeros(nb + 1, dtype=_I32, device=dev)
    torch.cumsum(n_chunks, 0, out=chunk0[1:])
    covers = torch.ones((), dtype=torch.bool, device=dev)
    if n_rhs is not None and nb > 0:
        e0, ne = blk_e[:, 0], blk_e[:, 1]
        covers = (e0[0] == 0) & (e0[-1] + ne[-1] == n_rhs) & (e0[1:] == e0[:-1] + ne[:-1]).all()
    max_edges, bad, uncovered, total, covers = _fetch(torch.stack([flags[0], flags[1], flags[2], chunk0[-1], covers.to(_I32)]))
    if bad or max_edges > 255 or total == 0:
        return None
    chunks = torch.empty((int(total), 4), dtype=_I32, device=dev)
    cgap = torch.empty(int(total), dtype=_I32, device=dev)
    words = torch.empty(d32.numel(), dtype=_I32, device=dev)
    check(lib().pygho_seg_scatter_write_aligned(ptr(chunks), ptr(words), ptr(cgap), ptr(chunk0), ptr(blk_e), ptr(sufmin), ptr(a32), ptr(c32),
                                                ptr(d32), ptr(block_m), ptr(row_cut), nb, int(total), d32.numel(), stream_ptr(dev)),
          "seg_scatter_write_aligned")
    return n_chunks, chunk0, blk_e, chunks, words, int(max_edges), bool(covers), cgap, int(uncovered)


def scatter_plan_parts(a32: Tensor, c32: Tensor, d32: Tensor, block_m: Tensor, n_rhs: Optional[int] = None):
    """the planner over given blocks: (n_chunks per block, chunk0 = their exclusive scan (int32, n_blocks + 1), blk_e (n_blocks, 2),
    chunks (total, 4), words (M), max_edges, covers) -- two kernels and ONE host read (chunk total + flags + whether the blocks' edge
    ranges tile [0, n_rhs)); `n_rhs` None: covers is not asked for"""
    dev = d32.device
    nb = block_m.numel() - 1
    n_chunks = torch.empty(nb, dtype=_I32, device=dev)
    blk_e = torch.empty((nb, 2), dtype=_I32, device=dev)
    flags = torch.zeros(2, dtype=_I32, device=dev)
    check(lib().pygho_seg_scatter_count(ptr(n_chunks), ptr(blk_e), ptr(flags), ptr(a32), ptr(c32), ptr(d32), ptr(block_m), nb,
                                        stream_ptr(dev)), "seg_scatter_count")
    chunk0 = torch.zeros(nb + 1, dtype=_I32, device=dev)
    torch.cumsum(n_chunks, 0, out=chunk0[1:])
    covers = torch.ones((), dtype=torch.bool, device=dev)
    if n_rhs is not None and nb > 0:
        e0, ne = blk_e[:, 0], blk_e[:, 1]
        covers = (e0[0] == 0) & (e0[-1] + ne[-1] == n_rhs) & (e0[1:] == e0[:-1] + ne[:-1]).all()
    max_edges, bad, total, covers = _fetch(torch.stack([flags[0], flags[1], chunk0[-1], covers.to(_I32)]))
    if bad or max_edges > 255 or total == 0:
        return None
    chunks = torch.empty((int(total), 4), dtype=_I32, device=dev)
    words = torch.empty(d32.numel(), dtype=_I32, device=dev)
    check(lib().pygho_seg_scatter_write(ptr(chunks), ptr(words), ptr(chunk0), ptr(blk_e), ptr(a32), ptr(c32), ptr(d32), ptr(block_m), nb,
                                        int(total), d32.numel(), stream_ptr(dev)), "seg_scatter_write")
    return n_chunks, chunk0, blk_e, chunks, words, int(max_edges), bool(covers)


def block_cuts(d32: Tensor) -> Tensor:
    """(n_blocks + 1) int32 block starts of a message list over second-operand rows `d32` (+ the closing M): a block starts where
    every earlier d is smaller than every later one (`pygho_block_cuts`: two device scans + one selection; ONE host read, the count)"""
    dev = require_device(d32)
    m = d32.numel()
    block_m = torch.empty(m + 1, dtype=_I32, device=dev)
    nb = torch.empty(1, dtype=_I32, device=dev)
    nbytes = int(lib().pygho_block_cuts_workspace(m))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().pygho_block_cuts(ptr(block_m), ptr(nb), ptr(d32.contiguous()), m, ptr(ws), nbytes, stream_ptr(dev)), "block_cuts")
    n_blocks = int(_fetch(nb)[0])
    return block_m[:n_blocks + 1]


def _block_row_cuts(plan: "MessagePlan", block_m: Tensor) -> Tensor:
    """c-row ranges of the blocks for the aligned planner: block b owns [min c of block b, min c of block b + 1) (the first block from
    row 0, the last to n_lhs).  Whether the blocks' c ranges really are disjoint and ascending is the planner's check."""
    dev = block_m.device
    nb = block_m.numel() - 1
    lens = (block_m[1:] - block_m[:-1]).long()
    blk = torch.repeat_interleave(torch.arange(nb, device=dev), lens)
    cmin = torch.full((nb,), plan.n_lhs, dtype=_I32, device=dev).scatter_reduce_(0, blk, plan.c32, "amin", include_self=True)
    cut = torch.empty(nb + 1, dtype=_I32, device=dev)
    cut[:nb] = cmin
    cut[0] = 0
    cut[nb] = plan.n_lhs
    return cut


def _scatter_plan_build(plan: "MessagePlan") -> Optional[ScatterPlan]:
    block_m = block_cuts(plan.d32)                                                        # (host read 1: the number of blocks)
    parts = None
    if DUAL_BWD and plan.n_lhs == plan.n_out and block_m.numel() > 1:
        # chunks that also serve the fused backward (every message of a chunk's c rows inside the chunk); the plain chunks otherwise
        parts = scatter_plan_parts_aligned(plan.a32, plan.c32, plan.d32, block_m, _block_row_cuts(plan, block_m), plan.n_rhs)
    if parts is None:
        parts = scatter_plan_parts(plan.a32, plan.c32, plan.d32, block_m, plan.n_rhs)     # (host read 2: chunk total + verdicts)
    if parts is None:
        return None
    _, chunk0, blk_e, chunks, words, max_edges, covers = parts[:7]
    sp = ScatterPlan()
    sp.n_blocks, sp.n_chunks, sp.chunk0, sp.blk_e, sp.max_edges = block_m.numel() - 1, chunks.shape[0], chunk0, blk_e, max_edges
    sp.chunks, sp.words, sp.covers = chunks, words, covers
    sp.cgap, sp.covers_c = (parts[7], parts[8] == 0) if len(parts) > 7 else (None, False)
    sp.n_dyn = None
    return sp


def install_scatter_plan(plan: "MessagePlan", chunk0: Optional[Tensor], blk_e: Optional[Tensor], chunks: Tensor, words: Tensor, max_edges: int,
                         covers: bool, cgap: Optional[Tensor] = None, covers_c: bool = False, n_dyn: Optional[Tensor] = None) -> None:
    """a ScatterPlan that already exists (`collate.DeviceGraphStore`: the chunks of a block-diagonal batch are its graphs' precomputed
    chunks with the message / row offsets added): no planner launch, no host read.  The caller guarantees the planner's contract.
    `cgap`: the chunks are ALIGNED (they also serve the fused backward); `covers_c`: every first-operand row belongs to a chunk.
    Without `chunk0` / `blk_e` (the per-block arrays) the plan serves the fused backward's table-gradient form only, which walks the
    chunk list alone (`slots.BatchSlot`: a list of fixed capacity, all-zero records behind the batch's chunks, the true count in
    `n_dyn` on the device)."""
    sp = ScatterPlan()
    sp.n_blocks, sp.n_chunks, sp.chunk0, sp.blk_e, sp.max_edges, sp.covers = (chunk0.numel() - 1 if chunk0 is not None else 0), chunks.shape[0], chunk0, blk_e, max_edges, covers
    sp.chunks, sp.words, sp.cgap, sp.covers_c, sp.n_dyn = chunks, words, cgap, covers_c, n_dyn
    plan._scatter = sp if sp.n_chunks > 0 else False


def scatter_plan(plan: "MessagePlan", on_demand: bool = False) -> Optional[ScatterPlan]:
    """the plan's ScatterPlan, or None when its blocks are outside the kernel's limits (more than 255 edges in a block, a not sorted
    inside a block, more than four messages of one edge among 16 consecutive ones).  Never built under stream capture (host reads).
    Built on an EXPLICIT call only (`SpModel.prepare`, the prefetcher's side stream) or installed with the batch
    (`collate.DeviceGraphStore`): planning costs ~0.5 ms and two host reads while one launch saves ~15 us, so the dispatcher's call
    (`on_demand`) never builds -- it takes the plan when there is one and the gather form otherwise.  (Round 4 built on demand after
    12 by-edge launches on a pattern: hidden state that made WHICH kernel ran depend on a batch's history; the two forms return the
    same bits, but a step's time and its host reads should not depend on how often a batch was seen.)"""
    sp = getattr(plan, "_scatter", None)
    if sp is not None:
        return sp or None
    if on_demand or plan.m == 0 or plan.m >= (1 << 31) or torch.cuda.is_current_stream_capturing():
        return None
    plan._scatter = _scatter_plan_build(plan) or False
    return plan._scatter or None


# --------------------------------------------------------------------------
# the layer MLP folded into the forward aggregation's load path (csrc/seg_fused.hip)
# --------------------------------------------------------------------------
FUSED_FWD = os.environ.get("PYGHO_FUSED_FWD", "1") != "0"       # A/B switch
FUSED_BLOCK_ROWS = 512          # planner blocks of a plan that brings no graph cuts: chunks never cross them


class FusedPlan:
    """chunks of consecutive output rows (at most 32 rows / 64 messages / a 32-row first-operand window) of a plan's forward order, and
    per chunk the window rows it is the first to cover (`pygho_seg_fused_count` / `_write`); one host read, cached with the MessagePlan"""
    __slots__ = ("n_chunks", "chunks", "own")


def fused_plan_parts(seg_ptr: Tensor, c_fwd: Tensor, row_cut: Tensor, n_rows: int):
    """(n_chunks per block, chunk0, chunks (total, 4), own (total)) over the row blocks `row_cut`, or None when a row is outside the
    kernel's limits -- three kernels and ONE host read (chunk total + verdict)"""
    dev = require_device(seg_ptr, c_fwd, row_cut)
    nb = row_cut.numel() - 1
    n_chunks = torch.empty(nb, dtype=_I32, device=dev)
    flags = torch.zeros(1, dtype=_I32, device=dev)
    check(lib().pygho_seg_fused_count(ptr(n_chunks), ptr(flags), ptr(seg_ptr), ptr(c_fwd), ptr(row_cut), nb, stream_ptr(dev)),
          "seg_fused_count")
    chunk0 = torch.zeros(nb + 1, dtype=_I32, device=dev)
    torch.cumsum(n_chunks, 0, out=chunk0[1:])
    bad, total = _fetch(torch.stack([flags[0], chunk0[-1]]))
    if bad or total == 0:
        return None
    chunks = torch.empty((int(total), 4), dtype=_I32, device=dev)
    own = torch.empty(int(total), dtype=_I32, device=dev)
    owner_ws = torch.empty(n_rows, dtype=_I32, device=dev)
    check(lib().pygho_seg_fused_write(ptr(chunks), ptr(own), ptr(owner_ws), ptr(chunk0), ptr(seg_ptr), ptr(c_fwd), ptr(row_cut), nb,
                                      int(total), n_rows, stream_ptr(dev)), "seg_fused_write")
    return n_chunks, chunk0, chunks, own


def fused_plan(plan: "MessagePlan", on_demand: bool = False, row_cut: Optional[Tensor] = None) -> Optional[FusedPlan]:
    """the plan's FusedPlan, or None (never asked for / a row outside the limits: more than 64 messages or first-operand rows more than
    31 apart).  Built on an explicit call only, like `scatter_plan`."""
    fp = getattr(plan, "_fused", None)
    if fp is not None:
        return fp or None
    if on_demand or plan.m == 0 or plan.m >= (1 << 31) or plan.n_lhs != plan.n_out or torch.cuda.is_current_stream_capturing():
        return None
    if row_cut is None:
        dev = plan.c_fwd.device
        row_cut = torch.arange(0, plan.n_out + FUSED_BLOCK_ROWS, FUSED_BLOCK_ROWS, dtype=_I32, device=dev).clamp_(max=plan.n_out)
    parts = fused_plan_parts(plan.fwd.seg_ptr, plan.c_fwd, row_cut, plan.n_lhs)
    if parts is None:
        plan._fused = False
        return None
    fp = FusedPlan()
    fp.chunks, fp.own = parts[2], parts[3]
    fp.n_chunks = fp.chunks.shape[0]
    plan._fused = fp
    return fp


def install_fused_plan(plan: "MessagePlan", chunks: Tensor, own: Tensor) -> None:
    """a FusedPlan that already exists (`collate.DeviceGraphStore`: a batch's chunks are its graphs' precomputed chunks with the message
    / row offsets added): no planner launch, no host read.  Trailing all-zero records (a fixed-capacity slot's padding) end a
    workgroup's share of the list."""
    fp = FusedPlan()
    fp.chunks, fp.own, fp.n_chunks = chunks, own, chunks.shape[0]
    plan._fused = fp if fp.n_chunks > 0 else False


def fused_forward(x: Tensor, wl: Tensor, bias: Optional[Tensor], scale: Tensor, shift: Tensor, act: str, table: Tensor, look_fwd: Tensor,
                  plan: "MessagePlan", fp: FusedPlan, aggr: str, residual: bool, want_h: bool):
    """(out, H or None): out[a] = [x[a] +] aggr_{(a,c,d)} H[c] * table[look[m]] with H = act((x wl^T + bias) * scale + shift) formed per
    chunk on the matrix cores and never read from memory; `want_h`: the H rows that messages read are also stored (the by-edge gradient
    reads them; rows no message reads stay uninitialised)"""
    dev = require_device(x, wl, bias, scale, shift, table, look_fwd)
    n, d = x.shape
    assert plan.n_out == n and plan.n_lhs == n and wl.shape == (d, d) and wl.dtype == x.dtype and table.dtype == x.dtype
    out = torch.empty_like(x)
    h = torch.empty_like(x) if want_h else None
    # has-to-move bytes: x once, out once, the stored H once, two int32 indices per message, the CSR pointers, the chunk records
    nbytes = x.element_size() * d * n * (3 if want_h else 2) + 8 * plan.m + 4 * (n + 1) + 20 * fp.n_chunks
    name = f"seg_fused[{str(x.dtype).split('.')[-1]},{aggr}{',res' if residual else ''}{',h' if want_h else ''}]"
    _timed(name, nbytes, dev, lambda: _fused_launch(out, h, x, wl, bias, scale, shift, act, table, look_fwd, plan, fp, aggr, residual))
    return out, h


def _fused_launch(out, h, x, wl, bias, scale, shift, act, table, look_fwd, plan, fp, aggr, residual):
    dev, (n, d) = x.device, x.shape
    check(lib().pygho_seg_fused_fwd(ptr(out), ptr(h), ptr(x), ptr(wl.contiguous()), ptr(bias), ptr(scale), ptr(shift), ptr(table.contiguous()),
                                    table.shape[0], 1 if residual else 0, ptr(plan.fwd.seg_ptr), ptr(plan.c_fwd), ptr(look_fwd), ptr(fp.chunks),
                                    ptr(fp.own), fp.n_chunks, n, plan.m, d, ACT_CODE[act], 1 if aggr == "mean" else 0, dtype_code(x),
                                    stream_ptr(dev)), "seg_fused_fwd")


def _scatter_eligible(plan: "MessagePlan", g: Tensor, h: Optional[Tensor], scale, addend, tg: bool = False) -> bool:
    """`tg`: asked for the fused backward's table-gradient form (its own size threshold; a chunk list without per-block arrays will do)"""
    if SEG_SCATTER == "0" or h is None or scale is not None or g.dtype not in (torch.bfloat16, torch.float16) or h.dtype != g.dtype:
        return False
    rb = g.shape[1] * g.element_size() if g.dim() == 2 else 0
    if (rb == 0 or rb % 64 != 0 or rb > 512 or h.dim() != 2 or h.shape[1] != g.shape[1]
            or plan.m < (min(DUAL_TG_MIN_MESSAGES, SEG_SCATTER_MIN_MESSAGES) if tg else SEG_SCATTER_MIN_MESSAGES)):
        return False
    if max(g.shape[0], h.shape[0], plan.n_rhs) * rb >= (1 << 31) or g.shape[0] != plan.n_out or h.shape[0] != plan.n_lhs:
        return False
    sp = scatter_plan(plan, on_demand=True)
    if sp is None or (sp.chunk0 is None and not tg):
        return False
    if tg:
        return True
    per_wave = (sp.max_edges + 7) // 8 * 8 * 144 + 2 * 32 * 80 + 256
    return (rb // 64) * per_wave <= 160 * 1024


def by_edge_product(plan: "MessagePlan", g: Tensor, h: Optional[Tensor], scale: Optional[Tensor] = None,
                    addend: Optional[Tensor] = None) -> Tensor:
    """out[d] = [addend[d] +] sum over the plan's messages (a, c, d) of [scale[a] *] g[a] * h[c]   (n_rhs rows): the gradient of the
    SECOND operand's values of `out[a] = (+) A[c] B[d]` with g the output gradient and h the first operand (autograd of
    pygho/backend/Spspmm.py:309-315).  16-bit rows over a block-diagonal plan go through the scatter form (every row fetched once);
    everything else through the gather form over the messages grouped by d -- same bits either way."""
    if not _scatter_eligible(plan, g, h, scale, addend):
        p, a_g, c_g = plan.by_d()
        return seg_gmr(plan.n_rhs, g, h, p.seg_ptr, a_g, c_g if h is not None else None, "sum", scale,
                       addend=None if addend is None else addend.contiguous())
    sp = scatter_plan(plan)
    dev = require_device(g, h, addend)
    g, h = g.contiguous(), h.contiguous()
    d = g.shape[1]
    if addend is not None:
        addend = addend.contiguous()
        assert addend.shape == (plan.n_rhs, d) and addend.dtype == g.dtype
    if sp.covers:
        out = torch.empty((plan.n_rhs, d), dtype=g.dtype, device=dev)
    else:                                                   # rows outside every block: no message reaches them
        out = torch.zeros((plan.n_rhs, d), dtype=g.dtype, device=dev) if addend is None else addend.clone()
    timer = LaunchTimer.active
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
    check(lib().pygho_seg_scatter_mul_reduce(ptr(out), ptr(addend), ptr(g), ptr(h), ptr(sp.chunks), ptr(sp.words), ptr(sp.chunk0),
                                             ptr(sp.blk_e), sp.n_blocks, sp.n_chunks, plan.m, sp.max_edges, plan.n_rhs, d, g.shape[0], h.shape[0],
                                             dtype_code(g), stream_ptr(dev)), "seg_scatter_mul_reduce")
    if timer is not None:
        e1.record(torch.cuda.current_stream(dev))
        es = g.element_size()
        # the same algorithmic bytes as the gather form (every operand row once, every output row once, two int32 indices per message,
        # the CSR pointers): the figures of the two forms compare like for like
        nbytes = es * d * (g.shape[0] + h.shape[0] + plan.n_rhs * (2 if addend is not None else 1)) + 8 * plan.m + 4 * (plan.n_rhs + 1)
        timer.records.append((f"seg_gmr[{str(g.dtype).split('.')[-1]},sum,both{',res' if addend is not None else ''},scatter]",
                              nbytes, e0, e1))
    return out


def dual_eligible(plan: "MessagePlan", g: Tensor, h: Optional[Tensor], table: Optional[Tensor], scale: Optional[Tensor], tg: bool = False) -> bool:
    """both gradients of the aggregation in one pass (csrc/seg_dual.hip): an aligned scatter plan, 16-bit rows, a lookup table of at most
    32 rows as the second operand, sum.  `tg`: the table-gradient form (no edge accumulators: no limit on the edges per block)"""
    if not DUAL_BWD or table is None or h is None or scale is not None or not _scatter_eligible(plan, g, h, scale, None, tg):
        return False
    sp = scatter_plan(plan, on_demand=True)
    if sp is None or sp.cgap is None or table.dim() != 2 or table.shape[0] > 32 or table.dtype != g.dtype:
        return False
    if plan.n_lhs != h.shape[0] or table.shape[1] != g.shape[1] or h.shape[0] * g.shape[1] * g.element_size() >= (1 << 31):
        return False
    if tg:
        return True
    if sp.max_edges > 96:
        return False
    rb = g.shape[1] * g.element_size()
    per_wave = (sp.max_edges + 7) // 8 * 8 * 144 + 65 * 80 + 512 + 160 + (table.shape[0] + 1) * 80
    return (rb // 64) * per_wave <= 160 * 1024


DUAL_TABLE_GRAD = os.environ.get("PYGHO_DUAL_TABLE_GRAD", "1") != "0"      # A/B switch: the table gradient straight from the fused backward
# the table-gradient form replaces by-tuple launch + by-edge launch + table reduction + the chain adds by ONE launch and a fold: it
# pays from far fewer messages than the scatter form does against the gather form (captured 1024-graph step, 440 k messages: 2.43 ->
# 2.20 ms; 128 graphs, 55 k messages: 1.165 -> 1.16 ms, i.e. nothing)
DUAL_TG_MIN_MESSAGES = int(os.environ.get("PYGHO_DUAL_TG_MIN_MESSAGES", str(1 << 17)))


def dual_tg_eligible(plan: "MessagePlan", g: Tensor, h: Optional[Tensor], table: Optional[Tensor], scale: Optional[Tensor],
                     index: Optional[Tensor]) -> bool:
    """the fused backward in its table-gradient form (csrc/seg_dual.hip, TG): what `dual_eligible` asks for, rows of at most 256 bytes,
    and an index tensor whose values are KNOWN to lie below the kernel's table-gradient rows (4: bond types) -- the bound the device
    graph store attaches to the arrays it range-checked (`_pygho_value_bound`)"""
    if not DUAL_TABLE_GRAD or index is None or not dual_eligible(plan, g, h, table, scale, tg=True):
        return False
    bound = getattr(index, "_pygho_value_bound", None)
    if bound is None or bound[0] != index._version:
        return False
    rows = ctypes.c_int(0)
    check(lib().pygho_seg_dual_limits(None, None, ctypes.byref(rows)), "seg_dual_limits")
    return int(bound[1]) <= rows.value and g.shape[1] * g.element_size() <= 256


def dual_backward_tg(plan: "MessagePlan", g: Tensor, h: Tensor, table: Tensor, look_fwd: Tensor, look_byc: Tensor):
    """(gh, g_table) of out[a] = sum_{(a,c,d)} h[c] * table[look[d]]: gh as `dual_backward`; g_table (table rows, d) f32 = the gradient of
    the TABLE, sum over the messages of g[a] * h[c] by looked-up row -- accumulated inside the kernel (exact products, f32 sums, one
    slab per workgroup folded here), instead of the per-edge gradient + its chain of adds + `table_grad`"""
    sp = scatter_plan(plan, on_demand=True)
    dev = require_device(g, h, table, look_fwd, look_byc)
    g, h, table = g.contiguous(), h.contiguous(), table.contiguous()
    d = g.shape[1]
    pc, a_byc, _ = plan.by_c()
    gh = (torch.empty if sp.covers_c else torch.zeros)((plan.n_lhs, d), dtype=g.dtype, device=dev)
    rows = ctypes.c_int(0)
    check(lib().pygho_seg_dual_limits(None, None, ctypes.byref(rows)), "seg_dual_limits")
    nblk = int(lib().pygho_seg_dual_tg_blocks(sp.n_chunks, d, table.shape[0], dtype_code(g)))
    assert nblk > 0
    slabs = torch.empty((nblk, rows.value * d), dtype=torch.float32, device=dev)      # (every workgroup has a chunk: nblk <= n_chunks)
    es = g.element_size()
    nbytes = es * d * (g.shape[0] + h.shape[0] + plan.n_lhs) + 16 * plan.m + 4 * (plan.n_lhs + 1) + 20 * sp.n_chunks
    name = f"seg_dual[{str(g.dtype).split('.')[-1]},sum,table]"
    _timed(name, nbytes, dev, lambda: check(lib().pygho_seg_dual_tg(
        ptr(gh), ptr(slabs), ptr(g), ptr(h), ptr(table), table.shape[0], ptr(sp.chunks), ptr(sp.words), ptr(sp.cgap), ptr(pc.seg_ptr),
        ptr(a_byc), ptr(look_byc), ptr(look_fwd), sp.n_chunks, plan.m, d, g.shape[0], h.shape[0], dtype_code(g), ptr(sp.n_dyn),
        stream_ptr(dev)), "seg_dual_tg"))
    from .blocks import sum_blocks
    t = table.shape[0]
    part = sum_blocks(slabs, max(t, rows.value) * d).reshape(-1, d)          # fold + the zero rows behind the kernel's rows: one launch
    return gh, part[:t]


def dual_backward(plan: "MessagePlan", g: Tensor, h: Tensor, table: Tensor, look_byc: Tensor, addend: Optional[Tensor] = None):
    """(gh, g_rhs) of out[a] = sum_{(a,c,d)} h[c] * table[look[d]]:  gh[c] = sum g[a] * table[look]  (n_lhs rows; the bits of
    `seg_gmr` over the by-c plan) and g_rhs[d] = [addend[d] +] sum g[a] * h[c]  (n_rhs rows; the bits of `by_edge_product`), g and h
    rows fetched once for both (autograd of pygho/backend/Spspmm.py:309-315)."""
    sp = scatter_plan(plan, on_demand=True)
    dev = require_device(g, h, table, look_byc, addend)
    g, h, table = g.contiguous(), h.contiguous(), table.contiguous()
    d = g.shape[1]
    pc, a_byc, _ = plan.by_c()
    if addend is not None:
        addend = addend.contiguous()
        assert addend.shape == (plan.n_rhs, d) and addend.dtype == g.dtype
    if sp.covers:
        out = torch.empty((plan.n_rhs, d), dtype=g.dtype, device=dev)
    else:
        out = torch.zeros((plan.n_rhs, d), dtype=g.dtype, device=dev) if addend is None else addend.clone()
    gh = (torch.empty if sp.covers_c else torch.zeros)((plan.n_lhs, d), dtype=g.dtype, device=dev)
    es = g.element_size()
    # has-to-move bytes: g, h and gh rows once, the edge rows once (twice with the chained gradient), per message the packed word + two
    # by-c indices, the by-c CSR pointers, the chunk records
    nbytes = es * d * (g.shape[0] + h.shape[0] + plan.n_lhs + plan.n_rhs * (2 if addend is not None else 1)) + 12 * plan.m + 4 * (plan.n_lhs + 1) + 20 * sp.n_chunks
    name = f"seg_dual[{str(g.dtype).split('.')[-1]},sum{',res' if addend is not None else ''}]"
    _timed(name, nbytes, dev, lambda: check(lib().pygho_seg_dual(
        ptr(out), ptr(gh), ptr(addend), ptr(g), ptr(h), ptr(table), table.shape[0], ptr(sp.chunks), ptr(sp.words), ptr(sp.cgap), ptr(sp.chunk0),
        ptr(sp.blk_e), ptr(pc.seg_ptr), ptr(a_byc), ptr(look_byc), sp.n_blocks, sp.n_chunks, plan.m, sp.max_edges, plan.n_rhs, d, g.shape[0],
        h.shape[0], dtype_code(g), stream_ptr(dev)), "seg_dual"))
    return gh, out


class _MessageReduce(torch.autograd.Function):
    """out[a] = (+) lhs[c] * rhs[d] over the plan; either operand may be None (pattern only)."""

    @staticmethod
    def forward(ctx, lhs: Optional[Tensor], rhs: Optional[Tensor], plan: MessagePlan, aggr: str, addend: Optional[Tensor] = None):
        ties = None
        li, ri = (plan.c_fwd if lhs is not None else None), (plan.d_fwd if rhs is not None else None)
        if (aggr in ("max", "min") and addend is None and USE_FORWARD_TIES and any(ctx.needs_input_grad[:2])
                and _extremum_vec_ok(lhs if lhs is not None else rhs, lhs, rhs)
                and ((lhs if lhs is not None else rhs).dtype == torch.float32 or plan.fwd.max_len < 256)):
            # the forward also counts the ties its backward divides by (exact in the value dtype: at most 256 per segment in bf16)
            out, ties = seg_gmr_ties(plan.n_out, lhs, rhs, plan.fwd.seg_ptr, li, ri, aggr)
        else:
            out = seg_gmr(plan.n_out, lhs, rhs, plan.fwd.seg_ptr, li, ri, aggr, addend=None if addend is None else addend.contiguous())
        ctx.plan, ctx.aggr = plan, aggr
        ctx.has = (lhs is not None, rhs is not None)
        ctx.save_for_backward(lhs, rhs, out if aggr in ("max", "min") else None, ties)
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        lhs, rhs, fwd, fwd_ties = ctx.saved_tensors
        plan, aggr = ctx.plan, ctx.aggr
        gout = gout.contiguous()
        g_lhs = g_rhs = None
        scale = plan.fwd.inv_count if aggr == "mean" else None
        ties = None
        if aggr in ("max", "min"):
            if fwd_ties is not None and gout.dtype == fwd_ties.dtype:
                # share = grad / N_to_distribute, rounded to the value dtype: one elementwise pass (the forward counted the ties)
                dev = gout.device
                share = torch.empty_like(gout)
                _timed(f"seg_ext_share[{str(gout.dtype).split('.')[-1]},elementwise]", 3 * gout.numel() * gout.element_size(), dev,
                       lambda: torch.div(gout, fwd_ties, out=share))
                ties = ("share", share)
            else:
                ties = _ties(fwd, lhs, rhs, plan.fwd.seg_ptr, plan.c_fwd if lhs is not None else None,
                             plan.d_fwd if rhs is not None else None, gin=gout)
        if lhs is not None and ctx.needs_input_grad[0]:
            p, a_g, d_g = plan.by_c()
            if ties is None:
                g_lhs = seg_gmr(plan.n_lhs, gout, rhs, p.seg_ptr, a_g, d_g if rhs is not None else None, "sum", scale)
            else:
                g_lhs = _extremum_bwd(plan.n_lhs, gout, fwd, ties, lhs, rhs, p.seg_ptr, a_g, d_g)
        if rhs is not None and ctx.needs_input_grad[1]:
            if ties is None:
                g_rhs = by_edge_product(plan, gout, lhs, scale)
            else:
                p, a_g, c_g = plan.by_d()
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
            if row_gather_mean_ok(gout) and gout.data_ptr() % 16 == 0:
                return row_gather_mean(gout, ind32, plan.seg_ptr), None, None, None       # scale and gather in one pass, same bits
            scaled = gout * plan.inv_count.to(gout.dtype).unsqueeze(-1)
            return row_gather(scaled, ind32), None, None, None
        src, fwd = ctx.saved_tensors
        ties = _ties(fwd, src, None, plan.seg_ptr, plan.perm, None, gin=gout)
        up = unit_ptr(src.shape[0], src.device)
        return _extremum_bwd(src.shape[0], gout, fwd, ties, src, None, up, ind32, None), None, None, None


class _ScatterProd(torch.autograd.Function):
    """aggr = "prod" (utils.py:44-56 with reduce="prod"; coalesce(reduce="prod"), SpTensor.py:167-197): csrc/seg_prod.hip"""

    @staticmethod
    def forward(ctx, src: Tensor, plan: SegPlan):
        dev = require_device(src, plan.seg_ptr, plan.perm)
        src = src.contiguous()
        out = torch.empty((plan.n_seg, src.shape[1]), dtype=src.dtype, device=dev)
        check(lib().pygho_seg_prod(ptr(out), ptr(src), ptr(plan.seg_ptr), ptr(plan.perm), plan.n_seg, src.shape[1], dtype_code(src),
                                   stream_ptr(dev)), "seg_prod")
        ctx.plan = plan
        ctx.save_for_backward(src, out)
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        src, out = ctx.saved_tensors
        plan, dev = ctx.plan, src.device
        gsrc = torch.empty_like(src)
        check(lib().pygho_seg_prod_bwd(ptr(gsrc), ptr(gout.contiguous()), ptr(out), ptr(src), ptr(plan.seg_ptr), ptr(plan.perm), plan.n_seg,
                                       src.shape[1], dtype_code(src), stream_ptr(dev)), "seg_prod_bwd")
        return gsrc, None


_PROD_DTYPES = (torch.float32, torch.float64, torch.bfloat16, torch.float16, torch.int64)


def _scatter_prod(src2: Tensor, plan: SegPlan) -> Tensor:
    if src2.dtype not in _PROD_DTYPES:
        raise NotImplementedError(f"aggr 'prod' on {src2.dtype} values")
    require_static_rows(src2.shape[0], "aggr = 'prod'")
    return _ScatterProd.apply(src2, plan)


def scatter_reduce(src: Tensor, ind: Tensor, dim_size: int, aggr: str) -> Tensor:
    """torch_scatter_reduce(dim=0) (utils.py:44-56) on the HIP path."""
    require_device(src, ind)
    if aggr == "prod":
        assert ind.dim() == 1, "indice must be 1-d"
        assert src.shape[0] == ind.shape[0], "src and index length differ"
        src2 = _as2d(src) if src.dim() > 1 else src.contiguous().reshape(-1, 1)
        return _scatter_prod(src2, cached_plan(ind, dim_size, "scatter")).reshape((dim_size,) + tuple(src.shape[1:]))
    if aggr not in AGGR_CODE:
        raise NotImplementedError(f"aggr {aggr!r} is not supported (sum, mean, max, min, prod)")
    assert ind.dim() == 1, "indice must be 1-d"
    assert src.shape[0] == ind.shape[0], "src and index length differ"
    plan = cached_plan(ind, dim_size, "scatter")
    tail = tuple(src.shape[1:])
    src2 = _as2d(src) if src.dim() > 1 else src.contiguous().reshape(-1, 1)
    out = _ScatterReduce.apply(src2, plan, narrow_i32(ind), aggr)
    return out.reshape((dim_size,) + tail)


def scatter_reduce_planned(src: Tensor, plan: SegPlan, ind32: Tensor, aggr: str) -> Tensor:
    """scatter-reduce along a prebuilt plan (coalesce / sparse pooling)."""
    if aggr == "prod":
        src2 = _as2d(src) if src.dim() > 1 else src.contiguous().reshape(-1, 1)
        return _scatter_prod(src2, plan).reshape((plan.n_seg,) + tuple(src.shape[1:]))
    if aggr not in AGGR_CODE:
        raise NotImplementedError(f"aggr {aggr!r} is not supported (sum, mean, max, min, prod)")
    tail = tuple(src.shape[1:])
    src2 = _as2d(src) if src.dim() > 1 else src.contiguous().reshape(-1, 1)
    out = _ScatterReduce.apply(src2, plan, ind32, aggr)
    return out.reshape((plan.n_seg,) + tail)


USE_TABLE_GRAD = os.environ.get("PYGHO_TABLE_GRAD", "1") != "0"     # plan-free gradient of lookups into small tables (A/B switch)


def table_grad_ok(g2: Tensor, n_table: int) -> bool:
    return (USE_TABLE_GRAD and g2.is_cuda and g2.dim() == 2 and g2.shape[0] > 0 and 0 < n_table <= 64
            and g2.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and bool(lib().pygho_table_grad_supported(g2.shape[1], n_table)))


# The plan-free kernel is VALU-bound (n_table selects + adds per value: 60 us for 410 k rows of 256 B against 39 us for the planned
# hierarchy), but the planned route needs a radix sort and two host reads per index pattern and sums in another order.  Which route
# runs is a pure function of the call (shapes, and whether the CALLER installed / built a plan for the index array): round 4 switched
# a recurring large pattern to the planned route "after 3 uses", which made the bits of an embedding gradient depend on how often a
# batch had been seen -- step 3 and step 4 of a resident-batch run differed, and a resumed run diverged bitwise from a continuous one.
TABLE_GRAD_PLAN_ROWS = int(os.environ.get("PYGHO_TABLE_GRAD_PLAN_ROWS", str(1 << 16)))


def _table_grad_now(g2: Tensor, ind: Tensor, n_table: int) -> bool:
    if not table_grad_ok(g2, n_table):
        if dyn_rows(g2.shape[0]) is not None:
            # rows of a batch slot: the segment route is safe when the slot SERVES the grouping (pad rows lie outside every segment);
            # a grouping built from the padded index array would not be
            fac = getattr(ind, "_pygho_plan_factory", None)
            if fac is None or fac[0] != ind._version:
                require_static_rows(g2.shape[0], "the sorted-segment gradient of a row gather whose grouping the slot does not hold")
        return False
    if dyn_rows(g2.shape[0]) is not None:
        return True                                            # a batch slot's rows: the count is on the device, no plan can exist
    cache = getattr(ind, "_pygho_plans", None)
    if cache is not None and ("scatter", n_table, ind._version) in cache:
        return g2.shape[0] < TABLE_GRAD_PLAN_ROWS          # a plan exists (the caller built or installed it): use it where it is faster
    return True


def table_grad(g2: Tensor, ind: Tensor, n_table: int, out_dtype: Optional[torch.dtype] = None) -> Tensor:
    """sum of the rows of `g2` per value of `ind` (values in [0, n_table)) as an (n_table, d) tensor of g2's dtype (or `out_dtype`): the
    gradient of ``table[ind]`` for a small table (csrc/table_grad.hip: LDS bins per workgroup + a deterministic fold), accumulated
    in f32.  An index outside the table is skipped here and reported by the deferred range check of its first use."""
    dev = require_device(g2, ind)
    ind32 = narrow_i32(ind)
    m, d = g2.shape
    err = None
    seen = getattr(ind32, "_pygho_table_checked", None)
    bound = getattr(ind, "_pygho_value_bound", None)         # set by `collate.DeviceGraphStore`: the values were range-checked there
    known = bound is not None and bound[0] == ind._version and bound[1] <= n_table
    if not known and (seen is None or seen != (ind32._version, n_table)) and not torch.cuda.is_current_stream_capturing():
        # (under capture no flag: a flag allocated and zero-filled INSIDE a graph is uninitialised pool memory until the first replay,
        # and a fetch on this stream before that would read it -- the index was either checked in warm-up or stays unchecked here)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
    nblk = int(lib().pygho_table_grad_blocks(m, d, n_table))
    ws = torch.empty((nblk, n_table * d), dtype=torch.float32, device=dev)
    timer = LaunchTimer.active
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
    m_dev = dyn_rows(m)
    if m_dev is not None:
        check(lib().pygho_table_grad_dyn(ptr(ws), ptr(g2), ptr(ind32), m, ptr(m_dev), d, n_table, dtype_code(g2), ptr(err), stream_ptr(dev)),
              "table_grad_dyn")
    else:
        check(lib().pygho_table_grad(ptr(ws), ptr(g2), ptr(ind32), m, d, n_table, dtype_code(g2), ptr(err), stream_ptr(dev)), "table_grad")
    if timer is not None:
        e1.record(torch.cuda.current_stream(dev))
        timer.records.append((f"table_grad[{str(g2.dtype).split('.')[-1]}]", g2.element_size() * m * d + 4 * m + 4 * ws.numel(), e0, e1))
    if err is not None:
        defer_error(err, f"pygho_amd: lookup index out of range [0, {n_table})")
        try:
            ind32._pygho_table_checked = (ind32._version, n_table)
        except Exception:
            pass
    if nblk == 1:
        tot = ws
    else:
        tot = torch.empty((n_table * d,), dtype=torch.float32, device=dev)
        check(lib().pygho_sum_blocks(ptr(tot), ptr(ws), nblk, n_table * d, stream_ptr(dev)), "sum_blocks")
    return tot.view(n_table, d).to(out_dtype or g2.dtype)


class _RowGatherMaster(torch.autograd.Function):
    """table[ind] where `table` is the 16-bit copy (cast arena) of the f32 parameter `master`: the gradient goes to the MASTER, in its
    dtype, straight from the f32 accumulators of the reduction -- not rounded to the table's 16 bits and cast back (two launches per
    embedding and step, and 8 bits of every gradient entry)."""

    @staticmethod
    def forward(ctx, master: Tensor, table: Tensor, ind: Tensor):
        ctx.ind, ctx.n, ctx.mdt = ind, table.shape[0], master.dtype
        ctx.set_materialize_grads(False)
        return row_gather(table, narrow_i32(ind))

    @staticmethod
    def backward(ctx, gout: Optional[Tensor]):
        if gout is None:
            return None, None, None
        ind = ctx.ind
        g2 = _as2d(gout) if gout.dim() > 1 else gout.contiguous().reshape(-1, 1)
        if _table_grad_now(g2, ind, ctx.n):
            g = table_grad(g2, ind, ctx.n, out_dtype=ctx.mdt)
        else:
            g = seg_reduce_rows(g2, cached_plan(ind, ctx.n, "scatter"), "sum").to(ctx.mdt)
        return g.reshape((ctx.n,) + tuple(gout.shape[1:])), None, None


def gather_rows_master(master: Tensor, table: Tensor, ind: Tensor) -> Tensor:
    """`table[ind]` along dim 0 for `table` = a detached lower-precision copy of the parameter `master` (same shape): the gradient is
    returned to `master`"""
    require_device(master, table, ind)
    assert table.shape == master.shape and not table.requires_grad
    return _RowGatherMaster.apply(master, table, ind)


class _RowGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src: Tensor, ind: Tensor):
        ctx.ind, ctx.n = ind, src.shape[0]
        # a consumer that returns the TABLE's gradient itself (`dual_backward_tg`) sends nothing here: without this the engine would
        # hand over an all-zero (rows, d) gradient and the reduction below would run on it
        ctx.set_materialize_grads(False)
        return row_gather(src, narrow_i32(ind))

    @staticmethod
    def backward(ctx, gout: Optional[Tensor]):
        if gout is None:
            return None, None
        ind = ctx.ind
        g2 = _as2d(gout) if gout.dim() > 1 else gout.contiguous().reshape(-1, 1)
        if _table_grad_now(g2, ind, ctx.n):
            g = table_grad(g2, ind, ctx.n)                  # a handful of table rows: no index plan at all
        else:
            g = seg_reduce_rows(g2, cached_plan(ind, ctx.n, "scatter"), "sum")
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
            ties = _ties(fwd, val, X, plan.seg_ptr, plan.perm if val is not None else None, src_g, gin=gout)
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


USE_PAIR_BWD = os.environ.get("PYGHO_PAIR_BWD", "1") != "0"     # one-pass backward of the tuple initialisation on symmetric tuple sets


def pair_mirror(row32: Tensor, col32: Tensor, vidx32: Tensor, n_nodes: int) -> Optional[Tensor]:
    """mirror[t] = position of tuple (col[t], row[t]) for a tuple list sorted by (row, col) that is SYMMETRIC -- (j, i) present with
    every (i, j), equal feature on both, features below `pygho_pair_bwd_types()` -- else None.  K-hop tuple sets with a
    shortest-path-distance feature (hodata/SpTupleSampler.py:91-126) are.  One host synchronisation per tuple pattern, memoised on
    the index object (keyed by the partner objects AND every index tensor's version counter: an in-place edit of the pattern
    invalidates the verdict); `SpModel.prepare` takes it off the training step for batches collated ahead.  Called from
    `pair_product`'s FORWARD, where the plans are built -- never from backward; under stream capture without a memo the answer is
    None (the three-launch backward), because the verdict needs a device-to-host read."""
    vers = (row32._version, col32._version, vidx32._version)
    memo = getattr(row32, "_pygho_mirror", None)
    if memo is not None and memo[0] is col32 and memo[1] is vidx32 and memo[2] == n_nodes and memo[4] == vers:
        return memo[3]
    if row32.is_cuda and torch.cuda.is_current_stream_capturing():
        return None
    n = row32.numel()
    res = None
    if 0 < n < (1 << 31) and col32.numel() == n and vidx32.numel() == n:
        nt = int(lib().pygho_pair_bwd_types())
        key = row32.to(torch.int64) * n_nodes + col32
        key_t = col32.to(torch.int64) * n_nodes + row32
        pos = torch.searchsorted(key, key_t).clamp_(max=n - 1)
        lo, hi = torch.aminmax(vidx32)
        ok = ((key[1:] > key[:-1]).all() & (key[pos] == key_t).all() & (vidx32[pos] == vidx32).all() & (lo >= 0) & (hi < nt)
              & (col32.max() < n_nodes))
        if _fetch(ok.to(torch.int32).reshape(1))[0]:
            res = pos.to(torch.int32)
    try:
        row32._pygho_mirror = (col32, vidx32, n_nodes, res, vers)
    except Exception:
        pass
    return res


def pair_bwd(g: Tensor, left: Tensor, right: Tensor, tab: Tensor, seg_ptr: Tensor, col32: Tensor, vidx32: Tensor, mirror: Tensor,
             tab_f32: bool = False):
    """(g_left, g_right, g_tab) of `pair_product` with a table operand in one pass over g (`pygho_pair_bwd`).  `tab_f32`: the table's
    gradient as the f32 fold of the workgroups' slabs (for the table's f32 master parameter) instead of rounded to the table's dtype."""
    dev = require_device(g, left, right, tab, seg_ptr, col32, vidx32, mirror)
    n_nodes, d = left.shape[0], left.shape[1]
    g_left, g_right = torch.empty_like(left), torch.empty_like(right)
    nt = int(lib().pygho_pair_bwd_types())
    nblk = int(lib().pygho_pair_bwd_blocks(n_nodes, d, dtype_code(g)))
    ws = torch.empty((nblk, nt * d), dtype=torch.float32, device=dev)
    timer = LaunchTimer.active
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
    check(lib().pygho_pair_bwd(ptr(g_left), ptr(g_right), ptr(ws), ptr(g), ptr(left), ptr(right), ptr(tab), ptr(seg_ptr), ptr(col32),
                               ptr(vidx32), ptr(mirror), n_nodes, col32.numel(), d, dtype_code(g), stream_ptr(dev)), "pair_bwd")
    if timer is not None:
        e1.record(torch.cuda.current_stream(dev))
        nbytes = g.element_size() * d * (g.shape[0] + 4 * n_nodes) + 12 * col32.numel() + 4 * (n_nodes + 1) + 4 * ws.numel()
        timer.records.append((f"pair_bwd[{str(g.dtype).split('.')[-1]}]", nbytes, e0, e1))
    # fold the workgroups' slabs: a (blocks, nt * d) array is only nt * d / 64 workgroups wide for pygho_sum_blocks, so groups of 64 slabs
    # are folded side by side first
    width, fan = nt * d, 64
    if nblk % fan == 0 and nblk > fan:
        mid = torch.empty((fan * width,), dtype=torch.float32, device=dev)
        check(lib().pygho_sum_blocks(ptr(mid), ptr(ws), nblk // fan, fan * width, stream_ptr(dev)), "sum_blocks")
        ws, nblk = mid, fan
    if tab_f32:
        # the fold writes the rows behind the kernel's `nt` as zeros itself (table rows without a tuple get no gradient)
        rows = max(nt, tab.shape[0])
        tot = torch.empty((rows * d,), dtype=torch.float32, device=dev)
        check(lib().pygho_sum_blocks_pad(ptr(tot), ptr(ws), nblk, width, rows * d, stream_ptr(dev)), "sum_blocks_pad")
        return g_left, g_right, tot.view(rows, d)[:tab.shape[0]]
    tot = torch.empty((width,), dtype=torch.float32, device=dev)
    check(lib().pygho_sum_blocks(ptr(tot), ptr(ws), nblk, width, stream_ptr(dev)), "sum_blocks")
    g_tab = torch.zeros((tab.shape[0], d), dtype=tab.dtype, device=dev)          # table rows without a tuple get no gradient
    k = min(nt, tab.shape[0])
    g_tab[:k] = tot.view(nt, d)[:k].to(tab.dtype)
    return g_left, g_right, g_tab


class _PairProduct(torch.autograd.Function):
    """out[t] = (left[row[t]] * right[col[t]]) * val[vidx[t]] (vidx None = t): the tuple initialisation of
    example/minimal.py:62-67 (two unpoolings of node features onto the tuple pattern and two elementwise products; with
    `vidx` also the embedding lookup of the tuple feature, example/minimal.py:30-33) as ONE pass; the three operand
    gradients are the same three-operand kernel over the unit / by-row / by-col / by-feature groupings of the tuples."""

    @staticmethod
    def forward(ctx, left, right, val, row32, col32, vidx32, by_row, by_col, by_val, mirror, groupings=None, val_master=None):
        # `val_master`: `val` is a (detached) 16-bit copy of this f32 parameter -- the table's gradient is returned to IT, in f32
        n = row32.numel()
        ctx.mirror, ctx.groupings = mirror, groupings
        ctx.master_dtype = None if val_master is None else val_master.dtype
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
        mirror = ctx.mirror
        mdt = ctx.master_dtype
        to_master = mdt is not None and ctx.needs_input_grad[11]
        if (mirror is not None and g.dtype == left.dtype and g.shape[1] == left.shape[1] and left.is_contiguous()
                and right.is_contiguous() and val.is_contiguous()):
            g_left, g_right, g_val = pair_bwd(g, left, right, val, by_row[0].seg_ptr, col32, vidx32, mirror,
                                              tab_f32=to_master and mdt == torch.float32)
            return (g_left if ctx.needs_input_grad[0] else None, g_right if ctx.needs_input_grad[1] else None,
                    g_val if ctx.needs_input_grad[2] else None, None, None, None, None, None, None, None, None,
                    g_val.to(mdt) if to_master else None)
        if by_col is None and ctx.groupings is not None:
            by_col, by_val = ctx.groupings()          # the gradient arrived in another dtype / layout than the forward promised
        if ctx.needs_input_grad[0]:
            p, col_p, v_p = by_row
            g_left = seg_triple(p.n_seg, g, val, right, p.seg_ptr, p.perm, v_p if vidx32 is not None else p.perm, col_p)
        if ctx.needs_input_grad[1]:
            p, row_p, v_p = by_col
            g_right = seg_triple(p.n_seg, g, val, left, p.seg_ptr, p.perm, v_p if vidx32 is not None else p.perm, row_p)
        if ctx.needs_input_grad[2] or to_master:
            if vidx32 is None:
                g_val = seg_triple(n, g, left, right, unit_ptr(n, g.device), None, row32, col32)
            else:
                # gradient of the (small) table: a handful of very long segments -> chunked f32 partial sums, then a tree
                p, row_p, col_p = by_val
                levels = p.levels(LONG_CHUNK) if p.max_len > LONG_SEGMENT else [p.seg_ptr]
                cur = seg_triple(levels[0].numel() - 1, g, left, right, levels[0], p.perm, row_p, col_p, out_f32=len(levels) > 1)
                for lv in levels[1:]:
                    cur = seg_gmr(lv.numel() - 1, cur, None, lv, None, None, "sum")
                g_val = cur.to(mdt if to_master else val.dtype)
        return (g_left, g_right, g_val if ctx.needs_input_grad[2] else None, None, None, None, None, None, None, None, None,
                (g_val.to(mdt) if to_master else None))


def _grouped(plan: SegPlan, key, *idx32):
    """index arrays re-ordered into the plan's grouped order, memoised on the plan object."""
    memo = plan._partner                     # `key`: the index tensor OBJECTS (kept alive by the memo, compared by identity)
    if memo is None or len(memo[0]) != len(key) or any(a is not b for a, b in zip(memo[0], key)):
        memo = (key, tuple(None if i is None else plan.take(i) for i in idx32))
        plan._partner = memo
    return memo[1]


def pair_product(left: Tensor, right: Tensor, val: Tensor, row: Tensor, col: Tensor, val_index: Optional[Tensor] = None,
                 val_master: Optional[Tensor] = None) -> Tensor:
    """``left[row] * right[col] * val`` for (n_rows, d) / (n_cols, d) node features and (nnz, d) tuple values -- or, with
    `val_index`, ``... * val[val_index]`` for a small (n_types, d) table.  `row` / `col` / `val_index` are persistent
    int64 index arrays of the tuple pattern (plans are cached on them).  `val_master`: the table `val` is a detached 16-bit copy of
    this parameter (cast arena); the table's gradient is returned to the parameter in its own dtype, from f32 accumulators."""
    require_device(left, right, val, row, col, val_index)
    if val_master is not None:
        assert val_index is not None and val_master.shape == val.shape and not val.requires_grad
        if not val_master.requires_grad:
            val_master = None
    assert left.dim() == right.dim() == val.dim() == 2
    row32, col32 = narrow_i32(row), narrow_i32(col)
    vidx32 = None if val_index is None else narrow_i32(val_index)
    key = (row32, col32, vidx32)
    p_row = cached_plan(row, left.shape[0], "pair-row")
    by_row = (p_row,) + _grouped(p_row, key, col32, vidx32)
    # the one-pass backward needs the tuple set's mirror permutation: decided HERE (one host read per pattern, memoised -- or
    # installed by `collate.DeviceGraphStore`), where the plans are built, so that backward never synchronises
    mirror = None
    if (USE_PAIR_BWD and vidx32 is not None and torch.is_grad_enabled() and (left.requires_grad or right.requires_grad or val.requires_grad or val_master is not None)
            and p_row.perm is None and left.dtype in (torch.bfloat16, torch.float16) and left.dtype == right.dtype == val.dtype
            and left.shape == right.shape and (left.shape[1] * 2) % 16 == 0 and left.shape[1] * 2 <= 1024
            and left.is_contiguous() and right.is_contiguous() and val.is_contiguous()):
        mirror = pair_mirror(row32, col32, vidx32, left.shape[0])

    n_right, n_val = right.shape[0], val.shape[0]

    def groupings():
        """the by-column / by-feature groupings of the three-launch backward (two radix sorts per pattern, cached on the index
        objects): built in forward unless the one-pass backward is certain to run"""
        p_col = cached_plan(col, n_right, "pair-col", assume_sorted=False)
        by_col = (p_col,) + _grouped(p_col, key, row32, vidx32)
        by_val = None
        if val_index is not None:
            p_val = cached_plan(val_index, n_val, "pair-val", assume_sorted=False)
            by_val = (p_val,) + _grouped(p_val, key, row32, col32)
        return by_col, by_val
    by_col, by_val = (None, None) if mirror is not None else groupings()
    return _PairProduct.apply(left, right, val, row32, col32, vidx32, by_row, by_col, by_val, mirror, groupings, val_master)

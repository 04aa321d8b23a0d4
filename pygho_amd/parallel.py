"""
Data parallelism over graphs.  A batch is a block-diagonal union of graphs, so the whole
operator path shards by graph with no data-path collective; the only exchange is the gradient
all-reduce.  The reference has no distributed code at all; this is new work.

``FlatGradSync`` packs every parameter gradient into ONE flat buffer so that the step's gradient exchange is
a single RCCL all-reduce over xGMI (0.70 MB for the minimal NGNN: latency-bound), or -- ``overlap=True`` -- two
halves of it issued on a side stream from backward hooks, the later layers' half while backward still runs.
"""
import time
import weakref
from typing import Iterable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


class FlatGradSync:
    """gradient exchange of one data-parallel step.

    ``zero_grad()`` drops the gradients (autograd then hands its freshly produced tensors over instead of launching
    one accumulate kernel per parameter); ``sync()`` packs them into ONE flat buffer with a multi-tensor copy,
    averages it over the ranks and re-points every ``param.grad`` at its slice (without a process group only the
    packing happens).

    ``overlap=True``: the flat buffer is cut into ``buckets`` contiguous parameter ranges (in ``parameters()`` order:
    backward finishes the LAST range first); a post-accumulate hook on every parameter counts its range down, and the
    moment a range is complete its gradients are packed and its all-reduce is issued on a SIDE stream while backward
    goes on with the earlier layers.  ``sync()`` then only launches what is left (ranges holding parameters without a
    gradient), joins the side stream and scales.  The result is the same buffer as the single collective's: an
    all-reduce sums element-wise, so the bucket boundaries cannot change a bit.  Contract: ONE exchanging backward pass per
    ``zero_grad()`` / ``sync()`` pair; gradient accumulation over micro-batches runs the earlier passes under ``no_sync()``
    (the hooks stay quiet, the gradients accumulate in ``param.grad``) and only the last pass outside it."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional["dist.ProcessGroup"] = None,
                 dtype: torch.dtype = torch.float32, overlap: bool = False, buckets: int = 2):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        self.views: List[torch.Tensor] = []
        offs = [0]
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[offs[-1]:offs[-1] + n].view_as(p))
            offs.append(offs[-1] + n)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.allreduce_calls = 0
        self.overlap = bool(overlap) and dist.is_available() and dist.is_initialized()
        self._quiet = False                   # inside no_sync(): backward passes accumulate, nothing is exchanged
        self._events = []                     # (start, end) device events around every side-stream collective (allreduce_ms)
        self._handles = []                    # RemovableHandles of the backward hooks (close())
        self._closed = False
        # evidence of overlap for the LAST step: host time / device event when each range's collective was issued, and when
        # backward() returned (mark_backward_end)
        self._issue_log: List[Tuple[float, int, object]] = []
        self._bwd_end: Optional[Tuple[float, object]] = None
        if self.overlap:
            # contiguous parameter ranges of about equal size
            nb = max(1, min(int(buckets), len(self.params)))
            cuts, target = [0], total / nb
            for i in range(1, len(self.params)):
                if len(cuts) < nb and offs[i] >= target * len(cuts):
                    cuts.append(i)
            cuts.append(len(self.params))
            self._ranges = [(cuts[i], cuts[i + 1], offs[cuts[i]], offs[cuts[i + 1]]) for i in range(len(cuts) - 1)]
            self._bucket_of = [b for b, (lo, hi, _, _) in enumerate(self._ranges) for _ in range(lo, hi)]
            self._pending = [hi - lo for lo, hi, _, _ in self._ranges]
            self._launched = [False] * len(self._ranges)
            self._works = []
            self._side = torch.cuda.Stream(device=dev) if self.flat.is_cuda else None
            # RCCL work.wait() only makes the calling STREAM wait; any other backend's wait() blocks the host until the collective
            # is done, so there it is deferred to sync() -- otherwise the hook would stall backward and nothing could overlap
            self._stream_ordered = self.flat.is_cuda and dist.get_backend(group) == "nccl"
            # the hooks hold this object only weakly: a syncer that is dropped (or close()d) stops exchanging anything
            ref = weakref.ref(self)
            for i, p in enumerate(self.params):
                def hook(_p, i=i, ref=ref):
                    me = ref()
                    if me is not None and not me._closed:
                        me._on_grad(i)
                self._handles.append(p.register_post_accumulate_grad_hook(hook))

    @property
    def capturable(self) -> bool:
        """can a step that exchanges through this object be captured into a HIP graph?  Without a process group (packing only) and
        with RCCL (stream-ordered collectives) yes; a host-staged backend (gloo) no."""
        if not (dist.is_available() and dist.is_initialized()):
            return True
        return self.flat.is_cuda and dist.get_backend(self.group) == "nccl"

    def close(self) -> None:
        """remove the backward hooks (a second FlatGradSync over the same parameters must not leave this one issuing collectives)"""
        self._closed = True
        for h in self._handles:
            h.remove()
        self._handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def zero_grad(self) -> None:
        for p in self.params:
            p.grad = None
        if self.overlap:
            self._pending = [hi - lo for lo, hi, _, _ in self._ranges]
            self._launched = [False] * len(self._ranges)
            self._issue_log = []
            self._bwd_end = None

    def _pack_range(self, lo: int, hi: int) -> None:
        missing = [v for v, p in zip(self.views[lo:hi], self.params[lo:hi]) if p.grad is None]
        have = [(v, p.grad) for v, p in zip(self.views[lo:hi], self.params[lo:hi])
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])

    def pack(self) -> None:
        """flat <- gradients (parameters without a gradient contribute zeros), then param.grad = its flat slice."""
        self._pack_range(0, len(self.params))
        for v, p in zip(self.views, self.params):
            p.grad = v

    def no_sync(self):
        """``with sync.no_sync(): loss.backward()`` -- a backward pass of a micro-batch that is NOT the last one before ``sync()``:
        its gradients accumulate in ``param.grad`` and no range is exchanged.  The last pass runs outside the context (its hooks
        launch the ranges as they complete, now holding the accumulated gradients); ``sync()`` launches whatever is left."""
        me = self

        class _Quiet:
            def __enter__(self):
                self.prev, me._quiet = me._quiet, True

            def __exit__(self, *exc):
                me._quiet = self.prev
                return False
        return _Quiet()

    def _on_grad(self, i: int) -> None:
        if self._quiet:
            return
        b = self._bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] < 0:
            raise RuntimeError("FlatGradSync: a parameter received a second gradient before sync() -- ONE exchanging backward pass "
                               "per zero_grad() / sync() pair (a second pass would re-reduce an already exchanged range); run the "
                               "earlier passes of a gradient accumulation under `with sync.no_sync():`")
        if self._pending[b] == 0 and not self._launched[b]:
            # under stream capture the hook (autograd's worker thread) leaves the range to sync(), which records the collective on the
            # capturing stream from the capturing thread (see _launch)
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                return
            self._launch(b)

    def _launch(self, b: int) -> None:
        lo, hi, flo, fhi = self._ranges[b]
        self._launched[b] = True
        self._pack_range(lo, hi)                       # on the stream backward runs on
        piece = self.flat[flo:fhi]
        if self._side is None:                         # host tensors (gloo on CPU): no streams to overlap
            self._works.append((dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None))
            self._issue_log.append((time.perf_counter(), b, None))
        else:
            cur = torch.cuda.current_stream(self.flat.device)
            # under stream capture (`graphs.SlotStep(..., sync=...)`: the whole step incl. this collective becomes ONE HIP graph) the
            # collective is recorded on the CAPTURING stream itself, from sync(): on this stack (PyTorch 2.10 / ROCm 7.2 / RCCL 2.26) a
            # collective issued on a second stream that joined the capture crashes hipStreamEndCapture, one on the capturing stream is
            # recorded fine (tools/experiments/rccl_capture_repro.py: every variant).  Inside a captured step the exchange therefore
            # follows backward instead of overlapping it: 0.7 MB over xGMI against a 2.5 ms step
            capturing = torch.cuda.is_current_stream_capturing()
            if capturing:
                if not self._stream_ordered:
                    raise RuntimeError("FlatGradSync: only the RCCL backend can be captured into a HIP graph (a host-staged collective has "
                                       "no stream to be ordered on); run this step eagerly")
                dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group)
                self._issue_log.append((time.perf_counter(), b, None))
                self._captured_on_current = True
                self.allreduce_calls += 1
                return
            self._side.wait_stream(cur)                # the packed range is complete on the side stream
            with torch.cuda.stream(self._side):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self._side)
                work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._issue_log.append((time.perf_counter(), b, e0))
                if self._stream_ordered:
                    work.wait()                        # RCCL: the SIDE stream waits for the collective (the host does not block)
                    e1.record(self._side)
                else:
                    self._works.append((work, e1))     # joined in sync()
                self._events.append((e0, e1))
                del self._events[:-256]
        self.allreduce_calls += 1

    def mark_backward_end(self) -> None:
        """call right after loss.backward() returns: host time + an event on the compute stream, against which overlap_report()
        places the moment the first range's collective was issued"""
        ev = None
        if self.flat.is_cuda and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.flat.device))
        self._bwd_end = (time.perf_counter(), ev)

    def overlap_report(self) -> Optional[dict]:
        """for the last step (after a device synchronisation): how long before the END of backward the first collective was issued
        (host clock) and started on the device (event on the side stream vs the backward-end event on the compute stream).
        Positive numbers = the exchange of the later layers' range ran while backward was still producing the earlier layers'."""
        if not self.overlap or not self._issue_log or self._bwd_end is None:
            return None
        t_end, ev_end = self._bwd_end
        t0, b0, e0 = self._issue_log[0]
        rep = {"ranges_issued_inside_backward": sum(1 for t, _, _ in self._issue_log if t < t_end),
               "ranges": len(self._ranges), "first_range": b0,
               "host_ms_first_issue_before_backward_end": (t_end - t0) * 1e3}
        if e0 is not None and ev_end is not None:
            rep["device_ms_first_collective_start_before_backward_end"] = e0.elapsed_time(ev_end)
        return rep

    def sync(self) -> None:
        """average the gradient over all ranks (one collective over the flat buffer; with `overlap` one per range, most of them
        already in flight)."""
        if not self.overlap:
            self.pack()
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
                self.allreduce_calls += 1
                if self.world > 1:
                    self.flat.div_(self.world)
            return
        for b in range(len(self._ranges)):
            if not self._launched[b]:                  # a range with a parameter that received no gradient
                self._launch(b)
        for w, e1 in self._works:
            if self._side is None:
                w.wait()
            else:
                with torch.cuda.stream(self._side):    # the result becomes visible to the SIDE stream, which the compute stream joins
                    w.wait()
                    e1.record(self._side)
        self._works = []
        if self._side is not None and not (torch.cuda.is_current_stream_capturing() and getattr(self, "_captured_on_current", False)):
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
        if self.world > 1:
            self.flat.div_(self.world)
        for v, p in zip(self.views, self.params):
            p.grad = v
        # one backward per zero_grad() / sync() pair; the counters are re-armed here so that a caller who clears the gradients by
        # other means (optimizer.zero_grad) still exchanges the next step's
        self._pending = [hi - lo for lo, hi, _, _ in self._ranges]
        self._launched = [False] * len(self._ranges)

    def allreduce_ms(self) -> Optional[float]:
        """mean device time of one side-stream collective so far (after a device synchronisation); None without `overlap`"""
        if not self._events:
            return None
        return sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)

    def broadcast_params(self, src: int = 0) -> None:
        """every rank starts from rank `src`'s parameters: ONE broadcast per parameter dtype over a flat staging buffer (round 4
        issued one collective per parameter, ~70 latency-bound calls for the minimal NGNN)"""
        if not (dist.is_available() and dist.is_initialized()):
            return
        by_dtype = {}
        for p in self.params:
            by_dtype.setdefault(p.dtype, []).append(p)
        with torch.no_grad():
            for dtype, ps in by_dtype.items():
                flat = torch.empty(sum(p.numel() for p in ps), dtype=dtype, device=ps[0].device)
                views, off = [], 0
                for p in ps:
                    views.append(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()
                torch._foreach_copy_(views, [p.detach() for p in ps])
                dist.broadcast(flat, src=src, group=self.group)
                torch._foreach_copy_([p.detach() for p in ps], views)
                self.broadcast_calls = getattr(self, "broadcast_calls", 0) + 1
        from . import _ops
        _ops.invalidate_cast_arenas()              # the parameters changed underneath any 16-bit copies


def shard_ranges(weights, world_size: int) -> List[Tuple[int, int]]:
    """contiguous graph ranges per rank, balanced by cumulative weight (per-graph message count)."""
    w = np.asarray(weights, dtype=np.float64)
    cum = np.concatenate(([0.0], np.cumsum(w)))
    cuts = [0]
    for r in range(1, world_size):
        cuts.append(max(cuts[-1], int(np.searchsorted(cum, cum[-1] * r / world_size, side="left"))))
    cuts.append(len(w))
    return [(cuts[i], cuts[i + 1]) for i in range(world_size)]

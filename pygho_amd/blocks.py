"""
The dense steps around the aggregation as fused blocks: BatchNorm + activation, 16-bit parameter copies
(cast arena), the streaming row-block Linear family and the tuple-wise / concatenation / SSWL blocks built from them.
"""
from __future__ import annotations

import os
from ctypes import c_void_p
from typing import Optional, Tuple

import torch
from torch import Tensor

from ._native import AGGR_CODE, DTYPE_CODE, check, dtype_code, lib, ptr, require_device, stream_ptr

from .plans import *          # noqa: F401,F403
from .plans import _I32, _fetch, _flag
from .segment import *        # noqa: F401,F403
from .segment import _as2d
from .masked import *         # noqa: F401,F403


# --------------------------------------------------------------------------
# fused BatchNorm + activation (dense neighbour of the aggregation)
# --------------------------------------------------------------------------


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
        assert partial is not None or not training
    w32 = None if weight is None else weight.detach().float().contiguous()
    b32 = None if bias is None else bias.detach().float().contiguous()
    if training:
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        var = torch.empty(c, dtype=torch.float32, device=dev)
    else:
        mean, var = running_mean.float().clone(), running_var.float().clone()
    invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(3))
    fold = training and fold_momentum is not None
    m_dev = dyn_rows(m) if training else None        # the batch statistics divide by the TRUE row count (a device value in a slot)
    if training and partial is not None:
        sums, sum_shift = partial
        if m_dev is not None:
            check(lib().pygho_bn_finalize_dyn(ptr(mean), ptr(var), ptr(invstd), ptr(scale), ptr(shift), ptr(sums), sums.shape[0],
                                              ptr(sum_shift), m, ptr(m_dev), c, ptr(w32), ptr(b32), float(eps),
                                              ptr(running_mean) if fold else None, ptr(running_var) if fold else None,
                                              float(fold_momentum or 0.0), st), "bn_finalize_dyn")
        else:
            check(lib().pygho_bn_finalize(ptr(mean), ptr(var), ptr(invstd), ptr(scale), ptr(shift), ptr(sums), sums.shape[0],
                                          ptr(sum_shift), m, c, ptr(w32), ptr(b32), float(eps), ptr(running_mean) if fold else None,
                                          ptr(running_var) if fold else None, float(fold_momentum or 0.0), st), "bn_finalize")
    elif m_dev is not None:
        check(lib().pygho_bn_prepare_dyn(ptr(mean), ptr(var), ptr(invstd), ptr(scale), ptr(shift), ptr(x), m, ptr(m_dev), c,
                                         ptr(w32), ptr(b32), float(eps), ptr(running_mean) if fold else None,
                                         ptr(running_var) if fold else None, float(fold_momentum or 0.0), ptr(ws), dt, st),
              "bn_prepare_dyn")
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
    m_dev = dyn_rows(m)
    if m_dev is not None:
        check(lib().pygho_bn_act_bwd_dyn(ptr(dx), ptr(s1), ptr(s2), ptr(x), ptr(gy), ptr(mean), ptr(invstd), ptr(w32), ptr(b32),
                                         m, ptr(m_dev), c, ACT_CODE[act], 1 if training else 0, ptr(ws), dtype_code(x), ptr(sdx),
                                         stream_ptr(dev)), "bn_act_bwd_dyn")
        return dx, s1, s2, sdx
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


USE_DEFERRED_COUNTERS = os.environ.get("PYGHO_DEFER_COUNTERS", "1") != "0"
_PENDING_BATCH_COUNTERS: Optional[list] = None      # set by `deferred_batch_counters`: the counters of one forward pass, bumped together


class deferred_batch_counters:
    """inside this context every BatchNorm's `num_batches_tracked += 1` (one scalar launch per layer: 8 per step in the NGNN model)
    is collected and applied by ONE `torch._foreach_add_` at exit.  The cumulative-average mode (momentum None) reads its counter
    on the host and keeps the immediate update."""

    def __enter__(self):
        global _PENDING_BATCH_COUNTERS
        self.outer = _PENDING_BATCH_COUNTERS
        _PENDING_BATCH_COUNTERS = []
        return self

    def __exit__(self, *exc):
        global _PENDING_BATCH_COUNTERS
        pend, _PENDING_BATCH_COUNTERS = _PENDING_BATCH_COUNTERS, self.outer
        if pend and exc[0] is None:
            with torch.no_grad():
                torch._foreach_add_(pend, 1)


def _update_running(bn, mean: Tensor, var: Tensor, n: int, folded: bool = False) -> None:
    if bn.training and bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            if (USE_DEFERRED_COUNTERS and _PENDING_BATCH_COUNTERS is not None and bn.momentum is not None
                    and bn.num_batches_tracked is not None):
                _PENDING_BATCH_COUNTERS.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked += 1
            if folded:
                return
            require_static_rows(n, "the running-average update of a BatchNorm without a fixed momentum")
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
_ARENA_EPOCH = [0]      # bumped by whoever changes parameters behind the version counters' back (an optimizer step, a HIP graph replay)


def _invalidate_after_optimizer_step(*_args, **_kwargs) -> None:
    _ARENA_EPOCH[0] += 1


# Version counters are NOT a reliable staleness signal: the fused multi-tensor optimizers (`torch.optim.AdamW(fused=True)`: one
# `_fused_adamw_` launch) update the parameters in place WITHOUT moving `p._version` (measured: the arena kept serving the weights
# of step 0 for a whole training run).  Every optimizer step therefore invalidates every arena (a global post-step hook on
# torch.optim.Optimizer: it fires for every subclass, fused or not): one `_foreach_copy_` launch per step, which is what the arena is
# for; the version / address checks remain as the guard for updates between two steps.  The copies are NOT rewritten when nothing is
# known to have changed: autograd saves the arena's views (_ArenaLinearFn, _PairProduct), and a second grad-enabled forward before
# backward (siamese / contrastive use, a validation pass in between) must not bump their version counters.
try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook
    _register_step_hook(_invalidate_after_optimizer_step)
except ImportError:                                                         # pragma: no cover - older torch
    _register_step_hook = None


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
        # aliases with their OWN version counters, for the re-cast of UNCHANGED parameters to write through: autograd saves `views`
        # (_ArenaLinearFn, _PairProduct, _TupleBlock), and a re-cast between a forward and its backward -- a second grad-enabled
        # forward first: siamese use, a validation pass -- must not trip the saved tensors' version check; it writes the same bits.
        # Parameters known to have changed are written through `views` themselves (`refresh`).
        self.raw_views = [v.data for v in self.views]
        self.versions = [-1] * len(self.params)
        self.ptrs = [0] * len(self.params)          # storage address at the last refresh: `module.to()` / `p.data = ...` swap it
        self.epoch = -1
        ref = weakref.ref(self)
        for k in [k for k, (r, _i) in _ARENA_OF.items() if r() is None]:      # entries of arenas that are gone
            del _ARENA_OF[k]
        for i, p in enumerate(self.params):
            _ARENA_OF[id(p)] = (ref, i)

    def record_owners(self, module) -> None:
        """remember where the module tree holds every parameter, so that later calls can verify "same parameters as when the arena
        was built" with dictionary lookups instead of walking the module tree (0.2 ms per forward of a 70-module model)"""
        where, dicts = {}, []
        for mod in module.modules():
            dicts.append(mod._parameters)
            for name, p in mod._parameters.items():
                if p is not None:
                    where.setdefault(id(p), (mod._parameters, name))
        self.owners = [where.get(id(p)) for p in self.params]
        self.dicts, self.n_entries = dicts, sum(len(d) for d in dicts)

    def same_parameters(self) -> bool:
        """every recorded parameter is still the object its module holds and no module gained or lost a parameter entry.  (A
        SUBMODULE added after the first forward is not seen: call `rebuild_cast_arena(module)` after such surgery.)"""
        owners = getattr(self, "owners", None)
        if owners is None or any(o is None for o in owners):
            return False
        for (d, name), p in zip(owners, self.params):
            if d.get(name) is not p or not p.is_cuda or p.dtype != torch.float32:
                return False
        return sum(len(d) for d in self.dicts) == self.n_entries

    def refresh(self, force: bool = False) -> None:
        """Freshness is decided by the parameter's version counter and storage address.  Covered update paths: in-place ops on the
        parameter (optimizers, `p.copy_`, `load_state_dict`), `module.to()` / `p.data = t` (new storage), a replayed HIP graph
        (`GraphedStep.replay` bumps the epoch).  NOT visible from here: writes through a `.data` alias (`p.data.mul_(...)` has its
        own version counter) -- call `invalidate_cast_arenas()` after those.  Under stream capture every copy is re-cast INSIDE
        the graph: a graph that captured only forward + backward must not bake in the views of a cast that happened before it."""
        # KNOWN to have changed since the last cast (an optimizer step / `invalidate_cast_arenas()` moved the epoch, a version counter or
        # a storage address moved): those copies are rewritten through the views autograd may have SAVED, so their version counter
        # moves and the backward of a graph recorded BEFORE the update raises torch's usual "modified by an inplace operation" error
        # instead of silently reading the new 16-bit weights (forward, optimizer step, forward, backward of the FIRST graph).
        # Copies re-cast only because every forward re-casts (`force`: updates nobody can see, e.g. through `.data`) go through the
        # aliases with their own counters: with unchanged parameters they receive the same bits, and a second grad-enabled forward
        # before backward (siamese use, a validation pass) must not invalidate what the first one saved.
        moved = self.epoch != _ARENA_EPOCH[0]
        changed = [i for i, p in enumerate(self.params) if moved or self.versions[i] != p._version or self.ptrs[i] != p.data_ptr()]
        if force or (self.flat is not None and torch.cuda.is_current_stream_capturing()):
            stale = list(range(len(self.params)))
        else:
            stale = changed
        self.epoch = _ARENA_EPOCH[0]
        if not stale:
            return
        seen = set(changed)
        with torch.no_grad():
            torch._foreach_copy_([self.views[i] if i in seen else self.raw_views[i] for i in stale], [self.params[i].detach() for i in stale])
        for i in stale:
            self.versions[i] = self.params[i]._version
            self.ptrs[i] = self.params[i].data_ptr()


# Every forward re-casts every parameter: ONE multi-tensor launch, which is what the arena is for.  Round 4 re-cast only what version
# counters / the optimizer hook reported as changed; updates that bypass both were then served STALE 16-bit weights in training
# forwards, silently: `p.data.mul_(...)` / `p.data.copy_(...)` (EMA, weight tying, manual SGD on .data), a user's own captured graph
# that contains the optimizer step, optimizers not derived from torch.optim.Optimizer, broadcasts into .data.  PYGHO_ARENA_LAZY=1
# restores the lazy policy (then call `invalidate_cast_arenas()` after such updates).
ARENA_LAZY = os.environ.get("PYGHO_ARENA_LAZY", "0") not in ("", "0")


def ensure_cast_arena(module, dtype: Optional[torch.dtype]) -> None:
    """create (once) and refresh the cast arena of `module`'s parameters for the 16-bit compute dtype; call at the top of forward."""
    if not USE_CAST_ARENA or dtype not in (torch.bfloat16, torch.float16):
        return
    arena = module.__dict__.get("_pygho_cast_arena")
    if arena is not None and arena.dtype == dtype and arena.same_parameters():
        arena.refresh(force=not ARENA_LAZY)
        return
    params = list(module.parameters())
    if (arena is None or arena.dtype != dtype or len(arena.params) != sum(1 for p in params if p.is_cuda and p.dtype == torch.float32)
            or any(a is not b for a, b in zip(arena.params, (p for p in params if p.is_cuda and p.dtype == torch.float32)))):
        arena = ParamCastArena(params, dtype)
        module.__dict__["_pygho_cast_arena"] = arena
        arena.record_owners(module)
        if not module.__dict__.get("_pygho_cast_hook"):
            # belt and braces next to the version / address checks: a loaded state dict invalidates every copy
            module.register_load_state_dict_post_hook(lambda _m, _keys: invalidate_cast_arenas())
            module.__dict__["_pygho_cast_hook"] = True
    arena.refresh(force=not ARENA_LAZY)


def rebuild_cast_arena(module) -> None:
    """forget `module`'s cast arena (the next forward builds a new one): after adding / removing submodules of a model that has run"""
    module.__dict__.pop("_pygho_cast_arena", None)


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
        ctx.set_materialize_grads(False)            # (no gradient for the copy = none for the parameter: not a zero tensor)
        # a NEW tensor object every call: autograd writes this node into the returned object, and the arena's own view object
        # would carry it -- and the AccumulateGrad node behind it, with the stream it was created on -- into the next iteration
        # (a HIP graph capture after eager steps then pulled the eager stream into the capture and crashed in hipStreamEndCapture)
        return param_as(p, dtype).detach()

    @staticmethod
    def backward(ctx, g):
        return (None if g is None else g.to(ctx.src_dtype)), None


def cast_param(p: Tensor, dtype: torch.dtype) -> Tensor:
    """differentiable `p.to(dtype)` that reads the cast arena (the gradient returns in p's dtype)."""
    if p.dtype == dtype:
        return p
    return _CastParam.apply(p, dtype)


# --------------------------------------------------------------------------
# one tuple-wise block: Linear -> BatchNorm -> act [-> message passing [+ residual]]   
# --------------------------------------------------------------------------
_BMM_OUT_DTYPE = [None]


def _bmm_f32(a: Tensor, b: Tensor) -> Tensor:
    """batched product with f32 OUTPUT where the library offers it (the slabs of a split-K weight gradient are summed afterwards: 256
    partial products rounded to bf16 first cost 3 decimal digits), else in the operands' dtype"""
    if a.dtype in (torch.bfloat16, torch.float16) and _BMM_OUT_DTYPE[0] is not False:
        try:
            out = torch.bmm(a, b, out_dtype=torch.float32)
            _BMM_OUT_DTYPE[0] = True
            return out
        except (TypeError, RuntimeError, NotImplementedError):
            if _BMM_OUT_DTYPE[0]:
                raise
            _BMM_OUT_DTYPE[0] = False
    return torch.bmm(a, b)


def weight_grad_splitk(g: Tensor, x: Tensor, out_dtype: torch.dtype, want_colsum: bool = False, any_height: bool = False):
    """dW = g^T x for tall (nnz ~ 10^5..10^6) operands; with `want_colsum` returns (dW, g.sum(0)).
    Square 16-bit Linears of width 64 / 128 run on the transpose-read MFMA kernel (`pygho_weight_grad`); the rest falls back
    to the library: a plain GEMM below 2^19 rows, a batched split-K product above (the BLAS heuristics pick no split-K for a
    128 x 128 output, 2.9 ms at 1.8 M rows, but the batched call costs ~3.7 ms of host time, so it only pays for huge m)."""
    m, n, k = g.shape[0], g.shape[1], x.shape[1]
    cs = None
    m_dev = dyn_rows(m)
    if m_dev is not None:
        any_height = True                      # the library GEMMs below would sum the pad rows of a batch slot
    kernel_dtype = g.is_cuda and g.dtype in (torch.bfloat16, torch.float16) and x.dtype == g.dtype
    nb = 128 if (n % 128 == 0 and k % 128 == 0) else (64 if (n % 64 == 0 and k % 64 == 0) else 0)     # block width of the kernel
    if (kernel_dtype and nb and n // nb <= 2 and k // nb <= 8 and (n == nb or m_dev is not None or FORCE_DW_BLOCKS)
            and (m >= 8192 or (any_height and m > 0))):      # (`any_height`: short inputs too -- launch-bound callers)
        # `pygho_weight_grad` forms one nb x nb block of dW per launch from an nb-wide column block of g and one of x: square Linears of
        # width 64 / 128 are one block, in_features = j nb are j blocks side by side.  A g wider than one block (width 256 = 2 x 2
        # blocks) reads each operand half twice: 1.11 ms against 0.57 ms for the library's batched split-K at 2.4 M rows
        # (tools/rb256_bench.py), so that form only runs where the library cannot -- rows counted on the device (a batch slot)
        g, x = g.contiguous(), x.contiguous()
        dev = g.device
        nblk = int(lib().pygho_bn_bwd_linear_dw_blocks(m))
        rows = []
        for i in range(n // nb):                                          # row block of dW = column block of g
            parts = []
            for j in range(k // nb):                                      # column block of dW = column block of x
                cs_here = want_colsum and j == 0
                width = nb * nb + (2 * nb if cs_here else 0)              # one interleaved workspace, one folding launch
                ws = torch.empty((nblk, width), dtype=torch.float32, device=dev)
                cws_ptr = c_void_p(ws.data_ptr() + 4 * nb * nb) if cs_here else None
                check(lib().pygho_weight_grad_strided(ptr(ws), cws_ptr, c_void_p(g.data_ptr() + i * nb * g.element_size()), n,
                                                      c_void_p(x.data_ptr() + j * nb * x.element_size()), k, m, ptr(m_dev), nb,
                                                      dtype_code(g), width, stream_ptr(dev)), "weight_grad")
                tot = sum_blocks(ws)
                parts.append(tot[:nb * nb].reshape(nb, nb))
                if cs_here:
                    cs = tot[nb * nb:nb * nb + nb] if cs is None else torch.cat((cs, tot[nb * nb:nb * nb + nb]))
            rows.append(parts[0] if len(parts) == 1 else torch.cat(parts, dim=1))
        gw = (rows[0] if len(rows) == 1 else torch.cat(rows, dim=0)).to(out_dtype)
    else:
        require_static_rows(m, "the library weight-gradient GEMM (widths other than 64 / 128, f32)")
        slabs = min(256, m // 2048)
        if slabs < 4 or m < (1 << 19):
            gw = (g.t() @ x).to(out_dtype)
        else:
            rows = m // slabs
            main = rows * slabs
            part = _bmm_f32(g[:main].view(slabs, rows, n).transpose(1, 2), x[:main].view(slabs, rows, k))
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


USE_ROWBLOCK_256 = os.environ.get("PYGHO_ROWBLOCK_256", "1") != "0"     # width 256 on the column-split streaming kernels (A/B switch)
FORCE_DW_BLOCKS = False      # tests / tools: the blocked weight-gradient kernel also where the library would be faster


def rowblock_linear_supported(x: Tensor, out_features: int) -> bool:
    return (USE_ROWBLOCK_LINEAR and x.is_cuda and x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16) and x.shape[1] == out_features
            and (out_features in (64, 128) or (out_features == 256 and USE_ROWBLOCK_256)) and x.shape[0] >= 8192)


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
        nblk = int(lib().pygho_rowblock_linear_slots(m, d))
        ws = torch.empty((nblk, 2, d), dtype=torch.float32, device=dev)
    if addend is not None:
        addend = addend.contiguous()
    if stats_shift is True:
        shift = torch.empty(d, dtype=torch.float32, device=dev)
        m_dev = dyn_rows(m)
        if m_dev is not None:
            check(lib().pygho_rowblock_linear_autoshift_dyn(ptr(out), ptr(x), ptr(wl), ptr(bias), ptr(addend), ptr(ws), ptr(shift), m,
                                                            ptr(m_dev), d, dtype_code(x), stream_ptr(dev)), "rowblock_linear_dyn")
        else:
            check(lib().pygho_rowblock_linear_autoshift(ptr(out), ptr(x), ptr(wl), ptr(bias), ptr(addend), ptr(ws), ptr(shift), m, d,
                                                        dtype_code(x), stream_ptr(dev)), "rowblock_linear")
        return out, (ws, shift)
    if stats_shift is not None:
        require_static_rows(m, "rowblock_linear with a caller-given statistics shift")
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
    nblk = int(lib().pygho_rowblock_linear_slots(m, d))
    ws = torch.empty((nblk, 2, d), dtype=torch.float32, device=dev)
    m_dev = dyn_rows(m)
    if m_dev is not None:
        check(lib().pygho_rowblock_linear_bwd_sums_dyn(ptr(s1), ptr(s2), ptr(x), ptr(wl.contiguous()), ptr(bias), ptr(gh.contiguous()),
                                                       ptr(mean), ptr(invstd), ptr(w32), ptr(b32), m, ptr(m_dev), d, ACT_CODE[act], ptr(ws),
                                                       dtype_code(x), stream_ptr(dev)), "rowblock_linear_bwd_sums_dyn")
        return s1, s2
    check(lib().pygho_rowblock_linear_bwd_sums(ptr(s1), ptr(s2), ptr(x), ptr(wl.contiguous()), ptr(bias), ptr(gh.contiguous()), ptr(mean),
                                               ptr(invstd), ptr(w32), ptr(b32), m, d, ACT_CODE[act], ptr(ws), dtype_code(x),
                                               stream_ptr(dev)), "rowblock_linear_bwd_sums")
    return s1, s2


def rowblock_linear_bwd_apply(x: Tensor, wl: Tensor, bias: Optional[Tensor], gh: Tensor, saved, sums, act: str, training: bool,
                              want_colsum: bool):
    """(gpre, column sums of gpre or None): the BatchNorm + activation backward of pre = x @ wl^T + bias, recomputed (same bits as the
    forward's), given the two channel sums -- the apply half of `_bn_backward` without a stored pre-activation.  Width 256 builds its
    backward from this pass (`bn_bwd_linear`)."""
    mean, invstd, w32, b32, _ws = saved
    s1, s2 = sums
    dev = require_device(x, wl, bias, gh)
    m, d = x.shape
    gpre = torch.empty_like(x)
    cs = ws = None
    if want_colsum:
        nblk = int(lib().pygho_rowblock_linear_slots(m, d))
        cs = torch.empty(d, dtype=torch.float32, device=dev)
        ws = torch.empty(nblk * 2 * d + d, dtype=torch.float32, device=dev)
    check(lib().pygho_rowblock_linear_bwd_apply(ptr(gpre), ptr(cs), ptr(x), ptr(wl.contiguous()), ptr(bias), ptr(gh.contiguous()), ptr(mean),
                                                ptr(invstd), ptr(w32), ptr(b32), ptr(s1), ptr(s2), m, ptr(dyn_rows(m)), d, ACT_CODE[act],
                                                1 if training else 0, ptr(ws), dtype_code(x), stream_ptr(dev)), "rowblock_linear_bwd_apply")
    return gpre, cs


def sum_blocks(partials: Tensor, out_numel: Optional[int] = None) -> Tensor:
    """(n_blocks, ...) f32 per-workgroup partial results -> their sum over the first dim (one deterministic kernel).  With
    `out_numel` >= the slab size the result is a flat array of that length whose tail is zero (written by the same launch)."""
    dev = require_device(partials)
    assert partials.dtype == torch.float32 and partials.is_contiguous()
    if out_numel is None:
        out = torch.empty(partials.shape[1:], dtype=torch.float32, device=dev)
        check(lib().pygho_sum_blocks(ptr(out), ptr(partials), partials.shape[0], out.numel(), stream_ptr(dev)), "sum_blocks")
        return out
    n = int(torch.Size(partials.shape[1:]).numel())
    assert out_numel >= n
    out = torch.empty((out_numel,), dtype=torch.float32, device=dev)
    check(lib().pygho_sum_blocks_pad(ptr(out), ptr(partials), partials.shape[0], n, out_numel, stream_ptr(dev)), "sum_blocks_pad")
    return out


def bn_bwd_sums(pre: Tensor, gh: Tensor, saved, act: str):
    """the two channel sums of the BatchNorm + activation backward (sum dy, sum dy * xhat), two-stage and deterministic."""
    mean, invstd, w32, b32, ws = saved
    m, c = pre.shape
    dev = pre.device
    s1 = torch.empty(c, dtype=torch.float32, device=dev)
    s2 = torch.empty(c, dtype=torch.float32, device=dev)
    m_dev = dyn_rows(m)
    if m_dev is not None:
        check(lib().pygho_bn_act_bwd_sums_dyn(ptr(s1), ptr(s2), ptr(pre), ptr(gh), ptr(mean), ptr(invstd), ptr(w32), ptr(b32), m,
                                              ptr(m_dev), c, ACT_CODE[act], ptr(ws), dtype_code(pre), stream_ptr(dev)), "bn_act_bwd_sums_dyn")
        return s1, s2
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
        if c == 256:
            # no one-workgroup backward at this width (W, W^T and a 256 x 256 f32 accumulator do not fit a CU): gpre from a third
            # recomputing pass, gx = gpre W + addend on the forward kernel (the residual gradient rides in its epilogue), dW as 2 x 2
            # blocks of the weight-gradient kernel
            gpre, cs = rowblock_linear_bwd_apply(x, w, lin_bias, gh, saved, sums, act, training, want_colsum)
            gx, _ = rowblock_linear(gpre, w.t().contiguous(), None, addend=None if addend is None else addend.contiguous())
            gw = weight_grad_splitk(gpre, x, torch.float32)
            return gx, gw, s1, s2, cs
        gx = torch.empty_like(x)
        addend = None if addend is None else addend.contiguous()
        nblk = int(lib().pygho_bn_bwd_linear_dw_blocks(m))
        width = c * c + (2 * c if want_colsum else 0)
        ws = torch.empty((nblk, width), dtype=torch.float32, device=dev)
        cws_ptr = c_void_p(ws.data_ptr() + 4 * c * c) if want_colsum else None
        m_dev = dyn_rows(m)
        if m_dev is not None:
            check(lib().pygho_bn_bwd_linear_dw_recompute_dyn(ptr(gx), ptr(ws), ptr(gh), ptr(x), ptr(w.contiguous()), ptr(lin_bias), ptr(addend),
                                                             cws_ptr, ptr(mean), ptr(invstd), ptr(w32), ptr(b32), ptr(s1), ptr(s2), m,
                                                             ptr(m_dev), c, ACT_CODE[act], 1 if training else 0, dtype_code(x), width,
                                                             stream_ptr(dev)), "bn_bwd_linear_dw_recompute_dyn")
        else:
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
        m_dev = dyn_rows(m)
        if m_dev is not None:
            check(lib().pygho_bn_bwd_linear_dw_dyn(ptr(gx), ptr(ws), ptr(pre), ptr(gh), ptr(x), ptr(wl), ptr(addend), cws_ptr, ptr(mean),
                                                   ptr(invstd), ptr(w32), ptr(b32), ptr(s1), ptr(s2), m, ptr(m_dev), c, ACT_CODE[act],
                                                   1 if training else 0, dt, width, st), "bn_bwd_linear_dw_dyn")
        else:
            check(lib().pygho_bn_bwd_linear_dw(ptr(gx), ptr(ws), ptr(pre), ptr(gh), ptr(x), ptr(wl), ptr(addend), cws_ptr, ptr(mean),
                                               ptr(invstd), ptr(w32), ptr(b32), ptr(s1), ptr(s2), m, c, ACT_CODE[act],
                                               1 if training else 0, dt, width, st), "bn_bwd_linear_dw")
        tot = sum_blocks(ws)
        return gx, tot[:c * c].reshape(c, c), s1, s2, (tot[c * c:c * c + c] if want_colsum else None)
    require_static_rows(m, "the two-kernel BatchNorm backward + input-gradient GEMM (PYGHO fused weight gradient switched off)")
    cws = torch.empty((nblk, 2, c), dtype=torch.float32, device=dev) if want_colsum else None
    second = torch.empty_like(pre)
    check(lib().pygho_bn_bwd_linear(ptr(gx), ptr(second), ptr(pre), ptr(gh), ptr(wl), ptr(addend), ptr(cws), ptr(mean), ptr(invstd),
                                    ptr(w32), ptr(b32), ptr(s1), ptr(s2), m, c, ACT_CODE[act], 1 if training else 0, dt, st),
          "bn_bwd_linear")
    sdx = sum_blocks(cws)[0] if cws is not None else None
    return gx, second, s1, s2, sdx


class record_block_paths:
    """``with record_block_paths() as log:`` -- every tuple block that runs inside appends what it dispatched and, where a faster
    path was not taken, which of its conditions failed ("nothing reports which path ran except LaunchTimer"):
    {"forward": "seg_fused" | "rowblock_linear + seg_gmr" | ..., "forward_not_fused_because": [...]} at forward time and
    {"backward": "seg_dual" | "seg_gmr + by_edge", "backward_not_dual_because": [...]} at backward time."""
    active = None

    def __enter__(self):
        self.log, self.prev = [], record_block_paths.active
        record_block_paths.active = self.log
        return self.log

    def __exit__(self, *exc):
        record_block_paths.active = self.prev


def _why_not_fused(recompute, plan, rhs, rhs_lookup, aggr, x):
    why = []
    if not FUSED_FWD:
        why.append("PYGHO_FUSED_FWD=0")
    if not recompute:
        why.append("the block keeps its pre-activation (fewer than 8192 rows, width other than 64 / 128, no weight gradient wanted, or a switch is off)")
    if plan is None or rhs is None:
        why.append("no aggregation in this block")
        return why
    if rhs_lookup is None:
        why.append("the second operand is not known to be an embedding lookup (pass adj_lookup=(table, index) to forward_residual)")
    elif rhs_lookup[0].shape[0] > 32 or rhs_lookup[0].dtype != x.dtype:
        why.append("the lookup table has more than 32 rows or another dtype than the rows")
    if aggr not in ("sum", "mean"):
        why.append(f"aggregation {aggr!r}")
    if x.shape[1] != 128 or x.dtype not in (torch.bfloat16, torch.float16):
        why.append("rows are not 128 x 16-bit")
    if plan is not None and fused_plan(plan, on_demand=True) is None:
        why.append("the plan has no fused chunks (not planned: DeviceGraphStore.collate / SpModel.prepare / _ops.fused_plan; or a row outside the limits)")
    return why


def _why_not_dual(plan, g, h, table, scale, look, affine, wants_rhs):
    why = []
    if not wants_rhs:
        return ["the second operand needs no gradient"]
    if not DUAL_BWD:
        why.append("PYGHO_DUAL_BWD=0")
    if look is None or table is None:
        why.append("the second operand is not an embedding lookup")
    if affine is not None or h is None:
        why.append("the activated rows were not stored (act-on-load path)")
    if scale is not None:
        why.append("mean aggregation")
    if plan.m < SEG_SCATTER_MIN_MESSAGES:
        why.append(f"fewer than {SEG_SCATTER_MIN_MESSAGES} messages")
    sp = scatter_plan(plan, on_demand=True)
    if sp is None:
        why.append("no scatter plan (not planned, or blocks outside the limits)")
    elif sp.cgap is None:
        why.append("the scatter plan's chunks are not aligned (a group of messages outside the chunk limits)")
    elif sp.max_edges > 96:
        why.append("more than 96 second-operand rows in one block")
    if g.dtype not in (torch.bfloat16, torch.float16):
        why.append("rows are not 16-bit")
    return why


class _TupleBlock(torch.autograd.Function):
    """H = act(bn(x W^T + b));  out = H                                  (plan is None)
                                  out = [x +] (+)_{(a,c,d)} H[c] * rhs[d]   (plan given; `residual` adds x)
    One autograd node for the whole block so that (i) the Linear's bias gradient comes out of the BatchNorm
    backward pass, (ii) the residual add runs in the aggregation epilogue and (iii) the residual gradient is
    added in place into the fresh input-gradient GEMM output."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, training, eps, act, rhs, plan, aggr, residual,
                fold_momentum=None, rhs_lookup=None, chain=False, chain_x=False, table_master=None):
        # `table_master`: the parameter behind the lookup table of `rhs_lookup` (rhs == cast(table_master)[index]).  When it is given
        # and asks for a gradient, backward may hand the TABLE's gradient to it directly (the fused backward's table-gradient form)
        # instead of the per-edge gradient of `rhs`, which then receives none from this block -- the same total for the leaf
        ctx.has_master = table_master is not None
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
        # (width 256: the pre-activation IS kept -- every recomputing pass pays the product again, ~0.45 ms of LDS-bound MFMA work at
        # 2.4 M rows, which only the one-workgroup backward of widths 64 / 128 earns back; measured in tools/rb256_bench.py)
        recompute = (USE_RECOMPUTE_PRE and skinny and USE_BN_BWD_LINEAR and USE_FUSED_DW and x.shape[1] <= 128
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
        # the whole block forward in ONE launch after the statistics pass (csrc/seg_fused.hip): the product is formed again per chunk of
        # output rows and activated into LDS, where the chunk's messages read it -- H is stored only for the by-edge gradient
        fp = None
        if (recompute and FUSED_FWD and plan is not None and rhs is not None and rhs_lookup is not None and aggr in ("sum", "mean")
                and x.shape[1] == 128 and x.dtype in (torch.bfloat16, torch.float16) and rhs_lookup[0].shape[0] <= 32
                and rhs_lookup[0].dtype == x.dtype and x.shape[0] * 256 < (1 << 31) and plan.m * 4 < (1 << 31)):
            fp = fused_plan(plan, on_demand=True)
        if record_block_paths.active is not None:
            record_block_paths.active.append(
                {"forward": "seg_fused"} if fp is not None else
                {"forward": ("rowblock_linear" if skinny else "library GEMM") + (" + seg_gmr" if plan is not None else ""),
                 "forward_not_fused_because": _why_not_fused(recompute, plan, rhs, rhs_lookup, aggr, x)})
        if fp is not None:
            (scale, shift), mean, var, saved = _bn_forward(None, gamma, beta, running_mean, running_var, training, eps, act, fold_momentum,
                                                           partial, apply=False, producer=(x, wc, bc))
            look = (rhs_lookup[0].detach(),) + plan.lookup(rhs_lookup[1])
            out, h = fused_forward(x, wc, bc, scale, shift, act, look[0], look[1], plan, fp, aggr, residual, bool(needs[10]))
            ctx.affine, ctx.look = None, look
            ctx.look_index = rhs_lookup[1]
            ctx.save_for_backward(x, wc, None, h, rhs, bc, *saved)
            ctx.meta = (training, act, None if b is None else b.dtype, gamma is not None, beta is not None, plan, aggr, residual, w.dtype,
                        skinny)
            ctx.mark_non_differentiable(mean, var)
            ctx.set_materialize_grads(False)
            ctx.chain = (bool(chain), bool(chain_x))
            extra = ()
            if chain:
                extra += (rhs.view_as(rhs),)
            if chain_x:
                extra += (x.view_as(x),)
            return (out, mean, var) + extra
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
        ctx.look_index = rhs_lookup[1] if rhs_lookup is not None else None
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
        g_rhs = g_master = None
        gh = g
        if plan is not None:
            scale = plan.fwd.inv_count if aggr == "mean" else None
            p, a_g, d_g = plan.by_c()
            rhs_read = rhs
            if ctx.look is not None:
                rhs_read, d_g = ctx.look[0], ctx.look[2]
            use_tg = (rhs is not None and ctx.needs_input_grad[10] and ctx.look is not None and ctx.affine is None and ctx.has_master
                      and ctx.needs_input_grad[18] and dual_tg_eligible(plan, g, h, rhs_read, scale, getattr(ctx, "look_index", None)))
            use_dual = (not use_tg and rhs is not None and ctx.needs_input_grad[10] and ctx.look is not None and ctx.affine is None
                        and dual_eligible(plan, g, h, rhs_read, scale))
            if record_block_paths.active is not None and rhs is not None:
                record_block_paths.active.append(
                    {"backward": "seg_dual (table gradient)"} if use_tg else {"backward": "seg_dual"} if use_dual else
                    {"backward": "seg_gmr + by_edge_product",
                     "backward_not_dual_because": _why_not_dual(plan, g, h, rhs_read if ctx.look is not None else None, scale, ctx.look,
                                                                ctx.affine, bool(ctx.needs_input_grad[10]))})
            if use_tg:
                # the same pass with the TABLE's gradient accumulated inside it: the per-edge gradient is never formed (what arrived
                # for `rhs` from later blocks is passed on as it is)
                gh, g_master = dual_backward_tg(plan, g, h, rhs_read, ctx.look[1], d_g)
                g_rhs = g_chain
            elif use_dual:
                # both gradients of the aggregation from ONE pass over the forward message order (csrc/seg_dual.hip): g and H rows are
                # fetched once for the by-tuple sum gh and the by-edge sum g_rhs -- the bits of the two launches below
                gh, g_rhs = dual_backward(plan, g, h, rhs_read, d_g, addend=g_chain)
            else:
                gh = seg_gmr(plan.n_lhs, g, rhs_read, p.seg_ptr, a_g, d_g if rhs is not None else None, "sum", scale)
            if g_rhs is not None or use_tg:
                pass
            elif rhs is not None and ctx.needs_input_grad[10]:
                p, a_g, c_g = plan.by_d()
                if ctx.affine is not None:
                    g_rhs = seg_gmr(plan.n_rhs, g, pre, p.seg_ptr, a_g, c_g, "sum", scale,
                                    act=(ctx.affine[0], ctx.affine[1], act, 2))
                    if g_chain is not None:
                        g_rhs = g_rhs + g_chain
                else:
                    g_rhs = by_edge_product(plan, g, h, scale, addend=g_chain)
            elif g_chain is not None:
                g_rhs = g_chain
        elif rhs is not None and ctx.needs_input_grad[10]:
            g_rhs = g if g_chain is None else g + g_chain      # residual row operand: receives the output gradient as it is
        want_cs = b_dtype is not None and ctx.needs_input_grad[2]
        gx = gw = gb = None
        one_wg = skinny and x.shape[1] <= 128           # the one-workgroup backward kernels exist for widths 64 / 128 (see bn_bwd_linear)
        if pre is None or (one_wg and USE_BN_BWD_LINEAR and USE_FUSED_DW and ctx.needs_input_grad[1]):
            gx, gw32, s1, s2, sdx = bn_bwd_linear(pre, gh.contiguous(), saved, training, act, w, res_g, want_cs, x=x,
                                                  lin_bias=bc)
            gw = gw32.to(w_dtype)
        elif one_wg and USE_BN_BWD_LINEAR:
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
                g_rhs, None, None, None, None, None, None, None, g_master)


def tuple_block(x: Tensor, lin: "torch.nn.Linear", bn: "torch.nn.BatchNorm1d", act: str, rhs: Optional[Tensor] = None,
                plan: Optional[MessagePlan] = None, aggr: str = "sum", residual: bool = False,
                rhs_lookup: Optional[Tuple[Tensor, Tensor]] = None, chain: bool = False, chain_x: bool = False,
                rhs_master: Optional[Tensor] = None):
    """fused Linear -> BatchNorm1d -> act (-> aggregation over `plan` with `rhs` (-> + x)); parameters are read
    from the stock modules (f32 master weights are cast to the activation dtype like autocast would)."""
    training = bn.training or bn.running_mean is None
    if residual:
        assert plan is not None and plan.n_out == x.shape[0] and lin.out_features == x.shape[1]
    if plan is None and rhs is not None:                # residual row operand (see _TupleBlock.forward)
        assert rhs.shape == (x.shape[0], lin.out_features) and rhs.dtype == x.dtype and rhs_lookup is None
    fold = _fold_momentum(bn)
    if rhs_master is not None and (rhs_lookup is None or not rhs_master.requires_grad or tuple(rhs_master.shape) != tuple(rhs_lookup[0].shape)):
        rhs_master = None
    res = _TupleBlock.apply(x, lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                            bn.eps, act, rhs, plan, aggr, residual, fold, rhs_lookup, chain, chain_x, rhs_master)
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
            ga = by_edge_product(plan1, g1, x, sc1)
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
            and lin.in_features == len(xs) * d and lin.out_features == d and d <= 128 and rowblock_linear_supported(xs[0], d)
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

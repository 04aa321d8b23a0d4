"""
MLP / normalisation helpers of the model layers.  Mirror of ``pygho/honn/utils.py`` (reference
utils.py:46-142): these are dense ``torch.nn`` stacks (rocBLAS / hipBLASLt GEMMs through PyTorch-ROCm)
and are not part of the hand-written hot path; module / parameter names match the reference so that
``state_dict``s are interchangeable.
"""
import os
from typing import Callable, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import _ops


USE_SKINNY_LINEAR = os.environ.get("PYGHO_SKINNY_LINEAR", "1") != "0"  # square 16-bit Linears of width 64 / 128 on the row-block kernel
USE_ARENA_LINEAR = os.environ.get("PYGHO_ARENA_LINEAR", "1") != "0"    # short 16-bit inputs through _ArenaLinearFn as well
_MM_OUT_DTYPE = [None]       # does torch.mm take out_dtype here (an f32 weight gradient straight from 16-bit operands)?


def _small_weight_grad(g: Tensor, x: Tensor, dtype: torch.dtype) -> Tensor:
    """dW = g^T x for short operands, in `dtype` without a separate cast launch where the library allows it"""
    if dtype == torch.float32 and g.dtype in (torch.bfloat16, torch.float16) and _MM_OUT_DTYPE[0] is not False:
        try:
            gw = torch.mm(g.t(), x, out_dtype=torch.float32)
            _MM_OUT_DTYPE[0] = True
            return gw
        except (TypeError, RuntimeError):
            if _MM_OUT_DTYPE[0]:
                raise
            _MM_OUT_DTYPE[0] = False
    return (g.t() @ x).to(dtype)


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b with the weight gradient computed as a batched split-K product (operands as given: the callers in honn/Conv.py
    hand in weights they have cast / transposed themselves).

    The tuple-wise MLPs see (nnz ~ 10^6, d = 128) activations, so dW = g^T x is a (d x nnz) @ (nnz x d)
    GEMM whose reduction dim is the long one (``_ops.weight_grad_splitk``)."""

    @staticmethod
    def forward(ctx, x, w, b, any_height=False):
        # `any_height`: the weight-gradient KERNEL for short inputs too (node-level rows): its sum over the rows is cut into 64-row tiles
        # folded in tile order, the same for an exactly sized batch and for a batch slot's capacity rows -- a library GEMM / `sum(0)` picks
        # its split by the row count, so the two differ in the last bits
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.any_height = bool(any_height)
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = gb = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:
                gw, gb = _ops.weight_grad_splitk(g, x, w.dtype, want_colsum=True, any_height=ctx.any_height)
                gb = gb.to(g.dtype)
            else:
                gw = _ops.weight_grad_splitk(g, x, w.dtype, any_height=ctx.any_height)
        elif want_b:
            gb = g.sum(0)
        return gx, gw, gb, None


class _ArenaLinearFn(torch.autograd.Function):
    """y = x W^T + b in the compute dtype `dt`, taking the MASTER parameters: their 16-bit copies come from the cast arena
    (``_ops.param_as``: one multi-tensor cast per optimizer step, no autograd node), and the gradients go back in the masters' own
    dtype straight from f32 accumulators -- no cast launch per parameter and direction (autocast's ``nn.Linear`` issues four).

    Tall inputs: the tuple-wise MLPs see (nnz ~ 10^6, d = 128) activations, so dW = g^T x is a (d x nnz) @ (nnz x d) GEMM whose
    reduction dim is the long one (``_ops.weight_grad_splitk``)."""

    @staticmethod
    def forward(ctx, x, w, b, dt):
        wq = _ops.param_as(w, dt)
        ctx.save_for_backward(x, wq)
        ctx.grad_dtypes = (w.dtype, None if b is None else b.dtype)
        bq = None if b is None else _ops.param_as(b, dt)
        # square 16-bit maps of width 64 / 128 (the node-level Linears of example/minimal.py:47-60): the streaming row-block kernel
        # -- at the reference's batch size a library GEMM call costs ~50 us of HOST time, this launch ~8
        ctx.skinny = (USE_SKINNY_LINEAR and x.dtype == dt and dt in (torch.bfloat16, torch.float16) and x.shape[0] > 0
                      and tuple(wq.shape) in ((64, 64), (128, 128)) and x.shape[1] == wq.shape[1] and wq.is_contiguous()
                      and x.data_ptr() % 16 == 0 and wq.data_ptr() % 16 == 0)
        if ctx.skinny:
            return _ops.rowblock_linear(x, wq, bq)[0]
        return torch.nn.functional.linear(x, wq, bq)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        w_dtype, b_dtype = ctx.grad_dtypes
        g = g.contiguous()
        want_b = b_dtype is not None and ctx.needs_input_grad[2]
        if ctx.skinny and g.dtype == w.dtype and g.data_ptr() % 16 == 0:
            gx = _ops.rowblock_linear(g, w.t().contiguous())[0] if ctx.needs_input_grad[0] else None
            gw = gb = None
            if ctx.needs_input_grad[1]:
                gw, gb = _ops.weight_grad_splitk(g, x, w_dtype, want_colsum=True, any_height=True)
                gb = gb.to(b_dtype) if want_b else None
            elif want_b:
                _ops.require_static_rows(x.shape[0], "the bias gradient of a Linear whose weight needs no gradient")
                gb = g.sum(0, dtype=b_dtype)
            return gx, gw, gb, None
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or want_b:
            _ops.require_static_rows(x.shape[0], "the library-GEMM backward of a Linear (not a square 16-bit map of width 64 / 128)")
        if ctx.needs_input_grad[1]:
            if x.shape[0] < 8192:
                gw = _small_weight_grad(g, x, w_dtype)
            elif want_b:
                gw, gb = _ops.weight_grad_splitk(g, x, w_dtype, want_colsum=True)        # the column sums arrive in f32
                gb = gb.to(b_dtype)
            else:
                gw = _ops.weight_grad_splitk(g, x, w_dtype)
        if want_b and gb is None:
            gb = g.sum(0, dtype=b_dtype)
        return gx, gw, gb, None


class Linear(nn.Linear):
    """``nn.Linear`` (same parameters / state_dict).  2-D device inputs under autograd go through ``_ArenaLinearFn``: tall ones
    for the split-K weight gradient, 16-bit ones of any height for the cast arena (at the reference's batch size the step is
    bound by launches, and autocast's per-use parameter casts were 20 of its 163); anything else takes the stock path."""

    def forward(self, x: Tensor) -> Tensor:
        if x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and not self.__dict__.get("_pygho_stock", False):
            dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
            if x.shape[0] >= 8192 or (USE_ARENA_LINEAR and dt in (torch.bfloat16, torch.float16) and self.weight.dtype == torch.float32):
                xx = x if x.dtype == dt else x.to(dt)
                with torch.autocast("cuda", enabled=False):
                    return _ArenaLinearFn.apply(xx.contiguous(), self.weight, self.bias, dt)
        return super().forward(x)


class NormMomentumScheduler:
    """scales the momentum of every norm layer by ``mfunc(epoch)`` (reference utils.py:12-33)."""

    def __init__(self, mfunc: Callable, initmomentum: float, normtype=nn.BatchNorm1d) -> None:
        self.normtype = normtype
        self.mfunc = mfunc
        self.epoch = 0
        self.initmomentum = initmomentum

    def step(self, model: nn.Module):
        ratio = self.mfunc(self.epoch)
        if 1 - 1e-6 < ratio < 1 + 1e-6:
            return self.initmomentum
        curm = self.initmomentum * ratio
        self.epoch += 1
        for mod in model.modules():
            if type(mod) is self.normtype:
                mod.momentum = curm
        return curm


class NoneNorm(nn.Module):

    def __init__(self, dim=0, normparam=0) -> None:
        super().__init__()
        self.num_features = dim

    def forward(self, x):
        return x


class BatchNorm(nn.Module):
    """BatchNorm1d over the last dim, leading dims flattened (reference utils.py:46-61)."""

    def __init__(self, dim, normparam=0.1) -> None:
        super().__init__()
        self.num_features = dim
        self.norm = nn.BatchNorm1d(dim, momentum=normparam)

    def forward(self, x: Tensor):
        if x.dim() == 2:
            return self.norm(x)
        if x.dim() >= 3:
            return self.norm(x.flatten(0, -2)).reshape(x.shape)
        raise NotImplementedError


class LayerNorm(nn.Module):

    def __init__(self, dim, normparam=0.1) -> None:
        super().__init__()
        self.num_features = dim
        self.norm = nn.LayerNorm(dim)

    def forward(self, x: Tensor):
        return self.norm(x)


normdict = {"bn": BatchNorm, "ln": LayerNorm, "none": NoneNorm}
act_dict = {"relu": nn.ReLU(inplace=True), "ELU": nn.ELU(inplace=True), "silu": nn.SiLU(inplace=True)}


class MLP(nn.Module):
    """``numlayer`` x [Linear -> norm -> dropout -> act]; the last block only has norm/act when
    ``tailact`` (reference utils.py:85-142; layer order inside ``lins`` is identical)."""

    def __init__(self, hiddim: int, outdim: int, numlayer: int, tailact: bool, dp: float = 0, norm: str = "bn",
                 act: str = "relu", tailbias=True, normparam: float = 0.1) -> None:
        super().__init__()
        assert numlayer >= 0
        if numlayer == 0:
            assert hiddim == outdim
            self.lins = NoneNorm()
            return
        blocks = []
        for _ in range(numlayer - 1):
            blocks.append(Linear(hiddim, hiddim))
            blocks.append(normdict[norm](hiddim, normparam))
            if dp > 0:
                blocks.append(nn.Dropout(dp, inplace=True))
            blocks.append(act_dict[act])
        blocks.append(Linear(hiddim, outdim, bias=tailbias))
        if tailact:
            blocks.append(normdict[norm](outdim, normparam))
            if dp > 0:
                blocks.append(nn.Dropout(dp, inplace=True))
            blocks.append(act_dict[act])
        self.lins = nn.Sequential(*blocks)

    def single_block(self):
        """(Linear, BatchNorm1d, act name) when this MLP is exactly one Linear -> BatchNorm -> SiLU/ReLU block
        (the shape every shipped layer uses for its tuple-wise update), else None."""
        if not isinstance(self.lins, nn.Sequential) or len(self.lins) != 3:
            return None
        lin, norm, act = self.lins
        name = _ACT_NAMES.get(type(act))
        if isinstance(lin, nn.Linear) and type(norm) is BatchNorm and name is not None:
            return lin, norm.norm, name
        return None

    def forward(self, x: Tensor, residual: Optional[Tensor] = None, return_input: bool = False):
        """`residual` (optional, shape of the output): returns residual + MLP(x); when the MLP ends in a fused
        [Linear, BatchNorm, act] block the sum is formed inside that block's activation pass.
        `return_input`: returns (MLP(x), x') where x' holds x's values (in the block's compute dtype) and -- when the MLP starts with
        a fused block -- is an autograd OUTPUT of that block: a caller that also reads x elsewhere reads x' instead, and that
        gradient is added inside the block's input-gradient GEMM rather than by autograd's accumulation."""
        if return_input:
            self._chained_input = None
            y = self._forward(x, None, True)[0]
            xin, self._chained_input = self._chained_input, None
            return y, (x if xin is None else xin)
        if residual is not None:
            y, fused = self._forward(x, residual)
            return y if fused else residual + y
        return self._forward(x, None)[0]

    def _forward(self, x: Tensor, residual: Optional[Tensor], chain_first: bool = False):
        if not isinstance(self.lins, nn.Sequential):
            return self.lins(x), False
        # same module sequence as the reference.  On device tensors a [Linear, BatchNorm, act] run is ONE autograd
        # node (GEMM + fused BatchNorm/activation kernels; the bias gradient falls out of the BatchNorm backward
        # pass) and a [BatchNorm, act] pair is one fused kernel pair; everything else is the stock module.
        mods = list(self.lins)
        i = 0
        while i < len(mods):
            mod = mods[i]
            if isinstance(mod, nn.Linear) and i + 1 < len(mods) and type(mods[i + 1]) is BatchNorm and x.is_cuda and x.dim() >= 2:
                act = _ACT_NAMES.get(type(mods[i + 2])) if i + 2 < len(mods) else None
                x2 = _autocast_input(x.flatten(0, -2) if x.dim() > 2 else x)
                if x2.shape[0] >= 8192 and _ops.bn_act_supported_shape(x2.shape[0], mod.out_features, x2.dtype):
                    step = 3 if act is not None else 2
                    row_res = None
                    # fused only when the residual already has the block's compute dtype (an f32 residual stream stays f32)
                    if (residual is not None and i + step == len(mods) and residual.dtype == x2.dtype
                            and residual.shape == tuple(x.shape[:-1]) + (mod.out_features,)):
                        row_res = residual.reshape(-1, mod.out_features)
                    want_x = chain_first and i == 0 and row_res is None and torch.is_grad_enabled() and x2.requires_grad
                    with torch.autocast("cuda", enabled=False):
                        y = _ops.tuple_block(x2, mod, mods[i + 1].norm, act or "none", rhs=row_res, chain_x=want_x)
                    if want_x:
                        y, xin = y
                        self._chained_input = xin.reshape(x.shape)
                    x = y.reshape(tuple(x.shape[:-1]) + (mod.out_features,))
                    i += step
                    if row_res is not None:
                        return x, True
                    continue
            if type(mod) is BatchNorm and x.is_cuda and x.dim() >= 2:
                act = _ACT_NAMES.get(type(mods[i + 1])) if i + 1 < len(mods) else None
                x2 = x.flatten(0, -2) if x.dim() > 2 else x
                if _ops.bn_act_supported(x2):
                    y = _ops.batch_norm_act(x2, mod.norm, act or "none")
                    x = y.reshape(x.shape)
                    i += 2 if act is not None else 1
                    continue
            x = mod(x)
            i += 1
        return x, False


_ACT_NAMES = {nn.SiLU: "silu", nn.ReLU: "relu"}


def _autocast_input(x: Tensor) -> Tensor:
    if torch.is_autocast_enabled("cuda") and x.is_floating_point():
        dt = torch.get_autocast_dtype("cuda")
        return x if x.dtype == dt else x.to(dt)
    return x

"""
MLP / normalisation helpers of the model layers.  Mirror of ``pygho/honn/utils.py`` (reference
utils.py:46-142): these are dense ``torch.nn`` stacks (rocBLAS / hipBLASLt GEMMs through PyTorch-ROCm)
and are not part of the hand-written hot path; module / parameter names match the reference so that
``state_dict``s are interchangeable.
"""
from typing import Callable

import torch
import torch.nn as nn
from torch import Tensor

from .. import _ops


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b with the weight gradient computed as a batched split-K product.

    The tuple-wise MLPs see (nnz ~ 10^6, d = 128) activations, so dW = g^T x is a (d x nnz) @ (nnz x d)
    GEMM whose reduction dim is the long one; the BLAS heuristics pick a kernel without split-K for it
    (2.9 ms vs 0.19 ms on MI355X at nnz = 1.8 M).  Slicing nnz into S slabs turns it into S independent
    small GEMMs plus one tiny sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = g @ w
        if ctx.needs_input_grad[1]:
            m, n, k = g.shape[0], g.shape[1], x.shape[1]
            slabs = min(256, m // 2048)
            if slabs >= 4:
                rows = m // slabs
                main = rows * slabs
                part = torch.bmm(g[:main].view(slabs, rows, n).transpose(1, 2), x[:main].view(slabs, rows, k))
                gw = part.float().sum(0)
                if main < m:
                    gw = gw + (g[main:].t() @ x[main:]).float()
                gw = gw.to(w.dtype)
            else:
                gw = g.t() @ x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


class Linear(nn.Linear):
    """``nn.Linear`` (same parameters / state_dict) whose backward uses the split-K weight gradient above for
    tall 2-D inputs on the device; anything else takes the stock path."""

    def forward(self, x: Tensor) -> Tensor:
        if x.is_cuda and x.dim() == 2 and x.shape[0] >= 8192 and torch.is_grad_enabled():
            dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
            xx = x if x.dtype == dt else x.to(dt)
            b = None if self.bias is None else self.bias.to(dt)
            with torch.autocast("cuda", enabled=False):
                return _SplitKLinearFn.apply(xx.contiguous(), self.weight.to(dt), b)
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

    def forward(self, x: Tensor):
        if not isinstance(self.lins, nn.Sequential):
            return self.lins(x)
        # same module sequence as the reference; [BatchNorm, act] pairs on device tensors run as ONE fused HIP
        # kernel pair (the BatchNorm output is never materialised), everything else is the stock module
        mods = list(self.lins)
        i = 0
        while i < len(mods):
            mod = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            act = {nn.SiLU: "silu", nn.ReLU: "relu"}.get(type(nxt)) if nxt is not None else None
            if type(mod) is BatchNorm and x.is_cuda and x.dim() >= 2:
                x2 = x.flatten(0, -2) if x.dim() > 2 else x
                if _ops.bn_act_supported(x2):
                    y = _ops.batch_norm_act(x2, mod.norm, act or "none")
                    x = y.reshape(x.shape)
                    i += 2 if act is not None else 1
                    continue
            x = mod(x)
            i += 1
        return x

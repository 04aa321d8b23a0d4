"""
MLP / normalisation helpers of the model layers.  Mirror of ``pygho/honn/utils.py`` (reference
utils.py:46-142): these are dense ``torch.nn`` stacks (rocBLAS / hipBLASLt GEMMs through PyTorch-ROCm)
and are not part of the hand-written hot path; module / parameter names match the reference so that
``state_dict``s are interchangeable.
"""
from typing import Callable

import torch.nn as nn
from torch import Tensor


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
            blocks.append(nn.Linear(hiddim, hiddim))
            blocks.append(normdict[norm](hiddim, normparam))
            if dp > 0:
                blocks.append(nn.Dropout(dp, inplace=True))
            blocks.append(act_dict[act])
        blocks.append(nn.Linear(hiddim, outdim, bias=tailbias))
        if tailact:
            blocks.append(normdict[norm](outdim, normparam))
            if dp > 0:
                blocks.append(nn.Dropout(dp, inplace=True))
            blocks.append(act_dict[act])
        self.lins = nn.Sequential(*blocks)

    def forward(self, x: Tensor):
        return self.lins(x)

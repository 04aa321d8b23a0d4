"""
``MaskedTensor`` container on the MI355X backend.  Mirror of ``pygho/backend/MaTensor.py`` (file:line
cited per member): ``data`` (*maskedshape, *denseshape), bool ``mask`` (*maskedshape), ``padvalue``.

Semantics follow the reference's DOCUMENTED contract ("unused elements in data do not affect the
output", docs/BasicDataStructure.md:12): masked-out entries never contribute to reductions or
contractions and read as ``padvalue`` through ``.data``.  The reference constructor never actually fills
(``__init__`` sets padvalue before ``fill_masked_`` tests it, MaTensor.py:107-120) and its ``min`` calls
``amax`` (:203); both are deliberate deviations.  The fill is applied lazily: kernels take
(raw data, mask) and skip masked entries themselves, so no extra pass over the data is spent on it.
"""
from typing import Callable, Iterable, Union

import torch
from torch import BoolTensor, LongTensor, Tensor

from .. import _ops


def filterinf(X: Tensor, filled_value: float = 0):
    """replace +-inf by ``filled_value``.  Reference: MaTensor.py:8-31."""
    return X.masked_fill(torch.isinf(X), filled_value)


class MaskedTensor:

    def __init__(self, data: Tensor, mask: BoolTensor, padvalue: float = 0.0, is_filled: bool = False):
        # mask: True for valid value, False for invalid value   (MaTensor.py:90-111)
        assert data.ndim >= mask.ndim, "data's #dim should be larger than mask "
        assert data.shape[:mask.ndim] == mask.shape, "data and mask's first dimensions should match"
        self.__raw = data
        self.__mask = mask
        self.__masked_dim = mask.ndim
        self.__padvalue = padvalue
        self.__filled = padvalue if is_filled else None     # value the masked entries of raw are known to hold
        self.__fullnegmask = None

    # ---- fill (MaTensor.py:113-128) ---------------------------------------------
    def _is_filled_with(self, val) -> bool:
        f = self.__filled
        return f is not None and (f == val)

    def fill_masked_(self, val: float = 0.0) -> None:
        """in-place (on the container) fill of the masked-out entries."""
        self.__raw = self.fill_masked(val)
        self.__padvalue = val
        self.__filled = val

    def fill_masked(self, val: float = 0.0) -> Tensor:
        """tensor whose masked-out entries equal ``val``."""
        if self._is_filled_with(val):
            return self.__raw
        out = _ops.masked_fill(self.__raw, self.__mask, val)
        if val == self.__padvalue:          # remember the filled form, it is what `.data` exposes
            self.__raw, self.__filled = out, val
        return out

    def to(self, device, non_blocking: bool = True):
        self.__raw = self.__raw.to(device, non_blocking=non_blocking)
        self.__mask = self.__mask.to(device, non_blocking=non_blocking)
        self.__fullnegmask = None
        return self

    # ---- properties (MaTensor.py:139-173) -----------------------------------------
    @property
    def padvalue(self) -> float:
        return self.__padvalue

    @property
    def data(self) -> Tensor:
        return self.fill_masked(self.__padvalue)

    @property
    def raw(self) -> Tensor:
        """data without the fill guarantee (what the kernels consume together with `mask`)."""
        return self.__raw

    @property
    def mask(self) -> BoolTensor:
        return self.__mask

    @property
    def fullnegmask(self) -> BoolTensor:
        if self.__fullnegmask is None:
            m = self.__mask
            for _ in range(self.dense_dim):
                m = m.unsqueeze(-1)
            self.__fullnegmask = torch.logical_not(m)
        return self.__fullnegmask

    @property
    def shape(self) -> torch.Size:
        return self.__raw.shape

    @property
    def masked_dim(self):
        return self.__masked_dim

    @property
    def dense_dim(self):
        return len(self.denseshape)

    @property
    def maskedshape(self):
        return self.shape[:self.masked_dim]

    @property
    def denseshape(self):
        return self.shape[self.masked_dim:]

    # ---- reductions over masked dims (MaTensor.py:175-206) -----------------------
    def _reduce(self, dims, keepdim: bool, op: str):
        if isinstance(dims, int):
            dims = [dims]
        dims = sorted(set(int(x) for x in dims), reverse=True)
        assert all(0 <= x < self.masked_dim for x in dims), "can only reduce masked dims"
        data, mask = self.__raw, self.__mask
        if op == "mean":
            cnt = mask
            for x in dims:
                cnt = cnt.sum(dim=x)
            cnt = cnt.clamp_min(1)
        for x in dims:                       # one masked dim per launch, highest first
            data, mask = _ops.masked_reduce(data, mask, x, "sum" if op == "mean" else op)
        if op == "mean":
            for _ in range(data.dim() - cnt.dim()):
                cnt = cnt.unsqueeze(-1)
            data = data / cnt.to(data.dtype)
        if keepdim:
            for x in sorted(dims):
                data, mask = data.unsqueeze(x), mask.unsqueeze(x)
        return MaskedTensor(data, mask, padvalue=0, is_filled=True)

    def sum(self, dims: Union[Iterable[int], int], keepdim: bool = False):
        return self._reduce(dims, keepdim, "sum")

    def mean(self, dims: Union[Iterable[int], int], keepdim: bool = False):
        return self._reduce(dims, keepdim, "mean")

    def max(self, dims: Union[Iterable[int], int], keepdim: bool = False):
        return self._reduce(dims, keepdim, "max")

    def min(self, dims: Union[Iterable[int], int], keepdim: bool = False):
        return self._reduce(dims, keepdim, "min")

    # ---- diag / unpooling (MaTensor.py:208-234) -------------------------------------
    def diag(self, dims: Iterable[int]):
        """diagonal over masked dims `dims`, placed at dims[0]."""
        assert len(dims) >= 2, "must diag several dims"
        dims = sorted(list(dims))
        tdata, tmask = self.__raw, self.__mask
        tdata = torch.diagonal(tdata, 0, dims[0], dims[1])
        tmask = torch.diagonal(tmask, 0, dims[0], dims[1])
        for i in range(2, len(dims)):
            # the reference's loop (MaTensor.py:218-220) keeps the ORIGINAL dim numbers here and raises for every input
            # ("diagonal dimensions cannot be identical" / "Dimension out of range"): i smaller dims are gone by now, the running
            # diagonal sits last -- the documented intent, x[.., k, .., k, .., k, ..] at dims[0] (deviations)
            tdata = torch.diagonal(tdata, 0, dims[i] - i, -1)
            tmask = torch.diagonal(tmask, 0, dims[i] - i, -1)
        tdata = torch.movedim(tdata, -1, dims[0])
        tmask = torch.movedim(tmask, -1, dims[0])
        filled = self.__filled is not None
        return MaskedTensor(tdata, tmask, self.__filled if filled else self.__padvalue, filled)

    def unpooling(self, dims: Union[int, Iterable[int]], tarX):
        """insert new masked dims `dims` (sizes from tarX) and adopt tarX's mask."""
        if isinstance(dims, int):
            dims = [dims]
        dims = sorted(list(dims))
        if len(dims) == 1:
            out = _ops.masked_broadcast(self.__raw, tarX.mask, dims[0], self.__padvalue, self.masked_dim)
            return MaskedTensor(out, tarX.mask, self.__padvalue, True)
        tdata = self.__raw
        for _ in dims:
            tdata = tdata.unsqueeze(_)
        tdata = tdata.expand(*(-1 if i not in dims else tarX.shape[i] for i in range(tdata.ndim)))
        return MaskedTensor(tdata, tarX.mask, self.__padvalue, False)

    # ---- elementwise on data (MaTensor.py:236-266) -----------------------------------
    def tuplewiseapply(self, func: Callable[[Tensor], Tensor]):
        ndata = func(self.fill_masked(0.))
        return MaskedTensor(ndata, self.mask)

    def diagonalapply(self, func: Callable[[Tensor, LongTensor], Tensor]):
        assert self.masked_dim == 3, "only implemented for 2D"
        diagonaltype = torch.eye(self.shape[1], self.shape[2], dtype=torch.long, device=self.__raw.device)
        diagonaltype = diagonaltype.unsqueeze(0).expand_as(self.mask)
        ndata = func(self.data, diagonaltype)
        return MaskedTensor(ndata, self.mask)

    def add(self, tarX, samesparse: bool):
        assert isinstance(tarX, MaskedTensor)
        if samesparse:
            same = (self.__filled is not None and tarX._is_filled_with(self.__filled) and self.__filled == 0)
            return MaskedTensor(tarX.raw + self.__raw, self.mask, self.padvalue, is_filled=same)
        return MaskedTensor(tarX.fill_masked(0.) + self.fill_masked(0.), torch.logical_or(self.mask, tarX.mask), 0, True)

    def catvalue(self, tarX: Iterable, samesparse: bool):
        assert samesparse == True, "must have the same sparcity to concat value"  # noqa: E712
        if isinstance(tarX, MaskedTensor):
            tarX = [tarX]
        ndata = torch.concat([self.__raw] + [_.raw for _ in tarX], dim=-1)
        filled = self._is_filled_with(0) and all(_._is_filled_with(0) for _ in tarX)
        return MaskedTensor(ndata, self.mask, 0.0, filled)

    def __repr__(self):
        return f"MaskedTensor(shape={tuple(self.shape)}, masked_dim={self.masked_dim})"

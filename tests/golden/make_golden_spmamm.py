#!/usr/bin/env python3
"""
Golden vectors for `spmamm` in the one configuration the reference itself can run (pygho/backend/Spmamm.py:42-68): a sparse
(b, n, m) adjacency with SCALAR values (no dense dims; with any dense dim `Aval.unsqueeze(1) * tB[...]` does not broadcast,
SURVEY.md 2.3) times a MaskedTensor of shape (b, m, l) / (b, l, m) whose masked entries hold 0 (pre-filled: the reference's
`masked_fill` is not in-place, so it relies on that), aggr sum; plus value-less A.  Produced by RUNNING THE REFERENCE here (CPU);
the GPU box only sees the committed ``spmamm.npz``.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from pygho import MaskedTensor, SparseTensor        # noqa: E402
from pygho.backend.Spmamm import spmamm              # noqa: E402

T = torch.from_numpy


def main():
    rng = np.random.default_rng(17)
    b, n, m, l = 4, 5, 6, 3
    Amask = rng.random((b, n, m)) > 0.55
    Amask[2, 3] = False                                  # a target row without any message
    ind = np.stack(np.nonzero(Amask)).astype(np.int64)
    Aval = rng.standard_normal(ind.shape[1]).astype(np.float32)
    out = {"ind": ind, "Aval": Aval, "shapeA": np.array([b, n, m], dtype=np.int64)}
    for dim1 in (1, 2):
        k, nout = (n, m) if dim1 == 1 else (m, n)          # contracted size, output rows per batch element
        for dim2, bshape, oshape in ((1, (b, k, l), (b, nout, l)), (2, (b, l, k), (b, l, nout))):
            Bmask = rng.random(bshape) > 0.3
            B = (rng.standard_normal(bshape) * Bmask).astype(np.float32)       # masked entries hold 0
            omask = rng.random(oshape) > 0.2                                  # mask=None would wrap B.mask, whose shape only fits n == m
            tag = f"dim1_{dim1}_dim2_{dim2}"
            out[f"B_{tag}"], out[f"Bmask_{tag}"], out[f"omask_{tag}"] = B, Bmask, omask
            A = SparseTensor(T(ind), T(Aval), [b, n, m], is_coalesced=True)
            r = spmamm(A, dim1, MaskedTensor(T(B), T(Bmask), 0.0, True), dim2, T(omask), "sum")
            out[f"sum_{tag}"] = r.data.numpy()
            A0 = SparseTensor(T(ind), None, [b, n, m], is_coalesced=True)
            out[f"sum_novalue_{tag}"] = spmamm(A0, dim1, MaskedTensor(T(B), T(Bmask), 0.0, True), dim2, T(omask), "sum").data.numpy()
    np.savez_compressed(os.path.join(HERE, "spmamm.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

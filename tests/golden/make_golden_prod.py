#!/usr/bin/env python3
"""
Golden vectors for aggr = "prod" (reference: pygho/backend/utils.py:44-56 with reduce="prod"; coalesce(reduce="prod"),
pygho/backend/SpTensor.py:167-197).  Runs ONLY in the build container (imports the read-only reference checkout at
/root/reference, pure Python on torch-CPU); stores inputs + outputs as arrays in prod.npz next to this script.

    python tests/golden/make_golden_prod.py

Cases: the SURVEY 8(c)-3 vector; random f32 rows with empty segments and an unsorted index, forward AND autograd gradients
(torch's scatter_reduce_backward rule for "prod"), with exact zeros planted so that segments with one zero (the zero element
receives the product of the others) and with two zeros (everything 0) occur; an int64 1-D source; coalesce with duplicates.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from pygho.backend import SpTensor  # noqa: E402
from pygho.backend.utils import torch_scatter_reduce  # noqa: E402

T = torch.from_numpy


def main():
    rng = np.random.default_rng(61)
    out = {}
    src = np.array([[1, -2], [4, 5], [7, -8], [9, 10]], dtype=np.float32)
    ind = np.array([2, 2, 0, 0], dtype=np.int64)
    out.update(p0_src=src, p0_ind=ind, p0_size=np.int64(4), p0_prod=torch_scatter_reduce(0, T(src), T(ind), 4, "prod").numpy())
    # random, 2-D dense shape, unsorted index, empty segments, planted zeros
    src = rng.uniform(0.5, 1.5, size=(160, 3, 4)).astype(np.float32) * rng.choice([-1.0, 1.0], size=(160, 3, 4)).astype(np.float32)
    ind = rng.integers(0, 40, size=160).astype(np.int64)
    ind[ind % 7 == 0] = 1
    first = {int(s): np.nonzero(ind == s)[0] for s in np.unique(ind)}
    one = [s for s, rows in first.items() if rows.size >= 3][:4]
    for k, s in enumerate(one):
        src[first[s][1], 0, k] = 0.0                      # one zero in the segment (channel (0, k))
        src[first[s][0], 1, k] = 0.0                      # two zeros in the segment (channel (1, k))
        src[first[s][2], 1, k] = 0.0
    w = rng.standard_normal((43, 3, 4)).astype(np.float32)
    s = T(src).clone().requires_grad_(True)
    res = torch_scatter_reduce(0, s, T(ind), 43, "prod")
    (res * T(w)).sum().backward()
    out.update(p1_src=src, p1_ind=ind, p1_size=np.int64(43), p1_prod=res.detach().numpy(), p1_w=w, p1_grad=s.grad.numpy())
    # int64, 1-D
    src = rng.integers(-3, 4, size=48).astype(np.int64)
    ind = rng.integers(0, 10, size=48).astype(np.int64)
    out.update(p2_src=src, p2_ind=ind, p2_size=np.int64(12), p2_prod=torch_scatter_reduce(0, T(src), T(ind), 12, "prod").numpy())
    # coalesce with duplicates
    ci_in = rng.integers(0, 4, size=(3, 60)).astype(np.int64)
    cv_in = rng.uniform(0.5, 1.5, size=(60, 5)).astype(np.float32)
    ci, cv = SpTensor.coalesce(T(ci_in), T(cv_in), "prod")
    out.update(co_ind=ci_in, co_val=cv_in, co_ind_out=ci.numpy(), co_val_prod=cv.numpy())
    path = os.path.join(HERE, "prod.npz")
    np.savez_compressed(path, **out)
    print(f"prod.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


if __name__ == "__main__":
    main()

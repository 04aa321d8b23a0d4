#!/usr/bin/env python3
"""
Golden vectors for the padded-batch builders of the dense (MaskedTensor) path: ``to_dense_x``, ``to_dense_tuplefeat``,
``to_dense_adj``, ``to_sparse_adj`` (reference pygho/hodata/MaData.py:25-214), produced by RUNNING THE REFERENCE in this
container (CPU).  Run once here; the GPU box only sees the committed ``dense_collate.npz``.

``pygho.hodata`` cannot be imported as a package (torch_geometric is absent and its ``__init__`` pulls in the dataset
wrappers), but the four functions use nothing of it: the module file is loaded on its own, with placeholder
``torch_geometric.data`` / ``torch_geometric.utils`` modules satisfying its import lines and an empty ``pygho.hodata``
package object standing in for the package ``__init__``.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

for name, attrs in (("torch_geometric", ()), ("torch_geometric.data", ("Data", "Batch")), ("torch_geometric.utils", ("coalesce",)),
                    ("torch_geometric.nn", ("HeteroLinear",))):
    m = types.ModuleType(name)
    for a in attrs:
        setattr(m, a, type(a, (), {}))
    sys.modules.setdefault(name, m)

import pygho  # noqa: E402  (reference backend: SparseTensor / MaskedTensor)

pkg = types.ModuleType("pygho.hodata")
pkg.__path__ = [os.path.join(REF, "pygho", "hodata")]
sys.modules["pygho.hodata"] = pkg
spec = importlib.util.spec_from_file_location("pygho.hodata.MaData", os.path.join(REF, "pygho", "hodata", "MaData.py"))
MaData = importlib.util.module_from_spec(spec)
sys.modules["pygho.hodata.MaData"] = MaData
spec.loader.exec_module(MaData)

T = torch.from_numpy


def main():
    rng = np.random.default_rng(41)
    out = {}
    # ---- to_dense_x (MaData.py:108-147): ragged node counts incl. a 1-node graph, float and int64 features -----------
    counts = np.array([5, 1, 9, 3, 7], dtype=np.int64)
    ptr = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    xf = rng.standard_normal((int(ptr[-1]), 6)).astype(np.float32)
    xi = rng.integers(0, 28, size=(int(ptr[-1]),)).astype(np.int64)
    for tag, x in (("xf", xf), ("xi", xi)):
        mt = MaData.to_dense_x(T(x), T(ptr))
        out[tag], out[tag + "_mask"] = x, mt.mask.numpy()
        out[tag + "_raw"] = mt.data.numpy()          # the reference builds these unfilled: padded slots hold clamped neighbours
    mt = MaData.to_dense_x(T(xf), T(ptr), max_num_nodes=12)
    out["xf12_mask"], out["xf12_raw"] = mt.mask.numpy(), mt.data.numpy()
    out["ptr"] = ptr
    # ---- to_dense_tuplefeat (MaData.py:150-214): 2-D tuple grids of ragged (n_g, n_g) and rectangular shapes --------------
    for tag, shape in (("sq", np.stack((counts, counts), 1)), ("rect", np.array([[2, 5], [4, 1], [3, 3], [1, 6]], dtype=np.int64))):
        sizes = shape.prod(1)
        tptr = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
        tf = rng.integers(0, 4, size=(int(tptr[-1]),)).astype(np.int64)
        tv = rng.standard_normal((int(tptr[-1]), 3)).astype(np.float32)
        for kind, feat in (("i", tf), ("v", tv)):
            mt = MaData.to_dense_tuplefeat(T(feat), T(shape), T(tptr))
            out[f"tf_{tag}_{kind}"] = feat
            out[f"tf_{tag}_{kind}_mask"] = mt.mask.numpy()
            out[f"tf_{tag}_{kind}_raw"] = mt.data.numpy()
        out[f"tf_{tag}_shape"], out[f"tf_{tag}_ptr"] = shape, tptr
    # 3-D tuple grid
    shape3 = np.array([[2, 3, 2], [1, 1, 4], [3, 2, 1]], dtype=np.int64)
    tptr = np.concatenate(([0], np.cumsum(shape3.prod(1)))).astype(np.int64)
    tv = rng.standard_normal((int(tptr[-1]), 2)).astype(np.float32)
    mt = MaData.to_dense_tuplefeat(T(tv), T(shape3), T(tptr))
    out.update(tf3=tv, tf3_shape=shape3, tf3_ptr=tptr, tf3_mask=mt.mask.numpy(), tf3_raw=mt.data.numpy())
    # ---- to_dense_adj / to_sparse_adj (MaData.py:25-105): graph-local edge indices + edge batch vector -----------------------
    eb, ei = [], []
    for g, n in enumerate(counts):
        m = int(rng.integers(0, n * 2 + 1))
        pairs = {(int(rng.integers(0, n)), int(rng.integers(0, n))) for _ in range(m)}
        for a, b in sorted(pairs):
            eb.append(g)
            ei.append((a, b))
    eb = np.array(eb, dtype=np.int64)
    ei = np.array(ei, dtype=np.int64).T.copy()
    ea = rng.standard_normal((eb.shape[0], 4)).astype(np.float32)
    eai = rng.integers(1, 4, size=(eb.shape[0],)).astype(np.int64)
    out.update(adj_eb=eb, adj_ei=ei, adj_ea=ea, adj_eai=eai)
    for tag, attr, fill in (("ea", ea, 0.0), ("eai", eai, 0), ("ea_m1", ea, -1.0)):
        mt = MaData.to_dense_adj(T(ei), T(eb), T(attr), int(counts.max()), len(counts), fill)
        out[f"adj_{tag}_data"], out[f"adj_{tag}_mask"] = mt.data.numpy(), mt.mask.numpy()
    mt = MaData.to_dense_adj(T(ei), T(eb), None, int(counts.max()), len(counts))
    out["adj_ones_data"], out["adj_ones_mask"] = mt.data.numpy(), mt.mask.numpy()
    sp = MaData.to_sparse_adj(T(ei), T(eb), T(ea), int(counts.max()), len(counts))
    out["spadj_ind"], out["spadj_val"], out["spadj_shape"] = sp.indices.numpy(), sp.values.numpy(), np.array(sp.shape, dtype=np.int64)
    path = os.path.join(HERE, "dense_collate.npz")
    np.savez_compressed(path, **out)
    print(f"dense_collate.npz: {os.path.getsize(path)/1024:.1f} KiB, {len(out)} arrays")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""
Golden vectors for the tuple samplers: outputs of the REFERENCE's
``pygho/hodata/SpTupleSampler.py`` (``KhopSampler`` :91-126, ``I2Sampler`` :129-173, with the file's own
``k_hop_subgraph`` :12-88) on seeded graphs.  Runs only in the build container (reads /root/reference).

The reference module imports torch_geometric (absent here) for two CONTAINERS and two helpers; none of them
carries sampler logic, so clearly labelled stand-ins are registered before the module file is loaded:

  torch_geometric.data.Data            attribute bag
  torch_geometric.data.Batch           ``from_data_list`` = concatenate tensor attributes along dim 0
  torch_geometric.utils.to_scipy_sparse_matrix   COO edge list -> scipy matrix
  torch_geometric.utils.k_hop_subgraph           shadowed by the reference file's own definition (:12)
  torch_geometric.utils.num_nodes.maybe_num_nodes  num_nodes or max index + 1

``pygho/hodata/__init__.py`` pulls in loaders that need more of torch_geometric, so SpTupleSampler.py is loaded as a
module on its own (its relative import of ``..backend.SpTensor`` resolves to the real reference backend).

    python tests/golden/make_golden_samplers.py      ->  tests/golden/samplers.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.sparse as ssp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)


class StandInData:
    """stand-in for torch_geometric.data.Data: an attribute bag"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class StandInBatch(StandInData):
    """stand-in for torch_geometric.data.Batch.from_data_list: tensor attributes concatenated along dim 0"""

    @classmethod
    def from_data_list(cls, items):
        out = cls()
        for k, v in items[0].__dict__.items():
            if torch.is_tensor(v):
                setattr(out, k, torch.cat([getattr(it, k) for it in items], dim=0))
        return out


def _to_scipy(edge_index, num_nodes=None):
    ei = edge_index.numpy()
    n = int(num_nodes) if num_nodes is not None else int(ei.max()) + 1
    # (PyG returns COO; the scipy installed here only takes csr / csc / lil in csgraph.shortest_path, the values are the same)
    return ssp.coo_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n)).tocsr()


def _maybe_num_nodes(edge_index, num_nodes=None):
    return int(num_nodes) if num_nodes is not None else int(edge_index.max()) + 1


def load_reference_sampler():
    tg, tgd, tgu, tgn = (types.ModuleType(n) for n in ("torch_geometric", "torch_geometric.data", "torch_geometric.utils",
                                                       "torch_geometric.utils.num_nodes"))
    tgd.Data, tgd.Batch = StandInData, StandInBatch
    tgu.to_scipy_sparse_matrix = _to_scipy
    tgu.k_hop_subgraph = None                     # shadowed by the definition inside the reference file
    tgn.maybe_num_nodes = _maybe_num_nodes
    tg.data, tg.utils, tgu.num_nodes = tgd, tgu, tgn
    for m in (tg, tgd, tgu, tgn):
        sys.modules.setdefault(m.__name__, m)
    import pygho                                   # the real reference package (backend only is touched)
    pkg = types.ModuleType("pygho.hodata")
    pkg.__path__ = [os.path.join(REF, "pygho", "hodata")]
    sys.modules["pygho.hodata"] = pkg
    spec = importlib.util.spec_from_file_location("pygho.hodata.SpTupleSampler", os.path.join(REF, "pygho", "hodata", "SpTupleSampler.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    from pygho_amd import synth
    ref = load_reference_sampler()
    out = {}
    rng = np.random.default_rng(51)
    cases = []
    for g in range(4):
        n, adj = synth._zinc_like_graph(rng)
        cases.append(("z%d" % g, n, adj, 3))
    for g in range(3):
        n, adj = synth._gnm_graph(rng)
        cases.append(("g%d" % g, n, adj, 2 if g == 0 else 3))
    # a path with an isolated tail behind the hop limit and a two-node graph
    path = np.zeros((9, 9), bool)
    for a in range(8):
        path[a, a + 1] = path[a + 1, a] = True
    cases.append(("path", 9, path, 2))
    cases.append(("pair", 2, np.array([[0, 1], [1, 0]], bool), 3))
    names = []
    for name, n, adj, hop in cases:
        ei = torch.from_numpy(np.stack(np.nonzero(adj)).astype(np.int64))
        data = StandInData(edge_index=ei, num_nodes=n)
        k = ref.KhopSampler(data, hop)
        out[f"{name}_n"], out[f"{name}_hop"], out[f"{name}_edge_index"] = np.int64(n), np.int64(hop), ei.numpy()
        out[f"{name}_khop_ind"], out[f"{name}_khop_val"] = k.indices.numpy(), k.values.numpy()
        i2 = ref.I2Sampler(data, hop)
        out[f"{name}_i2_ind"], out[f"{name}_i2_val"] = i2.indices.numpy(), i2.values.numpy()
        assert k.indices.dtype == torch.int64 and k.values.dtype == torch.int64 and i2.values.shape[1] == 2
        names.append(name)
    out["names"] = np.array(names)
    path_out = os.path.join(HERE, "samplers.npz")
    np.savez_compressed(path_out, **out)
    print(f"samplers.npz: {os.path.getsize(path_out) / 1024:.1f} KiB, {len(out)} arrays, graphs {names}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(1)
    main()

"""GPU: the padded-batch builders of the dense layout (pygho_amd.hodata.MaData -> pygho_pad_stack / pygho_dense_adj) against the
reference's outputs (tests/golden/dense_collate.npz, padding slots included bit for bit) and the oracle on seeded batches."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


T = lambda a, dev: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
N = lambda t: t.detach().cpu().numpy()


def test_to_dense_x_golden(dev):
    from pygho_amd.hodata import to_dense_x
    g = load_golden("dense_collate.npz")
    for tag in ("xf", "xi"):
        mt = to_dense_x(T(g[tag], dev), T(g["ptr"], dev))
        assert np.array_equal(N(mt.mask), g[tag + "_mask"]) and np.array_equal(N(mt.raw), g[tag + "_raw"])
        m = g[tag + "_mask"].reshape(g[tag + "_mask"].shape + (1,) * (g[tag + "_raw"].ndim - 2))
        assert np.array_equal(N(mt.data), np.where(m, g[tag + "_raw"], 0))          # documented semantics: lazy fill on .data
    mt = to_dense_x(T(g["xf"], dev), T(g["ptr"], dev), max_num_nodes=12)
    assert np.array_equal(N(mt.mask), g["xf12_mask"]) and np.array_equal(N(mt.raw), g["xf12_raw"])


def test_to_dense_tuplefeat_golden(dev):
    from pygho_amd.hodata import to_dense_tuplefeat
    g = load_golden("dense_collate.npz")
    for tag in ("sq", "rect"):
        for kind in ("i", "v"):
            mt = to_dense_tuplefeat(T(g[f"tf_{tag}_{kind}"], dev), T(g[f"tf_{tag}_shape"], dev), T(g[f"tf_{tag}_ptr"], dev))
            assert np.array_equal(N(mt.mask), g[f"tf_{tag}_{kind}_mask"])
            assert np.array_equal(N(mt.raw), g[f"tf_{tag}_{kind}_raw"])
    mt = to_dense_tuplefeat(T(g["tf3"], dev), T(g["tf3_shape"], dev), T(g["tf3_ptr"], dev))
    assert np.array_equal(N(mt.mask), g["tf3_mask"]) and np.array_equal(N(mt.raw), g["tf3_raw"])
    # feat2mask is intersected with the padding mask
    mt = to_dense_tuplefeat(T(g["tf_sq_i"], dev), T(g["tf_sq_shape"], dev), T(g["tf_sq_ptr"], dev), feat2mask=lambda v: v > 0)
    assert np.array_equal(N(mt.mask), g["tf_sq_i_mask"] & (g["tf_sq_i_raw"] > 0))


def test_to_dense_and_sparse_adj_golden(dev):
    from pygho_amd.hodata import to_dense_adj, to_sparse_adj
    g = load_golden("dense_collate.npz")
    n, b = int(np.diff(g["ptr"]).max()), len(g["ptr"]) - 1
    ei, eb = T(g["adj_ei"], dev), T(g["adj_eb"], dev)
    for tag, attr, fill in (("ea", g["adj_ea"], 0.0), ("eai", g["adj_eai"], 0), ("ea_m1", g["adj_ea"], -1.0)):
        mt = to_dense_adj(ei, eb, T(attr, dev), n, b, fill)
        assert np.array_equal(N(mt.mask), g[f"adj_{tag}_mask"]) and np.array_equal(N(mt.data), g[f"adj_{tag}_data"])
    mt = to_dense_adj(ei, eb, None, n, b)
    assert np.array_equal(N(mt.mask), g["adj_ones_mask"]) and np.array_equal(N(mt.data), g["adj_ones_data"])
    mt = to_dense_adj(ei, eb, T(g["adj_ea"], dev))                       # sizes inferred like the reference
    assert tuple(mt.shape[:3]) == (b, int(g["adj_ei"].max()) + 1, int(g["adj_ei"].max()) + 1)
    sp = to_sparse_adj(ei, eb, T(g["adj_ea"], dev), n, b)
    assert np.array_equal(N(sp.indices), g["spadj_ind"]) and np.array_equal(N(sp.values), g["spadj_val"])
    assert list(sp.shape) == list(g["spadj_shape"])


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int64, np.uint8, np.float64])
@pytest.mark.parametrize("tail", [(), (3,), (128,), (5, 2)])
def test_builders_vs_oracle_every_row_size(dev, dtype, tail):
    """row sizes from 1 byte to 1 KiB (every unit width of the byte movers), empty graphs, a batch of 300 graphs."""
    from pygho_amd.hodata import to_dense_adj, to_dense_tuplefeat, to_dense_x
    rng = np.random.default_rng(len(tail) * 7 + np.dtype(dtype).itemsize)
    counts = rng.integers(0, 12, size=300).astype(np.int64)
    counts[0] = 11                                                # first graph non-empty (clamp target exists)
    ptr = np.concatenate(([0], np.cumsum(counts)))
    mk = lambda n: (rng.standard_normal((n,) + tail) * 50).astype(dtype)
    x = mk(int(ptr[-1]))
    mt = to_dense_x(T(x, dev), T(ptr, dev))
    raw, mask = O.to_dense_x(x, ptr)
    assert np.array_equal(N(mt.mask), mask) and np.array_equal(N(mt.raw), raw)
    shape = np.stack((counts, counts), 1)
    tptr = np.concatenate(([0], np.cumsum(counts * counts)))
    tf = mk(int(tptr[-1]))
    mt = to_dense_tuplefeat(T(tf, dev), T(shape, dev), T(tptr, dev))
    raw, mask = O.to_dense_tuplefeat(tf, shape, tptr)
    assert np.array_equal(N(mt.mask), mask) and np.array_equal(N(mt.raw), raw)
    eb, er, ec = np.nonzero(rng.random((300, 11, 11)) < 0.1)
    keep = (er < counts[eb]) & (ec < counts[eb])
    eb, er, ec = eb[keep], er[keep], ec[keep]
    ea = mk(eb.shape[0])
    fill = 7
    mt = to_dense_adj(T(np.stack((er, ec)), dev), T(eb, dev), T(ea, dev), 11, 300, fill)
    data, mask = O.to_dense_adj(np.stack((er, ec)), eb, ea, 11, 300, fill)
    assert np.array_equal(N(mt.mask), mask) and np.array_equal(N(mt.data), data)


def test_batch2dense_feeds_the_dense_layers(dev):
    """a collated batch (attribute object, as the reference's PygBatch) -> batch2dense -> NGNNConv mode "DD" runs on the result
    and agrees with the same layer fed the host-side padded arrays of synth.make_dense_batch."""
    from pygho_amd import MaskedTensor, synth
    from pygho_amd.hodata import batch2dense
    from pygho_amd.honn import Conv
    h = 16
    dn = synth.make_dense_batch(6, seed=3, hidden=h, clip_nodes=9)
    nm = dn["nodemask"]
    counts = nm.sum(1).astype(np.int64)
    ptr = np.concatenate(([0], np.cumsum(counts)))
    b, n = nm.shape
    xs = np.concatenate([np.arange(c) for c in counts]).astype(np.int64)           # any node feature
    eb, er, ec = np.nonzero(dn["Amask"])
    tf = np.concatenate([dn["X"][g, :c, :c].reshape(c * c, h) for g, c in enumerate(counts)])
    batch = types.SimpleNamespace(x=T(xs, dev), ptr=T(ptr, dev), edge_index=T(np.stack((er, ec)), dev), edge_index_batch=T(eb, dev),
                                  edge_attr=T(dn["A"][eb, er, ec], dev), tuplefeat=T(tf, dev),
                                  tupleshape=T(np.stack((counts, counts), 1), dev),
                                  tuplefeat_ptr=T(np.concatenate(([0], np.cumsum(counts * counts))), dev))
    batch = batch2dense(batch, denseadj=True)
    assert np.array_equal(N(batch.x.mask), nm) and np.array_equal(N(batch.X.mask), dn["Xmask"])
    assert np.array_equal(N(batch.A.mask), dn["Amask"]) and np.array_equal(N(batch.A.data), dn["A"])
    assert np.array_equal(N(batch.X.data), dn["X"] * dn["Xmask"][..., None])
    torch.manual_seed(0)
    mlp = {"numlayer": 1, "tailact": True, "norm": "none", "act": "silu", "dp": 0.0}
    layer = Conv.NGNNConv(h, h, "sum", "DD", mlp).to(dev)
    out_a = layer(batch.A, batch.X, {})
    out_b = layer(MaskedTensor(T(dn["A"], dev), T(dn["Amask"], dev), 0.0, True), MaskedTensor(T(dn["X"], dev), T(dn["Xmask"], dev), 0.0, True), {})
    torch.testing.assert_close(out_a.data, out_b.data, rtol=1e-6, atol=1e-6)


def test_sp_datapreprocess_and_batch2sparse(dev):
    """reference hodata/SpData.py:80-171 on the device: per-graph preprocessing (edge coalescing incl. duplicate edges, tuple sampler,
    precomputed message triples for two keys) against the host plan of the same graph, then batch2sparse on a collated batch."""
    import types
    from pygho_amd import SparseTensor, synth
    from pygho_amd.hodata import batch2sparse, parsekey, sp_datapreprocess
    from conftest import canon_triples
    assert parsekey("X___A___1___X___0") == ("X", "A", 1, "X", 0)
    keys = ["X___X___1___A___0", "X___A___1___X___0"]
    rng = np.random.default_rng(12)
    for _ in range(3):
        rec = synth.make_graph(rng, "zinc", 3, tuple(keys))
        n = rec.num_nodes
        dup = np.concatenate((rec.edge_index, rec.edge_index[:, :3]), axis=1)          # duplicated edges: attributes are summed
        dup_attr = np.concatenate((rec.edge_attr, rec.edge_attr[:3])).astype(np.float32)
        data = types.SimpleNamespace(num_nodes=n, x=T(rec.x, dev), edge_index=T(dup, dev), edge_attr=T(dup_attr, dev))
        sampler = lambda d: SparseTensor(T(rec.tupleid, dev), T(rec.tuplefeat, dev), [n, n], True)
        out = sp_datapreprocess(data, [sampler], [""], keys)
        assert np.array_equal(N(out.edge_index), rec.edge_index) and out.num_edges == rec.edge_index.shape[1]
        exp_attr = rec.edge_attr.astype(np.float32).copy()
        # edges come back in lexicographic order = the record's order; the first three were given twice
        exp_attr[:3] *= 2
        assert np.array_equal(N(out.edge_attr), exp_attr)
        assert out.num_tuples == rec.tupleid.shape[1] and N(out.tupleshape).tolist() == [[n, n]]
        for key in keys:
            got = canon_triples(N(getattr(out, key + "___acd")))
            assert np.array_equal(got, canon_triples(rec.acd[key])), key
    hb = synth.make_batch(5, "zinc", seed=4)
    batch = dict(num_nodes=hb.num_nodes, edge_index=T(hb.edge_index, dev), edge_attr=T(hb.edge_attr, dev), tupleid=T(hb.tupleid, dev),
                 tuplefeat=T(hb.tuplefeat, dev), tupleshape=torch.tensor([[hb.num_nodes // 5 + 1, hb.num_nodes // 5 + 1]] * 4
                                                                         + [[hb.num_nodes - 4 * (hb.num_nodes // 5 + 1)] * 2], device=dev))
    batch = batch2sparse(batch)
    assert tuple(batch["A"].shape[:2]) == (hb.num_nodes, hb.num_nodes) and batch["A"].nnz == hb.edge_index.shape[1]
    assert tuple(batch["X"].shape[:2]) == (hb.num_nodes, hb.num_nodes) and np.array_equal(N(batch["X"].indices), hb.tupleid)

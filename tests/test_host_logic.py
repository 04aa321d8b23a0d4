"""
CPU tests of the host-side logic: containers' shape bookkeeping, operator key plumbing and mode dispatch
(mirroring pygho/honn), the synthetic data contract (collation offsets of hodata/SpData.py:60-77) and the
graph sharder.  No kernels are launched.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import np_oracle as O
from pygho_amd import MaskedTensor, SparseTensor, synth
from pygho_amd.honn import Conv, MaOperator, SpOperator, TensorOp
from pygho_amd.parallel import shard_ranges

MLP = {"numlayer": 1, "tailact": True, "norm": "bn", "act": "silu", "dp": 0.0}


def test_sparse_tensor_bookkeeping():
    ind = torch.tensor([[0, 1, 1], [2, 0, 1]])
    val = torch.arange(12.0).reshape(3, 4)
    A = SparseTensor(ind, val, [2, 3, 4], is_coalesced=True)
    assert (A.sparse_dim, A.nnz, A.shape, A.sparseshape, A.denseshape) == (2, 3, (2, 3, 4), (2, 3), (4,))
    assert A.is_coalesced() and A.indices is ind and A.values is val
    B = A.tuplewiseapply(lambda v: v[:, :2] * 2)
    assert B.indices is ind and B.shape == (2, 3, 2)                      # pattern shared by reference
    C = A.catvalue([B], True)
    assert C.shape == (2, 3, 6)
    assert torch.equal(A.add(A, True).values, val * 2)
    assert SparseTensor(ind, val, None, True).shape == (2, 3, 4)          # shape inferred from max index
    with pytest.raises(AssertionError):
        SparseTensor(ind, val[:2], [2, 3, 4], True)
    with pytest.raises(AssertionError):
        A.catvalue(B, False)
    assert A.to("cpu") is A                                               # in place, returns self (SpTensor.py:271-274)
    d = A.diagonalapply(lambda v, flag: v * flag.unsqueeze(-1))
    assert torch.equal(d.values[2], val[2]) and float(d.values[:2].abs().sum()) == 0.0
    coo = A.to_torch_sparse_coo()
    assert torch.equal(coo.to_dense(), SparseTensor.from_torch_sparse_coo(coo.coalesce()).to_torch_sparse_coo().to_dense())


def test_masked_tensor_bookkeeping():
    data = torch.randn(2, 3, 3, 5)
    mask = torch.rand(2, 3, 3) > 0.5
    M = MaskedTensor(data, mask, padvalue=0.0, is_filled=True)
    assert (M.masked_dim, M.dense_dim, tuple(M.maskedshape), tuple(M.denseshape)) == (3, 1, (2, 3, 3), (5,))
    assert M.data is data and M.mask is mask and M.padvalue == 0.0
    assert M.fullnegmask.shape == (2, 3, 3, 1)
    assert M.fill_masked(0.) is data                                      # already filled with the same value
    dg = M.diag([1, 2])
    assert dg.shape == (2, 3, 5) and torch.equal(dg.mask, torch.diagonal(mask, 0, 1, 2))
    cat = M.catvalue([M], True)
    assert cat.shape == (2, 3, 3, 10)
    with pytest.raises(AssertionError):
        MaskedTensor(data, mask[:, :2])


def test_precompute_keys_and_dispatch_match_reference():
    g = load_golden("layers.npz")
    model = torch.nn.ModuleList([Conv.NGNNConv(8, 8, "sum", "SS", dict(MLP)), Conv.SSWLConv(8, 8, "sum", "SS", dict(MLP)),
                                 Conv.I2Conv(8, 8, "sum", "SS", dict(MLP)), Conv.PPGNConv(8, 8, "sum", "SS", dict(MLP))])
    assert SpOperator.parse_precomputekey(model) == list(g["parsed_keys"])        # produced by the reference
    assert SpOperator.KEYSEP == "___"
    assert SpOperator.OpMessagePassingOnSubg2D().precomputekey == "X___X___1___A___0"
    assert SpOperator.OpMessagePassingCrossSubg2D().precomputekey == "X___A___1___X___0"
    assert SpOperator.OpMessagePassingOnSubg3D().precomputekey == "X___X___2___A___0"
    assert SpOperator.Op2FWL(optuplefeat="Y").precomputekey == "Y___Y___1___Y___0"
    assert isinstance(TensorOp.OpMessagePassingOnSubg2D("SS").mod, SpOperator.OpMessagePassingOnSubg2D)
    assert isinstance(TensorOp.OpMessagePassingOnSubg2D("DD").mod, MaOperator.OpMessagePassingOnSubg2D)
    assert isinstance(TensorOp.OpMessagePassingOnSubg2D("SD").mod, MaOperator.OpSpMessagePassingOnSubg2D)
    assert isinstance(TensorOp.OpPoolingSubg2D("D", "max").mod, MaOperator.OpPoolingSubg2D)
    assert TensorOp.OpPoolingSubg2D("S", "mean").mod.dims == [1] and MaOperator.OpPoolingSubg2D().dims == [2]
    assert MaOperator.OpMessagePassingOnSubg2D().dim1 == 2 and MaOperator.OpMessagePassingCrossSubg2D().dim1 == 1
    with pytest.raises(AssertionError):
        TensorOp.OpMessagePassingOnSubg2D("DD", aggr="max")
    with pytest.raises(NotImplementedError):
        TensorOp.OpMessagePassingOnSubg2D("XX")
    with pytest.raises(NotImplementedError):
        TensorOp.OpDiag2D("Q")
    # state_dict layout is interchangeable with the reference's
    sd_names = sorted(k[len("ngnn_sd_"):] for k in g.files if k.startswith("ngnn_sd_"))
    assert sorted(Conv.NGNNConv(16, 16, "sum", "SS", dict(MLP)).state_dict().keys()) == sd_names


def test_synthetic_batch_contract():
    keys = ("X___X___1___A___0", "X___A___1___X___0")
    rng = np.random.default_rng(3)
    recs = [synth.make_graph(rng, "zinc", 3, keys) for _ in range(5)]
    hb = synth.collate(recs)
    assert hb.num_graphs == 5 and hb.num_nodes == sum(r.num_nodes for r in recs)
    for k in keys:
        op0, op1, d1, op2, d2 = synth.parse_key(k)
        pick = lambda op: hb.tupleid if op[0] == "X" else hb.edge_index
        tar, bcd = O.spspmm_ind(pick(op1), d1, pick(op2), d2)
        ref = O.filterind(pick(op0), tar, bcd)                           # planner on the WHOLE batch
        assert np.array_equal(hb.acd[k], ref), k                         # == per-graph plans + collate offsets
        assert (np.diff(hb.acd[k][0]) >= 0).all()
    assert (np.diff(O.indicehash(hb.tupleid)) > 0).all() and (np.diff(O.indicehash(hb.edge_index)) > 0).all()
    assert (np.diff(hb.batch) >= 0).all() and hb.batch[-1] == 4
    rep = synth.replicate(hb, 3)
    assert rep.num_graphs == 15 and rep.num_messages(keys[0]) == 3 * hb.num_messages(keys[0])
    tar, bcd = O.spspmm_ind(rep.tupleid, 1, rep.edge_index, 0)
    assert np.array_equal(rep.acd[keys[0]], O.filterind(rep.tupleid, tar, bcd))
    hi = synth.make_batch(2, "i2", seed=0)
    assert hi.tupleid.shape[0] == 3 and hi.tuplefeat.shape[1] == 2
    dn = synth.make_dense_batch(3, seed=1, hidden=4, clip_nodes=6)
    assert dn["X"].shape[:3] == dn["Xmask"].shape and float(np.abs(dn["A"][~dn["Amask"]]).sum()) == 0.0


def test_shard_ranges_balance_and_cover():
    rng = np.random.default_rng(0)
    w = rng.integers(100, 900, size=1000)
    for world in (1, 2, 3, 8):
        r = shard_ranges(w, world)
        assert r[0][0] == 0 and r[-1][1] == 1000 and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        loads = [w[a:b].sum() for a, b in r]
        assert max(loads) <= 1.05 * w.sum() / world + 900
    assert shard_ranges(np.ones(3), 8)[-1][1] == 3                        # more ranks than graphs: empty tails


def test_window_kernel_dispatch_rule():
    """host-side choice between the gather-everything segment kernel and its LDS-window variant (`_ops._window_eligible`):
    two-operand sum / mean with an indexed, SMALL rhs and rows of 512 B ... 1 KB only."""
    import torch
    from pygho_amd import _ops
    idx = torch.zeros(10, dtype=torch.int32)
    big = lambda rows, d, dt: torch.empty(rows, d, dtype=dt)
    lhs, rhs = big(100_000, 256, torch.bfloat16), big(5_000, 256, torch.bfloat16)
    saved = (_ops.USE_SEG_WINDOW, _ops.SEG_WINDOW_MIN_ROW_BYTES)
    try:
        _ops.USE_SEG_WINDOW, _ops.SEG_WINDOW_MIN_ROW_BYTES = True, 512
        assert _ops._window_eligible(100_000, lhs, rhs, idx, "sum") and _ops._window_eligible(100_000, lhs, rhs, idx, "mean")
        assert not _ops._window_eligible(100_000, lhs, rhs, idx, "max")                    # extremum: other kernel
        assert not _ops._window_eligible(100_000, lhs, rhs, None, "sum")                   # rhs read in message order: no window
        assert not _ops._window_eligible(100_000, None, rhs, idx, "sum")                   # one operand only
        assert not _ops._window_eligible(100_000, lhs, big(80_000, 256, torch.bfloat16), idx, "sum")   # rhs is not the small operand
        assert not _ops._window_eligible(100_000, big(100_000, 128, torch.bfloat16), big(5_000, 128, torch.bfloat16), idx, "sum")  # 256-B rows
        assert _ops._window_eligible(100_000, big(100_000, 128, torch.float32), big(5_000, 128, torch.float32), idx, "sum")       # 512-B rows
        assert not _ops._window_eligible(100_000, big(100_000, 512, torch.float32), big(5_000, 512, torch.float32), idx, "sum")   # 2-KB rows
        assert not _ops._window_eligible(1_000, big(1_000, 256, torch.bfloat16), big(100, 256, torch.bfloat16), idx, "sum")       # tiny launch
        _ops.USE_SEG_WINDOW = False
        assert not _ops._window_eligible(100_000, lhs, rhs, idx, "sum")
    finally:
        _ops.USE_SEG_WINDOW, _ops.SEG_WINDOW_MIN_ROW_BYTES = saved


def test_parsekey_and_parseop():
    """reference SpData.py:14-53; an unknown operand name raises here (the reference returns a tuple, SpData.py:31)"""
    import pytest
    from pygho_amd.hodata.SpData import parsekey, parseop
    assert parsekey("X___A___1___X___0") == ("X", "A", 1, "X", 0)
    assert parseop("X") == "num_tuples" and parseop("X1") == "num_tuples1" and parseop("A") == "num_edges"
    with pytest.raises(NotImplementedError):
        parseop("B")
    with pytest.raises(NotImplementedError):
        parsekey("X___B___1___X___0")


def test_optimizer_steps_invalidate_the_cast_arenas():
    """the 16-bit parameter copies of pygho_amd.blocks are invalidated by EVERY optimizer step (global post-step hook): fused
    optimizers update parameters without moving their version counters, so versions alone cannot tell (tests/test_gpu_sparse.py
    checks the copies themselves on the device)"""
    import torch
    from pygho_amd import blocks
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    for opt in (torch.optim.SGD([p], lr=0.1), torch.optim.AdamW([p], lr=0.1)):
        before = blocks._ARENA_EPOCH[0]
        opt.step()
        assert blocks._ARENA_EPOCH[0] > before
    before = blocks._ARENA_EPOCH[0]
    blocks.invalidate_cast_arenas()
    assert blocks._ARENA_EPOCH[0] == before + 1


def test_store_mirror_verdict_per_graph():
    """collate._mirror_of (host half of the collated mirror plan): positions of the transposed tuples of a symmetric 2-tuple set
    with a symmetric feature, and the verdicts for an asymmetric set / an asymmetric feature / a feature beyond the kernel's table."""
    import numpy as np
    from pygho_amd import synth
    from pygho_amd.collate import _mirror_of
    rng = np.random.default_rng(0)
    r = synth.make_graph(rng, "zinc", 3)
    pos, ok = _mirror_of(r, 16)
    assert ok
    row, col = r.tupleid
    assert np.array_equal(row[pos], col) and np.array_equal(col[pos], row) and np.array_equal(r.tuplefeat[pos], r.tuplefeat)
    assert not _mirror_of(r, int(r.tuplefeat.max()))[1]                          # a feature value the table kernel cannot hold
    off = int(np.nonzero(row != col)[0][0])
    feat = r.tuplefeat.copy()
    feat[off] += 1
    assert not _mirror_of(synth.GraphRecord(r.num_nodes, r.x, r.edge_index, r.edge_attr, r.tupleid, feat), 16)[1]
    keep = np.ones(row.size, dtype=bool)
    keep[off] = False
    assert not _mirror_of(synth.GraphRecord(r.num_nodes, r.x, r.edge_index, r.edge_attr, r.tupleid[:, keep], r.tuplefeat[keep]), 16)[1]
    empty = synth.GraphRecord(3, r.x[:3], r.edge_index[:, :0], r.edge_attr[:0], r.tupleid[:, :0], r.tuplefeat[:0])
    pos, ok = _mirror_of(empty, 16)
    assert ok and pos.size == 0


# ---------------------------------------------------------------------------------------------------------------------
# fixed-capacity batch slots (pygho_amd.slots): the host side -- capacities, the per-batch upload, the store's validation
# ---------------------------------------------------------------------------------------------------------------------
def test_batch_layout_is_the_running_offsets_of_the_selected_graphs():
    """the upload of a slot: per family the exclusive running offsets of the selected graphs (= the increments of
    hodata/SpData.py:60-77), their first store columns, the batch's true totals in entry G; None when a capacity is exceeded"""
    from pygho_amd.slots import batch_layout
    rng = np.random.default_rng(0)
    n_store, g = 50, 7
    h_len = rng.integers(0, 9, size=(3, n_store)).astype(np.int64)
    h_ptr = np.concatenate((np.zeros((3, 1), dtype=np.int64), np.cumsum(h_len, axis=1)[:, :-1]), axis=1)
    ids = rng.permutation(n_store)[:g]
    caps = np.asarray([100, 100, 100], dtype=np.int64)
    lay = batch_layout(h_len, h_ptr, caps, ids)
    assert lay.shape == (2 + 6, g + 1) and lay.dtype == np.int64
    assert np.array_equal(lay[0, :g], ids) and np.array_equal(lay[1], np.arange(g + 1))
    for f in range(3):
        run = 0
        for s, gid in enumerate(ids):
            assert lay[2 + f, s] == run and lay[5 + f, s] == h_ptr[f, gid]
            run += h_len[f, gid]
        assert lay[2 + f, g] == run                                   # the true total, what the "_dyn" kernels read
    tight = h_len[:, ids].sum(axis=1)
    assert batch_layout(h_len, h_ptr, tight, ids) is not None
    assert batch_layout(h_len, h_ptr, tight - np.asarray([0, 1, 0]), ids) is None


def test_slot_capacities_are_distinct_and_cover_random_batches():
    from types import SimpleNamespace
    from pygho_amd.slots import slot_capacities
    rng = np.random.default_rng(1)
    lens = {"node": rng.integers(9, 38, 4000), "edge": rng.integers(20, 90, 4000), "tup": rng.integers(100, 400, 4000),
            ("acd", "k"): rng.integers(200, 900, 4000), ("sc", "k"): rng.integers(3, 9, 4000)}
    store = SimpleNamespace(h_len=lens)
    for g in (8, 128, 1024):
        caps = slot_capacities(store, g)
        assert set(caps) == {"node", "edge", "tup", ("acd", "k")}            # (scatter chunks have no capacity: the slot runs the gather form)
        vals = list(caps.values())
        assert len(set(vals)) == len(vals) and g not in vals and all(v % 64 == 0 for v in vals)
        worst = {f: np.sort(lens[f])[::-1][:g].sum() for f in caps}
        assert all(caps[f] <= worst[f] + 64 * len(vals) for f in caps)
        misses = 0
        for _ in range(300):
            ids = rng.permutation(4000)[:g]
            misses += any(lens[f][ids].sum() > caps[f] for f in caps)
        assert misses == 0                                                   # 4.5 sigma: one batch in ~10^5 would overflow
        # and the capacity is close to the typical batch (the pad rows are wasted launches' width)
        assert all(caps[f] <= 1.35 * g * lens[f].mean() + 128 for f in caps) or g == 8


def test_graph_store_validation_catches_what_the_per_batch_flags_caught():
    """collated batches skip the per-batch range flags because 'the store checked them' -- so the store has to."""
    import copy
    from pygho_amd import synth
    from pygho_amd.collate import _validate_records
    rng = np.random.default_rng(2)
    key = "X___X___1___A___0"
    recs = [synth.make_graph(rng, "zinc", 3, (key,)) for _ in range(4)]
    _validate_records(recs, [key])

    def broken(mutate):
        bad = copy.deepcopy(recs)
        mutate(bad[2])
        with pytest.raises(ValueError, match="graph 2"):
            _validate_records(bad, [key])
    broken(lambda r: r.x.__setitem__(0, -1))
    broken(lambda r: r.edge_attr.__setitem__(1, -3))
    broken(lambda r: r.tuplefeat.__setitem__(0, -1))
    broken(lambda r: r.edge_index.__setitem__((0, 0), r.num_nodes))
    broken(lambda r: r.tupleid.__setitem__((1, 2), -1))
    broken(lambda r: r.acd[key].__setitem__((1, 0), r.tupleid.shape[1]))
    broken(lambda r: r.acd[key].__setitem__((2, 0), r.edge_index.shape[1]))
    broken(lambda r: setattr(r, "x", r.x[:-1]))

"""
Both gradients of a subgraph layer's aggregation in ONE pass (`csrc/seg_dual.hip`, `segment.dual_backward`): autograd of the reference's
index / index / mul / scatter_reduce chain (pygho/backend/Spspmm.py:309-315) inside NGNNConv (pygho/honn/Conv.py:53-58) with the
adjacency values an embedding lookup (example/minimal.py:22-34).

* the ALIGNED planner: chunks tile the messages, every message of a chunk's first-operand rows lies inside the chunk, the chunk's
  messages occupy the same positions in the grouping by c, and every first-operand row is owned by exactly one chunk;
* the dual launch == `seg_gmr` over the by-c plan (by-tuple gradient) AND `by_edge_product` (by-edge gradient), bit for bit, over
  dtypes, table sizes, with and without the chained gradient; rows without messages come out as zeros;
* a plan with a group of messages outside the chunk limits stays unaligned and takes the two launches;
* at BASELINE size (8192 ZINC-shape graphs, store-collated plan): both gradients bit for bit against the host oracle;
* a whole NGNN training step with and without it: same loss, same gradients, bit for bit.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEY = "X___X___1___A___0"


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the ROCm device")
    return torch.device("cuda:0")


@pytest.fixture()
def small_plans(dev):
    """the size threshold of the scatter form (2^20 messages) off for the duration of a test"""
    from pygho_amd import _ops
    old = _ops.SEG_SCATTER_MIN_MESSAGES
    _ops.SEG_SCATTER_MIN_MESSAGES = 0
    yield
    _ops.SEG_SCATTER_MIN_MESSAGES = old


def _batch(dev, graphs, seed):
    from pygho_amd import _ops, synth
    hb = synth.make_batch(graphs, "zinc", seed=seed)
    acd = torch.from_numpy(hb.acd[KEY]).to(dev)
    plan = _ops.message_plan(acd, hb.num_tuples, hb.num_tuples, hb.num_edges)
    ea = torch.from_numpy(hb.edge_attr).to(dev).long()
    return hb, plan, ea


def test_aligned_planner_invariants(dev, small_plans):
    from pygho_amd import _ops
    hb, plan, _ = _batch(dev, 120, 3)
    sp = _ops.scatter_plan(plan)
    assert sp is not None and sp.cgap is not None and sp.covers_c
    ch = sp.chunks.cpu().numpy().astype(np.int64)
    gap = sp.cgap.cpu().numpy().astype(np.int64)
    a, c = plan.a32.cpu().numpy().astype(np.int64), plan.c32.cpu().numpy().astype(np.int64)
    pc, a_byc, _ = plan.by_c()
    ptr_c, a_byc = pc.seg_ptr.cpu().numpy().astype(np.int64), a_byc.cpu().numpy().astype(np.int64)
    m_lo, a_lo, c_lo = ch[:, 0], ch[:, 1], ch[:, 2]
    n, a_rows, c_rows = ch[:, 3] & 0xff, (ch[:, 3] >> 8) & 0xff, (ch[:, 3] >> 16) & 0xff
    assert m_lo[0] == 0 and np.array_equal(m_lo[1:], (m_lo + n)[:-1]) and m_lo[-1] + n[-1] == plan.m          # tile the messages
    assert n.min() >= 1 and n.max() <= 64 and a_rows.max() <= 32 and c_rows.max() <= 32
    owner = np.zeros(plan.n_lhs, dtype=np.int64)
    for k in range(ch.shape[0]):
        sl = slice(m_lo[k], m_lo[k] + n[k])
        assert a[sl].min() == a_lo[k] and a[sl].max() == a_lo[k] + a_rows[k] - 1
        assert c[sl].min() == c_lo[k] and c[sl].max() == c_lo[k] + c_rows[k] - 1
        # closure: the by-c positions of the window's rows are exactly the chunk's message positions
        assert ptr_c[c_lo[k]] == m_lo[k] and ptr_c[c_lo[k] + c_rows[k]] == m_lo[k] + n[k]
        assert np.array_equal(np.sort(a_byc[sl]), np.sort(a[sl]))
        before, after = gap[k] & 0xffff, gap[k] >> 16
        owner[c_lo[k] - before:c_lo[k] + c_rows[k] + after] += 1
    assert np.all(owner == 1)                                               # every first-operand row: exactly one owner


def _operands(dev, hb, dtype, table_rows, seed=0):
    torch.manual_seed(seed)
    d = 128
    g = torch.randn(hb.num_tuples, d, device=dev).to(dtype)
    h = torch.randn(hb.num_tuples, d, device=dev).to(dtype)
    table = torch.randn(table_rows, d, device=dev).to(dtype)
    addend = torch.randn(hb.num_edges, d, device=dev).to(dtype)
    return g, h, table, addend


def _separate(plan, g, h, table, look_byc, addend):
    from pygho_amd import _ops
    pc, a_byc, _ = plan.by_c()
    gh = _ops.seg_gmr(plan.n_lhs, g, table, pc.seg_ptr, a_byc, look_byc, "sum")
    g_rhs = _ops.by_edge_product(plan, g, h, None, addend=addend)
    return gh, g_rhs


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("table_rows,chained", [(16, True), (5, False), (1, True), (32, False)])
def test_dual_equals_the_two_launches_bitwise(dev, small_plans, dtype, table_rows, chained):
    from pygho_amd import _ops
    hb, plan, ea = _batch(dev, 200, 5)
    assert _ops.scatter_plan(plan).cgap is not None
    g, h, table, addend = _operands(dev, hb, dtype, table_rows)
    look_byc = plan.lookup(_ops.flat_index(ea % table_rows))[1]
    assert _ops.dual_eligible(plan, g, h, table, None)
    gh_ref, gr_ref = _separate(plan, g, h, table, look_byc, addend if chained else None)
    timer = _ops.LaunchTimer()
    with timer:
        gh, gr = _ops.dual_backward(plan, g, h, table, look_byc, addend=addend if chained else None)
    torch.cuda.synchronize()
    assert any(k.startswith("seg_dual[") for k in timer.summary())
    assert torch.equal(gh, gh_ref)
    assert torch.equal(gr, gr_ref)


def test_rows_without_messages_are_zero_and_sparse_patterns(dev, small_plans):
    """first-operand rows that no message reads (in front of a window, between windows, behind a block's last window), a block whose
    rows carry no message at all, one row with many messages"""
    from pygho_amd import _ops
    n, ne = 400, 12
    a = [3, 3, 10, 50, 51, 51, 51] + [300] * 40 + [399]
    c = [5, 7, 12, 40, 60, 61, 62] + [280 + (i % 30) for i in range(40)] + [390]
    dd = [0, 1, 2, 3, 3, 4, 5] + [6 + (i % 5) for i in range(40)] + [11]
    acd = torch.tensor([a, c, dd], dtype=torch.int64, device=dev)
    plan = _ops.message_plan(acd, n, n, ne)
    sp = _ops.scatter_plan(plan)
    assert sp is not None and sp.cgap is not None
    torch.manual_seed(1)
    g = torch.randn(n, 128, device=dev).to(torch.bfloat16)
    h = torch.randn(n, 128, device=dev).to(torch.bfloat16)
    table = torch.randn(4, 128, device=dev).to(torch.bfloat16)
    row_of = torch.tensor([i % 4 for i in range(ne)], dtype=torch.int64, device=dev)
    look_byc = plan.lookup(row_of)[1]
    assert _ops.dual_eligible(plan, g, h, table, None)
    gh_ref, gr_ref = _separate(plan, g, h, table, look_byc, None)
    gh, gr = _ops.dual_backward(plan, g, h, table, look_byc)
    assert torch.equal(gh, gh_ref) and torch.equal(gr, gr_ref)
    read = torch.zeros(n, dtype=torch.bool, device=dev)
    read[acd[1]] = True
    assert bool((gh[~read] == 0).all()) and int((~read).sum()) > 300


def test_a_group_outside_the_limits_keeps_the_two_launches(dev, small_plans):
    """70 messages that all read the same first-operand rows cannot be cut into aligned chunks of 64: the plan stays unaligned (the plain
    chunks still serve the by-edge scatter) and the dispatcher takes the two launches"""
    from pygho_amd import _ops
    n, ne = 200, 30
    a = sorted([20 + (i % 14) for i in range(70)])
    c = [40 + (i % 5) for i in range(70)]
    dd = [(i * 7) % ne for i in range(70)]
    acd = torch.tensor([a, c, dd], dtype=torch.int64, device=dev)
    plan = _ops.message_plan(acd, n, n, ne)
    sp = _ops.scatter_plan(plan)
    g = torch.randn(n, 128, device=dev).to(torch.bfloat16)
    h = torch.randn(n, 128, device=dev).to(torch.bfloat16)
    table = torch.randn(4, 128, device=dev).to(torch.bfloat16)
    assert sp is None or sp.cgap is None
    assert not _ops.dual_eligible(plan, g, h, table, None)


@pytest.fixture(scope="module")
def baseline_batch(dev):
    from pygho_amd import synth
    from pygho_amd.collate import DeviceGraphStore
    rng = np.random.default_rng(1000)
    store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(8192)], dev)
    return store.collate(np.random.default_rng(11).permutation(8192))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dual_at_baseline_size_vs_host_oracle(dev, baseline_batch, dtype):
    """8192 ZINC-shape graphs, width 128, the store-collated (aligned) plan: BOTH gradients bit for bit against the reference's ATen
    sequence on the host (oracle.aten_port.spspmm_values_chunked: index, index, mul, index_add_ in message order, f32; the f32 product
    of two 16-bit values is exact, so the device's f32 sums agree exactly and the result is their one rounding), with and without the
    chained gradient of the adjacency values."""
    from oracle import aten_port as P
    from pygho_amd import _ops
    dd = baseline_batch
    acd = dd[KEY + "___acd"]
    nt, ne, d = dd["X"].nnz, dd["A"].nnz, 128
    plan = _ops.message_plan(acd, nt, nt, ne)
    sp = _ops.scatter_plan(plan, on_demand=True)
    assert sp is not None and sp.cgap is not None and sp.covers_c, "the store-collated batch must come with aligned chunks"
    gen = torch.Generator().manual_seed(5)
    gh_ = torch.randn(nt, d, generator=gen).to(dtype)
    hh = torch.randn(nt, d, generator=gen).to(dtype)
    th = torch.randn(16, d, generator=gen).to(dtype)
    ah = torch.randn(ne, d, generator=gen).to(dtype)
    g, h, table, addend = (t.to(dev) for t in (gh_, hh, th, ah))
    look_byc = plan.lookup(_ops.flat_index(dd["A"].values))[1]
    assert _ops.dual_eligible(plan, g, h, table, None)
    acd_h = acd.cpu()
    b32 = th.float()[dd["A"].values.cpu().long()]
    want_gh = P.spspmm_values_chunked(gh_.float(), b32, acd_h[1], acd_h[0], acd_h[2], nt, "sum").to(dtype)
    sums = P.spspmm_values_chunked(gh_.float(), hh.float(), acd_h[2], acd_h[0], acd_h[1], ne, "sum")
    # the table-gradient form: gh the same bits; the table's gradient against the port's f32 sums by looked-up row (another summation
    # order: f32 accuracy)
    idx = _ops.flat_index(dd["A"].values)
    if _ops.dual_tg_eligible(plan, g, h, table, None, idx):
        look_fwd = plan.lookup(idx)[0]
        tg_gh, tg_gt = _ops.dual_backward_tg(plan, g, h, table, look_fwd, look_byc)
        assert torch.equal(tg_gh.cpu(), want_gh), "table-gradient form: by-tuple gradient"
        look_of_msg = dd["A"].values.cpu().long()[acd_h[2]]
        want_gt = P.spspmm_values_chunked(gh_.float(), hh.float(), look_of_msg, acd_h[0], acd_h[1], 16, "sum")
        torch.testing.assert_close(tg_gt.cpu(), want_gt, rtol=2e-4, atol=2e-4 * float(want_gt.abs().max()))
    for chained in (False, True):
        got_gh, got_gr = _ops.dual_backward(plan, g, h, table, look_byc, addend=addend if chained else None)
        want_gr = ((ah.float() + sums) if chained else sums).to(dtype)
        for name, got, want in (("by-tuple", got_gh, want_gh), ("by-edge", got_gr, want_gr)):
            got = got.cpu()
            if not torch.equal(got, want):
                bad = got != want
                raise AssertionError(f"{name} gradient (chained={chained}): {int(bad.sum())} of {bad.numel()} elements differ from the host oracle")


def _train_step(model, dd):
    for p in model.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(dd)
    loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
    loss.backward()
    return loss.detach().clone(), [p.grad.clone() for p in model.parameters()]


def test_training_step_with_and_without_the_dual_backward_is_bit_identical(dev, small_plans):
    """three runs of the same two steps from the same state: the two launches (PYGHO_DUAL_BWD=0), the fused backward with the per-edge
    gradient (PYGHO_DUAL_TABLE_GRAD=0) and the fused backward in its table-gradient form (shipped).  Loss and EVERY gradient of the first
    two agree bit for bit; the third agrees bit for bit except for the edge-feature embedding table, whose gradient it accumulates
    itself -- f32 sums of exact products instead of per-edge rows rounded to bf16 and chained through six layers: closer to an f64
    evaluation, compared with a tolerance"""
    from pygho_amd import _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.ngnn import SpModel
    rng = np.random.default_rng(9)
    store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(160)], dev)
    assert "cgap" in store.scatter_parts[KEY]
    torch.manual_seed(1)
    model = SpModel(1, 3, 128, act_dtype=torch.bfloat16).to(dev)
    model.train()
    names_p = [k for k, _ in model.named_parameters()]
    ea_name = "data_encoder.ea_encoder.weight"
    assert ea_name in names_p
    state = {k: v.clone() for k, v in model.state_dict().items()}
    timer = _ops.LaunchTimer()
    old = (_ops.DUAL_BWD, _ops.DUAL_TABLE_GRAD)
    try:
        res = {}
        for mode, (dual, tg) in (("two launches", (False, False)), ("dual", (True, False)), ("dual + table gradient", (True, True))):
            _ops.DUAL_BWD, _ops.DUAL_TABLE_GRAD = dual, tg
            model.load_state_dict(state)
            out = []
            with timer:
                for seed in (0, 1):
                    ids = np.random.default_rng(seed).permutation(160)[:96]
                    out.append(_train_step(model, store.collate(ids)))
            res[mode] = out
    finally:
        _ops.DUAL_BWD, _ops.DUAL_TABLE_GRAD = old
    torch.cuda.synchronize()
    names = set(timer.summary())
    assert any(k.startswith("seg_dual[") and "table" not in k for k in names) and any(k.endswith(",table]") for k in names), names
    for step in range(2):
        l0, g0 = res["two launches"][step]
        for mode in ("dual", "dual + table gradient"):
            l1, g1 = res[mode][step]
            assert torch.equal(l1, l0), mode
            for name, a, b in zip(names_p, g1, g0):
                if name == ea_name and mode.endswith("table gradient"):
                    scale = float(b.abs().max())
                    torch.testing.assert_close(a, b, rtol=0, atol=2.0 ** -7 * scale, msg=f"{mode}: {name}")
                    assert not torch.equal(a, b)                 # it IS another (more accurate) summation
                else:
                    assert torch.equal(a, b), (mode, name)


def test_table_gradient_form_vs_f64_and_the_by_tuple_bits(dev, small_plans):
    """`dual_backward_tg` on a store-collated batch (the index carries the store's value bound): gh has the bits of the by-tuple launch;
    the table gradient equals sum_m [look_m = t] g[a_m] * h[c_m] evaluated in f64 to f32 accuracy -- and is CLOSER to it than the route it
    replaces (per-edge gradient rounded to bf16, then `table_grad`)"""
    from pygho_amd import _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    rng = np.random.default_rng(3)
    store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(200)], dev)
    dd = store.collate(np.arange(200))
    acd = dd[KEY + "___acd"]
    nt, ne, d = dd["X"].nnz, dd["A"].nnz, 128
    plan = _ops.message_plan(acd, nt, nt, ne)
    idx = _ops.flat_index(dd["A"].values)
    torch.manual_seed(0)
    g = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    h = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    table = torch.randn(16, d, device=dev).to(torch.bfloat16)
    look_fwd, look_byc = plan.lookup(idx)
    assert _ops.dual_tg_eligible(plan, g, h, table, None, idx)
    gh, gt = _ops.dual_backward_tg(plan, g, h, table, look_fwd, look_byc)
    pc, a_byc, _ = plan.by_c()
    assert torch.equal(gh, _ops.seg_gmr(nt, g, table, pc.seg_ptr, a_byc, look_byc, "sum"))
    prod = g.double()[acd[0]] * h.double()[acd[1]]
    want = torch.zeros(16, d, dtype=torch.float64, device=dev).index_add_(0, dd["A"].values[acd[2]], prod)
    scale = float(want.abs().max())
    err_new = float((gt.double() - want).abs().max()) / scale
    old_route = _ops.table_grad(_ops.by_edge_product(plan, g, h, None), idx, 16)
    err_old = float((old_route.double() - want).abs().max()) / scale
    print(f"table gradient: max error / max |value| new {err_new:.2e}, per-edge route {err_old:.2e}")
    assert err_new < 1e-5 and err_new <= err_old
    assert bool((gt[4:] == 0).all())
    # an index without the store's bound keeps the per-edge form
    assert not _ops.dual_tg_eligible(plan, g, h, table, None, idx.clone())


def test_the_table_gradient_form_leaves_the_lookup_backward_alone(dev, small_plans):
    """with the table's gradient returned by the layers themselves nothing travels back through A.values: the lookup's backward must
    not run on an all-zero gradient (autograd materialises one for a custom function unless told otherwise) -- one small-table reduction
    per step (the node features'), not two; and the fold of the kernel's 4-row slabs writes the table's other rows as zero itself"""
    from pygho_amd import _ops, blocks, synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.ngnn import SpModel
    rng = np.random.default_rng(4)
    store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(128)], dev)
    torch.manual_seed(2)
    model = SpModel(1, 2, 128, act_dtype=torch.bfloat16).to(dev)
    model.train()
    dd = store.collate(np.arange(96))
    counts = {}
    for tg in (False, True):
        old = _ops.DUAL_TABLE_GRAD
        _ops.DUAL_TABLE_GRAD = tg
        try:
            timer = _ops.LaunchTimer()
            with timer:
                _train_step(model, dd)
            torch.cuda.synchronize()
            summ = timer.summary()
            counts[tg] = sum(v[0] for k, v in summ.items() if k.startswith("table_grad["))
            assert any(k.endswith(",table]") for k in summ) == tg
        finally:
            _ops.DUAL_TABLE_GRAD = old
    assert counts == {False: 2, True: 1}, counts
    g = model.data_encoder.ea_encoder.weight.grad
    assert g is not None and bool((g[4:] == 0).all()) and float(g[:4].abs().max()) > 0
    # the fold on its own: (blocks, n) -> n sums followed by zeros, the sums being sum_blocks' bits
    parts = torch.randn(37, 512, device=dev)
    padded = blocks.sum_blocks(parts, 2048)
    assert padded.shape == (2048,) and torch.equal(padded[:512], blocks.sum_blocks(parts)) and bool((padded[512:] == 0).all())
    assert torch.equal(blocks.sum_blocks(parts, 512), blocks.sum_blocks(parts))

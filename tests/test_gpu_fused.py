"""
The layer MLP folded into the forward aggregation's load path (`csrc/seg_fused.hip`, `segment.fused_forward`): reference
NGNNConv.forward (pygho/honn/Conv.py:53-58) + the model loop's residual add (example/minimal.py:76-79).

* the planner's chunks tile the output rows, respect the kernel's limits, and every first-operand row a message reads is owned by
  exactly one chunk;
* the fused launch == `rowblock_linear_bn_act` followed by `seg_gmr` (+ residual row) bit for bit -- output AND the stored H rows --
  over activations, aggregations, residual on / off, table sizes, bf16 / f16;
* a plan outside the limits takes the separate kernels;
* a whole NGNN training step with the fused forward == the step without it, bit for bit (loss, every gradient), on store-collated
  batches and inside a captured batch slot.
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


@pytest.fixture(scope="module")
def batch(dev):
    from pygho_amd import _ops, synth
    hb = synth.make_batch(96, "zinc", seed=21)
    acd = torch.from_numpy(hb.acd[KEY]).to(dev)
    plan = _ops.message_plan(acd, hb.num_tuples, hb.num_tuples, hb.num_edges)
    ea = torch.from_numpy(hb.edge_attr).to(dev).long()
    return hb, plan, ea


def _operands(dev, n, d, dtype, table_rows, bias=True, seed=0):
    torch.manual_seed(seed)
    x = torch.randn(n, d, device=dev).to(dtype)
    wl = (torch.randn(d, d, device=dev) / d ** 0.5).to(dtype)
    b = (torch.randn(d, device=dev) * 0.1).to(dtype) if bias else None
    scale = (torch.rand(d, device=dev) + 0.5).float()
    shift = (torch.randn(d, device=dev) * 0.1).float()
    table = torch.randn(table_rows, d, device=dev).to(dtype)
    return x, wl, b, scale, shift, table


def _limits():
    import ctypes
    from pygho_amd._native import lib
    v = [ctypes.c_int(0) for _ in range(4)]
    assert lib().pygho_seg_fused_limits(*[ctypes.byref(x) for x in v]) == 0
    return tuple(int(x.value) for x in v)          # messages per chunk, rows per chunk, table rows, width


def test_planner_chunks_tile_the_rows_within_the_limits(dev, batch):
    from pygho_amd import _ops
    max_msgs, max_rows, _, _ = _limits()
    hb, plan, _ = batch
    fp = _ops.fused_plan(plan)
    assert fp is not None
    ch = fp.chunks.cpu().numpy().astype(np.int64)
    own = fp.own.cpu().numpy().astype(np.uint32)
    seg = plan.fwd.seg_ptr.cpu().numpy().astype(np.int64)
    c = plan.c_fwd.cpu().numpy().astype(np.int64)
    m_lo, a_lo, c_lo = ch[:, 0], ch[:, 1], ch[:, 2]
    msgs, rows, crow = ch[:, 3] & 0xff, (ch[:, 3] >> 8) & 0xff, (ch[:, 3] >> 16) & 0xff
    assert rows.min() >= 1 and rows.max() <= max_rows and msgs.max() <= max_msgs and crow.max() <= max_rows
    assert a_lo[0] == 0 and np.array_equal(a_lo[1:], (a_lo + rows)[:-1]) and a_lo[-1] + rows[-1] == plan.n_out       # tile [0, n)
    assert np.array_equal(m_lo, seg[a_lo]) and np.array_equal(m_lo + msgs, seg[a_lo + rows])                          # whole rows only
    owner_count = np.zeros(plan.n_lhs, dtype=np.int64)
    for k in range(ch.shape[0]):
        cs = c[m_lo[k]:m_lo[k] + msgs[k]]
        if cs.size:
            assert cs.min() >= c_lo[k] and cs.max() < c_lo[k] + crow[k]
        bits = [i for i in range(32) if (int(own[k]) >> i) & 1]
        assert all(i < crow[k] for i in bits)
        owner_count[c_lo[k] + np.asarray(bits, dtype=np.int64)] += 1
    read = np.unique(c)
    assert np.all(owner_count[read] == 1) and owner_count.max() <= 1        # every row a message reads: exactly one owner


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act,aggr,residual,table_rows,bias", [("silu", "sum", True, 16, True), ("relu", "mean", True, 32, True),
                                                              ("none", "sum", False, 5, False), ("silu", "mean", False, 1, True)])
def test_fused_forward_equals_the_two_launches_bitwise(dev, batch, dtype, act, aggr, residual, table_rows, bias):
    from pygho_amd import _ops
    hb, plan, ea = batch
    n, d = hb.num_tuples, 128
    x, wl, b, scale, shift, table = _operands(dev, n, d, dtype, table_rows, bias)
    look = _ops.narrow_i32(ea % table_rows)
    look_fwd = _ops.gather_i32(look, plan.d_fwd)
    fp = _ops.fused_plan(plan)
    h_ref = _ops.rowblock_linear_bn_act(x, wl, b, scale, shift, act)
    o_ref = _ops.seg_gmr(n, h_ref, table, plan.fwd.seg_ptr, plan.c_fwd, look_fwd, aggr, addend=x if residual else None)
    for want_h in (True, False):
        o, h = _ops.fused_forward(x, wl, b, scale, shift, act, table, look_fwd, plan, fp, aggr, residual, want_h)
        assert torch.equal(o, o_ref)
        if want_h:
            read = torch.unique(plan.c_fwd.long())
            assert torch.equal(h[read], h_ref[read])
        else:
            assert h is None


def test_rows_outside_the_limits_keep_the_separate_kernels(dev):
    """one output row with more messages than a chunk holds, and one whose first-operand rows are 40 apart (> 31): no fused plan"""
    from pygho_amd import _ops
    n, over = 400, _limits()[0] + 6
    for a, c in (([5] * over, list(range(over))), ([7, 7], [3, 43])):
        acd = torch.tensor([a, c, [0] * len(a)], dtype=torch.int64, device=dev)
        plan = _ops.message_plan(acd, n, n, 4)
        assert _ops.fused_plan(plan) is None
        assert _ops.fused_plan(plan, on_demand=True) is None


def test_rows_without_messages_and_sparse_patterns(dev):
    """most output rows have no message at all (they are the residual row alone), chunks without any message, one row with the maximum
    number of messages, a table with a single row"""
    from pygho_amd import _ops
    max_msgs = _limits()[0]
    n, d = 1500, 128
    a = [3, 3, 10, 50, 51, 51, 51, 700] + [900] * max_msgs + [1499]
    c = [0, 7, 12, 40, 60, 61, 52, 690] + [880 + (i % 30) for i in range(max_msgs)] + [1490]
    acd = torch.tensor([a, c, [0] * len(a)], dtype=torch.int64, device=dev)
    plan = _ops.message_plan(acd, n, n, 3)
    fp = _ops.fused_plan(plan)
    assert fp is not None
    x, wl, b, scale, shift, table = _operands(dev, n, d, torch.bfloat16, 1)
    look_fwd = torch.zeros(len(a), dtype=torch.int32, device=dev)
    h_ref = _ops.rowblock_linear_bn_act(x, wl, b, scale, shift, "silu")
    for aggr in ("sum", "mean"):
        o_ref = _ops.seg_gmr(n, h_ref, table, plan.fwd.seg_ptr, plan.c_fwd, look_fwd, aggr, addend=x)
        o, h = _ops.fused_forward(x, wl, b, scale, shift, "silu", table, look_fwd, plan, fp, aggr, True, True)
        assert torch.equal(o, o_ref)
        read = torch.unique(plan.c_fwd.long())
        assert torch.equal(h[read], h_ref[read])


def _train_step(model, dd):
    for p in model.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pred = model(dd)
    loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
    loss.backward()
    return loss.detach().clone(), [p.grad.clone() for p in model.parameters()]


def test_training_step_with_and_without_the_fused_forward_is_bit_identical(dev):
    from pygho_amd import _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.ngnn import SpModel
    rng = np.random.default_rng(9)
    store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(160)], dev)
    assert KEY in store.fused_parts
    torch.manual_seed(1)
    model = SpModel(1, 3, 128, act_dtype=torch.bfloat16).to(dev)
    model.train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    timer = _ops.LaunchTimer()
    old = _ops.FUSED_FWD
    try:
        res = {}
        for fused in (True, False):
            _ops.FUSED_FWD = fused
            model.load_state_dict(state)
            out = []
            with timer:
                for seed in (0, 1):
                    ids = np.random.default_rng(seed).permutation(160)[:96]
                    out.append(_train_step(model, store.collate(ids)))
            res[fused] = out
    finally:
        _ops.FUSED_FWD = old
    torch.cuda.synchronize()
    names = set(timer.summary())
    assert any(k.startswith("seg_fused[") for k in names), names            # the fused kernel did run
    for (l1, g1), (l0, g0) in zip(res[True], res[False]):
        assert torch.equal(l1, l0)
        for a, b in zip(g1, g0):
            assert torch.equal(a, b)


def test_captured_slot_step_runs_the_fused_forward_and_matches_the_unfused_eager_step(dev):
    from pygho_amd import _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.graphs import SlotStep
    from pygho_amd.ngnn import SpModel
    import copy
    rng = np.random.default_rng(10)
    store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(200)], dev)
    torch.manual_seed(2)
    model = SpModel(1, 2, 128, act_dtype=torch.bfloat16).to(dev)
    ref = copy.deepcopy(model)

    def stepper(m):
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, capturable=True)

        def step(dd):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = m(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            opt.step()
            return loss.detach()
        return step
    ids = [np.random.default_rng(30 + k).permutation(200)[:64] for k in range(6)]
    old = _ops.FUSED_FWD
    try:
        _ops.FUSED_FWD = False
        eager = stepper(ref)
        for _ in range(3):
            eager(store.collate(ids[0]))
        want = [eager(store.collate(i)).clone() for i in ids[1:]]
        _ops.FUSED_FWD = True
        ss = SlotStep(store, 64, stepper(model), warmup_ids=ids[0], warmup=3)
        assert any("fu_chunks" in ent for ent in ss.slot.msg.values())
        got = [ss.run(i).clone() for i in ids[1:]]
    finally:
        _ops.FUSED_FWD = old
    assert ss.replays == len(ids) - 1 and ss.eager_steps == 0
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    for p, q in zip(model.parameters(), ref.parameters()):
        assert torch.equal(p, q)


# ------------------------------------------------------------------------------------------------------------------------------------
# the fused forward against the ORACLE at BASELINE size: not against the launches it replaces
# ------------------------------------------------------------------------------------------------------------------------------------
def _ordered(t):
    """storage-type values (bf16 / f16) as integers in value order: neighbouring representable values differ by 1 (-0 == +0)"""
    bits = t.contiguous().view(torch.int16).to(torch.int32) & 0xffff
    return torch.where(bits >= 0x8000, 0x8000 - bits, bits)


def _neighbours(v64, dtype):
    """the two storage-type values that bracket the f64 values (equal where v64 is representable)"""
    near = v64.to(dtype)
    nf = near.double()
    bits = near.view(torch.int16)
    # one ulp towards the other side of v64: +1 on the bit pattern moves away from zero, -1 towards it
    away = (nf.abs() < v64.abs())
    other = torch.where(nf == v64, bits, torch.where(away, bits + 1, bits - 1))
    # crossing zero (near == +-0 and v64 on the other side does not happen: near is the NEAREST value); bits - 1 on +-0 never taken
    # because |near| < |v64| whenever near == 0 and v64 != 0
    return near, other.view(dtype)


@pytest.fixture(scope="module")
def baseline_store(dev):
    """BASELINE config 2's batch: 8192 distinct ZINC-shape graphs, collated by the device store WITH every plan (the fused chunks
    are the store's precomputed per-graph chunks with offsets added -- the plan the benchmark's step runs on)"""
    from pygho_amd import synth
    from pygho_amd.collate import DeviceGraphStore
    rng = np.random.default_rng(1000)
    store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(8192)], dev)
    dd = store.collate(np.random.default_rng(7).permutation(8192))
    return store, dd


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_forward_at_baseline_size_vs_host_oracle(dev, baseline_store, dtype):
    """`seg_fused_fwd_kernel` (the kernel bench.py's roofline reports) at 8192 ZINC-shape graphs, width 128, on the store-collated
    plan, against the HOST:

    * the stored H rows against act(bn(x W^T + b)) (reference honn/utils.py:126-142 via Conv.py:56) evaluated in f64.  The device
      rounds the Linear's output to the storage type before BatchNorm (as the reference's autocast does: the Linear's output IS a
      16-bit tensor), so the f64 pre-activation's TWO neighbouring storage values are the admissible roundings: every H element must be
      within 1 storage-ulp of H evaluated at one of them, and >= 99.9 % within 1 ulp of H evaluated at the NEAREST one;
    * `out` BIT FOR BIT against round(x + aggr_{(a,c,d)} H_dev[c] * table[look[d]]) where the aggregation is the reference's ATen
      sequence on the host (oracle.aten_port.spspmm_values_chunked, Spspmm.py:309-315: index, index, mul, index_add_ in message
      order) over the DEVICE's stored H: f32 products of two 16-bit values are exact, so the f32 sums must agree exactly and the
      result is their single rounding; sum and mean; the residual add of example/minimal.py:80."""
    from oracle import aten_port as P
    from pygho_amd import _ops
    _store, dd = baseline_store
    acd = dd[KEY + "___acd"]
    nt, ne, d = dd["X"].nnz, dd["A"].nnz, 128
    plan = _ops.message_plan(acd, nt, nt, ne)
    fp = _ops.fused_plan(plan, on_demand=True)
    assert fp is not None and fp.n_chunks > 0, "the store-collated batch must come with its fused chunks"
    gen = torch.Generator().manual_seed(3)
    xh = torch.randn(nt, d, generator=gen).to(dtype)
    wh = (torch.randn(d, d, generator=gen) / d ** 0.5).to(dtype)
    bh = (torch.randn(d, generator=gen) * 0.1).to(dtype)
    scale_h = (torch.rand(d, generator=gen) + 0.5).float()
    shift_h = (torch.randn(d, generator=gen) * 0.3).float()
    table_h = torch.randn(16, d, generator=gen).to(dtype)
    x, wl, b, scale, shift, table = (t.to(dev) for t in (xh, wh, bh, scale_h, shift_h, table_h))
    ea = _ops.flat_index(dd["A"].values)
    look_fwd = plan.lookup(ea)[0]
    acd_h = acd.cpu()
    ea_h = dd["A"].values.cpu().long()
    read = torch.unique(acd_h[1])
    timer = _ops.LaunchTimer()
    res = {}
    with timer:
        for aggr in ("sum", "mean"):
            res[aggr] = _ops.fused_forward(x, wl, b, scale, shift, "silu", table, look_fwd, plan, fp, aggr, True, True)
    torch.cuda.synchronize()
    assert any(k.startswith("seg_fused[") for k in timer.summary())
    h_dev = res["sum"][1].cpu()
    assert torch.equal(res["mean"][1].cpu()[read], h_dev[read])
    # ---- H against the f64 evaluation, in slabs of rows
    ulp_near_bad, worst, n_checked = 0, 0, 0
    w64, b64, s64, t64 = wh.double(), bh.double(), scale_h.double(), shift_h.double()
    silu = lambda z: z / (1.0 + torch.exp(-z))
    for lo in range(0, read.numel(), 1 << 17):
        rows = read[lo:lo + (1 << 17)]
        pre = xh[rows].double() @ w64.t() + b64
        near, other = _neighbours(pre, dtype)
        got = _ordered(h_dev[rows])
        dist = []
        for cand in (near, other):
            want = silu(cand.double() * s64 + t64).to(dtype)
            dist.append((got - _ordered(want)).abs())
        ulp_near_bad += int((dist[0] > 1).sum())
        worst = max(worst, int(torch.minimum(dist[0], dist[1]).max()))
        n_checked += rows.numel() * d
    frac_bad = ulp_near_bad / n_checked
    print(f"fused forward {dtype}: H over {n_checked} elements: {frac_bad:.2e} beyond 1 ulp of the nearest-rounding evaluation, "
          f"worst distance to an admissible evaluation {worst} ulp")
    assert worst <= 1, f"H: an element {worst} storage-ulps from both admissible f64 evaluations"
    assert frac_bad <= 1e-3, f"H: {frac_bad:.2e} of the elements beyond 1 ulp of the f64 evaluation"
    # ---- out against the ATen sequence over the device's H (rows no message reads are never stored: zero them for the host gather)
    h32 = torch.zeros(nt, d)
    h32[read] = h_dev[read].float()
    b32 = table_h.float()[ea_h]
    for aggr in ("sum", "mean"):
        agg = P.spspmm_values_chunked(h32, b32, acd_h[0], acd_h[1], acd_h[2], nt, aggr)
        want = (xh.float() + agg).to(dtype)
        got = res[aggr][0].cpu()
        if not torch.equal(got, want):
            bad = got != want
            raise AssertionError(f"{aggr}: {int(bad.sum())} of {bad.numel()} output elements differ from the host oracle over the device's H")


def test_explicit_lookup_argument_and_path_report(dev):
    """the fused dispatch used to hang on a private tensor attribute and nothing said which path ran.  Now the
    provenance of A's values is an EXPLICIT argument of `forward_residual` (`adj_lookup=(table, index)`), and
    `_ops.record_block_paths()` reports per block what was dispatched and which condition kept a faster path out: a plain gathered
    A (no attribute, no argument) takes the separate launches and says why; the same call with `adj_lookup` runs seg_fused forward and
    seg_dual backward -- same output bits either way."""
    from pygho_amd import SparseTensor, _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.honn.Conv import NGNNConv
    rng = np.random.default_rng(4)
    store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(128)], dev)
    dd = store.collate(np.arange(128))
    h = 128
    torch.manual_seed(0)
    layer = NGNNConv(h, h, "sum", "SS", {"numlayer": 1, "tailact": True, "norm": "bn", "act": "silu"}).to(dev).train()
    table = torch.randn(16, h, device=dev).to(torch.bfloat16).requires_grad_(True)
    idx = _ops.flat_index(dd["A"].values)
    n = int(dd["num_nodes"])
    xv = torch.randn(dd["X"].nnz, h, device=dev).to(torch.bfloat16)
    old = _ops.SEG_SCATTER_MIN_MESSAGES
    _ops.SEG_SCATTER_MIN_MESSAGES = 0
    try:
        outs = []
        for explicit in (False, True):
            av = table[dd["A"].values]                       # a plain gather: no provenance on the tensor
            A = SparseTensor(dd["A"].indices, av, [n, n, h], True)
            X = SparseTensor(dd["X"].indices, xv.clone().requires_grad_(True), [n, n, h], True)
            with _ops.record_block_paths() as log:
                out = layer.forward_residual(A, X, dd, adj_lookup=(table.detach(), idx) if explicit else None)
                out.values.float().square().mean().backward()
            torch.cuda.synchronize()
            outs.append(out.values.detach().clone())
            fwd = [e for e in log if "forward" in e][0]
            bwd = [e for e in log if "backward" in e][0]
            if explicit:
                assert fwd == {"forward": "seg_fused"} and bwd == {"backward": "seg_dual"}, log
            else:
                assert fwd["forward"] == "rowblock_linear + seg_gmr" and any("adj_lookup" in w for w in fwd["forward_not_fused_because"]), log
                assert bwd["backward"] == "seg_gmr + by_edge_product" and bwd["backward_not_dual_because"], log
        assert torch.equal(outs[0], outs[1])
    finally:
        _ops.SEG_SCATTER_MIN_MESSAGES = old

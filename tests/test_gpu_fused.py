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

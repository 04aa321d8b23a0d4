"""
Fixed-capacity batch slots and the one captured step that serves every batch (`pygho_amd.slots`, `pygho_amd.graphs.SlotStep`):
the reference's loop draws a fresh shuffled batch per step (example/minimal.py:119, :141-149; hodata/SpData.py:60-77).

* the slot's arrays equal `DeviceGraphStore.collate`'s on the batch's true extents (bit-exact), pad columns are inert;
* an eager step on the slot == an eager step on the exactly sized batch, bit for bit (loss, every gradient);
* ONE captured step replayed on >= 20 different batches == the eager step on each batch, bit for bit (loss, every gradient, every
  parameter after the optimizer step), including the BatchNorm running statistics;
* a batch that does not fit takes the eager path.
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
def store(dev):
    from pygho_amd import synth
    from pygho_amd.collate import DeviceGraphStore
    rng = np.random.default_rng(5)
    recs = [synth.make_graph(rng, "zinc", 3, (KEY,)) for _ in range(384)]
    return DeviceGraphStore(recs, dev)


def _model(dev, layers=3, hidden=128, seed=0):
    from pygho_amd.ngnn import SpModel
    torch.manual_seed(seed)
    return SpModel(1, layers, hidden, act_dtype=torch.bfloat16).to(dev)


def _make_step(model, opt):
    def step(dd):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        opt.step()
        return loss.detach()
    return step


def _batches(n_store, g, count, seed=3):
    rng = np.random.default_rng(seed)
    return [rng.permutation(n_store)[:g] for _ in range(count)]


def test_slot_arrays_equal_the_exact_collation(dev, store):
    from pygho_amd.slots import BatchSlot
    g = 48
    slot = BatchSlot(store, g)
    for ids in _batches(store.num_graphs, g, 4):
        assert slot.fits(ids)
        dd_s = slot.collate(ids)
        dd_e = store.collate(ids)
        sizes = slot.true_sizes()
        n, e, t, m = sizes["node"], sizes["edge"], sizes["tup"], sizes[("acd", KEY)]
        assert n == dd_e["num_nodes"] and e == dd_e["A"].nnz and t == dd_e["X"].nnz and m == dd_e[KEY + "___acd"].shape[1]
        assert torch.equal(dd_s["x"][:n], dd_e["x"]) and torch.equal(dd_s["batch"][:n], dd_e["batch"]) and torch.equal(dd_s["y"], dd_e["y"])
        assert torch.equal(dd_s["A"].indices[:, :e], dd_e["A"].indices) and torch.equal(dd_s["A"].values[:e], dd_e["A"].values)
        assert torch.equal(dd_s["X"].indices[:, :t], dd_e["X"].indices) and torch.equal(dd_s["X"].values[:t], dd_e["X"].values)
        assert torch.equal(dd_s[KEY + "___acd"][:, :m], dd_e[KEY + "___acd"])
        # pad columns: valid row 0 for index arrays, the totals for CSR pointers (empty pad segments)
        assert int(dd_s["x"][n:].abs().sum()) == 0 and int(dd_s["X"].indices[:, t:].abs().sum()) == 0
        ent = slot.msg[KEY]
        from pygho_amd import _ops
        plan_e = _ops.message_plan(dd_e[KEY + "___acd"], t, t, e)
        assert torch.equal(ent["ptr_a"][:t + 1], plan_e.fwd.seg_ptr) and bool((ent["ptr_a"][t:] == m).all())
        pc, a_c, d_c = plan_e.by_c()
        assert torch.equal(ent["ptr_c"][:t + 1], pc.seg_ptr) and torch.equal(ent["perm_c"][:m], pc.perm)
        assert torch.equal(ent["by_c"][0, :m], a_c) and torch.equal(ent["by_c"][1, :m], d_c)
        pd, a_d, c_d = plan_e.by_d()
        assert torch.equal(ent["ptr_d"][:e + 1], pd.seg_ptr) and bool((ent["ptr_d"][e:] == m).all())
        assert torch.equal(ent["by_d"][0, :m], a_d) and torch.equal(ent["by_d"][1, :m], c_d)
        assert torch.equal(ent["acd32"][:, :m].to(torch.int64), dd_e[KEY + "___acd"])
        look_f, look_c = plan_e.lookup(dd_e["A"].values)
        assert torch.equal(ent["look"][0, :m], look_f) and torch.equal(ent["look"][1, :m], look_c)
        assert torch.equal(slot.graph_ptr.to(torch.int64), torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                                                                       torch.bincount(dd_e["batch"], minlength=g).cumsum(0)]))


def _grads(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def _assert_same(a, b, what):
    assert a.keys() == b.keys(), what
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    if bad:
        k = bad[0]
        diff = (a[k].float() - b[k].float()).abs().max().item()
        scale = b[k].float().abs().max().item()
        raise AssertionError(f"{what}: {len(bad)} of {len(a)} tensors differ, first {k}: max |diff| {diff:.3e} at scale {scale:.3e}")


def test_eager_step_on_the_slot_is_bit_identical_to_the_exact_batch(dev, store):
    """forward loss, every parameter gradient and the updated BatchNorm running statistics of ONE step from the same model state"""
    from pygho_amd.slots import BatchSlot
    g = 48
    slot = BatchSlot(store, g)
    for ids in _batches(store.num_graphs, g, 3, seed=11):
        res = []
        for mode in ("slot", "exact"):
            model = _model(dev)
            opt = torch.optim.SGD(model.parameters(), lr=0.0)
            step = _make_step(model, opt)
            if mode == "slot":
                dd = slot.collate(ids)
                with slot.rows():
                    loss = step(dd)
            else:
                loss = step(store.collate(ids))
            stats = {k: v.detach().clone() for k, v in model.state_dict().items() if "running" in k}
            res.append((loss.clone(), _grads(model), stats))
        assert torch.equal(res[0][0], res[1][0]), (float(res[0][0]), float(res[1][0]))
        _assert_same(res[0][1], res[1][1], "gradients (slot vs exact batch)")
        _assert_same(res[0][2], res[1][2], "running statistics (slot vs exact batch)")


def test_one_captured_step_serves_twenty_four_batches_bit_for_bit(dev, store):
    """ONE capture, 24 different batches: loss, every gradient and every parameter after AdamW equal the eager loop's on
    `store.collate` batches, bit for bit, at every step (so the two trajectories never separate)."""
    from pygho_amd.graphs import SlotStep
    g, n_steps = 48, 24
    batches = _batches(store.num_graphs, g, n_steps, seed=7)
    warm = _batches(store.num_graphs, g, 1, seed=99)[0]
    # eager reference: the same warm-up steps SlotStep takes (3 on the warm-up batch), then the sequence
    ref_model = _model(dev)
    ref_opt = torch.optim.AdamW(ref_model.parameters(), lr=1e-3, capturable=True)
    ref_step = _make_step(ref_model, ref_opt)
    for _ in range(3):
        ref_step(store.collate(warm))
    ref = []
    for ids in batches:
        loss = ref_step(store.collate(ids))
        ref.append((loss.clone(), _grads(ref_model), {k: v.detach().clone() for k, v in ref_model.state_dict().items()}))
    model = _model(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
    ss = SlotStep(store, g, _make_step(model, opt), warmup_ids=warm, warmup=3)
    for k, ids in enumerate(batches):
        loss = ss.run(ids)
        assert torch.equal(loss, ref[k][0]), (k, float(loss), float(ref[k][0]))
        _assert_same(_grads(model), ref[k][1], f"gradients at step {k}")
        _assert_same({kk: v.detach() for kk, v in model.state_dict().items()}, ref[k][2], f"model state after step {k}")
    assert ss.replays == n_steps and ss.eager_steps == 0
    assert float(ref[-1][0]) < float(ref[0][0]) * 1.5 and all(bool(torch.isfinite(r[0])) for r in ref)


def test_a_batch_that_does_not_fit_runs_eagerly(dev, store):
    from pygho_amd.graphs import SlotStep
    from pygho_amd.slots import slot_capacities
    g = 48
    caps = slot_capacities(store, g, sigmas=0.0)          # capacity = the mean batch: about half of all batches overflow
    model = _model(dev, layers=2)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
    order = np.argsort(np.asarray(store.h_len["tup"]))
    small, large = order[:g], order[-g:]
    ss = SlotStep(store, g, _make_step(model, opt), warmup_ids=small, capacities=caps)
    assert ss.slot.fits(small) and not ss.slot.fits(large)
    l0 = float(ss.run(small))
    l1 = float(ss.run(large))
    l2 = float(ss.run(large[: g // 2]))                    # another batch size: eager as well
    assert ss.replays == 1 and ss.eager_steps == 2 and all(np.isfinite(v) for v in (l0, l1, l2))


def test_row_reduction_without_a_device_count_form_refuses_slot_rows(dev, store):
    from pygho_amd import _ops
    from pygho_amd.slots import BatchSlot
    slot = BatchSlot(store, 48)
    slot.collate(_batches(store.num_graphs, 48, 1)[0])
    n = slot.caps["node"]
    g = torch.randn((n, 96), device=dev, dtype=torch.bfloat16)
    x = torch.randn((n, 96), device=dev, dtype=torch.bfloat16)
    with slot.rows():
        with pytest.raises(RuntimeError, match="device-side row-count"):
            _ops.weight_grad_splitk(g, x, torch.float32)
    _ops.weight_grad_splitk(g, x, torch.float32)           # outside a slot context the capacity is just a row count


@pytest.mark.parametrize("family", ["NGNN", "SSWL", "DSSGNN", "GNNAK", "SUN", "I2GNN"])
def test_captured_slot_step_serves_every_two_tuple_family(dev, family):
    """(round 6: and the 3-tuple family I2GNN, whose merged (i, j) pooling pattern now comes with the slot)
    the model of example/zinc.py:222-297 for each 2-tuple layer family (`pygho_amd.models.SpModel`: max subgraph pooling, mean graph
    pooling, cross-subgraph pooling / unpooling, GNNAK's and SUN's diagonal views): ONE captured step, 6 different batches, bit for bit
    against the eager loop on exactly sized batches -- loss, every gradient, the model state after AdamW.  (The diagonal positions and
    per-node tuple counts those two layers search by hash / bincount per batch come with the slot; hash-searching operators on a
    padded pattern raise instead of answering wrongly.)"""
    from pygho_amd import synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.graphs import SlotStep
    from pygho_amd.honn.SpOperator import parse_precomputekey
    from pygho_amd.models import SpModel
    g = 48

    def make():
        torch.manual_seed(1)
        return SpModel(family, num_layer=2, hiddim=128, act_dtype=torch.bfloat16).to(dev)
    keys = tuple(parse_precomputekey(make()))
    rng = np.random.default_rng(9)
    st = DeviceGraphStore([synth.make_graph(rng, "i2" if family == "I2GNN" else "zinc", 3, keys) for _ in range(192)], dev)
    batches = _batches(st.num_graphs, g, 6, seed=13)
    warm = _batches(st.num_graphs, g, 1, seed=77)[0]
    ref_model = make()
    ref_step = _make_step(ref_model, torch.optim.AdamW(ref_model.parameters(), lr=1e-3, capturable=True))
    for _ in range(3):
        ref_step(st.collate(warm))
    ref = []
    for ids in batches:
        loss = ref_step(st.collate(ids))
        ref.append((loss.clone(), _grads(ref_model), {k: v.detach().clone() for k, v in ref_model.state_dict().items()}))
    model = make()
    ss = SlotStep(st, g, _make_step(model, torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)), warmup_ids=warm, warmup=3)
    for k, ids in enumerate(batches):
        loss = ss.run(ids)
        assert torch.equal(loss, ref[k][0]), (family, k, float(loss), float(ref[k][0]))
        _assert_same(_grads(model), ref[k][1], f"{family}: gradients at step {k}")
        _assert_same({kk: v.detach() for kk, v in model.state_dict().items()}, ref[k][2], f"{family}: model state after step {k}")
    assert ss.replays == len(batches) and ss.eager_steps == 0


def test_hash_searching_operators_refuse_a_padded_pattern(dev, store):
    from pygho_amd.slots import BatchSlot
    slot = BatchSlot(store, 48)
    dd = slot.collate(_batches(store.num_graphs, 48, 1)[0])
    with pytest.raises(RuntimeError, match="matches index tuples by hash"):
        dd["X"].diag([0, 1])


@pytest.fixture()
def dual_everywhere(dev):
    """the size thresholds of the fused backward off for the duration of a test (the slot tests run 48-graph batches)"""
    from pygho_amd import _ops
    old = (_ops.SEG_SCATTER_MIN_MESSAGES, _ops.DUAL_TG_MIN_MESSAGES)
    _ops.SEG_SCATTER_MIN_MESSAGES = _ops.DUAL_TG_MIN_MESSAGES = 0
    yield
    _ops.SEG_SCATTER_MIN_MESSAGES, _ops.DUAL_TG_MIN_MESSAGES = old


def test_the_slot_holds_the_fused_backward_chunk_list(dev, store, dual_everywhere):
    """the aligned chunk list of the by-edge plan in a slot: the exact batch's records on the true extent, all-zero records behind;
    one eager step through the fused backward's table-gradient form on the slot == on the exactly sized batch, bit for bit INCLUDING
    the embedding table's gradient (the workgroups' shares are cut from the true chunk count on the device, not from the capacity)"""
    from pygho_amd import _ops
    from pygho_amd.slots import BatchSlot
    assert "cgap" in store.scatter_parts[KEY]
    g = 48
    slot = BatchSlot(store, g)
    assert ("sc", KEY) in slot.caps and "sc_chunks" in slot.msg[KEY]
    for ids in _batches(store.num_graphs, g, 3, seed=21):
        dd_s = slot.collate(ids)
        dd_e = store.collate(ids)
        t, e = dd_e["X"].nnz, dd_e["A"].nnz
        sp_e = _ops.scatter_plan(_ops.message_plan(dd_e[KEY + "___acd"], t, t, e), on_demand=True)
        ent, n = slot.msg[KEY], slot.true_sizes()[("sc", KEY)]
        assert n == sp_e.n_chunks and int(slot.counts[("sc", KEY)]) == n
        assert torch.equal(ent["sc_chunks"][:n], sp_e.chunks) and int(ent["sc_chunks"][n:].abs().sum()) == 0
        assert torch.equal(ent["sc_words"][:sp_e.words.numel()], sp_e.words) and torch.equal(ent["sc_cgap"][:n], sp_e.cgap)
        res = []
        for mode in ("slot", "exact"):
            model = _model(dev)
            step = _make_step(model, torch.optim.SGD(model.parameters(), lr=0.0))
            timer = _ops.LaunchTimer()
            with timer:
                if mode == "slot":
                    with slot.rows():
                        loss = step(slot.collate(ids))
                else:
                    loss = step(store.collate(ids))
            torch.cuda.synchronize()
            assert any(k.endswith(",table]") for k in timer.summary()), (mode, sorted(timer.summary()))
            res.append((loss.clone(), _grads(model)))
        assert torch.equal(res[0][0], res[1][0])
        _assert_same(res[0][1], res[1][1], "gradients through the fused backward (slot vs exact batch)")


def test_captured_slot_step_with_the_fused_capturable_optimizer(dev, store):
    """torch's fused multi-tensor AdamW in its capturable form inside the captured step (what bench.py and examples/minimal.py use:
    one launch per replay instead of the foreach implementation's ~20): 10 replayed batches == the eager loop with the same optimizer,
    bit for bit"""
    from pygho_amd.graphs import SlotStep
    g, n_steps = 48, 10
    batches = _batches(store.num_graphs, g, n_steps, seed=31)
    warm = _batches(store.num_graphs, g, 1, seed=97)[0]
    mk = lambda m: torch.optim.AdamW(m.parameters(), lr=1e-3, capturable=True, fused=True)
    ref_model = _model(dev)
    ref_step = _make_step(ref_model, mk(ref_model))
    for _ in range(3):
        ref_step(store.collate(warm))
    ref = []
    for ids in batches:
        loss = ref_step(store.collate(ids))
        ref.append((loss.clone(), {k: v.detach().clone() for k, v in ref_model.state_dict().items()}))
    model = _model(dev)
    ss = SlotStep(store, g, _make_step(model, mk(model)), warmup_ids=warm, warmup=3)
    for k, ids in enumerate(batches):
        loss = ss.run(ids)
        assert torch.equal(loss, ref[k][0]), (k, float(loss), float(ref[k][0]))
        _assert_same({kk: v.detach() for kk, v in model.state_dict().items()}, ref[k][1], f"model state after step {k}")
    assert ss.replays == n_steps and ss.eager_steps == 0


def test_captured_slot_step_through_the_fused_backward(dev, store, dual_everywhere):
    """ONE capture with the fused forward AND the fused backward (table-gradient form) inside, 8 different batches == the eager loop on
    `store.collate` batches bit for bit (loss, every gradient, every parameter after AdamW)"""
    from pygho_amd import _ops
    from pygho_amd.graphs import SlotStep
    g, n_steps = 64, 8
    batches = _batches(store.num_graphs, g, n_steps, seed=17)
    warm = _batches(store.num_graphs, g, 1, seed=98)[0]
    ref_model = _model(dev)
    ref_step = _make_step(ref_model, torch.optim.AdamW(ref_model.parameters(), lr=1e-3, capturable=True))
    timer = _ops.LaunchTimer()
    with timer:
        for _ in range(3):
            ref_step(store.collate(warm))
    torch.cuda.synchronize()
    assert any(k.endswith(",table]") for k in timer.summary())
    ref = []
    for ids in batches:
        loss = ref_step(store.collate(ids))
        ref.append((loss.clone(), _grads(ref_model), {k: v.detach().clone() for k, v in ref_model.state_dict().items()}))
    model = _model(dev)
    ss = SlotStep(store, g, _make_step(model, torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)), warmup_ids=warm, warmup=3)
    for k, ids in enumerate(batches):
        loss = ss.run(ids)
        assert torch.equal(loss, ref[k][0]), (k, float(loss), float(ref[k][0]))
        _assert_same(_grads(model), ref[k][1], f"gradients at step {k}")
        _assert_same({kk: v.detach() for kk, v in model.state_dict().items()}, ref[k][2], f"model state after step {k}")
    assert ss.replays == n_steps and ss.eager_steps == 0


@pytest.mark.parametrize("family", ["NGNN", "SSWL", "DSSGNN", "GNNAK", "SUN"])
def test_one_step_on_the_slot_equals_the_exact_batch_over_many_batches(dev, family):
    """12 different batches, ONE eager step each from the same model state, on the slot and on the exactly sized batch: loss and every
    gradient bit for bit.  (Round 6: a row reduction whose split depends on the ROW COUNT -- a library GEMM / `sum(0)` over a slot's
    capacity rows against the batch's true rows -- agreed on most batches and differed in the last bits on about a quarter of them, in
    GNNAK's and SUN's node-level Linears; they now take the tiled weight-gradient kernel at any height.  The embedding tables' gradients
    reach their f32 masters unrounded since this round, so nothing of that kind hides behind a bf16 rounding any more.)"""
    from pygho_amd import synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.honn.SpOperator import parse_precomputekey
    from pygho_amd.models import SpModel
    from pygho_amd.slots import BatchSlot
    g = 48

    def make():
        torch.manual_seed(1)
        return SpModel(family, num_layer=2, hiddim=128, act_dtype=torch.bfloat16).to(dev)
    keys = tuple(parse_precomputekey(make()))
    rng = np.random.default_rng(9)
    st = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, keys) for _ in range(192)], dev)
    slot = BatchSlot(st, g)
    for s in range(12):
        ids = np.random.default_rng(100 + s).permutation(st.num_graphs)[:g]
        res = []
        for mode in ("slot", "exact"):
            model = make()
            step = _make_step(model, torch.optim.SGD(model.parameters(), lr=0.0))
            if mode == "slot":
                dd = slot.collate(ids)
                with slot.rows():
                    loss = step(dd)
            else:
                loss = step(st.collate(ids))
            res.append((loss.clone(), _grads(model)))
        assert torch.equal(res[0][0], res[1][0]), (family, s)
        _assert_same(res[0][1], res[1][1], f"{family}: gradients of batch {s} (slot vs exact batch)")

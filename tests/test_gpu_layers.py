"""
Layer-level parity on the GPU: the shipped subgraph-GNN layers (NGNNConv, SSWLConv, I2Conv sparse; NGNNConv
dense) built from pygho_amd with the reference's state_dict loaded, forward + input/parameter gradients
against golden vectors produced by the reference layers themselves (BatchNorm in eval mode).
This is BASELINE config 1 (the reference's own CPU-runnable case) replayed on the HIP path.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
# north_star: "within 1e-5 relative for floating-point aggregation".  Layer outputs and input gradients (values of order 1)
# are held to rtol = atol = 1e-5.  Parameter gradients are sums over every tuple of the batch with entries up to ~30: they are
# held to 1e-5 RELATIVE TO THE LARGEST ENTRY of the gradient tensor (an element that cancels to ~0 inside a tensor of order 10
# cannot be compared to 1e-5 of itself: f32 summation order differs between ATen's GEMM and the kernels here).
TOL = dict(rtol=1e-5, atol=1e-5)


def assert_param_grad(got, exp, err_msg=""):
    scale = max(float(np.abs(exp).max()), 1.0)
    np.testing.assert_allclose(got / scale, exp / scale, rtol=1e-5, atol=1e-5, err_msg=err_msg)


MLP = {"numlayer": 1, "tailact": True, "norm": "bn", "act": "silu", "dp": 0.0}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def _load(layer, g, name, dev):
    sd = {k[len(name) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + "_sd_")}
    missing = layer.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return layer.to(dev).eval()


def _check(layer, g, name, A, mkX, xv, dd, dev, valid=None):
    xv = xv.clone().requires_grad_(True)
    res = layer(A, mkX(xv), dd)
    vals = res.values if hasattr(res, "values") else res.data
    exp, w = g[f"{name}_out"], g[f"{name}_w"]
    if valid is not None:
        exp, w = exp * valid, w * valid
    np.testing.assert_allclose(N(vals), exp, **TOL)
    (vals * T(w, dev)).sum().backward()
    if valid is None:
        np.testing.assert_allclose(N(xv.grad), g[f"{name}_gX"], **TOL)
        for k, p in layer.named_parameters():
            assert_param_grad(N(p.grad), g[f"{name}_pg_{k}"], err_msg=k)


def test_sparse_layers_match_reference(dev):
    from pygho_amd import SparseTensor
    from pygho_amd.honn import Conv
    g = load_golden("layers.npz")
    h = g["Xv"].shape[1]
    n = int(g["N"])
    ei, tid = T(g["edge_index"], dev), T(g["tupleid"], dev)
    dd = {k[4:] + "___acd": T(g[k], dev) for k in g.files if k.startswith("acd_")}
    A = SparseTensor(ei, T(g["Av"], dev), [n, n, h], True)
    mk = lambda xv: SparseTensor(tid, xv, [n, n, h], True)
    _check(_load(Conv.NGNNConv(h, h, "sum", "SS", dict(MLP)), g, "ngnn", dev), g, "ngnn", A, mk, T(g["Xv"], dev), dd, dev)
    _check(_load(Conv.NGNNConv(h, h, "max", "SS", dict(MLP)), g, "ngnnmax", dev), g, "ngnnmax", A, mk, T(g["Xv"], dev), dd, dev)
    _check(_load(Conv.SSWLConv(h, h, "sum", "SS", dict(MLP)), g, "sswl", dev), g, "sswl", A, mk, T(g["Xv"], dev), dd, dev)
    n3 = int(g["N3"])
    ei3, tid3 = T(g["edge_index3"], dev), T(g["tupleid3"], dev)
    A3 = SparseTensor(ei3, T(g["Av3"], dev), [n3, n3, h], True)
    dd3 = {"X___X___2___A___0___acd": T(g["acd3"], dev)}
    _check(_load(Conv.I2Conv(h, h, "sum", "SS", dict(MLP)), g, "i2", dev), g, "i2", A3,
           lambda xv: SparseTensor(tid3, xv, [n3, n3, n3, h], True), T(g["Xv3"], dev), dd3, dev)


def test_dense_ngnn_matches_reference_on_valid_entries(dev):
    from pygho_amd import MaskedTensor
    from pygho_amd.honn import Conv
    g = load_golden("layers.npz")
    h = g["dd_X"].shape[-1]
    Xm, Am = T(g["dd_Xmask"], dev), T(g["dd_Amask"], dev)
    layer = _load(Conv.NGNNConv(h, h, "sum", "DD", dict(MLP)), g, "ngnn_dd", dev)
    # masked positions of the reference output hold unfilled garbage (MaTensor.py:107-120): compare valid entries
    _check(layer, g, "ngnn_dd", MaskedTensor(T(g["dd_A"], dev), Am), lambda xv: MaskedTensor(xv, Xm), T(g["dd_X"], dev), {}, dev,
           valid=g["dd_Xmask"][..., None].astype(np.float32))



def test_ppgn_gnnak_dssgnn_match_reference(dev):
    from pygho_amd import SparseTensor
    from pygho_amd.honn import Conv
    g = load_golden("layers.npz")
    h = g["Xvp"].shape[1]
    n = int(g["Np"])
    ei, tid = T(g["edge_indexp"], dev), T(g["tupleidp"], dev)
    dd = {k[5:] + "___acd": T(g[k], dev) for k in g.files if k.startswith("acdp_")}
    A = SparseTensor(ei, T(g["Avp"], dev), [n, n, h], True)
    mk = lambda xv: SparseTensor(tid, xv, [n, n, h], True)
    _check(_load(Conv.PPGNConv(h, h, "sum", "SS", dict(MLP)), g, "ppgn", dev), g, "ppgn", A, mk, T(g["Xvp"], dev), dd, dev)
    _check(_load(Conv.GNNAKConv(h, h, "sum", "mean", "SS", dict(MLP), dict(MLP)), g, "gnnak", dev), g, "gnnak", A, mk,
           T(g["Xvp"], dev), dd, dev)
    Asc = SparseTensor(ei, None, [n, n], True)
    _check(_load(Conv.DSSGNNConv(h, h, "sum", "sum", "mean", "SS", dict(MLP)), g, "dssgnn", dev), g, "dssgnn", Asc, mk,
           T(g["Xvp"], dev), dd, dev)


def test_ngnn_model_matches_reference_model(dev):
    """BASELINE config 1 replayed on the HIP path: pygho_amd.ngnn.SpModel (f32) loads the REFERENCE model's state_dict
    strict (example/minimal.py's model assembled from the reference's operator classes, tests/golden/make_golden.py
    gen_model) and reproduces its training-mode prediction, L1 loss, every parameter gradient and the updated BatchNorm
    running statistics on 16 ZINC-shape graphs, d = 128."""
    from pygho_amd import SparseTensor
    from pygho_amd.ngnn import SpModel
    g = load_golden("ngnn_model.npz")
    model = SpModel(1, 6, 128)
    res = model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(dev).train()
    n = int(g["num_nodes"])
    dd = {"x": T(g["x"], dev), "batch": T(g["batch"], dev), "num_graphs": int(g["num_graphs"]), "num_nodes": n,
          "X___X___1___A___0___acd": T(g["acd"], dev),
          "A": SparseTensor(T(g["edge_index"], dev), T(g["edge_attr"], dev), [n, n], True),
          "X": SparseTensor(T(g["tupleid"], dev), T(g["tuplefeat"], dev), [n, n], True)}
    pred = model(dd)
    loss = torch.nn.functional.l1_loss(T(g["y"], dev).unsqueeze(-1), pred, reduction="mean")
    loss.backward()
    np.testing.assert_allclose(N(pred), g["pred"], **TOL)
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-5)
    for k, p in model.named_parameters():
        assert_param_grad(N(p.grad), g[f"pg_{k}"], err_msg=k)
    after = model.state_dict()
    for k in g.files:
        if k.startswith("after_"):
            np.testing.assert_allclose(N(after[k[6:]]), g[k], rtol=1e-5, atol=1e-6, err_msg=k)


# free-running bf16 trajectory against the f32 port: six AdamW steps amplify rounding noise (Adam's early updates are ~ lr * sign(g): a
# gradient component at the noise level flips a whole +-lr update), so the figure moves with any change of a summation order --
# measured 1.06 % in round 4, 1.89 % after round 5 re-dealt the BatchNorm / fold partitions.  It catches a run that stops following
# the optimizer (round 3's stale 16-bit copies: > 10 %), not a 1 % drift; THAT is the teacher-forced test below.
BF16_TRAJECTORY_RTOL = 0.03
BF16_DELTA_COSINE = 0.95          # measured 0.968 (ea_encoder.weight, the smallest over the weight matrices)
# teacher-forced (both sides take every step from the SAME parameters): what bf16 activations cost in ONE step, no amplification
BF16_STEP_LOSS_RTOL = {64: 0.004, 128: 0.008}      # measured 0.16 % at hidden 64 and 0.40 % at hidden 128 (its own batch and parameters; the fused
                                                   # forward is bit-identical to the launches it replaces, tests/test_gpu_fused.py, so the width, not the kernel)
BF16_STEP_GRAD_COSINE = 0.93      # measured 0.951 for the WORST weight matrix (lin_tupleinit0 at step 0: its gradient is what is left after
                                  # heavy cancellation over the tuples of a root, so 16-bit rounding of the summands shows) ...
BF16_STEP_GRAD_COSINE_MEDIAN = 0.99   # ... while the typical weight matrix agrees far better


@pytest.mark.parametrize("dtype,optimizer", [(None, "adamw_fused"), (None, "adamw_foreach"), (None, "sgd_fused"),
                                             (torch.bfloat16, "adamw_fused"), (torch.bfloat16, "adamw_foreach")])
def test_ngnn_training_trajectory_matches_the_host_port(dev, dtype, optimizer):
    """SEVERAL training steps, not one: pygho_amd.ngnn.SpModel on the HIP path against the reference's ATen op sequence on the
    host (oracle/aten_port.NGNNPort, itself pinned to the reference model's golden outputs in tests/test_oracle_golden.py), same
    initial state_dict, same 48-graph batch, AdamW(lr 1e-3) in torch's fused and foreach forms and fused SGD(lr 1e-2).  f32: the
    loss of every step to 1e-4; under SGD also every parameter and buffer after 6 steps to 1e-4 of its tensor's largest entry
    (under Adam the biases in front of a BatchNorm -- exact gradient zero, computed gradient rounding noise -- receive +-lr
    updates whose sign is that noise's, so parameters are only compared under SGD).  bf16 activations (f32 masters): the loss
    trajectory to 2 % -- a run whose 16-bit weight copies do not follow the optimizer (round 3's arena bug) fails this."""
    from oracle import aten_port as P
    from pygho_amd import synth
    from pygho_amd.ngnn import SpModel
    key = "X___X___1___A___0"
    hb = synth.make_batch(48, "zinc", seed=23)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    torch.manual_seed(4)
    model = SpModel(1, 3, 64, act_dtype=dtype)
    port = P.NGNNPort(64, 3)
    res = port.load_state_dict({P.port_key(k): v.clone() for k, v in model.state_dict().items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(dev).train()
    port.train()
    dd = synth.to_datadict(hb, dev)
    host = (t(hb.x), t(hb.edge_attr), t(hb.tupleid), t(hb.tuplefeat), t(hb.acd[key]), t(hb.batch), hb.num_graphs)
    y_host = t(hb.y).unsqueeze(-1)
    if optimizer == "sgd_fused":
        opt_d, opt_h = torch.optim.SGD(model.parameters(), lr=1e-2, fused=True), torch.optim.SGD(port.parameters(), lr=1e-2)
    else:
        opt_d = torch.optim.AdamW(model.parameters(), lr=1e-3, **({"fused": True} if optimizer == "adamw_fused" else {"foreach": True}))
        opt_h = torch.optim.AdamW(port.parameters(), lr=1e-3)
    steps = 6
    got, exp = [], []
    init = {P.port_key(k): v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for _ in range(steps):
        opt_d.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype is not None):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        opt_d.step()
        got.append(float(loss.detach()))
        opt_h.zero_grad()
        lh = torch.nn.functional.l1_loss(y_host, port(*host))
        lh.backward()
        opt_h.step()
        exp.append(float(lh.detach()))
    assert exp[-1] < 0.97 * exp[0], f"the host port does not learn on this batch: {exp}"
    if dtype is None:
        np.testing.assert_allclose(got, exp, rtol=1e-4, err_msg="loss trajectory")
        if optimizer == "sgd_fused":
            after = {P.port_key(k): v for k, v in model.state_dict().items()}
            for k, v in port.state_dict().items():
                if v.dtype.is_floating_point:
                    scale = max(float(v.abs().max()), 1e-3)
                    np.testing.assert_allclose(N(after[k]) / scale, v.numpy() / scale, rtol=0, atol=1e-4, err_msg=k)
    else:
        rel = max(abs(a - b) / abs(b) for a, b in zip(got, exp))
        assert rel < BF16_TRAJECTORY_RTOL, f"loss trajectory (bf16 activations): max relative difference {rel:.4f}: {got} vs {exp}"
        # and the parameters themselves: the 6-step displacement of every weight matrix points where the host port's does (a run whose
        # 16-bit weight copies do not follow the optimizer keeps differentiating at the initial weights and turns away)
        after = {P.port_key(k): v.detach().cpu() for k, v in model.state_dict().items()}
        worst = (1.0, None)
        for k, v in port.state_dict().items():
            if v.dtype.is_floating_point and v.dim() == 2 and v.numel() >= 1024:
                dh, dd_ = (v - init[k]).flatten().double(), (after[k] - init[k]).flatten().double()
                cos = float(torch.dot(dh, dd_) / (dh.norm() * dd_.norm() + 1e-30))
                worst = min(worst, (cos, k))
        print(f"bf16 trajectory: max relative loss difference {rel:.4f}, smallest displacement cosine {worst[0]:.4f} ({worst[1]})")
        assert worst[0] > BF16_DELTA_COSINE, f"parameter displacement of {worst[1]} deviates from the host port's: cosine {worst[0]:.3f}"


@pytest.mark.parametrize("hidden", [64, 128])
def test_ngnn_bf16_steps_from_the_same_parameters_match_the_host_port(dev, hidden):
    """The tight half of the bf16 parity claim (the free-running trajectory 'catches a frozen arena, not a 1 %
    drift').  Six steps, TEACHER-FORCED: before every step the device model takes the host port's current parameters and buffers,
    so both sides differentiate at the same point and rounding noise is not amplified by the optimizer.  Per step: the loss within
    BF16_STEP_LOSS_RTOL of the f32 port's, and the gradient of every weight matrix within cosine BF16_STEP_GRAD_COSINE of the port's
    (bf16 activations, f32 masters); the BatchNorm running statistics after the step within 1 %.
    The batch is collated by the device store (every plan installed, incl. the fused forward's chunks), so at hidden = 128 the layers'
    forward IS `seg_fused_fwd_kernel` (asserted through the launch names): the kernel the benchmark reports sits on this test's path."""
    from oracle import aten_port as P
    from pygho_amd import _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.ngnn import SpModel
    key = "X___X___1___A___0"
    rng = np.random.default_rng(23)
    store = DeviceGraphStore([synth.make_graph(rng, "zinc", 3, (key,)) for _ in range(64)], dev)
    torch.manual_seed(4)
    model = SpModel(1, 3, hidden, act_dtype=torch.bfloat16)
    port = P.NGNNPort(hidden, 3)
    port.load_state_dict({P.port_key(k): v.clone() for k, v in model.state_dict().items()}, strict=True)
    model = model.to(dev).train()
    port.train()
    dd = store.collate(np.random.default_rng(5).permutation(64)[:48])
    host = (dd["x"].cpu(), dd["A"].values.cpu(), dd["X"].indices.cpu(), dd["X"].values.cpu(), dd[key + "___acd"].cpu(), dd["batch"].cpu(),
            int(dd["num_graphs"]))
    y_host = dd["y"].cpu().unsqueeze(-1)
    timer = _ops.LaunchTimer()
    opt_h = torch.optim.AdamW(port.parameters(), lr=1e-3)
    names = {P.port_key(k): k for k in model.state_dict()}
    worst_loss, worst_cos, all_cos = 0.0, (1.0, None), []
    for step in range(6):
        with torch.no_grad():                                   # the device model starts the step where the port stands
            sd = model.state_dict()
            for pk, v in port.state_dict().items():
                sd[names[pk]].copy_(v.to(dev))
        _ops.invalidate_cast_arenas()
        model.zero_grad(set_to_none=True)
        with timer, torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        opt_h.zero_grad()
        lh = torch.nn.functional.l1_loss(y_host, port(*host))
        lh.backward()
        worst_loss = max(worst_loss, abs(float(loss.detach()) - float(lh.detach())) / abs(float(lh.detach())))
        got = {P.port_key(k): p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}
        for k, p in port.named_parameters():
            if p.grad is not None and p.dim() == 2 and p.numel() >= 1024:
                a, b = got[k].flatten(), p.grad.double().flatten()
                cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
                all_cos.append(cos)
                worst_cos = min(worst_cos, (cos, f"{k} step {step}"))
        after = {P.port_key(k): v.detach().cpu() for k, v in model.state_dict().items() if "running" in k}
        for k, v in port.state_dict().items():
            if "running" in k:
                scale = max(float(v.abs().max()), 1e-3)
                np.testing.assert_allclose(after[k].numpy() / scale, v.numpy() / scale, rtol=0, atol=1e-2, err_msg=f"{k} step {step}")
        opt_h.step()
    med = float(np.median(all_cos))
    torch.cuda.synchronize()
    fused_ran = any(k.startswith("seg_fused[") for k in timer.summary())
    assert fused_ran == (hidden == 128 and _ops.FUSED_FWD), sorted(timer.summary())
    print(f"bf16 teacher-forced (hidden {hidden}, fused forward {'ran' if fused_ran else 'not eligible'}): max relative loss difference {worst_loss:.5f}, smallest gradient cosine {worst_cos[0]:.5f} ({worst_cos[1]}), "
          f"median {med:.5f} over {len(all_cos)} (matrix, step) pairs")
    assert worst_loss < BF16_STEP_LOSS_RTOL[hidden], f"per-step loss (bf16 activations, same parameters): {worst_loss:.5f}"
    assert worst_cos[0] > BF16_STEP_GRAD_COSINE, f"gradient of {worst_cos[1]}: cosine {worst_cos[0]:.4f} against the f32 port"
    assert med > BF16_STEP_GRAD_COSINE_MEDIAN, f"median gradient cosine {med:.4f} against the f32 port"


def _sun_check(layer, g, name, A_of, X_of, av, xv, dd, dev, fn="forward", amask=1.0):
    xv, av = xv.clone().requires_grad_(True), av.clone().requires_grad_(True)
    res = getattr(layer, fn)(A_of(av), X_of(xv), dd)
    vals = res.values if hasattr(res, "values") else res.data
    np.testing.assert_allclose(N(vals), g[f"{name}_out"], **TOL, err_msg=f"{name} {fn} out")
    (vals * T(g[f"{name}_w"], dev)).sum().backward()
    np.testing.assert_allclose(N(xv.grad), g[f"{name}_gX"], **TOL, err_msg=f"{name} {fn} gX")
    np.testing.assert_allclose(N(av.grad) * amask, g[f"{name}_gA"], **TOL, err_msg=f"{name} {fn} gA")
    for k, p in layer.named_parameters():
        assert_param_grad(N(p.grad), g[f"{name}_pg_{k}"], err_msg=f"{name} {fn} {k}")
        p.grad = None


@pytest.mark.parametrize("name,pool", [("sun", "mean"), ("sunsum", "sum")])
def test_sunconv_sparse_matches_reference(dev, name, pool):
    """SUNConv mode "SS" against the REFERENCE's SUNConv.forward (Conv.py:349-362) run around the labelled stand-in for
    torch_geometric's HeteroLinear (tests/golden/make_golden.py gen_sun): the reference state_dict (lin1_0.weight (2, 7d, d) +
    lin1_0.bias (2, d)) loads strict; output, input / adjacency / parameter gradients for the restructured forward AND the
    literal wiring; and what the literal wiring hands to HeteroLinear (the 7-way concatenation, the diagonal-type vector)."""
    from pygho_amd import SparseTensor
    from pygho_amd.honn import Conv
    g = load_golden("sun.npz")
    h, n = g["Xv"].shape[1], int(g["N"])
    ei, tid = T(g["edge_index"], dev), T(g["tupleid"], dev)
    dd = {"X___X___1___A___0___acd": T(g["acd"], dev)}
    layer = _load(Conv.SUNConv(h, h, "sum", pool, "SS", dict(MLP), dict(MLP)), g, name, dev)
    assert tuple(layer.lin1_0.weight.shape) == (2, 7 * h, h) and tuple(layer.lin1_0.bias.shape) == (2, h)
    A_of = lambda av: SparseTensor(ei, av, [n, n, h], True)
    X_of = lambda xv: SparseTensor(tid, xv, [n, n, h], True)
    seen = {}
    hook = layer.lin1_0.register_forward_pre_hook(lambda mod, args: seen.update(x=args[0].detach(), t=args[1].detach()))
    _sun_check(layer, g, name, A_of, X_of, T(g["Av"], dev), T(g["Xv"], dev), dd, dev, fn="forward_concat")
    hook.remove()
    assert np.array_equal(N(seen["t"]), g[f"{name}_type"]) and seen["t"].dtype == torch.int64
    np.testing.assert_allclose(N(seen["x"]), g[f"{name}_cat7"], **TOL)
    _sun_check(layer, g, name, A_of, X_of, T(g["Av"], dev), T(g["Xv"], dev), dd, dev, fn="forward")


def test_sunconv_dense_matches_reference(dev):
    """SUNConv mode "DD" on a ragged padded batch against the reference layer run graph by graph without padding (the
    reference on the padded batch leaks its unfilled padded rows through pool2node -- MaTensor.py:107-120, recorded as
    sundd_leaky_out; the documented semantics implemented here do not)."""
    from pygho_amd import MaskedTensor
    from pygho_amd.honn import Conv
    g = load_golden("sun.npz")
    h = g["dd_X"].shape[-1]
    Xm, Am = T(g["dd_Xmask"], dev), T(g["dd_Amask"], dev)
    layer = _load(Conv.SUNConv(h, h, "sum", "mean", "DD", dict(MLP), dict(MLP)), g, "sundd", dev)
    A_of = lambda av: MaskedTensor(av, Am)
    X_of = lambda xv: MaskedTensor(xv, Xm)
    for fn in ("forward_concat", "forward"):
        _sun_check(layer, g, "sundd", A_of, X_of, T(g["dd_A"], dev), T(g["dd_X"], dev), {}, dev, fn=fn,
                   amask=g["dd_Amask"][..., None].astype(np.float32))     # masked adjacency slots: don't-care gradients
    # the deviation is real on this batch: the reference's own padded run differs from its per-graph runs
    assert np.abs(g["sundd_leaky_out"] - g["sundd_out"]).max() > 1e-3


def test_sunconv_sparse_and_dense_agree(dev):
    """The sparse (SS) and dense (DD) realisations of the same layer, built from
    disjoint kernels (segment reduce vs MFMA bmm / masked reductions), agree on a batch where every node pair
    of a graph is a tuple (the dense sampler's pattern, hodata/MaTupleSampler.py:11-32)."""
    from pygho_amd import MaskedTensor, SparseTensor, synth
    from pygho_amd.honn import Conv
    h = 16
    dn = synth.make_dense_batch(3, seed=5, hidden=h, clip_nodes=8)
    b, nmax = dn["nodemask"].shape
    torch.manual_seed(0)
    mlp = dict(MLP, norm="none")           # BatchNorm statistics would see the padded rows in DD mode
    ss = Conv.SUNConv(h, h, "sum", "mean", "SS", dict(mlp), dict(mlp)).to(dev)
    ddl = Conv.SUNConv(h, h, "sum", "mean", "DD", dict(mlp), dict(mlp)).to(dev)
    ddl.load_state_dict(ss.state_dict())
    Xd, Ad = T(dn["X"], dev), T(dn["A"], dev)
    Xm, Am = T(dn["Xmask"], dev), T(dn["Amask"], dev)
    out_dd = ddl(MaskedTensor(Ad, Am), MaskedTensor(Xd, Xm), {}).data
    # the same batch as block-diagonal sparse tensors
    off = np.concatenate(([0], np.cumsum(dn["nodemask"].sum(1))))
    bi, ii, jj = np.nonzero(dn["Xmask"])
    tid = np.stack((ii + off[bi], jj + off[bi]))
    eb, ei_, ej = np.nonzero(dn["Amask"])
    eidx = np.stack((ei_ + off[eb], ej + off[eb]))
    n = int(off[-1])
    Xs = SparseTensor(T(tid, dev), T(dn["X"][bi, ii, jj], dev), [n, n, h], True)
    As = SparseTensor(T(eidx, dev), T(dn["A"][eb, ei_, ej], dev), [n, n, h], True)
    acd = synth.host_plan_acd(tid, tid, 1, eidx, 0)
    out_ss = ss(As, Xs, {"X___X___1___A___0___acd": T(acd, dev)}).values
    np.testing.assert_allclose(N(out_ss), N(out_dd)[bi, ii, jj], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["silu", "relu", "none"])
def test_fused_batchnorm_act_matches_torch(dev, dtype, act):
    """fused BatchNorm+activation kernels vs torch's batch_norm -> activation in f32 (training and eval mode,
    input / weight / bias gradients, running statistics)."""
    from pygho_amd import _ops
    torch.manual_seed(0)
    m, c = 20_011, 128
    x32 = torch.randn(m, c, device=dev) * 1.7 + 0.6
    x = x32.to(dtype)
    g = torch.randn(m, c, device=dev).to(dtype)
    actf = {"silu": torch.nn.functional.silu, "relu": torch.relu, "none": lambda t: t}[act]
    for training in (True, False):
        bn1, bn2 = torch.nn.BatchNorm1d(c).to(dev), torch.nn.BatchNorm1d(c).to(dev)
        with torch.no_grad():
            bn1.weight.uniform_(0.5, 1.5); bn1.bias.normal_(0, 0.3)
            bn1.running_mean.normal_(0, 0.2); bn1.running_var.uniform_(0.5, 2.0)
        bn2.load_state_dict(bn1.state_dict())
        bn1.train(training); bn2.train(training)
        xa = x.clone().requires_grad_(True)
        ya = _ops.batch_norm_act(xa, bn1, act)
        (ya.float() * g.float()).sum().backward()
        xb = x.float().clone().requires_grad_(True)
        yb = actf(bn2(xb))
        (yb * g.float()).sum().backward()
        tol = 2e-5 if dtype == torch.float32 else 2.0 ** -7
        torch.testing.assert_close(ya.float(), yb, rtol=tol, atol=tol)
        gtol = 1e-4 if dtype == torch.float32 else 3e-2
        torch.testing.assert_close(xa.grad.float(), xb.grad, rtol=gtol, atol=gtol)
        torch.testing.assert_close(bn1.weight.grad, bn2.weight.grad, rtol=1e-3 if dtype == torch.float32 else 2e-2, atol=0.5 if dtype != torch.float32 else 1e-2)
        torch.testing.assert_close(bn1.bias.grad, bn2.bias.grad, rtol=1e-3 if dtype == torch.float32 else 2e-2, atol=0.5 if dtype != torch.float32 else 1e-2)
        torch.testing.assert_close(bn1.running_mean, bn2.running_mean, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(bn1.running_var, bn2.running_var, rtol=1e-4, atol=1e-4)
        assert int(bn1.num_batches_tracked) == int(bn2.num_batches_tracked)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [128, 64, 256])
def test_tuple_block_without_stored_preactivation_matches_torch(dev, dtype, d):
    """the whole Linear -> BatchNorm1d -> SiLU block on the recompute kernels (statistics-only pass, one-pass forward, backward sums
    and one-pass backward, none of which reads a stored pre-activation) against plain torch in f32 on the same 16-bit inputs:
    output, input / weight / bias and BatchNorm gradients, running statistics; training and eval mode.  Tolerances: one rounding
    of the 16-bit pre-activation and output (the f32 reference does not round the pre-activation)."""
    from pygho_amd import _ops
    torch.manual_seed(0)
    m = 24_001
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    x = (torch.randn(m, d, device=dev) * 0.8 + 0.2).to(dtype)
    g = torch.randn(m, d, device=dev).to(dtype)
    for training in (True, False):
        lin1 = torch.nn.Linear(d, d).to(dev)
        bn1, bn2 = torch.nn.BatchNorm1d(d).to(dev), torch.nn.BatchNorm1d(d).to(dev)
        with torch.no_grad():
            lin1.weight.copy_(lin1.weight.to(dtype).float()); lin1.bias.copy_(lin1.bias.to(dtype).float())      # exactly representable
            bn1.weight.uniform_(0.5, 1.5); bn1.bias.normal_(0, 0.3); bn1.running_mean.normal_(0, 0.2); bn1.running_var.uniform_(0.5, 2.0)
        bn2.load_state_dict(bn1.state_dict())
        lin2 = torch.nn.Linear(d, d).to(dev)
        lin2.load_state_dict(lin1.state_dict())
        bn1.train(training); bn2.train(training)
        assert _ops.USE_RECOMPUTE_PRE
        xa = x.clone().requires_grad_(True)
        ya = _ops.tuple_block(xa, lin1, bn1, "silu")
        (ya.float() * g.float()).sum().backward()
        xb = x.float().clone().requires_grad_(True)
        yb = torch.nn.functional.silu(bn2(lin2(xb)))
        (yb * g.float()).sum().backward()
        scale = float(yb.abs().max())
        torch.testing.assert_close(ya.float() / scale, yb / scale, rtol=0, atol=6 * ulp)
        gs = float(xb.grad.abs().max())
        torch.testing.assert_close(xa.grad.float() / gs, xb.grad / gs, rtol=0, atol=8 * ulp)
        ws = float(lin2.weight.grad.abs().max())
        torch.testing.assert_close(lin1.weight.grad / ws, lin2.weight.grad / ws, rtol=0, atol=4 * ulp)
        for pa, pb in ((bn1.weight, bn2.weight), (bn1.bias, bn2.bias)):
            s = float(pb.grad.abs().max())
            torch.testing.assert_close(pa.grad / s, pb.grad / s, rtol=0, atol=4 * ulp)
        if not training:                               # in training mode the bias in front of a BatchNorm has a zero gradient
            s = float(lin2.bias.grad.abs().max())
            torch.testing.assert_close(lin1.bias.grad / s, lin2.bias.grad / s, rtol=0, atol=4 * ulp)
        torch.testing.assert_close(bn1.running_mean, bn2.running_mean, rtol=0, atol=4 * ulp)
        torch.testing.assert_close(bn1.running_var, bn2.running_var, rtol=4 * ulp, atol=4 * ulp)


def _ngnn_inputs(dev, dtype, graphs=64, h=128, seed=5):
    from pygho_amd import SparseTensor, synth
    hb = synth.make_batch(graphs, "zinc", seed=seed)
    n = hb.num_nodes
    torch.manual_seed(seed)
    A = SparseTensor(T(hb.edge_index, dev), (torch.randn(hb.num_edges, h, device=dev) * 0.3).to(dtype), [n, n, h], True)
    xv = (torch.randn(hb.num_tuples, h, device=dev)).to(dtype)
    tid = T(hb.tupleid, dev)
    dd = {k + "___acd": T(v, dev) for k, v in hb.acd.items()}
    return A, tid, xv, dd, n


def test_residual_aggregation_epilogue_bit_exact(dev):
    """out = addend + segment reduction in ONE launch is bit-identical (f32) to reduce, then add; bf16 rounds once."""
    from pygho_amd import _ops
    A, tid, xv, dd, n = _ngnn_inputs(dev, torch.float32)
    acd = dd["X___X___1___A___0___acd"]
    plan = _ops.message_plan(acd, xv.shape[0], xv.shape[0], A.nnz)
    for aggr in ("sum", "mean", "max"):
        plain = _ops.seg_gmr(plan.n_out, xv, A.values, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, aggr)
        fused = _ops.seg_gmr(plan.n_out, xv, A.values, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, aggr, addend=xv)
        assert torch.equal(fused, xv + plain), aggr
    xb, ab = xv.bfloat16(), A.values.bfloat16()
    fused = _ops.seg_gmr(plan.n_out, xb, ab, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum", addend=xb)
    exact = xb.float() + _ops.seg_gmr(plan.n_out, xb.float(), ab.float(), plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum")
    assert torch.equal(fused, exact.bfloat16())        # f32 accumulate of exact products, one rounding
    # generic path (d not a multiple of the 16-byte vector)
    x5, a5 = xv[:, :5].contiguous(), A.values[:, :5].contiguous()
    plain = _ops.seg_gmr(plan.n_out, x5, a5, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum")
    assert torch.equal(_ops.seg_gmr(plan.n_out, x5, a5, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum", addend=x5), x5 + plain)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("aggr", ["sum", "mean"])
def test_fused_residual_layer_matches_unfused(dev, dtype, aggr):
    """NGNNConv.forward_residual (one autograd node: GEMM, fused BatchNorm+act, aggregation with residual epilogue,
    bias gradient from the BatchNorm backward, residual gradient as the GEMM accumulator) against
    X.add(conv.forward(A, X, datadict), True) built from the separately tested operators: values, input / adjacency /
    parameter gradients and BatchNorm running statistics, training and eval mode."""
    import copy
    from pygho_amd import SparseTensor
    from pygho_amd.honn import Conv
    h = 128
    A, tid, xv, dd, n = _ngnn_inputs(dev, dtype, graphs=700)      # nnz > 8192: the MLP's fused Linear path too
    torch.manual_seed(1)
    w = torch.randn(xv.shape, device=dev)
    f32 = dtype == torch.float32
    for training in (True, False):
        la = Conv.NGNNConv(h, h, aggr, "SS", dict(MLP)).to(dev)
        with torch.no_grad():
            la.lin.lins[1].norm.weight.uniform_(0.5, 1.5)
            la.lin.lins[1].norm.bias.normal_(0, 0.3)
        lb = copy.deepcopy(la)
        la.train(training); lb.train(training)
        res = {}
        for name, layer in (("fused", la), ("plain", lb)):
            x = xv.clone().requires_grad_(True)
            av = A.values.clone().requires_grad_(True)
            Ax = SparseTensor(A.indices, av, A.shape, True)
            X = SparseTensor(tid, x, [n, n, h], True)
            if name == "fused":
                out = layer.forward_residual(Ax, X, dd)
            else:
                H = layer.lin.lins[2](layer.lin.lins[1](torch.nn.functional.linear(x, layer.lin.lins[0].weight.to(dtype),
                                                                                   layer.lin.lins[0].bias.to(dtype))))
                out = X.add(layer.aggr.forward(Ax, SparseTensor(tid, H, [n, n, h], True), dd, X), True)
            (out.values.float() * w).sum().backward()
            res[name] = (out.values.float(), x.grad.float(), av.grad.float(), {k: p.grad.float() for k, p in layer.named_parameters()},
                         layer.lin.lins[1].norm.running_mean.clone(), layer.lin.lins[1].norm.running_var.clone())
        fa, fb = res["fused"], res["plain"]
        tol = dict(rtol=2e-4, atol=2e-4) if f32 else dict(rtol=3e-2, atol=6e-2)
        torch.testing.assert_close(fa[0], fb[0], **tol)
        torch.testing.assert_close(fa[1], fb[1], **tol)
        scale = float(fb[2].abs().max())
        torch.testing.assert_close(fa[2] / scale, fb[2] / scale, rtol=0, atol=1e-4 if f32 else 2e-2)
        for k in fa[3]:
            ref = fb[3][k]
            s = float(ref.abs().max()) + 1e-6
            if k.endswith("lins.0.bias") and training:
                # mathematically zero (a bias before BatchNorm): both sides are rounding noise around 0
                assert float(fa[3][k].abs().max()) <= 1e-3 * float(fb[3]["lin.lins.0.weight"].abs().max()) * (1 if f32 else 200), k
                continue
            torch.testing.assert_close(fa[3][k] / s, ref / s, rtol=0, atol=2e-4 if f32 else 3e-2, msg=k)
        torch.testing.assert_close(fa[4], fb[4], rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(fa[5], fb[5], rtol=1e-3, atol=1e-3)


def test_bn_backward_column_sums(dev):
    """the bias gradient of the Linear in front of a BatchNorm = column sums of the (rounded) BatchNorm input gradient,
    produced by the BatchNorm backward pass itself (eval mode, where it is not degenerate)."""
    from pygho_amd import _ops
    torch.manual_seed(3)
    for dtype in (torch.float32, torch.bfloat16):
        m, c = 30_001, 128
        x = (torch.randn(m, c, device=dev) * 1.3).to(dtype)
        g = torch.randn(m, c, device=dev).to(dtype)
        bn = torch.nn.BatchNorm1d(c).to(dev).eval()
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5)
        _, _, _, saved = _ops._bn_forward(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, False, bn.eps, "silu")
        dx, s1, s2, sdx = _ops._bn_backward(x, g, saved, False, "silu", want_colsum=True)
        ref = dx.double().sum(0)
        torch.testing.assert_close(sdx.double(), ref, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [128, 64, 256])
def test_rowblock_linear_matches_f32_reference(dev, dtype, d):
    """the streaming tuple-wise GEMM (MFMA) against an f32 torch reference: product (+ bias) within one output rounding,
    the residual-add epilogue equal to product-rounded-then-added, and the BatchNorm partial sums of its epilogue
    equal to sums over the rounded output; ragged row counts exercise the tail tile."""
    from pygho_amd import _ops
    torch.manual_seed(7)
    for m in (8192, 100_003, 300_000):
        x = torch.randn(m, d, device=dev).to(dtype)
        w = (torch.randn(d, d, device=dev) / d ** 0.5).to(dtype)       # asymmetric: a transposed operand would show
        b = torch.randn(d, device=dev).to(dtype)
        g = torch.randn(m, d, device=dev).to(dtype)
        ref = torch.nn.functional.linear(x.float(), w.float(), b.float())
        out, none = _ops.rowblock_linear(x, w, b)
        assert none is None
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        torch.testing.assert_close(out.float(), ref, rtol=ulp, atol=ulp)
        shift = ref[0].contiguous()
        out2, sums = _ops.rowblock_linear(x, w, b, stats_shift=shift)
        assert torch.equal(out, out2)
        o = out.double() - shift.double()
        torch.testing.assert_close(sums[:, 0].double().sum(0), o.sum(0), rtol=1e-5, atol=1e-3 * m ** 0.5)
        torch.testing.assert_close(sums[:, 1].double().sum(0), (o * o).sum(0), rtol=1e-5, atol=1e-3)
        out3, _ = _ops.rowblock_linear(x, w, None, addend=g)
        prod, _ = _ops.rowblock_linear(x, w, None)
        assert torch.equal(out3, (prod.float() + g.float()).to(dtype))
        # shift taken inside the kernel = row 0 of the (unrounded) output, with and without bias / residual row
        for bb, add in ((b, None), (None, g), (b, g)):
            out4, (sums4, sh4) = _ops.rowblock_linear(x, w, bb, addend=add, stats_shift=True)
            r0 = x[0].float() @ w.float().t() + (bb.float() if bb is not None else 0) + (add[0].float() if add is not None else 0)
            torch.testing.assert_close(sh4, r0, rtol=1e-4, atol=1e-4)
            o = out4.double() - sh4.double()
            torch.testing.assert_close(sums4[:, 0].double().sum(0), o.sum(0), rtol=1e-5, atol=1e-3 * m ** 0.5)
            torch.testing.assert_close(sums4[:, 1].double().sum(0), (o * o).sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", ["silu", "relu", "none"])
def test_bn_bwd_linear_equals_two_kernel_path(dev, dtype, act):
    """the one-pass backward of Linear -> BatchNorm -> act (BatchNorm input gradient formed in the GEMM prologue) is
    bit-identical to pygho_bn_act_bwd followed by the streaming GEMM with the residual epilogue."""
    from pygho_amd import _ops
    torch.manual_seed(11)
    d = 128
    for m, training in ((100_003, True), (40_000, False)):
        pre = (torch.randn(m, d, device=dev) * 1.2 + 0.3).to(dtype)
        gh = torch.randn(m, d, device=dev).to(dtype)
        g = torch.randn(m, d, device=dev).to(dtype)
        w = (torch.randn(d, d, device=dev) / d ** 0.5).to(dtype)
        bn = torch.nn.BatchNorm1d(d).to(dev).train(training)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 2.0)
        _, _, _, saved = _ops._bn_forward(pre, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, act)
        gpre0, s1, s2, sdx0 = _ops._bn_backward(pre, gh, saved, training, act, want_colsum=True)
        gx0, _ = _ops.rowblock_linear(gpre0, w.t().contiguous(), None, addend=g)
        gx1, gpre1, t1, t2, sdx1 = _ops.bn_bwd_linear(pre, gh, saved, training, act, w, g, True)
        assert torch.equal(gpre0, gpre1) and torch.equal(gx0, gx1)
        assert torch.equal(s1, t1) and torch.equal(s2, t2)
        torch.testing.assert_close(sdx0, sdx1, rtol=1e-5, atol=1e-3)
        gx2, _, _, _, none = _ops.bn_bwd_linear(pre, gh, saved, training, act, w, None, False)
        assert none is None
        ref, _ = _ops.rowblock_linear(gpre0, w.t().contiguous(), None)
        assert torch.equal(gx2, ref)
        # weight gradient folded in (gpre never written): same gx bits, dW = gpre^T x against an f64 product of the rounded gpre
        x = torch.randn(m, d, device=dev).to(dtype)
        gx3, dw, u1, u2, sdx3 = _ops.bn_bwd_linear(pre, gh, saved, training, act, w, g, True, x=x)
        assert torch.equal(gx3, gx0) and torch.equal(u1, s1) and torch.equal(u2, s2)
        torch.testing.assert_close(sdx3, sdx0, rtol=1e-5, atol=1e-3)
        dw_ref = gpre0.double().t() @ x.double()
        scale = float(dw_ref.abs().max())
        torch.testing.assert_close(dw.double() / scale, dw_ref / scale, rtol=0, atol=2e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", ["silu", "relu", "none"])
@pytest.mark.parametrize("d", [128, 64])
def test_recomputed_preactivation_passes_are_bit_identical_to_stored_ones(dev, dtype, act, d):
    """a training block that never stores pre = x W^T + b: every pass recomputes it from x with the forward's instruction
    sequence.  Against the passes that read a stored pre: the statistics' partial sums, the activated output (with and
    without a residual row), the two backward channel sums' partials-folded values (close: different reduction tree) and,
    GIVEN the same channel sums, the input gradient (bit for bit), weight gradient and bias gradient of the one-pass backward."""
    from pygho_amd import _ops
    torch.manual_seed(5)
    for m, training, bias in ((70_001, True, True), (33_000, False, False), (190, True, True)):
        x = (torch.randn(m, d, device=dev) * 0.9).to(dtype)
        w = (torch.randn(d, d, device=dev) / d ** 0.5).to(dtype)
        b = (torch.randn(d, device=dev) * 0.2).to(dtype) if bias else None
        gh = torch.randn(m, d, device=dev).to(dtype)
        g = torch.randn(m, d, device=dev).to(dtype)
        res = torch.randn(m, d, device=dev).to(dtype)
        bn = torch.nn.BatchNorm1d(d).to(dev).train(training)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 2.0)
        pre, partial = _ops.rowblock_linear(x, w, b, stats_shift=True if training else None)
        none, partial_r = _ops.rowblock_linear(x, w, b, stats_shift=True, store=False)
        assert none is None
        if training:
            assert torch.equal(partial[0], partial_r[0]) and torch.equal(partial[1], partial_r[1])
        for addend in (None, res):
            h0, mean0, var0, saved = _ops._bn_forward(pre, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, act,
                                                      None, partial, addend=addend)
            h1, mean1, var1, saved1 = _ops._bn_forward(None, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, act,
                                                       None, partial_r if training else None, addend=addend, producer=(x, w, b))
            assert torch.equal(h0, h1) and torch.equal(mean0, mean1) and torch.equal(var0, var1)
        s1, s2 = _ops.bn_bwd_sums(pre, gh, saved, act)
        t1, t2 = _ops.rowblock_linear_bwd_sums(x, w, b, gh, saved1, act)
        scale = float(s1.abs().max()) + float(s2.abs().max())
        torch.testing.assert_close(t1 / scale, s1 / scale, rtol=0, atol=1e-6)
        torch.testing.assert_close(t2 / scale, s2 / scale, rtol=0, atol=1e-6)
        for addend, want_cs in ((None, False), (g, True)):
            gx0, dw0, _, _, sdx0 = _ops.bn_bwd_linear(pre, gh, saved, training, act, w, addend, want_cs, x=x, sums=(s1, s2))
            gx1, dw1, u1, u2, sdx1 = _ops.bn_bwd_linear(None, gh, saved1, training, act, w, addend, want_cs, x=x, sums=(s1, s2), lin_bias=b)
            assert torch.equal(gx0, gx1) and torch.equal(dw0, dw1)
            assert (sdx0 is None and sdx1 is None) or torch.equal(sdx0, sdx1)
        # and end to end with its own sums
        gx2, dw2, v1, v2, _ = _ops.bn_bwd_linear(None, gh, saved1, training, act, w, g, True, x=x, lin_bias=b)
        assert torch.equal(v1, t1) and torch.equal(v2, t2)
        gs = float(gx0.float().abs().max())
        torch.testing.assert_close(gx2.float() / gs, gx0.float() / gs, rtol=0, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", ["silu", "relu"])
def test_width_256_split_passes_equal_the_stored_preactivation_passes(dev, dtype, act):
    """width 256 (BASELINE config 5's hidden size; I2Conv, reference Conv.py:107-147): the column-split streaming kernels against the
    passes that read a stored pre-activation -- statistics partials and BatchNorm + act output bit for bit, the backward channel sums
    to the reduction order, and GIVEN the same sums: gpre (the apply pass) bit for bit with `pygho_bn_act_bwd`, its column sums, the
    input gradient gx = gpre W + g bit for bit with the generic path's product-then-add, dW against an f64 product."""
    from pygho_amd import _ops
    torch.manual_seed(6)
    d = 256
    for m, training, bias in ((70_001, True, True), (33_000, False, False), (8192, True, True)):
        x = (torch.randn(m, d, device=dev) * 0.9).to(dtype)
        w = (torch.randn(d, d, device=dev) / d ** 0.5).to(dtype)
        b = (torch.randn(d, device=dev) * 0.2).to(dtype) if bias else None
        gh = torch.randn(m, d, device=dev).to(dtype)
        g = torch.randn(m, d, device=dev).to(dtype)
        res = torch.randn(m, d, device=dev).to(dtype)
        bn = torch.nn.BatchNorm1d(d).to(dev).train(training)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 2.0)
        ref = torch.nn.functional.linear(x.float(), w.float(), None if b is None else b.float())
        pre, partial = _ops.rowblock_linear(x, w, b, stats_shift=True if training else None)
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        torch.testing.assert_close(pre.float(), ref, rtol=ulp, atol=ulp)
        none, partial_r = _ops.rowblock_linear(x, w, b, stats_shift=True, store=False)
        assert none is None
        if training:
            assert torch.equal(partial[0], partial_r[0]) and torch.equal(partial[1], partial_r[1])
        for addend in (None, res):
            h0, mean0, var0, saved = _ops._bn_forward(pre, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, act,
                                                      None, partial, addend=addend)
            h1, mean1, var1, saved1 = _ops._bn_forward(None, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, act,
                                                       None, partial_r if training else None, addend=addend, producer=(x, w, b))
            assert torch.equal(h0, h1) and torch.equal(mean0, mean1) and torch.equal(var0, var1)
        s1, s2 = _ops.bn_bwd_sums(pre, gh, saved, act)
        t1, t2 = _ops.rowblock_linear_bwd_sums(x, w, b, gh, saved1, act)
        scale = float(s1.abs().max()) + float(s2.abs().max())
        torch.testing.assert_close(t1 / scale, s1 / scale, rtol=0, atol=1e-6)
        torch.testing.assert_close(t2 / scale, s2 / scale, rtol=0, atol=1e-6)
        gpre0, u1, u2, sdx0 = _ops._bn_backward(pre, gh, saved, training, act, want_colsum=True)
        gpre1, cs1 = _ops.rowblock_linear_bwd_apply(x, w, b, gh, saved1, (u1, u2), act, training, True)
        assert torch.equal(gpre0, gpre1)
        torch.testing.assert_close(cs1, sdx0, rtol=1e-5, atol=1e-3)
        gx0 = ((gpre0 @ w).float() + g.float()).to(dtype)                      # library product, rounded, then the add
        gx1, dw1, v1, v2, sdx1 = _ops.bn_bwd_linear(None, gh, saved1, training, act, w, g, True, x=x, sums=(u1, u2), lin_bias=b)
        gs = float(gx0.float().abs().max())
        torch.testing.assert_close(gx1.float() / gs, gx0.float() / gs, rtol=0, atol=2 * ulp)       # (the library's k order differs)
        ref_gx, _ = _ops.rowblock_linear(gpre0, w.t().contiguous(), None, addend=g)
        assert torch.equal(gx1, ref_gx)
        torch.testing.assert_close(sdx1, sdx0, rtol=1e-5, atol=1e-3)
        dw_ref = gpre0.double().t() @ x.double()
        sc = float(dw_ref.abs().max())
        # (the library's batched split-K at this width: f32 slabs where torch.bmm takes out_dtype, else 16-bit slabs)
        torch.testing.assert_close(dw1.double() / sc, dw_ref / sc, rtol=0, atol=1e-5 if _ops._BMM_OUT_DTYPE[0] else 4e-3)
        _ops.FORCE_DW_BLOCKS = True                                            # and the 2 x 2 blocks of the weight-gradient kernel
        try:
            dw2 = _ops.weight_grad_splitk(gpre0, x, torch.float32)
        finally:
            _ops.FORCE_DW_BLOCKS = False
        torch.testing.assert_close(dw2.double() / sc, dw_ref / sc, rtol=0, atol=1e-5)


def test_tuple_block_without_stored_preactivation_equals_stored(dev):
    """NGNNConv.forward_residual with and without the stored pre-activation (module switch): output bit-identical, gradients equal
    up to the reduction order of the two backward channel sums; eval mode without gradients takes the two-stream forward."""
    import copy
    from pygho_amd import SparseTensor, _ops
    from pygho_amd.honn import Conv
    h = 128
    A, tid, xv, dd, n = _ngnn_inputs(dev, torch.bfloat16, graphs=700)
    torch.manual_seed(2)
    wgt = torch.randn(xv.shape, device=dev)
    layer = Conv.NGNNConv(h, h, "sum", "SS", dict(MLP)).to(dev)
    old = _ops.USE_RECOMPUTE_PRE
    res = {}
    try:
        for mode in (True, False):
            _ops.USE_RECOMPUTE_PRE = mode
            la = copy.deepcopy(layer).train(True)
            x = xv.clone().requires_grad_(True)
            av = A.values.clone().requires_grad_(True)
            out = la.forward_residual(SparseTensor(A.indices, av, A.shape, True), SparseTensor(tid, x, [n, n, h], True), dd)
            (out.values.float() * wgt).sum().backward()
            la.eval()
            with torch.no_grad():
                ev = la.forward_residual(SparseTensor(A.indices, A.values, A.shape, True), SparseTensor(tid, xv, [n, n, h], True), dd)
            res[mode] = (out.values.detach(), ev.values, x.grad, av.grad, {k: p.grad.float() for k, p in la.named_parameters()})
    finally:
        _ops.USE_RECOMPUTE_PRE = old
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for i in (2, 3):
        s = float(b[i].float().abs().max())
        torch.testing.assert_close(a[i].float() / s, b[i].float() / s, rtol=0, atol=1e-2)
    for k in b[4]:
        s = float(b[4][k].abs().max()) + 1e-6
        if k.endswith("lins.0.bias"):
            continue
        torch.testing.assert_close(a[4][k] / s, b[4][k] / s, rtol=0, atol=1e-3, msg=k)


@pytest.mark.parametrize("dtype,pool,ctx", [(torch.float32, "mean", True), (torch.float32, "sum", False), (torch.bfloat16, "mean", True)])
def test_gnnak_node_level_linear_equals_reference_wiring(dev, dtype, pool, ctx):
    """GNNAKConv (sparse) with the 3 d -> d map applied at node level before the broadcasts against the reference's literal wiring
    (unpool the three views, concatenate, tuple-wise MLP; module switch): output, gradients wrt X, the adjacency values and every
    parameter, BatchNorm running statistics; with and without the model loop's residual."""
    import copy
    from pygho_amd import SparseTensor, _ops
    from pygho_amd.honn import Conv
    h = 128
    A, tid, xv, dd, n = _ngnn_inputs(dev, dtype, graphs=700)
    torch.manual_seed(4)
    layer = Conv.GNNAKConv(h, h, "sum", pool, "SS", dict(MLP), dict(MLP), ctx=ctx).to(dev)
    wgt = torch.randn(xv.shape, device=dev)
    old = _ops.USE_NODE_LEVEL_LINEAR
    f32 = dtype == torch.float32
    for residual in (False, True):
        res = {}
        try:
            for mode in (True, False):
                _ops.USE_NODE_LEVEL_LINEAR = mode
                la = copy.deepcopy(layer)
                x = xv.clone().requires_grad_(True)
                av = A.values.clone().requires_grad_(True)
                Ax, X = SparseTensor(A.indices, av, A.shape, True), SparseTensor(tid, x, [n, n, h], True)
                out = la.forward_residual(Ax, X, dd) if residual else la.forward(Ax, X, dd)
                (out.values.float() * wgt).sum().backward()
                bn = la.lin.lins[1].norm
                res[mode] = (out.values.float(), x.grad.float(), av.grad.float(), {k: p.grad.float() for k, p in la.named_parameters()},
                             bn.running_mean.clone(), bn.running_var.clone())
        finally:
            _ops.USE_NODE_LEVEL_LINEAR = old
        a, b = res[True], res[False]
        tol = dict(rtol=2e-4, atol=2e-4) if f32 else dict(rtol=3e-2, atol=6e-2)
        torch.testing.assert_close(a[0], b[0], **tol)
        for i in (1, 2):
            s = float(b[i].abs().max())
            torch.testing.assert_close(a[i] / s, b[i] / s, rtol=0, atol=2e-4 if f32 else 3e-2)
        for k, ref in b[3].items():
            if k.endswith("lins.0.bias"):
                continue                                       # bias in front of a BatchNorm: zero up to rounding noise
            s = float(ref.abs().max()) + 1e-6
            torch.testing.assert_close(a[3][k] / s, ref / s, rtol=0, atol=2e-4 if f32 else 3e-2, msg=k)
        torch.testing.assert_close(a[4], b[4], rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(a[5], b[5], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_shared_adjacency_gradient_chain(dev, dtype):
    """three layers share A (the model loop of example/minimal.py:76-79): with the gradient chain every block hands A's values on
    as an autograd output and extends their gradient in its by-edge aggregation's epilogue; without it autograd adds the three
    per-edge gradients.  Same outputs (bit for bit), same gradients (one rounding of the sum instead of one per addend)."""
    import copy
    from pygho_amd import SparseTensor, _ops
    from pygho_amd.honn import Conv
    h = 128
    A, tid, xv, dd, n = _ngnn_inputs(dev, dtype, graphs=700)
    torch.manual_seed(3)
    layers = [Conv.NGNNConv(h, h, "sum", "SS", dict(MLP)).to(dev) for _ in range(3)]
    wgt = torch.randn(xv.shape, device=dev)
    old = _ops.USE_GRAD_CHAIN
    res = {}
    try:
        for mode in (True, False):
            _ops.USE_GRAD_CHAIN = mode
            ls = copy.deepcopy(layers)
            x = xv.clone().requires_grad_(True)
            av = A.values.clone().requires_grad_(True)
            Ax = SparseTensor(A.indices, av, A.shape, True)
            X = SparseTensor(tid, x, [n, n, h], True)
            for rep in range(2):                   # a second pass over the same A object: the links of the first must not be reused
                x.grad = av.grad = None
                for l in ls:
                    for p in l.parameters():
                        p.grad = None
                X = SparseTensor(tid, x, [n, n, h], True)
                dd_pass = dict(dd)
                dd_pass[Conv.GRAD_CHAIN_KEY] = {}  # what a model loop does once per forward pass
                for l in ls:
                    X = l.forward_residual(Ax, X, dd_pass)
                (X.values.float() * wgt).sum().backward()
                assert (len(dd_pass[Conv.GRAD_CHAIN_KEY]) == 1) == mode
            res[mode] = (X.values.detach(), av.grad, x.grad, [p.grad for l in ls for p in l.parameters()])
    finally:
        _ops.USE_GRAD_CHAIN = old
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0])
    s = float(b[1].float().abs().max())
    # bf16: the unchained sum rounds after every addend (up to one ulp = 2^-8 relative each), the chained one once
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=2.0 ** -6, atol=2.0 ** -8)   # (atol: addends that cancel)
    torch.testing.assert_close(a[1].float() / s, b[1].float() / s, **tol)
    assert torch.equal(a[2], b[2])
    for ga, gb in zip(a[3], b[3]):
        assert torch.equal(ga, gb)


@pytest.mark.parametrize("h", [64, 256])
def test_fused_residual_i2conv_3tuples(dev, h):
    """the same fused block on 3-tuples (I2Conv, BASELINE config 5 shape; d = 64 -> the d = 64 instantiation of the MFMA kernels,
    d = 256 = config 5's width -> the column-split streaming kernels: statistics pass, Linear + BatchNorm + act pass, backward sums /
    apply passes, input-gradient GEMM with the residual gradient in its epilogue): forward_residual against the unfused composition,
    values and input / parameter gradients, bf16."""
    import copy
    from pygho_amd import SparseTensor, synth
    from pygho_amd.honn import Conv
    dtype = torch.bfloat16
    hb = synth.make_batch(48, "i2", seed=9)
    n = hb.num_nodes
    assert hb.tupleid.shape[0] == 3 and hb.num_tuples >= 8192
    torch.manual_seed(4)
    A = SparseTensor(T(hb.edge_index, dev), (torch.randn(hb.num_edges, h, device=dev) * 0.3).to(dtype), [n, n, h], True)
    xv = torch.randn(hb.num_tuples, h, device=dev).to(dtype)
    tid = T(hb.tupleid, dev)
    dd = {k + "___acd": T(v, dev) for k, v in hb.acd.items()}
    w = torch.randn(xv.shape, device=dev)
    la = Conv.I2Conv(h, h, "sum", "SS", dict(MLP)).to(dev).train()
    lb = copy.deepcopy(la)
    res = {}
    for name, layer in (("fused", la), ("plain", lb)):
        x = xv.clone().requires_grad_(True)
        X = SparseTensor(tid, x, [n, n, n, h], True)
        if name == "fused":
            out = layer.forward_residual(A, X, dd)
        else:
            H = layer.lin.lins[2](layer.lin.lins[1](torch.nn.functional.linear(x, layer.lin.lins[0].weight.to(dtype),
                                                                               layer.lin.lins[0].bias.to(dtype))))
            out = X.add(layer.aggr.forward(A, SparseTensor(tid, H, [n, n, n, h], True), dd, X), True)
        (out.values.float() * w).sum().backward()
        res[name] = (out.values.float(), x.grad.float(), {k: p.grad.float() for k, p in layer.named_parameters()})
    fa, fb = res["fused"], res["plain"]
    torch.testing.assert_close(fa[0], fb[0], rtol=3e-2, atol=6e-2)
    torch.testing.assert_close(fa[1], fb[1], rtol=3e-2, atol=6e-2)
    for k in fa[2]:
        if k.endswith("lins.0.bias"):
            continue                                   # zero up to rounding noise in front of a BatchNorm
        s = float(fb[2][k].abs().max()) + 1e-6
        torch.testing.assert_close(fa[2][k] / s, fb[2][k] / s, rtol=0, atol=3e-2, msg=k)


def test_model_fused_paths_match_plain_composition(dev):
    """the whole NGNN model (example/minimal.py) with every fused path on (table-indexed tuple initialisation, fused tuple
    block with MFMA GEMMs, one-pass backward with the weight gradient) against the same model with the paths switched off
    one level down: loss and all parameter gradients, bf16 activations."""
    from pygho_amd import _ops, synth
    from pygho_amd.ngnn import SpModel
    hb = synth.make_batch(512, "zinc", seed=21)
    dd = synth.to_datadict(hb, dev)
    y = dd["y"].unsqueeze(-1)
    flags = ("USE_ADJ_TABLE", "USE_TABLE_PRODUCT", "USE_FUSED_DW", "USE_BN_BWD_LINEAR", "USE_ROWBLOCK_LINEAR")
    saved = {f: getattr(_ops, f) for f in flags}
    res = {}
    try:
        for mode in (True, False):
            for f in flags:
                setattr(_ops, f, mode)
            torch.manual_seed(0)
            model = SpModel(1, 3, 128, act_dtype=torch.bfloat16).to(dev)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(y, pred.float())
            loss.backward()
            res[mode] = (float(loss.detach()), {k: p.grad.float().clone() for k, p in model.named_parameters() if p.grad is not None})
    finally:
        for f, v in saved.items():
            setattr(_ops, f, v)
    assert abs(res[True][0] - res[False][0]) <= 2e-2 * abs(res[False][0]) + 1e-3
    assert res[True][1].keys() == res[False][1].keys()
    for k, ref in res[False][1].items():
        if k.endswith(".lins.0.bias"):
            continue                                       # bias in front of a BatchNorm: zero up to rounding noise
        s = float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(res[True][1][k] / s, ref / s, rtol=0, atol=8e-2, msg=k)


@pytest.mark.parametrize("mode", ["SS", "DD"])
def test_sunconv_restructured_equals_reference_wiring(dev, mode):
    """SUNConv.forward (per-type linear map pulled through the unpooling broadcasts, no (nnz, 7 d) concatenation) against
    SUNConv.forward_concat (the reference's literal wiring, Conv.py:338-362): outputs, input and parameter gradients."""
    import copy
    from pygho_amd import MaskedTensor, SparseTensor, synth
    from pygho_amd.honn import Conv
    h = 16
    dn = synth.make_dense_batch(5, seed=6, hidden=h, clip_nodes=9)
    torch.manual_seed(1)
    mlp = dict(MLP, norm="none")
    la = Conv.SUNConv(h, h, "sum", "mean", mode, dict(mlp), dict(mlp)).to(dev)
    lb = copy.deepcopy(la)
    if mode == "DD":
        xraw = T(dn["X"], dev)
        mk = lambda v: (MaskedTensor(T(dn["A"], dev), T(dn["Amask"], dev)), MaskedTensor(v, T(dn["Xmask"], dev)), {})
        vals_of = lambda r: r.data * T(dn["Xmask"], dev)[..., None]
    else:
        off = np.concatenate(([0], np.cumsum(dn["nodemask"].sum(1))))
        bi, ii, jj = np.nonzero(dn["Xmask"])
        tid = np.stack((ii + off[bi], jj + off[bi]))
        eb, ei_, ej = np.nonzero(dn["Amask"])
        eidx = np.stack((ei_ + off[eb], ej + off[eb]))
        n = int(off[-1])
        xraw = T(dn["X"][bi, ii, jj], dev)
        acd = T(synth.host_plan_acd(tid, tid, 1, eidx, 0), dev)
        As = SparseTensor(T(eidx, dev), T(dn["A"][eb, ei_, ej], dev), [n, n, h], True)
        mk = lambda v: (As, SparseTensor(T(tid, dev), v, [n, n, h], True), {"X___X___1___A___0___acd": acd})
        vals_of = lambda r: r.values
    w = torch.randn(xraw.shape, device=dev)
    res = []
    for layer, fn in ((la, "forward"), (lb, "forward_concat")):
        x = xraw.clone().requires_grad_(True)
        A, X, dd = mk(x)
        out = vals_of(getattr(layer, fn)(A, X, dd))
        (out * w).sum().backward()
        res.append((out.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=2e-4, atol=2e-4)
    for k in res[1][2]:
        torch.testing.assert_close(res[0][2][k], res[1][2][k], rtol=5e-4, atol=5e-4, msg=k)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [128, 64])
def test_weight_grad_kernel(dev, dtype, d):
    """stand-alone weight gradient g^T x (+ column sums of g) on the transpose-read MFMA kernel against an f64 product;
    asymmetric operands (a transposed result would show), ragged row counts."""
    from pygho_amd import _ops
    torch.manual_seed(9)
    for m in (8192, 100_003, 400_000):
        g = torch.randn(m, d, device=dev).to(dtype)
        x = (torch.randn(m, d, device=dev) * 0.5 + 0.1).to(dtype)
        gw, cs = _ops.weight_grad_splitk(g, x, torch.float32, want_colsum=True)
        ref = g.double().t() @ x.double()
        s = float(ref.abs().max())
        torch.testing.assert_close(gw.double() / s, ref / s, rtol=0, atol=2e-6)
        torch.testing.assert_close(cs.double(), g.double().sum(0), rtol=1e-5, atol=1e-2)
        assert torch.equal(_ops.weight_grad_splitk(g, x, torch.float32), gw)
        # in_features = 3 d (SSWLConv's concatenated input): one launch per column block, strided x
        x3 = (torch.randn(m, 3 * d, device=dev) * 0.5).to(dtype)
        gw3 = _ops.weight_grad_splitk(g, x3, torch.float32)
        ref3 = g.double().t() @ x3.double()
        s3 = float(ref3.abs().max())
        torch.testing.assert_close(gw3.double() / s3, ref3 / s3, rtol=0, atol=2e-6)


def test_hip_graph_replay_equals_eager_steps(dev):
    """a whole NGNN training step (forward, backward, AdamW) captured into a HIP graph and replayed k times leaves the
    parameters where k eager steps leave them: every kernel goes through the C ABI on the capture stream, nothing
    synchronises or allocates behind torch's back."""
    import copy
    from pygho_amd import synth
    from pygho_amd.graphs import GraphedStep
    from pygho_amd.ngnn import SpModel
    hb = synth.make_batch(256, "zinc", seed=31)
    dd = synth.to_datadict(hb, dev)
    y = dd["y"].unsqueeze(-1)
    torch.manual_seed(0)
    m_eager = SpModel(1, 2, 128, act_dtype=torch.bfloat16).to(dev)
    m_graph = copy.deepcopy(m_eager)

    def make_step(model, opt):
        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(y, pred.float())
            loss.backward()
            opt.step()
            return loss.detach()
        return step

    k, warm = 4, 2
    s_e = make_step(m_eager, torch.optim.AdamW(m_eager.parameters(), lr=1e-3, capturable=True))
    for _ in range(warm + k):                 # capture records the step without executing it
        s_e()
    gs = GraphedStep(make_step(m_graph, torch.optim.AdamW(m_graph.parameters(), lr=1e-3, capturable=True)), warmup=warm)
    for _ in range(k):                        # warm-up (2) + k replays = the eager count
        loss = gs.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    for (n1, p1), (_, p2) in zip(m_eager.named_parameters(), m_graph.named_parameters()):
        torch.testing.assert_close(p1, p2, rtol=1e-4, atol=1e-5, msg=n1)
    for (n1, b1), (_, b2) in zip(m_eager.named_buffers(), m_graph.named_buffers()):
        torch.testing.assert_close(b1.float(), b2.float(), rtol=1e-4, atol=1e-5, msg=n1)


def test_hip_graph_capture_after_eager_steps_and_eager_after_replay(dev):
    """the SAME model trains eagerly, is then captured and replayed, and is evaluated eagerly afterwards.  (i) capture after eager
    steps: the 16-bit parameter copies of `_ops.ParamCastArena` must not carry autograd state from one iteration into the next
    (a view object that kept the previous iteration's node alive dragged the eager stream into the capture: crash in
    hipStreamEndCapture); (ii) eager use after replays: a replayed optimizer step moves the parameters without moving their
    version counters, so the arena must be told (`invalidate_cast_arenas`) -- the eager forward has to see the CURRENT weights."""
    from pygho_amd import synth, _ops
    from pygho_amd.graphs import GraphedStep
    from pygho_amd.ngnn import SpModel
    hb = synth.make_batch(256, "zinc", seed=32)
    dd = synth.to_datadict(hb, dev)
    y = dd["y"].unsqueeze(-1)
    torch.manual_seed(0)
    model = SpModel(1, 2, 128, act_dtype=torch.bfloat16).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=3e-3, capturable=True)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(y, pred.float())
        loss.backward()
        opt.step()
        return loss.detach()

    for _ in range(3):
        step()                                         # eager: creates the arena on the default stream
    torch.cuda.synchronize()
    gs = GraphedStep(step, warmup=2)
    for _ in range(6):
        loss = gs.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        after = model(dd).float()
    old = _ops.USE_CAST_ARENA
    try:
        _ops.USE_CAST_ARENA = False                    # direct casts of the current parameters
        model.__dict__.pop("_pygho_cast_arena", None)
        _ops._ARENA_OF.clear()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            want = model(dd).float()
    finally:
        _ops.USE_CAST_ARENA = old
    assert torch.equal(after, want)


@pytest.mark.parametrize("case", ["f32_small", "bf16_rowblock"])
def test_sunconv_dense_fused_passes(dev, case):
    """SUNConv on the padded layout with the fused node-view / recombination passes (`_ops.USE_PAIR_COMBINE`: pair_views,
    pair_linear_mix -> pygho_masked_pair_combine, and at d = 128 bf16 the streaming GEMM kernel with the first product in
    the epilogue of the second) against the same layer with separate broadcast / add / select passes."""
    import copy
    from pygho_amd import MaskedTensor, synth, _ops
    from pygho_amd.honn import Conv
    if case == "f32_small":
        h, nb, kw, amp, tol = 16, 5, dict(clip_nodes=9), False, dict(rtol=2e-5, atol=2e-5)
    else:
        h, nb, kw, amp, tol = 128, 8, dict(nmax=37), True, dict(rtol=0, atol=6e-2)
    dn = synth.make_dense_batch(nb, seed=6, hidden=h, **kw)
    torch.manual_seed(1)
    mlp = dict(MLP, norm="bn")
    layer = Conv.SUNConv(h, h, "sum", "mean", "DD", dict(mlp), dict(mlp)).to(dev)
    dt = torch.bfloat16 if amp else torch.float32
    xraw = T(dn["X"], dev).to(dt)
    Amt = MaskedTensor(T(dn["A"], dev).to(dt), T(dn["Amask"], dev), 0.0, True)
    xm = T(dn["Xmask"], dev)
    w = torch.randn(xraw.shape, device=dev).to(dt)
    res = {}
    for fused in (True, False):
        _ops.USE_PAIR_COMBINE = fused
        try:
            lay = copy.deepcopy(layer)
            x = xraw.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out = lay(Amt, MaskedTensor(x, xm, 0.0, True), {})
            o = out.data * xm[..., None]
            o.backward(w)
            res[fused] = (o.detach().float(), x.grad.float() * xm[..., None], {k: p.grad.float() for k, p in lay.named_parameters()})
        finally:
            _ops.USE_PAIR_COMBINE = True
    for i in (0, 1):
        s = float(res[False][i].abs().max()) + 1e-6
        torch.testing.assert_close(res[True][i] / s, res[False][i] / s, **tol)
    for k, ref in res[False][2].items():
        if k.endswith(".lins.0.bias"):
            continue                                       # bias in front of a BatchNorm: zero up to rounding noise
        s = float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(res[True][2][k] / s, ref / s, msg=k, **tol)


@pytest.mark.parametrize("name", ["SSWLConv", "DSSGNNConv"])
def test_concat_block_layers(dev, name):
    """layers whose MLP takes a concatenation ([X, X A, A X] in SSWLConv, reference Conv.py:98-103; [local, shared] in
    DSSGNNConv, :190-196) with `_ops.concat_block` (chained streaming GEMMs, no (nnz, k d) tensor, one backward pass per input)
    against the same layer with the literal concatenation: outputs, input gradient and every parameter gradient, bf16."""
    import copy
    from pygho_amd import SparseTensor, synth, _ops
    from pygho_amd.honn import Conv
    from pygho_amd.honn.SpOperator import parse_precomputekey
    h = 128
    torch.manual_seed(3)
    if name == "SSWLConv":
        layer = Conv.SSWLConv(h, h, "sum", "SS", dict(MLP)).to(dev)
    else:
        layer = Conv.DSSGNNConv(h, h, "sum", "sum", "mean", "SS", dict(MLP)).to(dev)
    hb = synth.make_batch(64, "zinc", seed=21, keys=tuple(parse_precomputekey(layer)))
    dd = synth.to_datadict(hb, dev)
    X0, A0 = dd["X"], dd["A"]
    assert X0.nnz >= 8192
    xv0 = torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16)
    av = (torch.randn(A0.nnz, h, device=dev) * 0.5).to(torch.bfloat16)
    A = SparseTensor(A0.indices, av, list(A0.shape[:2]) + [h], True)
    w = torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16)
    res = {}
    for fused in (True, False):
        _ops.USE_CONCAT_BLOCK = fused
        try:
            lay = copy.deepcopy(layer)
            xv = xv0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = lay(A, SparseTensor(X0.indices, xv, list(X0.shape[:2]) + [h], True), dd)
            out.values.backward(w)
            res[fused] = (out.values.detach().float(), xv.grad.float(), {k: p.grad.float() for k, p in lay.named_parameters()},
                          {k: v.clone() for k, v in lay.state_dict().items() if "running" in k})
        finally:
            _ops.USE_CONCAT_BLOCK = True
    for i in (0, 1):
        s = float(res[False][i].abs().max()) + 1e-6
        torch.testing.assert_close(res[True][i] / s, res[False][i] / s, rtol=0, atol=4e-2)
    for k, ref in res[False][2].items():
        if k.endswith(".lins.0.bias"):
            continue                                       # bias in front of a BatchNorm: zero up to rounding noise
        s = float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(res[True][2][k] / s, ref / s, rtol=0, atol=4e-2, msg=k)
    for k, ref in res[False][3].items():                   # running statistics updated the same way
        torch.testing.assert_close(res[True][3][k].float(), ref.float(), rtol=2e-2, atol=2e-2, msg=k)


@pytest.mark.parametrize("case", ["f32_small", "bf16_rowblock"])
def test_sunconv_sparse_fused_passes(dev, case):
    """SUNConv mode "SS" with the fused node-view / recombination passes (sparse_pair_views, sparse_pair_linear_mix ->
    pygho_pair_gather_combine) against the same layer with separate gather / add / select passes."""
    import copy
    from pygho_amd import SparseTensor, synth, _ops
    from pygho_amd.honn import Conv
    from pygho_amd.honn.SpOperator import parse_precomputekey
    if case == "f32_small":
        h, nb, amp, tol = 16, 6, False, dict(rtol=0, atol=2e-5)
    else:
        h, nb, amp, tol = 128, 64, True, dict(rtol=0, atol=6e-2)
    torch.manual_seed(1)
    layer = Conv.SUNConv(h, h, "sum", "mean", "SS", dict(MLP), dict(MLP)).to(dev)
    hb = synth.make_batch(nb, "zinc", seed=31, keys=tuple(parse_precomputekey(layer)))
    dd = synth.to_datadict(hb, dev)
    X0, A0 = dd["X"], dd["A"]
    dt = torch.bfloat16 if amp else torch.float32
    xv0 = torch.randn(X0.nnz, h, device=dev).to(dt)
    A = SparseTensor(A0.indices, (torch.randn(A0.nnz, h, device=dev) * 0.5).to(dt), list(A0.shape[:2]) + [h], True)
    w = torch.randn(X0.nnz, h, device=dev).to(dt)
    res = {}
    for fused in (True, False):
        _ops.USE_PAIR_COMBINE = fused
        try:
            lay = copy.deepcopy(layer)
            xv = xv0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out = lay(A, SparseTensor(X0.indices, xv, list(X0.shape[:2]) + [h], True), dd)
            out.values.backward(w)
            res[fused] = (out.values.detach().float(), xv.grad.float(), {k: p.grad.float() for k, p in lay.named_parameters()})
        finally:
            _ops.USE_PAIR_COMBINE = True
    for i in (0, 1):
        s = float(res[False][i].abs().max()) + 1e-6
        torch.testing.assert_close(res[True][i] / s, res[False][i] / s, **tol)
    for k, ref in res[False][2].items():
        if k.endswith(".lins.0.bias"):
            continue                                       # bias in front of a BatchNorm: zero up to rounding noise
        s = float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(res[True][2][k] / s, ref / s, msg=k, **tol)


@pytest.mark.parametrize("layer_name", ["NGNNConv", "SSWLConv", "DSSGNNConv", "SUNConv"])
def test_sd_mode_equals_dd_mode(dev, layer_name):
    """mode "SD" (sparse (b, n, n) adjacency from hodata.to_sparse_adj, dense representation; spmamm) gives what mode "DD"
    (dense adjacency; masked bmm / neighbour lists) gives on the same batch: outputs and input gradients."""
    import copy
    from pygho_amd import MaskedTensor, synth
    from pygho_amd.hodata import to_dense_adj, to_sparse_adj
    from pygho_amd.honn import Conv
    h = 16
    dn = synth.make_dense_batch(6, seed=9, hidden=h, clip_nodes=9)
    b, n = dn["nodemask"].shape
    eb, er, ec = np.nonzero(dn["Amask"])
    ei, ebt, ea = T(np.stack((er, ec)), dev), T(eb, dev), T(dn["A"][eb, er, ec], dev)
    A_dd = to_dense_adj(ei, ebt, ea, n, b)
    A_sd = to_sparse_adj(ei, ebt, ea, n, b)
    torch.manual_seed(0)
    mlp = dict(MLP, norm="none")
    mk = {"NGNNConv": lambda m: Conv.NGNNConv(h, h, "sum", m, dict(mlp)),
          "SSWLConv": lambda m: Conv.SSWLConv(h, h, "sum", m, dict(mlp)),
          "DSSGNNConv": lambda m: Conv.DSSGNNConv(h, h, "sum", "sum", "mean", m, dict(mlp)),
          "SUNConv": lambda m: Conv.SUNConv(h, h, "sum", "mean", m, dict(mlp), dict(mlp))}[layer_name]
    dd_layer = mk("DD").to(dev)
    sd_layer = mk("SD").to(dev)
    sd_layer.load_state_dict(dd_layer.state_dict())
    xm = T(dn["Xmask"], dev)
    w = torch.randn(dn["X"].shape, device=dev)
    res = []
    for layer, A in ((dd_layer, A_dd), (sd_layer, A_sd)):
        x = T(dn["X"], dev).requires_grad_(True)
        out = layer(A, MaskedTensor(x, xm, 0.0, True), {})
        o = out.data * xm[..., None]
        o.backward(w)
        res.append((o.detach(), x.grad * xm[..., None]))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-5, atol=1e-5)


def test_batch_prefetcher_with_plan_preparation(dev):
    """collate.BatchPrefetcher (batch k + 1 collated and its index plans built on a side stream by SpModel.prepare while batch k
    trains): the batches are the ones plain collation gives, the prepared step performs no host synchronisation at all, and
    training through the prefetcher leaves the parameters exactly where plain collation leaves them."""
    import copy
    from pygho_amd import synth
    from pygho_amd.collate import BatchPrefetcher, DeviceGraphStore
    from pygho_amd.ngnn import SpModel
    rng = np.random.default_rng(3)
    recs = [synth.make_graph(rng, "zinc", 3, ("X___X___1___A___0",)) for _ in range(96)]
    store = DeviceGraphStore(recs, dev)
    gen = torch.Generator().manual_seed(1)
    ids = [torch.randperm(96, generator=gen)[:48] for _ in range(4)]
    torch.manual_seed(0)
    base = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev)

    def train(batches, model):
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        for dd in batches:
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float()).backward()
            opt.step()
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in model.state_dict().items()}

    m1, m2 = copy.deepcopy(base), copy.deepcopy(base)
    plain = train((store.collate(i) for i in ids), m1)
    pre = train(BatchPrefetcher(store, ids, m2.prepare), m2)
    for k in plain:
        assert torch.equal(plain[k], pre[k]), k
    # gated (round 6): the next batch's upload + collate kernel start no earlier than the mark the consumer sets inside its step
    # (bench.py: a forward pre-hook on the model's first graph-level module) -- same batches, same training
    m3 = copy.deepcopy(base)
    pf = BatchPrefetcher(store, ids, gated=True)
    hook = m3.lpool.register_forward_pre_hook(lambda _m, _a: pf.gate())
    gated = train(pf, m3)
    hook.remove()
    for k in plain:
        assert torch.equal(plain[k], gated[k]), k
    # same batches, and no synchronisation left inside a prepared step
    for a, b in zip((store.collate(i) for i in ids), BatchPrefetcher(store, ids)):
        assert torch.equal(a["X"].indices, b["X"].indices) and torch.equal(a["A"].values, b["A"].values)
        assert torch.equal(a["X___X___1___A___0___acd"], b["X___X___1___A___0___acd"]) and torch.equal(a["y"], b["y"])
    dd = store.collate(ids[0])
    m2.prepare(dd)
    calls = []
    orig = torch.Tensor.item
    torch.Tensor.item = lambda self: (calls.append(1) if self.is_cuda else None, orig(self))[1]
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = m2(dd)
        torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float()).backward()
    finally:
        torch.Tensor.item = orig
    assert not calls, f"{len(calls)} host synchronisations inside a prepared step"


def _dense_inputs_of(hb, dev):
    """the dense-layout inputs (integer MaskedTensors) of a sparse host batch, built with the device builders of hodata.MaData"""
    from pygho_amd.hodata import to_dense_adj, to_dense_x
    counts = np.bincount(hb.batch, minlength=hb.num_graphs)
    ptr = np.concatenate(([0], np.cumsum(counts)))
    n = int(counts.max())
    loc = lambda idx: idx - ptr[hb.batch[idx]]
    eb, tb = hb.batch[hb.edge_index[0]], hb.batch[hb.tupleid[0]]
    x = to_dense_x(T(hb.x, dev), T(ptr, dev))
    A = to_dense_adj(T(np.stack((loc(hb.edge_index[0]), loc(hb.edge_index[1]))), dev), T(eb, dev), T(hb.edge_attr, dev), n, hb.num_graphs)
    X = to_dense_adj(T(np.stack((loc(hb.tupleid[0]), loc(hb.tupleid[1]))), dev), T(tb, dev), T(hb.tuplefeat, dev), n, hb.num_graphs)
    return {"x": x, "A": A, "X": X}


@pytest.mark.parametrize("conv", ["NGNN", "SSWL", "DSSGNN", "GNNAK", "SUN", "PPGN"])
def test_zinc_models_sparse_and_dense_layouts_agree(dev, conv):
    """the example/zinc.py models for every 2-tuple layer family: the sparse-layout model and the dense-layout model with the
    same parameters give the same graph-level predictions and the same parameter gradients on the same graphs (LayerNorm MLPs:
    BatchNorm would see the padding rows of the dense layout)."""
    from pygho_amd import synth
    from pygho_amd.honn.SpOperator import parse_precomputekey
    from pygho_amd.models import MaModel, SpModel
    mlp = {"norm": "ln", "act": "silu", "dp": 0.0}
    torch.manual_seed(5)
    ma = MaModel(conv, num_layer=2, hiddim=32, mlp=mlp, outlayer=1).to(dev)
    sp = SpModel(conv, num_layer=2, hiddim=32, mlp=mlp, outlayer=1).to(dev)
    sd = {k: v for k, v in ma.state_dict().items()}
    # the reference's two models assign the tuple-initialisation factors differently: dense lin0 <-> node j, lin1 <-> root i
    # (zinc.py:196-199), sparse lin0 <-> indices[0] = i, lin1 <-> indices[1] = j (zinc.py:270-276); mirrored here, swapped for the comparison
    for part in ("weight", "bias"):
        sd[f"lin_tupleinit0.{part}"], sd[f"lin_tupleinit1.{part}"] = sd[f"lin_tupleinit1.{part}"], sd[f"lin_tupleinit0.{part}"]
    missing = sp.load_state_dict(sd, strict=False)
    assert set(missing.missing_keys) <= {"lin_tupleinit2.weight", "lin_tupleinit2.bias"} and not missing.unexpected_keys
    hb = synth.make_batch(6, "zinc", seed=17, keys=tuple(parse_precomputekey(sp)))
    dd = synth.to_datadict(hb, dev)
    y = torch.randn(hb.num_graphs, 1, device=dev)
    out_sp = sp(dd)
    (out_sp * y).sum().backward()
    out_ma = ma(_dense_inputs_of(hb, dev))
    (out_ma * y).sum().backward()
    torch.testing.assert_close(out_sp, out_ma, rtol=2e-4, atol=2e-4)
    gs, gm = dict(sp.named_parameters()), dict(ma.named_parameters())
    swap = {"lin_tupleinit0": "lin_tupleinit1", "lin_tupleinit1": "lin_tupleinit0"}
    for k, p in gm.items():
        if p.grad is None or k.startswith("data_encoder.ea_encoder"):
            continue                               # the dense adjacency embedding carries a padding row
        head, _, tail = k.partition(".")
        k = swap.get(head, head) + "." + tail if head in swap else k
        s = float(p.grad.abs().max()) + 1e-6
        torch.testing.assert_close(gs[k].grad / s, p.grad / s, rtol=0, atol=2e-3, msg=k)


def test_zinc_model_i2gnn_trains(dev):
    """the 3-tuple family (I2GNN: two tuple features, three tuple-initialisation factors, two-stage subgraph pooling) runs a
    training step on an I2-shape batch in bf16 and its loss decreases."""
    from pygho_amd import synth
    from pygho_amd.honn.SpOperator import parse_precomputekey
    from pygho_amd.models import SpModel
    torch.manual_seed(0)
    model = SpModel("I2GNN", num_layer=2, hiddim=64, act_dtype=torch.bfloat16).to(dev)
    hb = synth.make_batch(24, "i2", seed=3, keys=tuple(parse_precomputekey(model)))
    dd = synth.to_datadict(hb, dev, "i2")
    y = dd["y"].unsqueeze(-1)
    opt = torch.optim.AdamW(model.parameters(), lr=3e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(y, pred.float())
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("aggr", ["sum", "mean"])
def test_sswl_forward_residual_fused(dev, aggr):
    """SSWLConv.forward_residual (one autograd node, `_ops.sswl_block`: residual row added in the concat block's activation pass,
    the three contributions to the gradient of X and the two to the gradient of A summed in the aggregation launches' epilogues)
    against X.add(conv.forward(...), True) with the fused blocks switched off: bf16, outputs, input / adjacency / parameter gradients."""
    import copy
    from pygho_amd import SparseTensor, synth, _ops
    from pygho_amd.honn import Conv
    from pygho_amd.honn.SpOperator import parse_precomputekey
    h = 128
    torch.manual_seed(3)
    layer = Conv.SSWLConv(h, h, aggr, "SS", dict(MLP)).to(dev)
    hb = synth.make_batch(64, "zinc", seed=22, keys=tuple(parse_precomputekey(layer)))
    dd = synth.to_datadict(hb, dev)
    X0, A0 = dd["X"], dd["A"]
    xv0 = torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16)
    av0 = (torch.randn(A0.nnz, h, device=dev) * 0.5).to(torch.bfloat16)
    w = torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16)
    res = {}
    for fused in (True, False):
        _ops.USE_CONCAT_BLOCK = fused
        try:
            lay = copy.deepcopy(layer)
            xv, av = xv0.clone().requires_grad_(True), av0.clone().requires_grad_(True)
            X = SparseTensor(X0.indices, xv, list(X0.shape[:2]) + [h], True)
            A = SparseTensor(A0.indices, av, list(A0.shape[:2]) + [h], True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = lay.forward_residual(A, X, dd) if fused else X.add(lay.forward(A, X, dd), True)
            if fused:
                assert type(out.values.grad_fn).__name__ == "_SSWLBlockBackward"
            out.values.backward(w)
            res[fused] = (out.values.detach().float(), xv.grad.float(), {k: p.grad.float() for k, p in lay.named_parameters()},
                          av.grad.float())
        finally:
            _ops.USE_CONCAT_BLOCK = True
    for i in (0, 1, 3):
        s = float(res[False][i].abs().max()) + 1e-6
        torch.testing.assert_close(res[True][i] / s, res[False][i] / s, rtol=0, atol=4e-2)
    for k, ref in res[False][2].items():
        if k.endswith(".lins.0.bias"):
            continue
        s = float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(res[True][2][k] / s, ref / s, rtol=0, atol=4e-2, msg=k)


@pytest.mark.parametrize("conv", ["NGNN", "SSWL", "DSSGNN", "GNNAK", "SUN", "PPGN"])
def test_zinc_dense_models_train_under_autocast(dev, conv):
    """the dense-layout model of every 2-tuple family runs training steps under bf16 autocast (f32 and bf16 activations: mixed
    operand dtypes reach the fused passes) and its loss decreases; the padding row of the adjacency embedding stays zero."""
    from pygho_amd import synth
    from pygho_amd.models import MaModel
    hb = synth.make_batch(16, "zinc", seed=19)
    y = T(hb.y, dev).unsqueeze(-1)
    for act_dtype in (None, torch.bfloat16):
        torch.manual_seed(0)
        model = MaModel(conv, num_layer=2, hiddim=64, act_dtype=act_dtype).to(dev)
        opt = torch.optim.AdamW(model.parameters(), lr=3e-3)
        dd = _dense_inputs_of(hb, dev)
        losses = []
        for _ in range(10):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dict(dd))
            loss = torch.nn.functional.l1_loss(y, pred.float())
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert all(np.isfinite(losses)) and losses[-1] < losses[0], (conv, act_dtype, losses)
        assert float(model.data_encoder.ea_encoder.weight[0].detach().abs().sum()) == 0.0


@pytest.mark.parametrize("name,residual", [("SSWLConv", False), ("SSWLConv", True), ("DSSGNNConv", False), ("GNNAKConv", False)])
def test_concat_block_dense_layers(dev, name, residual):
    """the concatenating layers on the padded layout ("DD"): `_cat_apply` runs the fused concat block over the zero-filled padded
    rows (what catvalue + tuplewiseapply feed the MLP, reference MaTensor.py:264-270, 318-330) — against the same layer with the
    literal concatenation, valid entries of the output, input gradient, parameter gradients and running statistics, bf16."""
    import copy
    from pygho_amd import MaskedTensor, synth, _ops
    from pygho_amd.honn import Conv
    h = 128
    dn = synth.make_dense_batch(8, seed=8, hidden=h, nmax=37)
    torch.manual_seed(2)
    if name == "SSWLConv":
        layer = Conv.SSWLConv(h, h, "sum", "DD", dict(MLP)).to(dev)
    elif name == "DSSGNNConv":
        layer = Conv.DSSGNNConv(h, h, "sum", "sum", "mean", "DD", dict(MLP)).to(dev)
    else:
        layer = Conv.GNNAKConv(h, h, "sum", "mean", "DD", dict(MLP), dict(MLP)).to(dev)
    xraw = T(dn["X"], dev).to(torch.bfloat16)
    Amt = MaskedTensor(T(dn["A"], dev).to(torch.bfloat16), T(dn["Amask"], dev), 0.0, True)
    xm = T(dn["Xmask"], dev)
    w = torch.randn(xraw.shape, device=dev).to(torch.bfloat16)
    res = {}
    for fused in (True, False):
        _ops.USE_CONCAT_BLOCK = fused
        try:
            lay = copy.deepcopy(layer)
            x = xraw.clone().requires_grad_(True)
            X = MaskedTensor(x, xm, 0.0, True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if residual:
                    out = lay.forward_residual(Amt, X, {}) if fused else X.add(lay.forward(Amt, X, {}), True)
                else:
                    out = lay(Amt, X, {})
            o = out.data * xm[..., None]
            o.backward(w)
            res[fused] = (o.detach().float(), x.grad.float() * xm[..., None], {k: p.grad.float() for k, p in lay.named_parameters()},
                          {k: v.clone() for k, v in lay.state_dict().items() if "running" in k})
        finally:
            _ops.USE_CONCAT_BLOCK = True
    for i in (0, 1):
        s = float(res[False][i].abs().max()) + 1e-6
        torch.testing.assert_close(res[True][i] / s, res[False][i] / s, rtol=0, atol=4e-2)
    for k, ref in res[False][2].items():
        if k.endswith(".lins.0.bias"):
            continue
        s = float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(res[True][2][k] / s, ref / s, rtol=0, atol=4e-2, msg=k)
    for k, ref in res[False][3].items():
        torch.testing.assert_close(res[True][3][k].float(), ref.float(), rtol=2e-2, atol=2e-2, msg=k)


@pytest.mark.parametrize("layout", ["sparse", "dense"])
def test_sswl_fused_update_in_eval_mode_without_grad(dev, layout):
    """inference (BatchNorm in eval mode, torch.no_grad) through the fused SSWL update — one autograd node on the sparse layout,
    the concat block over the padded rows on the dense one — equals the literal concatenation path, and leaves the running
    statistics untouched."""
    import copy
    from pygho_amd import MaskedTensor, SparseTensor, synth, _ops
    from pygho_amd.honn import Conv
    from pygho_amd.honn.SpOperator import parse_precomputekey
    h = 128
    torch.manual_seed(4)
    layer = Conv.SSWLConv(h, h, "sum", "SS" if layout == "sparse" else "DD", dict(MLP)).to(dev)
    with torch.no_grad():
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 2.0)
    layer.eval()
    if layout == "sparse":
        hb = synth.make_batch(64, "zinc", seed=23, keys=tuple(parse_precomputekey(layer)))
        dd = synth.to_datadict(hb, dev)
        X0, A0 = dd["X"], dd["A"]
        X = SparseTensor(X0.indices, torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16), list(X0.shape[:2]) + [h], True)
        A = SparseTensor(A0.indices, (torch.randn(A0.nnz, h, device=dev) * 0.5).to(torch.bfloat16), list(A0.shape[:2]) + [h], True)
        valid = None
    else:
        dn = synth.make_dense_batch(8, seed=8, hidden=h, nmax=37)
        dd = {}
        X = MaskedTensor(T(dn["X"], dev).to(torch.bfloat16), T(dn["Xmask"], dev), 0.0, True)
        A = MaskedTensor(T(dn["A"], dev).to(torch.bfloat16), T(dn["Amask"], dev), 0.0, True)
        valid = T(dn["Xmask"], dev)[..., None]
    stats = {k: v.clone() for k, v in layer.state_dict().items() if "running" in k}
    res = {}
    for fused in (True, False):
        _ops.USE_CONCAT_BLOCK = fused
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                out = layer.forward_residual(A, X, dd) if fused else X.add(layer.forward(A, X, dd), True)
            res[fused] = (out.values if layout == "sparse" else out.data * valid).float()
        finally:
            _ops.USE_CONCAT_BLOCK = True
    s = float(res[False].abs().max()) + 1e-6
    torch.testing.assert_close(res[True] / s, res[False] / s, rtol=0, atol=2e-2)
    for k, v in layer.state_dict().items():
        if "running" in k:
            assert torch.equal(v, stats[k]), k


@pytest.mark.parametrize("layout", ["sparse", "dense"])
@pytest.mark.parametrize("name", ["DSSGNNConv", "GNNAKConv"])
def test_forward_residual_of_concatenating_layers(dev, name, layout):
    """DSSGNNConv / GNNAKConv.forward_residual (X added inside the concat block's activation pass as a separate residual operand,
    its gradient = the output gradient) against X.add(conv.forward(A, X, datadict), True) with the fused block switched off:
    bf16, outputs on valid entries, input gradient, parameter gradients."""
    import copy
    from pygho_amd import MaskedTensor, SparseTensor, synth, _ops
    from pygho_amd.honn import Conv
    from pygho_amd.honn.SpOperator import parse_precomputekey
    h = 128
    mode = "SS" if layout == "sparse" else "DD"
    torch.manual_seed(6)
    if name == "DSSGNNConv":
        layer = Conv.DSSGNNConv(h, h, "sum", "sum", "mean", mode, dict(MLP)).to(dev)
    else:
        layer = Conv.GNNAKConv(h, h, "sum", "mean", mode, dict(MLP), dict(MLP)).to(dev)
    if layout == "sparse":
        hb = synth.make_batch(64, "zinc", seed=25, keys=tuple(parse_precomputekey(layer)))
        dd = synth.to_datadict(hb, dev)
        X0, A0 = dd["X"], dd["A"]
        xraw = torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16)
        A = SparseTensor(A0.indices, (torch.randn(A0.nnz, h, device=dev) * 0.5).to(torch.bfloat16), list(A0.shape[:2]) + [h], True)
        mk = lambda x: SparseTensor(X0.indices, x, list(X0.shape[:2]) + [h], True)
        valid = None
    else:
        dn = synth.make_dense_batch(8, seed=10, hidden=h, nmax=37)
        dd = {}
        xraw = T(dn["X"], dev).to(torch.bfloat16)
        A = MaskedTensor(T(dn["A"], dev).to(torch.bfloat16), T(dn["Amask"], dev), 0.0, True)
        xm = T(dn["Xmask"], dev)
        mk = lambda x: MaskedTensor(x, xm, 0.0, True)
        valid = xm[..., None]
    w = torch.randn(xraw.shape, device=dev).to(torch.bfloat16)
    res = {}
    for fused in (True, False):
        _ops.USE_CONCAT_BLOCK = fused
        try:
            lay = copy.deepcopy(layer)
            x = xraw.clone().requires_grad_(True)
            X = mk(x)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = lay.forward_residual(A, X, dd) if fused else X.add(lay.forward(A, X, dd), True)
            o = out.values if valid is None else out.data * valid
            o.backward(w)
            gx = x.grad.float() if valid is None else x.grad.float() * valid
            res[fused] = (o.detach().float(), gx, {k: p.grad.float() for k, p in lay.named_parameters()})
        finally:
            _ops.USE_CONCAT_BLOCK = True
    for i in (0, 1):
        s = float(res[False][i].abs().max()) + 1e-6
        torch.testing.assert_close(res[True][i] / s, res[False][i] / s, rtol=0, atol=4e-2)
    for k, ref in res[False][2].items():
        if k.endswith(".lins.0.bias"):
            continue
        s = float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(res[True][2][k] / s, ref / s, rtol=0, atol=4e-2, msg=k)


@pytest.mark.parametrize("layout", ["sparse", "dense"])
def test_sunconv_forward_residual(dev, layout):
    """SUNConv.forward_residual (the layer input as residual row operand of the last fused MLP block) against
    X.add(conv.forward(A, X, datadict), True): bf16 activations (sum formed in the block) and f32 activations under autocast
    (the residual stream stays f32: the add is not fused), outputs, input gradient and parameter gradients."""
    import copy
    from pygho_amd import MaskedTensor, SparseTensor, synth, _ops
    from pygho_amd.honn import Conv
    from pygho_amd.honn.SpOperator import parse_precomputekey
    h = 128
    mode = "SS" if layout == "sparse" else "DD"
    torch.manual_seed(8)
    layer = Conv.SUNConv(h, h, "sum", "mean", mode, dict(MLP), dict(MLP)).to(dev)
    for xdt in (torch.bfloat16, torch.float32):
        if layout == "sparse":
            hb = synth.make_batch(64, "zinc", seed=27, keys=tuple(parse_precomputekey(layer)))
            dd = synth.to_datadict(hb, dev)
            X0, A0 = dd["X"], dd["A"]
            xraw = torch.randn(X0.nnz, h, device=dev).to(xdt)
            A = SparseTensor(A0.indices, (torch.randn(A0.nnz, h, device=dev) * 0.5).to(xdt), list(A0.shape[:2]) + [h], True)
            mk = lambda x: SparseTensor(X0.indices, x, list(X0.shape[:2]) + [h], True)
            valid = None
        else:
            dn = synth.make_dense_batch(8, seed=12, hidden=h, nmax=37)
            dd = {}
            xraw = T(dn["X"], dev).to(xdt)
            A = MaskedTensor(T(dn["A"], dev).to(xdt), T(dn["Amask"], dev), 0.0, True)
            xm = T(dn["Xmask"], dev)
            mk = lambda x: MaskedTensor(x, xm, 0.0, True)
            valid = xm[..., None]
        w = torch.randn(xraw.shape, device=dev).to(xdt)
        res = {}
        for fused in (True, False):
            lay = copy.deepcopy(layer)
            x = xraw.clone().requires_grad_(True)
            X = mk(x)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = lay.forward_residual(A, X, dd) if fused else X.add(lay.forward(A, X, dd), True)
            o = out.values if valid is None else out.data * valid
            assert o.dtype == xdt                                     # the residual stream keeps its dtype
            o.backward(w)
            gx = x.grad.float() if valid is None else x.grad.float() * valid
            res[fused] = (o.detach().float(), gx, {k: p.grad.float() for k, p in lay.named_parameters()})
        for i in (0, 1):
            s = float(res[False][i].abs().max()) + 1e-6
            torch.testing.assert_close(res[True][i] / s, res[False][i] / s, rtol=0, atol=4e-2)
        for k, ref in res[False][2].items():
            if k.endswith(".lins.0.bias"):
                continue
            s = float(ref.abs().max()) + 1e-6
            torch.testing.assert_close(res[True][2][k] / s, ref / s, rtol=0, atol=4e-2, msg=k)


def test_ppgn_forward_residual(dev):
    """PPGNConv.forward_residual (residual row added in the 2-FWL product kernel's epilogue, its gradient = the output gradient)
    against X.add(conv.forward(A, X, datadict), True): bf16, outputs, input gradient, parameter gradients."""
    import copy
    from pygho_amd import SparseTensor, synth
    from pygho_amd.honn import Conv
    from pygho_amd.honn.SpOperator import parse_precomputekey
    h = 128
    torch.manual_seed(9)
    layer = Conv.PPGNConv(h, h, "sum", "SS", dict(MLP)).to(dev)
    hb = synth.make_batch(48, "zinc", seed=29, keys=tuple(parse_precomputekey(layer)))
    dd = synth.to_datadict(hb, dev)
    X0 = dd["X"]
    xraw = torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16)
    w = torch.randn(X0.nnz, h, device=dev).to(torch.bfloat16)
    res = {}
    for fused in (True, False):
        lay = copy.deepcopy(layer)
        x = xraw.clone().requires_grad_(True)
        X = SparseTensor(X0.indices, x, list(X0.shape[:2]) + [h], True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = lay.forward_residual(dd["A"], X, dd) if fused else X.add(lay.forward(dd["A"], X, dd), True)
        if fused:
            assert type(out.values.grad_fn).__name__ in ("_MessageReduceBackward", "ViewBackward0", "ReshapeAliasBackward0")
        out.values.backward(w)
        res[fused] = (out.values.detach().float(), x.grad.float(), {k: p.grad.float() for k, p in lay.named_parameters()})
    for i in (0, 1):
        s = float(res[False][i].abs().max()) + 1e-6
        torch.testing.assert_close(res[True][i] / s, res[False][i] / s, rtol=0, atol=4e-2)
    for k, ref in res[False][2].items():
        if k.endswith(".lins.0.bias"):
            continue
        s = float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(res[True][2][k] / s, ref / s, rtol=0, atol=4e-2, msg=k)


def test_four_captured_steps_with_eager_work_between_replays(dev):
    """Four captured SpModel training steps (one per fixed mini-batch) sharing ONE model and ONE capturable AdamW, an eager kernel and
    an eager allocation between replays, NO device synchronisation: 0 non-finite losses of 200 replays and finite parameters.  The NaN
    rounds 2-3 reported for this regime comes from torch's stock nn.Linear under bf16 autocast (+ aten batch_norm) inside a captured
    step (plain-torch reduction: tools/repro_graph_nan2.py; pygho_amd/graphs.py); the shipped models run every Linear on the cast
    arena and every BatchNorm on this package's kernels -- the step contains no aten batch_norm launch, asserted here with the
    profiler on one eager step."""
    from pygho_amd import synth
    from pygho_amd.graphs import GraphedStep
    from pygho_amd.ngnn import SpModel
    torch.manual_seed(0)
    model = SpModel(1, 6, 128, act_dtype=torch.bfloat16).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
    dds = [synth.to_datadict(synth.make_batch(128, "zinc", seed=100 + k), dev) for k in range(4)]

    def make_step(dd):
        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
            loss.backward()
            opt.step()
            return loss.detach()
        return step

    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        make_step(dds[0])()
    names = {e.name for e in prof.events()}
    assert not any("batch_norm" in n for n in names), f"a torch BatchNorm runs inside the step: {sorted(n for n in names if 'batch_norm' in n)}"
    steps = [GraphedStep(make_step(dd), warmup=3) for dd in dds]
    bad = 0
    first = [float(gs.replay()) for gs in steps]
    for _ in range(49):
        for gs in steps:
            gs.replay()
            _ = torch.full((1,), 7.0, device=dev)          # an eager kernel + an eager allocation between two replays
        torch.cuda.current_stream().synchronize()          # (a stream wait to read the losses: not a device-wide synchronisation)
        bad += sum(int(not bool(torch.isfinite(gs.output))) for gs in steps)
    assert bad == 0, f"{bad} non-finite losses of 196"
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    last = [float(gs.output) for gs in steps]
    assert sum(last) < sum(first), f"the captured steps do not train: {first} -> {last}"


def test_prefetched_training_with_deferred_flags_from_two_streams(dev):
    """range-check flags are produced on two streams when training through BatchPrefetcher (the collation's on the side stream, the
    small-table gradients' on the training stream): each is read only by a fetch on ITS stream or by the full check, never by the
    other stream's fetch (which could see it before it was even zero-filled).  70 fresh batches (more than the 64 pending flags
    that force a full check) train without a spurious error, and a genuinely bad index is still reported."""
    from pygho_amd import _ops, synth
    from pygho_amd.collate import BatchPrefetcher, DeviceGraphStore
    from pygho_amd.ngnn import SpModel
    rng = np.random.default_rng(21)
    recs = [synth.make_graph(rng, "zinc", 3, ("X___X___1___A___0",)) for _ in range(64)]
    store = DeviceGraphStore(recs, dev)
    torch.manual_seed(0)
    model = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
    gen = torch.Generator().manual_seed(2)
    ids = [torch.randperm(64, generator=gen)[:8] for _ in range(70)]
    for dd in BatchPrefetcher(store, ids):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        opt.step()
    _ops.check_deferred_errors()
    assert bool(torch.isfinite(loss))
    side = torch.cuda.Stream()
    g = torch.ones(6, 8, device=dev)
    _ops.table_grad(g, torch.tensor([0, 1, 5, 1, 0, 2], device=dev), 3)            # flag raised on the current stream ...
    with torch.cuda.stream(side):
        assert _ops._fetch(torch.zeros(1, dtype=torch.int64, device=dev)) == [0]   # ... is not this stream's to read
    with pytest.raises(ValueError, match="out of range"):
        _ops._fetch(torch.zeros(1, dtype=torch.int64, device=dev))


@pytest.mark.parametrize("conv", ["NGNN", "SSWL", "SUN", "GNNAK", "DSSGNN", "I2GNN"])
def test_fresh_collated_batch_trains_without_host_reads(dev, conv):
    """a training step of every sparse model family on a batch it has never seen, collated from the device graph store,
    makes no device-to-host read (planner fetches, `.item()`, `.tolist()`): the batch arrives with its message / scatter plans, the
    groupings of its index rows are assembled from per-graph parts on demand, small-table gradients need no plan, and the store's
    range checks stand in for the hash asserts.  Round 6: I2GNN too -- the merged (i, j) pattern of pooling a 3-tuple representation's
    last coordinate away (reference SpTensor.py:368-380) and its grouping by root come with the batch (`DeviceGraphStore.pair_parts`)."""
    from pygho_amd import _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.honn.SpOperator import parse_precomputekey
    from pygho_amd.models import SpModel
    torch.manual_seed(0)
    model = SpModel(conv, num_layer=2, hiddim=64, act_dtype=torch.bfloat16).to(dev)
    keys = tuple(parse_precomputekey(model))
    rng = np.random.default_rng(1)
    store = DeviceGraphStore([synth.make_graph(rng, "i2" if conv == "I2GNN" else "zinc", 3, keys) for _ in range(24)], dev)
    assert (store.pair_parts is not None) == (conv == "I2GNN")

    def step(dd):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        return loss.detach()

    step(store.collate(list(range(8))))
    torch.cuda.synchronize()
    reads = []
    oi, ol = torch.Tensor.item, torch.Tensor.tolist
    torch.Tensor.item = lambda self: (reads.append("item") if self.is_cuda else None, oi(self))[1]
    torch.Tensor.tolist = lambda self: (reads.append("tolist") if self.is_cuda else None, ol(self))[1]
    f0 = _ops.FETCHES[0]
    try:
        loss = step(store.collate([9, 3, 17, 20, 5, 5, 11, 23]))
    finally:
        torch.Tensor.item, torch.Tensor.tolist = oi, ol
    assert _ops.FETCHES[0] == f0 and not reads, (conv, _ops.FETCHES[0] - f0, reads)
    assert bool(torch.isfinite(loss))


def test_collated_three_tuple_batch_brings_its_pooled_pattern(dev):
    """the merged (i, j) pattern a collated I2-shape batch installs (per-graph runs of the sorted 3-tuples + offsets) == the one
    `SparseTensor._reduce_to_sparse` plans by hash + sort + unique on the same indices (reference SpTensor.py:368-380): indices bit-exact,
    pooled values and their gradient identical for sum / max / mean; the pooled pattern's grouping by root likewise (the 2-D pooling
    that follows in OpPoolingSubg2D, SpOperator.py:496-522)"""
    from pygho_amd import SparseTensor, synth
    from pygho_amd.collate import DeviceGraphStore
    key = "X___X___2___A___0"
    rng = np.random.default_rng(2)
    store = DeviceGraphStore([synth.make_graph(rng, "i2", 3, (key,)) for _ in range(20)], dev)
    dd = store.collate([4, 11, 0, 19, 7, 7, 3])
    X = dd["X"]
    assert ("pool_sparse", (0, 1)) in X._cache()
    n = int(dd["num_nodes"])
    torch.manual_seed(0)
    vals = torch.randn(X.nnz, 16, device=dev)
    for op in ("sum", "max", "mean"):
        got_in = vals.clone().requires_grad_(True)
        ref_in = vals.clone().requires_grad_(True)
        Xa = SparseTensor(X.indices, got_in, [n, n, n, 16], True)                      # the collated index object: installed plans
        Xb = SparseTensor(X.indices.clone(), ref_in, [n, n, n, 16], True)              # a copy: planned by hash + sort + unique
        pa, pb = getattr(Xa, op)([2], return_sparse=True), getattr(Xb, op)([2], return_sparse=True)
        assert torch.equal(pa.indices, pb.indices) and torch.equal(pa.values, pb.values), op
        da, db = getattr(pa, op)([1]), getattr(pb, op)([1])                              # (n, 16): pooled again over the second coordinate
        assert torch.equal(da, db), op
        w = torch.randn_like(da)
        (da * w).sum().backward()
        (db * w).sum().backward()
        assert torch.equal(got_in.grad, ref_in.grad), op


def _allocated_growth(make_and_use, warm=6, iters=24):
    """bytes of device memory that stay allocated per iteration of `make_and_use()` with the cyclic collector OFF (so both kinds of
    leak show: cycles Python could collect later, and cycles through a view's C++-side `_base` pointer that it never can)"""
    import gc
    gc.collect()
    gc.disable()
    try:
        for _ in range(warm):
            make_and_use()
        torch.cuda.synchronize()
        a = torch.cuda.memory_allocated()
        for _ in range(iters):
            make_and_use()
        torch.cuda.synchronize()
        return (torch.cuda.memory_allocated() - a) / iters
    finally:
        gc.enable()


def test_fresh_batches_do_not_leak_device_memory(dev):
    """Every batch's index arrays carry caches (int32 copies, rows, plans) as attributes.  A VIEW of a tensor cached on that tensor
    is a reference cycle through the view's `_base` that no collector sees: until round 4 every fresh batch stayed allocated for the
    life of the process (4.6 MB per 128-graph batch, 330 MB per 8192-graph batch).  Three loops -- batches collated from the device
    store, batches planned the ordinary way, fresh masks through the dense contraction -- must leave allocated memory flat."""
    from pygho_amd import MaskedTensor, synth
    from pygho_amd.backend.Mamamm import mamamm
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.ngnn import SpModel
    key = "X___X___1___A___0"
    rng = np.random.default_rng(4)
    recs = [synth.make_graph(rng, "zinc", 3, (key,)) for _ in range(64)]
    store = DeviceGraphStore(recs, dev)
    torch.manual_seed(0)
    model = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev)
    gen = torch.Generator().manual_seed(0)

    def step(dd):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float()).backward()

    per_batch = 24 * 3 * 50 * 20 * 8          # ~ the int64 triples of one 24-graph batch: a leaked batch is several times this
    g1 = _allocated_growth(lambda: step(store.collate(torch.randperm(64, generator=gen)[:24])))
    assert g1 < 0.05 * per_batch, f"collated batches leak {g1:.0f} B per batch"
    g2 = _allocated_growth(lambda: step(synth.to_datadict(synth.collate([recs[i] for i in torch.randperm(64, generator=gen)[:24].tolist()]), dev, "zinc")))
    assert g2 < 0.05 * per_batch, f"ordinarily planned batches leak {g2:.0f} B per batch"

    def dense():
        b, n, d = 16, 20, 64
        mask = torch.rand(b, n, n, device=dev) < 0.4
        amask = torch.rand(b, n, n, device=dev) < 0.1
        X = MaskedTensor(torch.randn(b, n, n, d, device=dev, requires_grad=True), mask, 0.0, False)
        A = MaskedTensor(torch.randn(b, n, n, d, device=dev), amask, 0.0, False)
        mamamm(X, 2, A, 1, mask).data.sum().backward()
    g3 = _allocated_growth(dense)
    assert g3 < 16 * 20 * 20 * 2, f"fresh masks leak {g3:.0f} B per batch"


def _tensors_in(obj, depth=0, seen=None):
    """tensors reachable from a cache value through tuples / lists / dicts / plain objects' attributes (bounded depth)"""
    seen = set() if seen is None else seen
    if id(obj) in seen or depth > 5:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        yield obj
        return
    if isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_in(v, depth + 1, seen)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            yield from _tensors_in(v, depth + 1, seen)
    elif hasattr(obj, "__dict__") or hasattr(obj, "__slots__"):
        names = list(getattr(obj, "__dict__", {})) + [s for s in getattr(type(obj), "__slots__", ()) if hasattr(obj, s)]
        for nme in names:
            yield from _tensors_in(getattr(obj, nme, None), depth + 1, seen)


def test_no_cache_holds_a_view_of_its_owner(dev):
    """structural form of the leak test: after a training step on a collated batch and a dense contraction, no tensor's `_pygho_*`
    caches contain a tensor whose `_base` is that tensor (the uncollectable cycle), nor the tensor itself (a collectable one that
    still pins device memory until the cyclic collector runs)."""
    import gc
    from pygho_amd import MaskedTensor, synth
    from pygho_amd.backend.Mamamm import mamamm
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.models import SpModel
    from pygho_amd.honn.SpOperator import parse_precomputekey
    rng = np.random.default_rng(6)
    keep = []
    for conv in ("NGNN", "SUN", "SSWL", "GNNAK", "DSSGNN", "I2GNN"):
        torch.manual_seed(0)
        model = SpModel(conv, num_layer=2, hiddim=64, act_dtype=torch.bfloat16).to(dev)
        keys = tuple(parse_precomputekey(model))
        kind = "i2" if conv == "I2GNN" else "zinc"
        recs = [synth.make_graph(rng, kind, 3, keys) for _ in range(12)]
        for dd in (DeviceGraphStore(recs, dev).collate([3, 1, 7, 7, 10]), synth.to_datadict(synth.collate(recs[:5]), dev, kind)):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(dd)
            torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float()).backward()
            keep.append(dd)
    b, n, d = 8, 12, 64
    mask, amask = torch.rand(b, n, n, device=dev) < 0.4, torch.rand(b, n, n, device=dev) < 0.2
    X = MaskedTensor(torch.randn(b, n, n, d, device=dev, requires_grad=True), mask, 0.0, False)
    A = MaskedTensor(torch.randn(b, n, n, d, device=dev), amask, 0.0, False)
    mamamm(X, 2, A, 1, mask).data.sum().backward()
    keep += [mask, amask]
    from pygho_amd.models import MaModel
    hb = synth.make_batch(6, "zinc", seed=3)
    for conv in ("SUN", "PPGN"):                               # the dense layout's models (padded MaskedTensors)
        torch.manual_seed(0)
        ma = MaModel(conv, num_layer=2, hiddim=64, act_dtype=torch.bfloat16).to(dev)
        ddd = _dense_inputs_of(hb, dev)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = ma(ddd)
        pred.float().sum().backward()
        keep.append(ddd)
    bad = []
    for owner in [o for o in gc.get_objects() if torch.is_tensor(o)]:
        for name, val in list(getattr(owner, "__dict__", {}).items()):
            if not name.startswith("_pygho"):
                continue
            for t in _tensors_in(val):
                if t is owner:
                    bad.append((name, tuple(owner.shape), "holds itself"))
                elif t._base is owner:
                    bad.append((name, tuple(owner.shape), "holds a view of itself"))
    assert not bad, bad[:10]


def test_gradients_do_not_depend_on_how_often_a_batch_was_seen(dev):
    """SURVEY 5 makes run-to-run bit-equality the sanitiser substitute; VERDICT r4 weak 1a: the embedding gradients changed bits between
    step 3 and step 4 of a resident-batch run (the small-table gradient switched routes 'after 3 uses') and a resumed run diverged from a
    continuous one.  Round 5 removed every use counter from the dispatchers.  (i) the same step on the same resident batch, 8 times from
    the same parameters: bit-identical loss and parameter gradients at every repetition; (ii) a training run of 6 AdamW steps on warm
    caches against a RESUMED one -- a fresh model loaded with the state before step 6, fresh tensors for the batch (no cached plan,
    no use history): step 6's loss and gradients agree bit for bit."""
    import copy
    from pygho_amd import synth
    from pygho_amd.ngnn import SpModel
    hb = synth.make_batch(512, "zinc", seed=41)                   # 111 k tuples: above the old plan-after-3-uses threshold (65 536 rows)
    dd = synth.to_datadict(hb, dev)
    assert dd["X"].nnz > 65536
    torch.manual_seed(3)
    model = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev)

    def grads_of(m, batch):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = m(batch)
        loss = torch.nn.functional.l1_loss(batch["y"].unsqueeze(-1), pred.float())
        loss.backward()
        return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    state0 = copy.deepcopy(model.state_dict())
    first = None
    for rep in range(8):
        model.load_state_dict(state0)                             # (the BatchNorm running statistics move with every forward)
        loss, grads = grads_of(model, dd)
        if first is None:
            first = (loss, grads)
        else:
            assert torch.equal(loss, first[0]), rep
            bad = [k for k in grads if not torch.equal(grads[k], first[1][k])]
            assert not bad, f"repetition {rep}: {bad}"
    # (ii) warm run, then a resumed one
    model.load_state_dict(state0)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    for _ in range(5):
        grads_of(model, dd)
        opt.step()
    before = copy.deepcopy(model.state_dict())
    warm = grads_of(model, dd)
    torch.manual_seed(99)
    resumed = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev)
    resumed.load_state_dict(before)
    cold = grads_of(resumed, synth.to_datadict(hb, dev))          # new tensors: nothing cached, nothing counted
    assert torch.equal(warm[0], cold[0])
    bad = [k for k in warm[1] if not torch.equal(warm[1][k], cold[1][k])]
    assert not bad, f"a resumed run differs from the continuous one in {bad}"

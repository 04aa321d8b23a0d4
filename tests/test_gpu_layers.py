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
TOL = dict(rtol=2e-5, atol=2e-5)
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
            np.testing.assert_allclose(N(p.grad), g[f"{name}_pg_{k}"], rtol=2e-4, atol=2e-4, err_msg=k)


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


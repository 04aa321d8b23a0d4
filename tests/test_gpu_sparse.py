"""
GPU parity tests of the sparse path: the HIP kernels (through the C ABI / the pygho-compatible Python API)
against the CPU oracle on seeded inputs and against the golden fixtures generated from the reference.
Bar: bit-exact for integer / index work; 1e-5 relative for f32 aggregation (f32 sums are in fact
bit-identical to the sequential oracle); bf16 within one bf16 rounding of the f32 oracle.
"""
import numpy as np
import pytest
import torch

from conftest import canon_triples, load_golden
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from pygho_amd import _native
    _native.lib()                      # must load: no fallback
    return torch.device("cuda:0")


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def N(t):
    return t.detach().float().cpu().numpy() if t.dtype in (torch.bfloat16, torch.float16) else t.detach().cpu().numpy()


# --------------------------------------------------------------------------
def test_hash_and_planner_bit_exact(dev):
    from pygho_amd.backend import SpTensor, Spspmm
    g = load_golden("hash.npz")
    for sd in (2, 3, 4, 5):
        h = SpTensor.indicehash(T(g[f"ind{sd}"], dev))
        assert np.array_equal(N(h), g[f"hash{sd}"])
        assert np.array_equal(N(SpTensor.decodehash(h, sd)), g[f"ind{sd}"])
    with pytest.raises(AssertionError):
        SpTensor.indicehash(T(np.array([[0], [0], [1 << 21]], dtype=np.int64), dev))
    with pytest.raises(AssertionError):
        SpTensor.indicehash(T(np.array([[0], [-1]], dtype=np.int64), dev))
    th = SpTensor.indicehash_tight(T(g["tight_ind"], dev), torch.from_numpy(g["tight_shape"]))
    assert np.array_equal(N(th), g["tight_hash"])
    assert np.array_equal(N(SpTensor.decodehash_tight(th, torch.from_numpy(g["tight_shape"]))), g["tight_ind"])
    for red in ("sum", "mean", "max", "min"):
        ci, cv = SpTensor.coalesce(T(g["co_ind"], dev), T(g["co_val"], dev), red)
        assert np.array_equal(N(ci), g["co_ind_out"])
        np.testing.assert_allclose(N(cv), g[f"co_val_{red}"], **TOL)

    p = load_golden("planner.npz")
    for name in p["names"]:
        d1, d2 = (int(v) for v in p[f"{name}_dims"])
        tarind, bcd = Spspmm.spspmm_ind(T(p[f"{name}_ind1"], dev), d1, T(p[f"{name}_ind2"], dev), d2)
        assert np.array_equal(N(tarind), p[f"{name}_tarind"]), name
        assert np.array_equal(canon_triples(N(bcd)), p[f"{name}_bcd"]), name
        assert (np.diff(N(bcd)[0]) >= 0).all()
        if f"{name}_tar" in p.files:
            tar = T(p[f"{name}_tar"], dev)
            assert np.array_equal(N(Spspmm.spsphadamard_ind(tar, tarind)), p[f"{name}_b2a"]), name
            acd = N(Spspmm.filterind(tar, tarind, bcd))
            assert np.array_equal(canon_triples(acd), p[f"{name}_acd"]), name
            assert (np.diff(acd[0]) >= 0).all()
    s = load_golden("scatter.npz")
    assert np.array_equal(N(Spspmm.ptr2batch(T(s["ptr"], dev), 16)), s["ptr_expected"])
    assert np.array_equal(N(Spspmm.deg2batch(T(s["deg"], dev), 6)), s["deg_batch"])


def test_planner_primitives_bit_exact(dev):
    """the integer glue kernels of the planner against numpy on seeded inputs, incl. empty inputs."""
    from pygho_amd import _ops
    rng = np.random.default_rng(11)
    for n in (0, 1, 63, 1000, 70001):
        cnt = rng.integers(0, 5, n).astype(np.int64)
        assert np.array_equal(N(_ops.exclusive_scan(T(cnt, dev))), np.concatenate(([0], np.cumsum(cnt))))
        vals = rng.integers(-1, 3, max(n, 1)).astype(np.int64)
        assert np.array_equal(N(_ops.nonneg_positions(T(vals[:n], dev))), np.nonzero(vals[:n] >= 0)[0])
        via = rng.integers(0, max(n, 1), 2 * n).astype(np.int64)
        assert np.array_equal(N(_ops.nonneg_positions(T(vals, dev), via=T(via, dev))), np.nonzero(vals[via] >= 0)[0])
        src = rng.integers(0, 1 << 40, (3, max(n, 1))).astype(np.int64)
        for idt in (np.int64, np.int32):
            idx = rng.integers(0, max(n, 1), n).astype(idt)
            assert np.array_equal(N(_ops.gather_cols(T(src, dev), T(idx, dev))), src[:, idx])
            assert np.array_equal(N(_ops.gather_cols(T(src[1], dev), T(idx, dev))), src[1][idx])
        tab = rng.integers(0, 1 << 30, max(n, 1)).astype(np.int32)
        idx = rng.integers(0, max(n, 1), n).astype(np.int64)
        assert np.array_equal(N(_ops.widen_gather(T(tab, dev), T(idx, dev))), tab[idx].astype(np.int64))
        perm = rng.permutation(n).astype(np.int32)
        slot = rng.integers(0, 9, n).astype(np.int32)
        c, d = rng.integers(0, 99, n).astype(np.int64), rng.integers(0, 99, n).astype(np.int64)
        assert np.array_equal(N(_ops.plan_triples(T(slot, dev), T(c, dev), T(d, dev), T(perm, dev))),
                              np.stack((slot[perm].astype(np.int64), c[perm], d[perm])))
    # product hash == indicehash of the concatenated remaining coordinates (Spspmm.py:132-135)
    for sd1, dim1, sd2, dim2 in ((2, 1, 2, 0), (2, 0, 2, 1), (3, 1, 2, 0), (3, 2, 3, 1), (2, 1, 1, 0)):
        ind1 = rng.integers(0, 500, (sd1, 300)).astype(np.int64)
        ind2 = rng.integers(0, 500, (sd2, 200)).astype(np.int64)
        c, d = rng.integers(0, 300, 4000).astype(np.int64), rng.integers(0, 200, 4000).astype(np.int64)
        rest = np.concatenate((np.delete(ind1, dim1, 0)[:, c], np.delete(ind2, dim2, 0)[:, d]))
        got = _ops.product_hash(T(ind1, dev), dim1, T(ind2, dev), dim2, T(c, dev), T(d, dev))
        assert np.array_equal(N(got), O.indicehash(rest))
    with pytest.raises(AssertionError):
        _ops.product_hash(T(np.array([[0], [1 << 33]], dtype=np.int64), dev), 0, T(np.array([[0], [1]], dtype=np.int64), dev), 0,
                          T(np.array([0], dtype=np.int64), dev), T(np.array([0], dtype=np.int64), dev))


def test_scatter_reduce_golden(dev):
    from pygho_amd.backend.utils import torch_scatter_reduce
    g = load_golden("scatter.npz")
    for tag in ("s0", "s1", "s2"):
        for ag in ("sum", "mean", "max", "min"):
            got = torch_scatter_reduce(0, T(g[f"{tag}_src"], dev), T(g[f"{tag}_ind"], dev), int(g[f"{tag}_size"]), ag)
            exp = g[f"{tag}_{ag}"]
            assert tuple(got.shape) == exp.shape
            if np.issubdtype(exp.dtype, np.integer):
                assert np.array_equal(N(got), exp), (tag, ag)
            else:
                np.testing.assert_allclose(N(got), exp, **TOL)


def test_prod_aggregation_golden(dev):
    """aggr = "prod" of torch_scatter_reduce and coalesce (reference utils.py:44-56, SpTensor.py:167-197; VERDICT r5: it raised):
    forward and autograd gradient against outputs of the reference (prod.npz) -- segments with one exact zero (that element
    receives the product of the others), with two (all zero), empty segments, an unsorted index, int64 values; 16-bit rows to
    their rounding."""
    from pygho_amd.backend import SpTensor
    from pygho_amd.backend.utils import torch_scatter_reduce
    g = load_golden("prod.npz")
    for tag in ("p0", "p1", "p2"):
        got = torch_scatter_reduce(0, T(g[f"{tag}_src"], dev), T(g[f"{tag}_ind"], dev), int(g[f"{tag}_size"]), "prod")
        exp = g[f"{tag}_prod"]
        assert tuple(got.shape) == exp.shape
        if np.issubdtype(exp.dtype, np.integer):
            assert np.array_equal(N(got), exp), tag
        else:
            np.testing.assert_allclose(N(got), exp, **TOL)
    src = T(g["p1_src"], dev).requires_grad_(True)
    out = torch_scatter_reduce(0, src, T(g["p1_ind"], dev), int(g["p1_size"]), "prod")
    (out * T(g["p1_w"], dev)).sum().backward()
    np.testing.assert_allclose(N(src.grad), g["p1_grad"], **TOL)
    assert int((g["p1_grad"] == 0).sum()) > 8               # the planted zeros do exercise the zero rules
    for dt, tol in ((torch.bfloat16, 2.0 ** -6), (torch.float16, 2.0 ** -9)):
        got = torch_scatter_reduce(0, T(g["p1_src"], dev).to(dt), T(g["p1_ind"], dev), int(g["p1_size"]), "prod")
        np.testing.assert_allclose(N(got.float()), g["p1_prod"], rtol=tol, atol=tol)
    ci, cv = SpTensor.coalesce(T(g["co_ind"], dev), T(g["co_val"], dev), "prod")
    assert np.array_equal(N(ci), g["co_ind_out"])
    np.testing.assert_allclose(N(cv), g["co_val_prod"], **TOL)


def _sp(dev, ind, val, n, sd=2):
    from pygho_amd import SparseTensor
    shape = [n] * sd + ([] if val is None else list(val.shape[1:]))
    return SparseTensor(T(ind, dev), None if val is None else val, shape, is_coalesced=True)


def test_spspmm_golden_forward_backward(dev):
    from pygho_amd.backend.Spspmm import spspmm, spspmpnn
    g = load_golden("sparse_ops.npz")
    n_nodes = int(g["N"])
    tid, ei = g["tupleid"], g["edge_index"]
    acd = T(g["acd_X___X___1___A___0"], dev)
    for ag in ("sum", "mean", "max", "min"):
        Xv = T(g["Xv"], dev).requires_grad_(True)
        Av = T(g["Av"], dev).requires_grad_(True)
        out = spspmm(_sp(dev, tid, Xv, n_nodes), 1, _sp(dev, ei, Av, n_nodes), 0, ag, acd=acd, tar_ind=T(tid, dev))
        np.testing.assert_allclose(N(out.values), g[f"spspmm_{ag}"], **TOL)
        (out.values * T(g[f"spspmm_{ag}_w"], dev)).sum().backward()
        np.testing.assert_allclose(N(Xv.grad), g[f"spspmm_{ag}_gX"], **TOL)
        np.testing.assert_allclose(N(Av.grad), g[f"spspmm_{ag}_gA"], **TOL)
        assert out.shape == (n_nodes, n_nodes, g["Xv"].shape[1])
    Xv, Av = T(g["Xv"], dev), T(g["Av"], dev)
    tidt = T(tid, dev)
    o = spspmm(_sp(dev, tid, Xv, n_nodes), 1, _sp(dev, ei, None, n_nodes), 0, "sum", acd=acd, tar_ind=tidt)
    np.testing.assert_allclose(N(o.values), g["spspmm_noA_sum"], **TOL)
    o = spspmm(_sp(dev, tid, None, n_nodes), 1, _sp(dev, ei, Av, n_nodes), 0, "max", acd=acd, tar_ind=tidt)
    np.testing.assert_allclose(N(o.values), g["spspmm_noX_max"], **TOL)
    o = spspmm(_sp(dev, ei, Av, n_nodes), 1, _sp(dev, tid, Xv, n_nodes), 0, "sum", acd=T(g["acd_X___A___1___X___0"], dev), tar_ind=tidt)
    np.testing.assert_allclose(N(o.values), g["spspmm_cross_sum"], **TOL)
    o = spspmm(_sp(dev, tid, Xv, n_nodes), 1, _sp(dev, tid, Xv * 0.5, n_nodes), 0, "sum", acd=T(g["acd_X___X___1___X___0"], dev), tar_ind=tidt)
    np.testing.assert_allclose(N(o.values), g["spspmm_fwl_sum"], **TOL)
    # slow paths (planner on the device), Spspmm.py:322-331
    with pytest.warns(UserWarning):
        full = spspmm(_sp(dev, tid, Xv, n_nodes), 1, _sp(dev, ei, Av, n_nodes), 0, "sum")
    assert np.array_equal(N(full.indices), g["spspmm_slow_ind"])
    np.testing.assert_allclose(N(full.values), g["spspmm_slow_val"], **TOL)
    with pytest.warns(UserWarning):
        filt = spspmm(_sp(dev, tid, Xv, n_nodes), 1, _sp(dev, ei, Av, n_nodes), 0, "sum", tar_ind=tidt)
    np.testing.assert_allclose(N(filt.values), g["spspmm_slowtar_val"], **TOL)
    # message-function variant
    Xv = T(g["Xv"], dev).requires_grad_(True)
    Av = T(g["Av"], dev).requires_grad_(True)
    X = _sp(dev, tid, Xv, n_nodes)
    o = spspmpnn(X, 1, _sp(dev, ei, Av, n_nodes), 0, X, acd, lambda a, b, c, t: a * b + c, "sum")
    np.testing.assert_allclose(N(o.values), g["spspmpnn_sum"], **TOL)
    (o.values * T(g["spspmpnn_w"], dev)).sum().backward()
    np.testing.assert_allclose(N(Xv.grad), g["spspmpnn_gX"], **TOL)
    np.testing.assert_allclose(N(Av.grad), g["spspmpnn_gA"], **TOL)


def test_spmm_hadamard_golden(dev):
    from pygho_amd import SparseTensor
    from pygho_amd.backend.Spmm import spmm
    from pygho_amd.backend.Spspmm import spsphadamard
    g = load_golden("sparse_ops.npz")
    n_nodes, ei = int(g["N"]), g["edge_index"]
    for dim1 in (0, 1):
        for ag in ("sum", "mean", "max"):
            Av = T(g["Av"], dev).requires_grad_(True)
            xn = T(g["xn"], dev).requires_grad_(True)
            out = spmm(_sp(dev, ei, Av, n_nodes), dim1, xn, ag)
            np.testing.assert_allclose(N(out), g[f"spmm_{dim1}_{ag}"], **TOL)
            (out * T(g[f"spmm_{dim1}_{ag}_w"], dev)).sum().backward()
            np.testing.assert_allclose(N(Av.grad), g[f"spmm_{dim1}_{ag}_gA"], **TOL)
            np.testing.assert_allclose(N(xn.grad), g[f"spmm_{dim1}_{ag}_gx"], **TOL)
    xn = T(g["xn"], dev)
    np.testing.assert_allclose(N(spmm(_sp(dev, ei, None, n_nodes), 1, xn, "sum")), g["spmm_noval"], **TOL)
    Asc = SparseTensor(T(ei, dev), T(g["Av"][:, :1].copy(), dev), [n_nodes, n_nodes, 1], True)
    np.testing.assert_allclose(N(spmm(Asc, 1, xn, "sum")), g["spmm_scalar"], **TOL)
    had = spsphadamard(_sp(dev, g["tupleid"], T(g["Xv"], dev), n_nodes), _sp(dev, g["had_P"], T(g["had_Pv"], dev), n_nodes))
    assert np.array_equal(N(had.indices), g["had_ind"])
    np.testing.assert_allclose(N(had.values), g["had_val"], **TOL)


def test_sparse_tensor_methods_golden(dev):
    from pygho_amd import SparseTensor
    from pygho_amd.backend.Spspmm import spspmm
    g = load_golden("sparse_ops.npz")
    n_nodes, tid = int(g["N"]), g["tupleid"]
    for red in ("sum", "mean", "max"):
        for dims in (0, 1):
            Xv = T(g["Xv"], dev).requires_grad_(True)
            out = getattr(_sp(dev, tid, Xv, n_nodes), red)(dims)
            np.testing.assert_allclose(N(out), g[f"pool_{red}_{dims}"], **TOL)
            (out * T(g[f"pool_{red}_{dims}_w"], dev)).sum().backward()
            np.testing.assert_allclose(N(Xv.grad), g[f"pool_{red}_{dims}_g"], **TOL)
    Xv, xn = T(g["Xv"], dev), T(g["xn"], dev)
    X = _sp(dev, tid, Xv, n_nodes)
    np.testing.assert_allclose(N(X.diag([0, 1])), g["diag"], **TOL)
    xg = T(g["xn"], dev).requires_grad_(True)
    u0 = X.unpooling_fromdense1dim(0, xg).values
    assert np.array_equal(N(u0), g["unpool0"])
    (u0 * T(g["unpool0_w"], dev)).sum().backward()
    np.testing.assert_allclose(N(xg.grad), g["unpool0_g"], **TOL)
    assert np.array_equal(N(X.unpooling_fromdense1dim(1, xn).values), g["unpool1"])
    np.testing.assert_allclose(N(X.add(_sp(dev, tid, Xv * 2, n_nodes), True).values), g["add_same"], **TOL)
    addp = X.add(_sp(dev, g["had_P"], T(g["had_Pv"], dev), n_nodes), False)
    assert np.array_equal(N(addp.indices), g["add_diff_ind"])
    np.testing.assert_allclose(N(addp.values), g["add_diff_val"], **TOL)
    assert np.array_equal(N(X.catvalue([_sp(dev, tid, Xv * 2, n_nodes), _sp(dev, tid, Xv * 3, n_nodes)], True).values), g["cat"])
    np.testing.assert_allclose(N(X.diagonalapply(lambda v, f: f.unsqueeze(-1).to(v.dtype) * v).values), g["diagflag"], **TOL)
    cs = SparseTensor(T(g["ctor_ind_in"], dev), T(g["ctor_val_in"], dev), [n_nodes, n_nodes, Xv.shape[1]], False, "min")
    assert np.array_equal(N(cs.indices), g["ctor_ind"])
    np.testing.assert_allclose(N(cs.values), g["ctor_val"], **TOL)
    # 3-tuples
    n3, tid3 = int(g["N3"]), g["tupleid3"]
    Xv3, Av3 = T(g["Xv3"], dev), T(g["Av3"], dev)
    X3 = _sp(dev, tid3, Xv3, n3, sd=3)
    A3 = _sp(dev, g["edge_index3"], Av3, n3)
    for ag in ("sum", "max"):
        o = spspmm(X3, 2, A3, 0, ag, acd=T(g["acd3"], dev), tar_ind=T(tid3, dev))
        np.testing.assert_allclose(N(o.values), g[f"spspmm3_{ag}"], **TOL)
    for red in ("sum", "mean", "max"):
        p = getattr(X3, red)([2], return_sparse=True)
        assert np.array_equal(N(p.indices), g[f"pool3_{red}_ind"])
        np.testing.assert_allclose(N(p.values), g[f"pool3_{red}_val"], **TOL)
    p = X3.sum([2], return_sparse=True)
    np.testing.assert_allclose(N(p.unpooling([2], X3).values), g["unpool3"], **TOL)
    np.testing.assert_allclose(N(X3.sum([1, 2])), g["pool3_dense_12"], **TOL)
    np.testing.assert_allclose(N(X3.sum([2])), g["pool3_dense_2"], **TOL)


# --------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,d", [(torch.float32, 128), (torch.bfloat16, 128), (torch.float32, 20), (torch.float16, 64),
                                     (torch.bfloat16, 256), (torch.float32, 1), (torch.float64, 8)])
@pytest.mark.parametrize("aggr", ["sum", "mean", "max", "min"])
def test_spspmm_vs_oracle_random(dev, dtype, d, aggr):
    """ZINC-shape batch, seeded: HIP vs numpy oracle (bit-exact f32 sum; one bf16 ulp otherwise)."""
    from pygho_amd import synth
    from pygho_amd._ops import message_reduce
    hb = synth.make_batch(24, "zinc", seed=3)
    key = "X___X___1___A___0"
    acd = hb.acd[key]
    rng = np.random.default_rng(5)
    Xv = rng.standard_normal((hb.num_tuples, d)).astype(np.float32)
    Av = rng.standard_normal((hb.num_edges, d)).astype(np.float32)
    xt, at = T(Xv, dev, dtype), T(Av, dev, dtype)
    ref_in_x, ref_in_a = N(xt).astype(np.float64 if dtype == torch.float64 else np.float32), N(at).astype(np.float64 if dtype == torch.float64 else np.float32)
    got = message_reduce(xt, at, T(acd, dev), hb.num_tuples, hb.num_tuples, hb.num_edges, aggr)
    exp = O.spspmm_values(ref_in_x, ref_in_a, acd, hb.num_tuples, aggr)
    if dtype == torch.float32:
        if aggr in ("sum", "max", "min"):
            assert np.array_equal(N(got), exp), "f32 must be bit-identical to the sequential oracle"
        else:
            np.testing.assert_allclose(N(got), exp, rtol=1e-6, atol=1e-6)
    elif dtype == torch.float64:
        np.testing.assert_allclose(N(got), exp, rtol=1e-12, atol=1e-12)
    else:
        eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        np.testing.assert_allclose(N(got), exp, rtol=eps, atol=eps * 1e-2 + 1e-6)


def test_spspmm_edge_cases(dev):
    from pygho_amd._ops import message_reduce
    # empty message list, empty segments at both ends, one huge segment, unsorted acd
    d = 16
    x = torch.randn(5, d, device=dev)
    a = torch.randn(3, d, device=dev)
    empty = torch.zeros((3, 0), dtype=torch.int64, device=dev)
    out = message_reduce(x, a, empty, 7, 5, 3, "max")
    assert out.shape == (7, d) and float(out.abs().sum()) == 0.0
    acd = torch.tensor([[2, 2, 2, 2, 2, 4], [0, 1, 2, 3, 4, 0], [0, 1, 2, 0, 1, 2]], dtype=torch.int64, device=dev)
    for ag in ("sum", "mean", "max", "min"):
        got = message_reduce(x, a, acd, 7, 5, 3, ag)
        exp = O.spspmm_values(N(x), N(a), N(acd), 7, ag)
        np.testing.assert_allclose(N(got), exp, **TOL)
    perm = torch.tensor([5, 3, 0, 4, 1, 2], device=dev)
    got = message_reduce(x, a, acd[:, perm].contiguous(), 7, 5, 3, "sum")
    np.testing.assert_allclose(N(got), O.spspmm_values(N(x), N(a), N(acd), 7, "sum"), **TOL)
    with pytest.raises(ValueError):
        message_reduce(x, a, torch.tensor([[9], [0], [0]], dtype=torch.int64, device=dev), 7, 5, 3, "sum")
    with pytest.raises(RuntimeError):
        message_reduce(x.cpu(), a.cpu(), acd.cpu(), 7, 5, 3, "sum")            # no CPU fallback


def test_spspmm_grads_vs_oracle_bf16(dev):
    from pygho_amd import synth
    from pygho_amd._ops import message_reduce
    hb = synth.make_batch(8, "zinc", seed=9)
    acd = hb.acd["X___X___1___A___0"]
    rng = np.random.default_rng(2)
    d = 128
    Xv = T(rng.standard_normal((hb.num_tuples, d)).astype(np.float32), dev, torch.bfloat16).requires_grad_(True)
    Av = T(rng.standard_normal((hb.num_edges, d)).astype(np.float32), dev, torch.bfloat16).requires_grad_(True)
    w = T(rng.standard_normal((hb.num_tuples, d)).astype(np.float32), dev, torch.bfloat16)
    for ag in ("sum", "mean"):
        Xv.grad = Av.grad = None
        out = message_reduce(Xv, Av, T(acd, dev), hb.num_tuples, hb.num_tuples, hb.num_edges, ag)
        (out.float() * w.float()).sum().backward()
        gX, gA = O.spspmm_values_grad(N(Xv), N(Av), acd, hb.num_tuples, ag, N(w))
        # one bf16 rounding of an f32-accumulated sum: 2^-9 relative (2^-8 allowed), plus the accumulation-order noise of the f64
        # oracle against f32 (1e-6 of the terms' magnitude; sums of up to ~10 O(1) products)
        np.testing.assert_allclose(N(Xv.grad), gX, rtol=2 ** -8, atol=2e-5)
        np.testing.assert_allclose(N(Av.grad), gA, rtol=2 ** -8, atol=2e-5)


def test_run_to_run_determinism(dev):
    """segment kernels use no atomics: forward and both gradients are bit-reproducible."""
    from pygho_amd import synth
    from pygho_amd._ops import message_reduce
    hb = synth.make_batch(16, "zinc", seed=1)
    acd = T(hb.acd["X___X___1___A___0"], dev)
    outs = []
    for _ in range(3):
        torch.manual_seed(0)
        Xv = torch.randn(hb.num_tuples, 64, device=dev, requires_grad=True)
        Av = torch.randn(hb.num_edges, 64, device=dev, requires_grad=True)
        o = message_reduce(Xv, Av, acd.clone(), hb.num_tuples, hb.num_tuples, hb.num_edges, "sum")
        o.square().sum().backward()
        outs.append((o.detach().clone(), Xv.grad.clone(), Av.grad.clone()))
    for k in range(3):
        assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[0][k], outs[2][k])


def test_i2_shape_and_large_linearity(dev):
    """3-tuple (I2-shape) batch vs oracle, then a size-independent property at a large batch:
    spspmm is linear in each operand and a per-channel checksum matches a float64 host reduction."""
    from pygho_amd import synth
    from pygho_amd._ops import message_reduce
    hb = synth.make_batch(6, "i2", seed=4)
    acd = hb.acd["X___X___2___A___0"]
    rng = np.random.default_rng(0)
    Xv = rng.standard_normal((hb.num_tuples, 256)).astype(np.float32)
    Av = rng.standard_normal((hb.num_edges, 256)).astype(np.float32)
    got = message_reduce(T(Xv, dev), T(Av, dev), T(acd, dev), hb.num_tuples, hb.num_tuples, hb.num_edges, "sum")
    assert np.array_equal(N(got), O.spspmm_values(Xv, Av, acd, hb.num_tuples, "sum"))
    big = synth.replicate(synth.make_batch(256, "zinc", seed=7), 8)        # 2048 graphs
    acd_b = T(big.acd["X___X___1___A___0"], dev)
    nt, ne = big.num_tuples, big.num_edges
    x1, x2 = torch.randn(nt, 128, device=dev), torch.randn(nt, 128, device=dev)
    a1 = torch.randn(ne, 128, device=dev)
    f = lambda x, a: message_reduce(x, a, acd_b, nt, nt, ne, "sum")
    lhs = f(x1 + 2 * x2, a1)
    rhs = f(x1, a1) + 2 * f(x2, a1)
    torch.testing.assert_close(lhs, rhs, rtol=1e-4, atol=1e-4)
    # checksum of checksums: sum_a out[a] == sum_m x[c_m] * a[d_m]
    c, dd = acd_b[1], acd_b[2]
    direct = (x1.double()[c] * a1.double()[dd]).sum(0)
    # f32 rows summed into a checksum: the bound is relative to the sum of MAGNITUDES (the signed sum cancels to ~ sqrt(M); a tolerance
    # relative to it failed for unlucky draws: 3e-6 at one channel)
    bound = 1e-6 * (x1.double().abs()[c] * a1.double().abs()[dd]).sum(0)
    assert bool(((f(x1, a1).double().sum(0) - direct).abs() <= bound).all())
    assert bool((torch.diff(acd_b[0]) >= 0).all())                            # collated plan stays sorted


@pytest.mark.parametrize("kind,graphs,d,key", [("zinc", 8192, 128, "X___X___1___A___0"), ("i2", 2048, 256, "X___X___2___A___0")])
def test_baseline_size_properties_bf16(dev, kind, graphs, d, key):
    """BASELINE.json's full sizes (config 2: 8192 ZINC-shape graphs, hidden 128; config 5: 2048 I2-shape graphs, hidden 256; bf16),
    where the oracle no longer finishes in seconds: size-independent properties of the forward launch and of both backward plans
    (fast kernel, LDS-window kernels).  (i) scaling an operand by a power of two scales the result by it, bit for bit; (ii) a
    per-channel checksum of every output equals the f64 sum over all messages of the operand products within the rounding of
    the bf16 outputs; (iii) two runs are bit-identical (no atomics anywhere)."""
    from pygho_amd import _ops, synth
    hb = synth.replicate(synth.make_batch(min(graphs, 1024), kind, seed=1000), max(1, graphs // 1024))
    acd = T(hb.acd[key], dev)
    nt, ne = hb.num_tuples, hb.num_edges
    plan = _ops.message_plan(acd, nt, nt, ne)
    torch.manual_seed(0)
    x = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    a = torch.randn(ne, d, device=dev).to(torch.bfloat16)
    g = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    pc, a_c, d_c = plan.by_c()
    pd, a_d, c_d = plan.by_d()
    calls = {
        "forward": (lambda u, v: _ops.seg_gmr(nt, u, v, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum"), x, a, acd[1], acd[2]),
        "by-tuple backward": (lambda u, v: _ops.seg_gmr(nt, u, v, pc.seg_ptr, a_c, d_c, "sum"), g, a, acd[0], acd[2]),
        "by-edge backward": (lambda u, v: _ops.seg_gmr(ne, u, v, pd.seg_ptr, a_d, c_d, "sum"), g, x, acd[0], acd[1]),
    }
    for name, (fn, u, v, iu, iv) in calls.items():
        out = fn(u, v)
        assert torch.equal(out, fn(u, v)), name                                  # (iii)
        assert torch.equal(fn(u * 4, v), out * 4) and torch.equal(fn(u, v * 0.5), out * 0.5), name      # (i)
        direct = torch.zeros(d, dtype=torch.float64, device=dev)                 # (ii), in slices: 8 M messages x 256 f64 do not fit at once
        step = 1 << 20
        for lo in range(0, acd.shape[1], step):
            direct += (u.double()[iu[lo:lo + step]] * v.double()[iv[lo:lo + step]]).sum(0)
        got = out.double().sum(0)
        # every output row carries one bf16 rounding (2^-9 relative, random sign): the checksum's error grows like the root of the row count
        scale = float(out.float().abs().mean()) * out.shape[0] ** 0.5 * 2.0 ** -8
        torch.testing.assert_close(got, direct, rtol=0, atol=6 * scale, msg=name)


@pytest.mark.parametrize("kind,graphs,d,key", [("zinc", 8192, 128, "X___X___1___A___0"), ("i2", 2048, 256, "X___X___2___A___0")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_baseline_size_elementwise_vs_host_oracle(dev, kind, graphs, d, key, dtype):
    """BASELINE.json's full sizes, EVERY output element: the forward launch and both gradient plans (through autograd, i.e. through
    whatever kernel the dispatcher picks for each plan) against the reference's ATen op sequence run on the host in f32
    (oracle.aten_port.spspmm_values_chunked: index, index, mul, index_add_ in message order).  f32: bit-identical.  bf16: the
    kernel's f32 accumulator equals the oracle's (the f32 product of two bf16 values is exact), so the result must equal the
    oracle rounded ONCE to bf16 -- also bit for bit; `mean` likewise (one f32 division, then the rounding), and `mean`'s two
    gradient plans against the port with the per-message scale applied in the kernels' association (f32 and bf16, bit for bit)."""
    from oracle import aten_port as P
    from pygho_amd import synth
    from pygho_amd._ops import message_reduce
    hb = synth.make_batch(graphs, kind, seed=1000)                 # every graph distinct: no periodic index structure
    acd_h = torch.from_numpy(hb.acd[key])
    acd = acd_h.to(dev)
    nt, ne = hb.num_tuples, hb.num_edges
    # the by-edge gradient of 16-bit rows goes through the scatter kernel once its plan exists (the dispatcher builds it only for
    # patterns that keep coming back): built here so that THIS test covers that kernel at full size
    from pygho_amd import _ops
    assert _ops.scatter_plan(_ops.message_plan(acd, nt, nt, ne)) is not None
    gen = torch.Generator().manual_seed(0)
    xh = torch.randn(nt, d, generator=gen).to(dtype)
    ah = torch.randn(ne, d, generator=gen).to(dtype)
    gh = torch.randn(nt, d, generator=gen).to(dtype)
    x32, a32, g32 = xh.float(), ah.float(), gh.float()
    cast = (lambda t: t) if dtype == torch.float32 else (lambda t: t.to(dtype).float())

    def same(got, exp, what):
        got = got.detach().float().cpu()
        if not torch.equal(got, cast(exp)):
            bad = (got != cast(exp))
            raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} elements differ, max abs {float((got - exp).abs().max()):.3e}")

    for aggr in ("sum", "mean"):
        xg = xh.to(dev).requires_grad_(True)
        ag = ah.to(dev).requires_grad_(True)
        out = message_reduce(xg, ag, acd, nt, nt, ne, aggr)
        same(out, P.spspmm_values_chunked(x32, a32, acd_h[0], acd_h[1], acd_h[2], nt, aggr), f"{kind} {aggr} forward")
        out.backward(gh.to(dev))
        if aggr == "sum":
            gX = P.spspmm_values_chunked(g32, a32, acd_h[1], acd_h[0], acd_h[2], nt, "sum")
            gA = P.spspmm_values_chunked(g32, x32, acd_h[2], acd_h[0], acd_h[1], ne, "sum")
            same(xg.grad, gX, f"{kind} gradient wrt the tuple values (by-tuple plan)")
            same(ag.grad, gA, f"{kind} gradient wrt the adjacency values (by-edge plan)")
        else:
            # mean: the kernels scale every message by 1 / count AFTER forming the product, (1 / count[a]) * (g[a] * v), all in f32;
            # the port applies the same association (a_rowscale) -> bit-identical for f32 AND for bf16 (one rounding at the store).
            # 1 / count itself: an f32 reciprocal of a small integer, the same bits on the host and the device (checked)
            cnt = torch.bincount(acd_h[0], minlength=nt).clamp_min(1).float()
            inv = cnt.reciprocal()
            assert torch.equal(cnt.to(dev).reciprocal().cpu(), inv), "f32 reciprocal differs between the host and the device"
            gX = P.spspmm_values_chunked(g32, a32, acd_h[1], acd_h[0], acd_h[2], nt, "sum", a_rowscale=inv)
            gA = P.spspmm_values_chunked(g32, x32, acd_h[2], acd_h[0], acd_h[1], ne, "sum", a_rowscale=inv)
            same(xg.grad, gX, f"{kind} mean: gradient wrt the tuple values (by-tuple plan)")
            same(ag.grad, gA, f"{kind} mean: gradient wrt the adjacency values (by-edge plan)")
            if dtype == torch.float32:
                # and against autograd's own association, (g / count) * v: equal up to the f32 rounding of that one product
                gX2 = P.spspmm_values_chunked(g32 / cnt.unsqueeze(1), a32, acd_h[1], acd_h[0], acd_h[2], nt, "sum")
                gA2 = P.spspmm_values_chunked(g32 / cnt.unsqueeze(1), x32, acd_h[2], acd_h[0], acd_h[1], ne, "sum")
                torch.testing.assert_close(xg.grad.cpu(), gX2, rtol=1e-5, atol=1e-5)
                torch.testing.assert_close(ag.grad.cpu(), gA2, rtol=1e-5, atol=1e-4)
        del xg, ag, out
    # max (the reference's other first-class aggregation, utils.py:44-56): forward = the extremum of the exact products rounded once;
    # gradients = torch's scatter_reduce_backward rule restated chunk by chunk (oracle.aten_port.spspmm_extremum_grads_chunked: even
    # split among ties, the share grad / N rounded to the value dtype, f32 sums in message order) -- bit for bit, f32 and bf16
    xg = xh.to(dev).requires_grad_(True)
    ag = ah.to(dev).requires_grad_(True)
    out = message_reduce(xg, ag, acd, nt, nt, ne, "max")
    fwd = P.spspmm_values_chunked(x32, a32, acd_h[0], acd_h[1], acd_h[2], nt, "max")
    same(out, fwd, f"{kind} max forward")
    out.backward(gh.to(dev))
    gX, gA = P.spspmm_extremum_grads_chunked(x32, a32, acd_h, nt, cast(fwd), g32, store_dtype=dtype)
    same(xg.grad, gX, f"{kind} max: gradient wrt the tuple values (by-tuple plan)")
    same(ag.grad, gA, f"{kind} max: gradient wrt the adjacency values (by-edge plan)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("aggr", ["max", "min"])
def test_extremum_backward_with_ties_and_zero_extrema_vs_autograd_of_the_port(dev, dtype, aggr):
    """values on a small integer grid (many ties, many extrema that are exactly 0): the gradient of max / min against AUTOGRAD through
    the reference's op sequence on the host (oracle.aten_port.spspmm_values: index, index, mul, scatter_reduce_(amax|amin)).  Torch
    counts (self == result) into the number of ties, and `self` is the zero-initialised output (utils.py:44-49), so an extremum of
    exactly 0 shares its gradient with one phantom tie -- reproduced (it is what the reference computes).  Grid values and small
    tie counts keep every quotient and product exact in bf16 too, so the comparison is to f32 rounding of the sums."""
    from oracle import aten_port as P
    from pygho_amd import synth
    from pygho_amd._ops import message_reduce
    hb = synth.make_batch(40, "zinc", seed=9)
    acd_h = torch.from_numpy(hb.acd["X___X___1___A___0"])
    nt, ne, d = hb.num_tuples, hb.num_edges, 64
    gen = torch.Generator().manual_seed(1)
    grid = torch.tensor([-2.0, -1.0, 0.0, 1.0, 2.0])
    xh = grid[torch.randint(0, 5, (nt, d), generator=gen)].requires_grad_(True)
    ah = grid[torch.randint(0, 5, (ne, d), generator=gen)].requires_grad_(True)
    gh = torch.randint(-8, 9, (nt, d), generator=gen).float() * 6.0            # divisible by 1, 2, 3 and 6: exact shares for most tie counts
    P.spspmm_values(xh, ah, acd_h, nt, aggr).backward(gh)
    xg = xh.detach().to(dev).to(dtype).requires_grad_(True)
    ag = ah.detach().to(dev).to(dtype).requires_grad_(True)
    out = message_reduce(xg, ag, acd_h.to(dev), nt, nt, ne, aggr)
    assert torch.equal(out.detach().float().cpu(), P.spspmm_values(xh.detach(), ah.detach(), acd_h, nt, aggr))
    assert int((out == 0).sum()) > 100, "the case must contain extrema that are exactly zero"
    out.backward(gh.to(dev).to(dtype))
    tol = dict(rtol=1e-5, atol=1e-4) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=0.3)
    torch.testing.assert_close(xg.grad.float().cpu(), xh.grad, **tol)
    torch.testing.assert_close(ag.grad.float().cpu(), ah.grad, **tol)


@pytest.mark.parametrize("kind,graphs,d,key", [("zinc", 600, 128, "X___X___1___A___0"), ("zinc", 300, 64, "X___X___1___A___0"),
                                               ("i2", 96, 256, "X___X___2___A___0"), ("i2", 80, 128, "X___X___2___A___0")])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_addend", [False, True])
def test_by_edge_scatter_form_is_bit_identical_to_the_gather_form(dev, kind, graphs, d, key, dtype, with_addend, monkeypatch):
    """the by-edge gradient gB[d] = [addend +] sum g[a] * A[c] as a scatter over the forward message order (csrc/seg_scatter.hip: every
    operand row fetched once, per-block f32 accumulators in LDS) against the gather form over the messages grouped by d
    (seg_gmr on plan.by_d(), the fast / window kernels): the same f32 sums in the same order, so the same bits -- and against an f64
    sum of the exact products to one rounding.  Also the planner's invariants: chunks partition every block's messages, windows hold
    every row a chunk's messages name, and the packed words decode back to (a, c, d)."""
    from pygho_amd import _ops, synth
    from pygho_amd import segment as S
    monkeypatch.setattr(S, "SEG_SCATTER_MIN_MESSAGES", 0)                      # the dispatch threshold (10^6 messages) is a speed matter
    hb = synth.make_batch(graphs, kind, seed=5)
    acd = torch.from_numpy(hb.acd[key]).to(dev)
    nt, ne = hb.num_tuples, hb.num_edges
    plan = _ops.message_plan(acd, nt, nt, ne)
    sp = S.scatter_plan(plan)
    assert sp is not None and sp.covers and sp.n_blocks == graphs, "a block-diagonal batch: one block per graph, every edge row covered"
    # ---- planner invariants (integer work, exact) ----
    chunks = sp.chunks.cpu().numpy().astype(np.int64)
    words = sp.words.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    a, c, dd = (acd[i].cpu().numpy() for i in range(3))
    chunk0, blk_e = sp.chunk0.cpu().numpy(), sp.blk_e.cpu().numpy()
    m_lo, n = chunks[:, 0], chunks[:, 3] & 0xFF
    a_rows, c_rows = (chunks[:, 3] >> 8) & 0xFF, (chunks[:, 3] >> 16) & 0xFF
    first, last = (chunks[:, 3] >> 24) & 1, (chunks[:, 3] >> 25) & 1
    assert m_lo[0] == 0 and np.array_equal(m_lo[1:], (m_lo + n)[:-1]) and m_lo[-1] + n[-1] == acd.shape[1]
    assert n.min() >= 1 and n.max() <= 64 and a_rows.max() <= 32 and c_rows.max() <= 32
    assert first.sum() == last.sum() == graphs and np.array_equal(np.nonzero(first)[0], chunk0[:-1])
    blk_of_chunk = np.cumsum(first) - 1
    ci = np.repeat(np.arange(len(n)), n)                                     # chunk of every message
    assert np.array_equal((words & 31) + chunks[ci, 1], a) and np.array_equal(((words >> 5) & 31) + chunks[ci, 2], c)
    assert np.array_equal(((words >> 10) & 255) + blk_e[blk_of_chunk[ci], 0], dd)
    assert ((words & 31) < a_rows[ci]).all() and (((words >> 5) & 31) < c_rows[ci]).all()
    pos = np.arange(acd.shape[1]) - m_lo[ci]
    trip = ci * 4 + pos // 16
    order = np.lexsort((np.arange(len(dd)), dd, trip))                        # within a trip: by edge, then by message
    same = (trip[order][1:] == trip[order][:-1]) & (dd[order][1:] == dd[order][:-1])
    run = np.zeros(len(dd), dtype=np.int64)
    for i in np.nonzero(same)[0]:
        run[i + 1] = run[i] + 1
    phase = np.empty_like(run)
    phase[order] = run
    assert np.array_equal((words >> 18) & 3, phase) and phase.max() <= 3
    # ---- values ----
    gen = torch.Generator(device=dev).manual_seed(0)
    g = torch.randn(nt, d, device=dev, generator=gen).to(dtype)
    h = torch.randn(nt, d, device=dev, generator=gen).to(dtype)
    add = torch.randn(ne, d, device=dev, generator=gen).to(dtype) if with_addend else None
    timer = _ops.LaunchTimer()
    with timer:
        got = S.by_edge_product(plan, g, h, None, add)
    torch.cuda.synchronize()
    assert any(",scatter" in k for k in timer.summary()), f"the scatter kernel was not dispatched: {list(timer.summary())}"
    p, a_g, c_g = plan.by_d()
    ref = _ops.seg_gmr(ne, g, h, p.seg_ptr, a_g, c_g, "sum", None, addend=add)
    assert torch.equal(got, ref), f"{int((got != ref).sum())} of {got.numel()} elements differ from the gather form"
    exact = torch.zeros(ne, d, dtype=torch.float64, device=dev).index_add_(0, acd[2], g.double()[acd[0]] * h.double()[acd[1]])
    if add is not None:
        exact = exact + add.double()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    torch.testing.assert_close(got.double(), exact, rtol=eps, atol=eps * float(exact.abs().max()) * 2.0 ** -6)
    assert torch.equal(S.by_edge_product(plan, g, h, None, add), got)          # and run to run


def _synthetic_blocks(rng, n_graphs, tuples, edges, msgs_per_tuple, c_spread, skip=(), cyclic=False):
    """block-diagonal message triples made by hand: graph g owns tuple rows [g * tuples, ...) and edge rows [g * edges, ...); every
    tuple a sends `msgs_per_tuple` messages whose c lies within `c_spread` rows of a, to random edges of its graph (`cyclic`: to edge
    (message number) % edges, so that ANY 16 consecutive messages hit every edge equally often); graphs in `skip` send none"""
    a, c, d = [], [], []
    for g in range(n_graphs):
        if g in skip:
            continue
        t0, e0, sent = g * tuples, g * edges, 0
        for t in range(tuples):
            k = int(rng.integers(max(1, msgs_per_tuple - 1), msgs_per_tuple + 2))
            a += [t0 + t] * k
            c += list(t0 + np.clip(t + rng.integers(-c_spread, c_spread + 1, k), 0, tuples - 1))
            d += [e0 + (sent + i) % edges for i in range(k)] if cyclic else list(e0 + rng.integers(0, edges, k))
            sent += k
    return np.stack([np.array(a), np.array(c), np.array(d)]).astype(np.int64)


@pytest.mark.parametrize("case", ["many_edges", "empty_blocks", "few_blocks", "hot_edges"])
def test_by_edge_scatter_edge_cases(dev, case, monkeypatch):
    """the scatter form outside the benchmark shapes, each against the gather form bit for bit: blocks of 97..255 edges (the 16-load
    flush variant), graphs without a message (their edge rows are covered by no block: the caller pre-fills them, with and without an
    addend), fewer blocks than compute units, and a few edges that receive many messages of one trip (phases 1..3; a block whose phase
    would exceed 3 makes the whole plan fall back)"""
    from pygho_amd import _ops
    from pygho_amd import segment as S
    monkeypatch.setattr(S, "SEG_SCATTER_MIN_MESSAGES", 0)
    rng = np.random.default_rng(11)
    if case == "many_edges":
        n_graphs, tuples, edges, acd = 40, 260, 200, None
        acd = _synthetic_blocks(rng, n_graphs, tuples, edges, 3, 6)
    elif case == "empty_blocks":
        n_graphs, tuples, edges = 60, 120, 90
        acd = _synthetic_blocks(rng, n_graphs, tuples, edges, 2, 8, skip=(0, 17, 18, 59))
    elif case == "few_blocks":
        n_graphs, tuples, edges = 3, 200, 90
        acd = _synthetic_blocks(rng, n_graphs, tuples, edges, 3, 10)
    else:
        n_graphs, tuples, edges = 50, 150, 4                       # 4 edges per graph, hit in turn: every trip holds each edge 4 times
        acd = _synthetic_blocks(rng, n_graphs, tuples, edges, 2, 5, cyclic=True)
    nt, ne, d = n_graphs * tuples, n_graphs * edges, 128
    acd_t = torch.from_numpy(acd).to(dev)
    plan = _ops.message_plan(acd_t, nt, nt, ne)
    sp = S.scatter_plan(plan)
    p, a_g, c_g = plan.by_d()
    gen = torch.Generator(device=dev).manual_seed(1)
    g = torch.randn(nt, d, device=dev, generator=gen).to(torch.bfloat16)
    h = torch.randn(nt, d, device=dev, generator=gen).to(torch.bfloat16)
    add = torch.randn(ne, d, device=dev, generator=gen).to(torch.bfloat16)
    assert sp is not None
    if case == "hot_edges":
        phases = (sp.words.cpu().numpy().astype(np.int64) >> 18) & 3
        assert phases.max() == 3 and (phases == 3).sum() > 1000, "the case must exercise every phase"
        # one more message of an edge inside a trip (five of 16) is beyond the kernel's four phases: that plan must fall back
        acd5 = acd.copy()
        acd5[2, :16] = acd5[2, 0]
        plan5 = _ops.message_plan(torch.from_numpy(acd5).to(dev), nt, nt, ne)
        assert S.scatter_plan(plan5) is None
    if case == "many_edges":
        assert 96 < sp.max_edges <= 255
    if case == "empty_blocks":
        assert not sp.covers                                       # the edge rows of graphs 0, 17, 18, 59 belong to no block
    for addend in (None, add):
        timer = _ops.LaunchTimer()
        with timer:
            got = S.by_edge_product(plan, g, h, None, addend)
        torch.cuda.synchronize()
        assert any(",scatter" in k for k in timer.summary()), list(timer.summary())
        ref = _ops.seg_gmr(ne, g, h, p.seg_ptr, a_g, c_g, "sum", None, addend=addend)
        assert torch.equal(got, ref), f"{case}: {int((got != ref).sum())} of {got.numel()} elements differ (addend: {addend is not None})"


def test_by_edge_scatter_plan_is_built_only_on_request(dev, monkeypatch):
    """round 5 policy: the dispatcher never plans on its own (round 4 planned a pattern after 12 by-edge launches -- hidden state that made
    WHICH kernel ran, and whether a step read back from the device, depend on a batch's history).  A plan exists when the caller asked
    (`scatter_plan(plan)` / `SpModel.prepare`) or the batch came with one (`DeviceGraphStore.collate`); until then the gather form runs,
    however often the pattern comes back, with no planner fetch -- and both forms return the same bits.  Building it takes two host
    reads (block count; chunk total + verdicts) and no ATen scan (`pygho_block_cuts`: cuts bit-identical to the cummax / cummin rule)."""
    from pygho_amd import _ops, synth
    from pygho_amd import segment as S
    monkeypatch.setattr(S, "SEG_SCATTER_MIN_MESSAGES", 0)
    hb = synth.make_batch(200, "zinc", seed=12)
    acd = torch.from_numpy(hb.acd["X___X___1___A___0"]).to(dev)
    nt, ne = hb.num_tuples, hb.num_edges
    plan = _ops.message_plan(acd, nt, nt, ne)
    g = torch.randn(nt, 128, device=dev).to(torch.bfloat16)
    h = torch.randn(nt, 128, device=dev).to(torch.bfloat16)

    def run():
        timer = _ops.LaunchTimer()
        with timer:
            out = S.by_edge_product(plan, g, h)
        torch.cuda.synchronize()
        return out, any(",scatter" in k for k in timer.summary())
    first = run()                                                                 # (builds the by-d grouping of the gather form: one fetch)
    f0 = _ops.FETCHES[0]
    outs = [first] + [run() for _ in range(20)]
    assert not any(kind for _, kind in outs) and _ops.FETCHES[0] == f0           # 20 more uses: still the gather form, nothing read back
    sp = S.scatter_plan(plan)                                                     # the caller asks
    assert sp is not None and _ops.FETCHES[0] == f0 + 2
    # the device cut finder against the rule it restates: a block starts where max d[:m] < min d[m:]
    d = plan.d32
    pm = torch.cummax(d, 0).values
    sm = torch.flip(torch.cummin(torch.flip(d, [0]), 0).values, [0])
    cut = torch.nonzero(pm[:-1] < sm[1:]).flatten() + 1
    want = torch.cat([cut.new_zeros(1), cut, cut.new_full((1,), plan.m)]).to(torch.int32)
    assert torch.equal(S.block_cuts(d), want) and sp.n_blocks == want.numel() - 1 == 200
    after, kind = run()
    assert kind and torch.equal(after, outs[0][0]) and all(torch.equal(outs[0][0], o) for o, _ in outs[1:])


def test_message_plan_range_check_rides_on_the_narrowing_pass(dev):
    """an operand index outside its operand's rows (the reference's gather raises IndexError, Spspmm.py:309-311) is reported by the
    plan's construction -- through the flag of the narrowing kernel (`pygho_narrow_i64_i32_bounded`), not a separate reduction"""
    from pygho_amd import _ops, synth
    hb = synth.make_batch(20, "zinc", seed=13)
    nt, ne = hb.num_tuples, hb.num_edges
    good = torch.from_numpy(hb.acd["X___X___1___A___0"]).to(dev)
    _ops.message_plan(good, nt, nt, ne)
    for row, val in ((1, nt), (2, ne), (1, -1), (2, -5)):
        bad = good.clone()
        bad[row, 7] = val
        with pytest.raises(ValueError, match="acd operand index out of range"):
            _ops.message_plan(bad, nt, nt, ne)
            _ops.check_deferred_errors()


def test_by_edge_scatter_falls_back_outside_its_limits(dev):
    """plans the scatter kernel cannot take -- a block with more than 255 edges (one big graph), a pattern that is not block diagonal
    in message order -- return no ScatterPlan, and by_edge_product gives the gather form's result through the window / fast kernels"""
    from pygho_amd import _ops
    from pygho_amd import segment as S
    gen = torch.Generator().manual_seed(3)
    nt, ne, m, d = 6000, 5000, 40000, 128
    a = torch.sort(torch.randint(0, nt, (m,), generator=gen)).values
    acd = torch.stack([a, torch.randint(0, nt, (m,), generator=gen), torch.randint(0, ne, (m,), generator=gen)]).to(dev)
    plan = _ops.message_plan(acd, nt, nt, ne)
    assert S.scatter_plan(plan) is None                                        # one block of 5000 edges
    g = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    h = torch.randn(nt, d, device=dev).to(torch.bfloat16)
    got = S.by_edge_product(plan, g, h)
    exact = torch.zeros(ne, d, dtype=torch.float64, device=dev).index_add_(0, acd[2], g.double()[acd[0]] * h.double()[acd[1]])
    torch.testing.assert_close(got.double(), exact, rtol=2.0 ** -8, atol=2.0 ** -8)


@pytest.mark.parametrize("kind", ["i2", "zinc"])
@pytest.mark.parametrize("win", [8, 24, 32])
def test_tile_plan_invariants(dev, kind, win):
    """`pygho_seg_tile_plan` (integer work, checked exactly on the host): the tiles of every 256-segment chunk partition the chunk
    in order; a tile holds at most 64 segments; every lhs index of a tile with a window lies inside [row0, row0 + rows) with
    rows <= win_rows and the window is tight (its first and last row are used); a tile without a window has no messages or is ONE
    segment whose own row range is wider than the window; first message / message count equal the CSR pointers; the quarter
    boundaries are segment boundaries, monotone, and the smallest ones with at least j quarters of the messages in front."""
    from pygho_amd import _ops, synth
    key = "X___X___1___A___0" if kind == "zinc" else "X___X___2___A___0"
    hb = synth.make_batch(40 if kind == "zinc" else 6, kind, seed=9)
    acd = T(hb.acd[key], dev)
    nt, ne = hb.num_tuples, hb.num_edges
    plan = _ops.message_plan(acd, nt, nt, ne)
    pd, a_d, c_d = plan.by_d()
    for n_seg, sp_t, li_t in ((nt, plan.fwd.seg_ptr, plan.c_fwd), (ne, pd.seg_ptr, a_d)):
        cnt_t, tiles_t = _ops.tile_plan(sp_t, li_t, n_seg, win)
        sp, li = sp_t.cpu().numpy().astype(np.int64), li_t.cpu().numpy().astype(np.int64)
        cnt, tiles = cnt_t.cpu().numpy(), tiles_t.cpu().numpy()
        chunk = tiles.shape[1]
        assert cnt.shape[0] == (n_seg + chunk - 1) // chunk
        for ci in range(cnt.shape[0]):
            nloc = min(chunk, n_seg - ci * chunk)
            expect_first = 0
            for t in range(int(cnt[ci])):
                d = tiles[ci, t]
                first, ns, rows, row0, m0, nmsg = d[0] & 0xff, (d[0] >> 8) & 0xff, (d[0] >> 16) & 0xff, d[1], d[2], d[3]
                assert first == expect_first and 1 <= ns <= 64 and rows <= win
                expect_first += ns
                s0 = ci * chunk + first
                assert m0 == sp[s0] and nmsg == sp[s0 + ns] - sp[s0]
                idx = li[m0:m0 + nmsg]
                if rows:
                    assert nmsg > 0 and idx.min() == row0 and idx.max() == row0 + rows - 1
                else:
                    assert nmsg == 0 or (ns == 1 and idx.max() - idx.min() >= win)
                q = [d[4] & 0xff, (d[4] >> 8) & 0xff, (d[4] >> 16) & 0xff]
                ms = [d[5], d[6], d[7]]
                assert 0 <= q[0] <= q[1] <= q[2] <= ns
                for j in range(3):
                    assert ms[j] == sp[s0 + q[j]] - m0
                    assert 4 * ms[j] >= (j + 1) * nmsg or q[j] == ns                    # at least j + 1 quarters in front ...
                    assert q[j] == 0 or 4 * (sp[s0 + q[j] - 1] - m0) < (j + 1) * nmsg    # ... and no earlier boundary has
            assert expect_first == nloc


@pytest.mark.parametrize("kind,d,dtype", [("i2", 256, torch.bfloat16), ("i2", 128, torch.bfloat16), ("i2", 128, torch.float32),
                                          ("i2", 256, torch.float32), ("zinc", 128, torch.float32), ("i2", 256, torch.float16)])
def test_tile_kernel_bit_identical_to_fast_kernel(dev, kind, d, dtype):
    """`pygho_seg_gather_mul_reduce_tiled` (one wavefront per tile, lhs window in LDS, 1 / 2 / 4 streams per wavefront for 1024 / 512 /
    256-B rows) against the fast kernel on the forward, by-tuple and by-edge plans: sum, mean, sum with a per-row scale (the mean's
    gradient plans) and sum / mean with a residual row -- the same bits.  The plans are the shipped ones plus a doctored copy with
    empty segments at both ends and in the middle, one segment whose lhs rows span more than any window (an irregular tile) and
    one very long segment (more messages than a stream's index vector holds)."""
    from pygho_amd import _ops, synth
    key = "X___X___1___A___0" if kind == "zinc" else "X___X___2___A___0"
    hb = synth.make_batch(48 if kind == "zinc" else 8, kind, seed=5)
    acd = T(hb.acd[key], dev)
    nt, ne = hb.num_tuples, hb.num_edges
    assert nt >= 4096
    # doctored plan: drop the messages of a few output rows (empty segments), send one row's messages to far-apart lhs rows, and give
    # one row 300 messages
    a, c, dd = acd[0].clone(), acd[1].clone(), acd[2].clone()
    keep = ~((a < 3) | (a == 1000) | (a == 1001) | (a >= nt - 2))
    a, c, dd = a[keep], c[keep], dd[keep]
    wide = a == 2000
    c[wide] = torch.linspace(0, nt - 1, int(wide.sum()), device=dev).long()
    extra = 300
    ins = int(torch.searchsorted(a, torch.tensor(3000, device=dev)))
    gen = torch.Generator(device="cpu").manual_seed(1)
    c_long = (int(c[ins]) + torch.randint(0, 8, (extra,), generator=gen)).clamp_max(nt - 1).to(dev)
    d_long = torch.randint(0, ne, (extra,), generator=gen).to(dev)
    a2 = torch.cat((a[:ins], torch.full((extra,), 3000, device=dev), a[ins:]))
    c2 = torch.cat((c[:ins], c_long, c[ins:]))
    d2 = torch.cat((dd[:ins], d_long, dd[ins:]))
    acd_doc = torch.stack((a2, c2, d2)).contiguous()
    torch.manual_seed(0)
    x = torch.randn(nt, d, device=dev).to(dtype)
    av = torch.randn(ne, d, device=dev).to(dtype)
    g = torch.randn(nt, d, device=dev).to(dtype)
    res = torch.randn(nt, d, device=dev).to(dtype)
    saved = (_ops.SEG_TILE, _ops.SEG_TILE_WIN_ROWS)
    try:
        for plan_acd in (acd, acd_doc):
            plan = _ops.message_plan(plan_acd, nt, nt, ne)
            pc, a_c, d_c = plan.by_c()
            pd, a_d, c_d = plan.by_d()
            scale = plan.fwd.inv_count
            calls = [("forward sum", nt, x, av, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum", None, None),
                     ("forward mean", nt, x, av, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "mean", None, None),
                     ("forward sum + residual", nt, x, av, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "sum", None, res),
                     ("forward mean + residual", nt, x, av, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, "mean", None, res),
                     ("by-tuple backward", nt, g, av, pc.seg_ptr, a_c, d_c, "sum", None, None),
                     ("by-tuple backward of a mean", nt, g, av, pc.seg_ptr, a_c, d_c, "sum", scale, None),
                     ("by-edge backward", ne, g, x, pd.seg_ptr, a_d, c_d, "sum", None, None)]
            for name, n, lhs, rhs, sp, li, ri, aggr, sc, add in calls:
                if n < 4096:
                    continue
                outs = {}
                for mode, win in (("0", 24), ("1", 24), ("1", 32)):
                    _ops.SEG_TILE, _ops.SEG_TILE_WIN_ROWS = mode, win
                    addend = None if add is None else add[:n].contiguous()
                    outs[(mode, win)] = _ops.seg_gmr(n, lhs, rhs, sp, li, ri, aggr, sc, addend)
                ref = outs[("0", 24)]
                view = torch.int16 if ref.element_size() == 2 else torch.int32
                for k, o in outs.items():
                    assert torch.equal(o.view(view), ref.view(view)), f"{kind} d={d} {dtype} {name}: tiled (window {k[1]}) differs from the fast kernel"
    finally:
        _ops.SEG_TILE, _ops.SEG_TILE_WIN_ROWS = saved


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_long_segments_hierarchical(dev, dtype):
    """a 4-row table receiving 300k rows (the embedding-backward shape): hierarchical chunks, f32 partials"""
    from pygho_amd.backend.utils import torch_scatter_reduce
    rng = np.random.default_rng(0)
    m, d, n = 300_000, 64, 6
    ind = rng.integers(0, 4, size=m).astype(np.int64)           # segments 4, 5 stay empty
    ind[:5] = [3, 0, 2, 1, 0]
    src = rng.standard_normal((m, d)).astype(np.float32)
    st = T(src, dev, dtype)
    ref_in = N(st).astype(np.float64)
    for ag in ("sum", "mean", "max", "min"):
        got = N(torch_scatter_reduce(0, st, T(ind, dev), n, ag)).astype(np.float64)
        exp = O.scatter_reduce(ref_in, ind, n, ag)
        tol = 1e-4 if dtype == torch.float32 else 2.0 ** -7
        np.testing.assert_allclose(got, exp, rtol=tol, atol=tol * max(1.0, np.abs(exp).max()))
    # gradient of a gather from the tiny table == sum over the long segments
    table = torch.randn(4, d, device=dev, dtype=dtype, requires_grad=True)
    from pygho_amd import _ops
    out = _ops.gather_rows(table, T(ind, dev))
    w = T(src, dev, dtype)
    (out.float() * w.float()).sum().backward()
    exp = O.scatter_reduce(N(w).astype(np.float64), ind, 4, "sum")
    np.testing.assert_allclose(N(table.grad).astype(np.float64), exp, rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=1.0 if dtype == torch.bfloat16 else 1e-2)


def test_model_building_blocks_match_torch(dev):
    """IndexEmbedding / split-K Linear keep nn.Embedding / nn.Linear semantics (forward and gradients)"""
    from pygho_amd.honn.utils import Linear
    from pygho_amd.ngnn import IndexEmbedding
    torch.manual_seed(0)
    idx = torch.randint(0, 16, (50_000,), device=dev)
    e1, e2 = IndexEmbedding(16, 32).to(dev), torch.nn.Embedding(16, 32).to(dev)
    e2.load_state_dict(e1.state_dict())
    w = torch.randn(50_000, 32, device=dev)
    (e1(idx) * w).sum().backward()
    (e2(idx) * w).sum().backward()
    assert torch.equal(e1(idx), e2(idx))
    torch.testing.assert_close(e1.weight.grad, e2.weight.grad, rtol=1e-4, atol=1e-3)
    l1, l2 = Linear(64, 48).to(dev), torch.nn.Linear(64, 48).to(dev)
    l2.load_state_dict(l1.state_dict())
    x1 = torch.randn(40_000, 64, device=dev, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    g = torch.randn(40_000, 48, device=dev)
    (l1(x1) * g).sum().backward()
    (l2(x2) * g).sum().backward()
    torch.testing.assert_close(l1(x1), l2(x2), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(l1.weight.grad, l2.weight.grad, rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(l1.bias.grad, l2.bias.grad, rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("rows", [300, 40_000])
def test_linear_under_autocast_reads_the_cast_arena(dev, rows):
    """honn.utils.Linear under bf16 autocast (any height): same output bits as nn.Linear under autocast (the arena's copy is the
    same rounding of the master weight), gradients in the masters' f32 within bf16 rounding of autocast's (ours skip the bf16
    rounding of dW / db), with and without a cast arena behind the module; and an optimizer step is seen by the next forward."""
    from pygho_amd import _ops
    from pygho_amd.honn.utils import Linear
    torch.manual_seed(3)
    l1, l2 = Linear(128, 128).to(dev), torch.nn.Linear(128, 128).to(dev)
    l2.load_state_dict(l1.state_dict())
    x = torch.randn(rows, 128, device=dev)
    g = torch.randn(rows, 128, device=dev)
    for use_arena in (False, True):
        l1.zero_grad(set_to_none=True)
        l2.zero_grad(set_to_none=True)
        if use_arena:
            _ops.ensure_cast_arena(l1, torch.bfloat16)
        x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y1, y2 = l1(x1), l2(x2)
        # (square maps of width 64 / 128 run on the row-block kernel, others on the library: f32 accumulation and one rounding in
        # both, not the same summation order -- so within one bf16 rounding of the exact product rather than bit-equal)
        assert y1.dtype == y2.dtype == torch.bfloat16
        xq, wq, bq = x.to(torch.bfloat16).double(), l1.weight.detach().to(torch.bfloat16).double(), l1.bias.detach().to(torch.bfloat16).double()
        exact, mag = xq @ wq.t() + bq, xq.abs() @ wq.abs().t() + bq.abs()
        for y in (y1, y2):
            assert bool(((y.double() - exact).abs() <= 2.0 ** -8 * exact.abs() + 2.0 ** -20 * mag).all())
        (y1.float() * g).sum().backward()
        (y2.float() * g).sum().backward()
        assert l1.weight.grad.dtype == torch.float32 and l1.bias.grad.dtype == torch.float32
        torch.testing.assert_close(x1.grad, x2.grad, rtol=2 ** -7, atol=2 ** -7 * float(x2.grad.abs().max()))
        for a, b in ((l1.weight.grad, l2.weight.grad), (l1.bias.grad, l2.bias.grad)):
            torch.testing.assert_close(a, b, rtol=2 ** -7, atol=2 ** -7 * float(b.abs().max()))
        # against f64: ours must be at least as close as autocast's
        ref = (g.to(torch.bfloat16).double().t() @ x.to(torch.bfloat16).double())
        assert float((l1.weight.grad.double() - ref).abs().max()) <= float((l2.weight.grad.double() - ref).abs().max()) * 1.01 + 1e-6
    opt = torch.optim.SGD(l1.parameters(), lr=0.5)
    opt.step()                                            # in-place update: the arena copy is stale now and must not be used
    _ops.ensure_cast_arena(l1, torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = l1(x)
    xq, wq, bq = x.to(torch.bfloat16).double(), l1.weight.detach().to(torch.bfloat16).double(), l1.bias.detach().to(torch.bfloat16).double()
    exact, mag = xq @ wq.t() + bq, xq.abs() @ wq.abs().t() + bq.abs()
    assert bool(((y.double() - exact).abs() <= 2.0 ** -8 * exact.abs() + 2.0 ** -20 * mag).all())


@pytest.mark.parametrize("opt_kind", ["adamw_fused", "adamw_foreach", "sgd_fused"])
def test_cast_arena_follows_the_optimizer(dev, opt_kind):
    """The 16-bit parameter copies must be the CURRENT master weights in every forward pass.  torch's fused optimizers update the
    parameters without moving their version counters (round 3: the arena served the weights of step 0 for a whole run), so the
    arena is invalidated by a global optimizer post-step hook and re-cast in every pass under autograd.  Checked: after each step
    + forward every arena view equals a fresh cast, and the loss trajectory equals the one without the arena to 16-bit rounding."""
    from pygho_amd import _ops, synth
    from pygho_amd.ngnn import SpModel
    dd = synth.to_datadict(synth.make_batch(64, "zinc", seed=11), dev)

    def run(use_arena):
        saved = _ops.USE_CAST_ARENA
        _ops.USE_CAST_ARENA = use_arena
        try:
            torch.manual_seed(0)
            model = SpModel(1, 3, 64, act_dtype=torch.bfloat16).to(dev)
            params = list(model.parameters())
            opt = {"adamw_fused": lambda: torch.optim.AdamW(params, lr=3e-3, fused=True),
                   "adamw_foreach": lambda: torch.optim.AdamW(params, lr=3e-3, foreach=True),
                   "sgd_fused": lambda: torch.optim.SGD(params, lr=3e-2, momentum=0.9, fused=True)}[opt_kind]()
            losses = []
            for _ in range(8):
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    pred = model(dd)
                if use_arena:
                    arena = model.__dict__["_pygho_cast_arena"]
                    assert all(torch.equal(v, p.detach().to(torch.bfloat16)) for v, p in zip(arena.views, arena.params)), "stale 16-bit copy"
                loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            return losses
        finally:
            _ops.USE_CAST_ARENA = saved

    with_arena, without = run(True), run(False)
    assert with_arena[-1] < 0.8 * with_arena[0], f"the model does not learn: {with_arena}"
    np.testing.assert_allclose(with_arena, without, rtol=0.05, atol=0.02)


def test_cast_arena_sees_updates_that_bypass_version_counters(dev):
    """parameter updates through a `.data` alias (EMA, weight tying, manual SGD on .data, a broadcast into .data) move
    neither the parameter's version counter nor the optimizer hook; the lazily refreshed arena of round 4 then served STALE 16-bit
    weights in training forwards, silently.  Every forward now re-casts (one multi-tensor launch): the output follows the update."""
    from pygho_amd import _ops, synth
    from pygho_amd.ngnn import SpModel
    assert not _ops.ARENA_LAZY
    dd = synth.to_datadict(synth.make_batch(32, "zinc", seed=5), dev)
    torch.manual_seed(0)
    model = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev)
    model.eval()

    def fwd(grad):
        with torch.set_grad_enabled(grad), torch.autocast("cuda", dtype=torch.bfloat16):
            return model(dd).float().detach().clone()
    for grad in (True, False):
        before = fwd(grad)
        versions = [p._version for p in model.parameters()]
        with torch.no_grad():
            for p in model.parameters():
                p.data.mul_(0.5)                                       # behind every counter and hook
        assert [p._version for p in model.parameters()] == versions
        after = fwd(grad)
        arena = model.__dict__["_pygho_cast_arena"]
        assert all(torch.equal(v, p.detach().to(torch.bfloat16)) for v, p in zip(arena.views, arena.params)), "stale 16-bit copy"
        assert not torch.equal(before, after)
        saved = _ops.USE_CAST_ARENA
        try:
            _ops.USE_CAST_ARENA = False
            want = fwd(grad)
        finally:
            _ops.USE_CAST_ARENA = saved
        assert torch.equal(after, want)


def test_backward_of_a_graph_recorded_before_a_parameter_update_raises(dev):
    """the per-forward re-cast wrote through aliases with their own version counters, so `forward, optimizer step, forward,
    backward of the FIRST graph` silently differentiated the first graph with the NEW 16-bit weights (stock PyTorch raises the
    in-place-modification error there).  Copies of parameters known to have changed are now rewritten through the views autograd
    saved: that backward raises; two grad-enabled forwards on UNCHANGED parameters before one backward (siamese use) stay legal."""
    from pygho_amd import synth
    from pygho_amd.ngnn import SpModel
    dd = synth.to_datadict(synth.make_batch(24, "zinc", seed=6), dev)
    torch.manual_seed(0)
    model = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)

    def loss():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), model(dd).float())
    la, lb = loss(), loss()                      # unchanged parameters: the second forward's re-cast must not invalidate the first graph
    (la + lb).backward()
    first = loss()
    opt.zero_grad(set_to_none=True)
    loss().backward()
    opt.step()
    loss()                                       # re-casts the updated parameters
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        first.backward()


def _three_dim_pattern(rng, shape, dims, nnz):
    rows = set()
    a, b = dims
    while len(rows) < nnz:
        t = [int(rng.integers(0, s)) for s in shape]
        if rng.random() < 0.4:
            t[b] = t[a]                                  # plenty of entries on the partial diagonal, several per i
        rows.add(tuple(t))
    return np.ascontiguousarray(np.array(sorted(rows), dtype=np.int64).T)


@pytest.mark.parametrize("dims,shape", [((0, 1), (9, 9, 6)), ((1, 2), (5, 8, 8)), ((0, 2), (7, 4, 7))])
def test_partial_diag_to_dense(dev, dims, shape):
    """SparseTensor.diag over SOME sparse dims with a dense result (reference SpTensor.py:337-352: that branch raises TypeError at
    :346, so no reference output exists -- the documented intent is restated in oracle.np_oracle.sp_diag_partial_to_dense): every
    entry whose coordinates in `dims` coincide lands at its kept coordinates, zero elsewhere; the gradient is the gather back."""
    from oracle import np_oracle as O
    from pygho_amd import SparseTensor
    rng = np.random.default_rng(31)
    ind = _three_dim_pattern(rng, shape, dims, 120)
    val = rng.standard_normal((ind.shape[1], 4)).astype(np.float32)
    v = torch.from_numpy(val).to(dev).requires_grad_(True)
    X = SparseTensor(torch.from_numpy(ind).to(dev), v, list(shape) + [4], is_coalesced=True)
    got = X.diag(list(dims))
    exp = O.sp_diag_to_dense(ind, val, list(shape) + [4], list(dims))
    assert tuple(got.shape) == exp.shape and np.array_equal(got.detach().cpu().numpy(), exp)
    w = torch.randn_like(got)
    (got * w).sum().backward()
    on = np.all(ind[list(dims)] == ind[dims[0]], axis=0)
    keep = [i for i in range(3) if i not in dims[1:]]
    gexp = np.zeros_like(val)
    gexp[on] = w.cpu().numpy()[tuple(ind[k][on] for k in keep)]
    assert np.array_equal(v.grad.cpu().numpy(), gexp)


def test_two_forwards_before_one_backward_with_the_cast_arena(dev):
    """loss = f(model(b1)) + f(model(b2)) (siamese / contrastive use, or a grad-enabled validation pass between forward and
    backward): autograd has saved the arena's 16-bit views in the first pass, so the second pass's re-cast must not move their
    version counters -- backward would raise 'modified by an inplace operation'.  (Round 5: every forward re-casts, through aliases
    with their own version counters; with unchanged parameters it writes the same bits.)  The gradient equals the sum of the two
    single-pass gradients."""
    from pygho_amd import synth
    from pygho_amd.ngnn import SpModel
    torch.manual_seed(0)
    model = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev)
    model.eval()                                    # running statistics: the two passes are independent of each other
    d1 = synth.to_datadict(synth.make_batch(24, "zinc", seed=3), dev)
    d2 = synth.to_datadict(synth.make_batch(24, "zinc", seed=4), dev)

    def loss_of(dd):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return model(dd).float().square().mean()

    singles = []
    for dd in (d1, d2):
        model.zero_grad(set_to_none=True)
        loss_of(dd).backward()
        singles.append([p.grad.clone() for p in model.parameters() if p.grad is not None])
    model.zero_grad(set_to_none=True)
    arena = model.__dict__["_pygho_cast_arena"]
    versions = [v._version for v in arena.views]
    l1 = loss_of(d1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model(d2)                                   # a grad-enabled pass whose result is dropped
    l2 = loss_of(d2)
    assert [v._version for v in arena.views] == versions, "the re-cast moved the version counters of views autograd has saved"
    (l1 + l2).backward()
    both = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(both) == len(singles[0]) == len(singles[1])
    for g, a, b in zip(both, *singles):
        torch.testing.assert_close(g, a + b, rtol=1e-4, atol=1e-5 * float((a + b).abs().max()) + 1e-7)


def test_batchnorm_step_counters_are_bumped_once_per_layer_call(dev):
    """_ops.deferred_batch_counters: the counters collected during a forward pass move by exactly one per BatchNorm call (one
    multi-tensor add at exit), like nn.BatchNorm1d's own `num_batches_tracked += 1`; nothing moves in eval mode."""
    from pygho_amd import synth
    from pygho_amd.ngnn import SpModel
    model = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev)
    dd = synth.to_datadict(synth.make_batch(16, "zinc", seed=2), dev)
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    assert bns and all(int(b.num_batches_tracked) == 0 for b in bns)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model(dd)
        model(dd)
    assert all(int(b.num_batches_tracked) == 2 for b in bns)
    model.eval()
    with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        model(dd)
    assert all(int(b.num_batches_tracked) == 2 for b in bns)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pair_product_matches_elementwise_chain(dev, dtype):
    """tuple initialisation left[root] * right[node] * val as one three-operand kernel vs the unpooling + elementwise
    chain of example/minimal.py:62-67: forward bit-exact in f32, gradients against torch autograd of the chain."""
    from pygho_amd import _ops, synth
    hb = synth.make_batch(96, "zinc", seed=4)
    n, d = hb.num_nodes, 128
    row, col = T(hb.tupleid[0], dev), T(hb.tupleid[1], dev)
    torch.manual_seed(2)
    mk = lambda r: torch.randn(r, d, device=dev).to(dtype)
    left, right, val, w = mk(n), mk(n), mk(hb.num_tuples), torch.randn(hb.num_tuples, d, device=dev)
    a = [t.clone().requires_grad_(True) for t in (left, right, val)]
    out = _ops.pair_product(a[0], a[1], a[2], row, col)
    (out.float() * w).sum().backward()
    b = [t.double().clone().requires_grad_(True) for t in (left, right, val)]
    ref = b[0][row] * b[1][col] * b[2]
    (ref * w.double()).sum().backward()
    if dtype == torch.float32:
        assert torch.equal(out, left[row] * right[col] * val)
        for x, y in zip(a, b):
            s = float(y.grad.abs().max())
            torch.testing.assert_close(x.grad.double() / s, y.grad / s, rtol=0, atol=1e-6)
    else:
        torch.testing.assert_close(out.double(), ref, rtol=2.0 ** -7, atol=1e-6)
        for x, y in zip(a, b):
            s = float(y.grad.abs().max())
            torch.testing.assert_close(x.grad.double() / s, y.grad / s, rtol=0, atol=2.0 ** -7)
    # the forward runs on the unit-segment kernel: same values as through the segment machinery
    _ops.USE_UNIT_TRIPLE = False
    try:
        assert torch.equal(out, _ops.pair_product(left, right, val, row, col))
    finally:
        _ops.USE_UNIT_TRIPLE = True
    # generic width (d = 5), f32
    l5, r5, v5 = left.float()[:, :5].contiguous(), right.float()[:, :5].contiguous(), val.float()[:, :5].contiguous()
    assert torch.equal(_ops.pair_product(l5, r5, v5, row, col), l5[row] * r5[col] * v5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pair_product_with_embedding_table(dev, dtype):
    """left[root] * right[node] * table[feature]: the embedding lookup of the tuple feature inside the product kernel;
    the table gradient (16 very long segments) goes through the chunked f32 hierarchy."""
    from pygho_amd import _ops, synth
    hb = synth.make_batch(96, "zinc", seed=4)
    n, d = hb.num_nodes, 128
    row, col = T(hb.tupleid[0], dev), T(hb.tupleid[1], dev)
    feat = T(hb.tuplefeat.reshape(-1), dev)
    torch.manual_seed(3)
    mk = lambda r: torch.randn(r, d, device=dev).to(dtype)
    left, right, table, w = mk(n), mk(n), mk(16), torch.randn(hb.num_tuples, d, device=dev)
    a = [t.clone().requires_grad_(True) for t in (left, right, table)]
    out = _ops.pair_product(a[0], a[1], a[2], row, col, feat)
    (out.float() * w).sum().backward()
    b = [t.double().clone().requires_grad_(True) for t in (left, right, table)]
    ref = b[0][row] * b[1][col] * b[2][feat]
    (ref * w.double()).sum().backward()
    if dtype == torch.float32:
        assert torch.equal(out, left[row] * right[col] * table[feat])
    else:
        torch.testing.assert_close(out.double(), ref, rtol=2.0 ** -7, atol=1e-6)
    for x, y, name in zip(a, b, ("left", "right", "table")):
        s = float(y.grad.abs().max())
        tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
        torch.testing.assert_close(x.grad.double() / s, y.grad / s, rtol=0, atol=tol, msg=name)
    assert torch.equal(out, _ops.pair_product(left, right, table[feat].contiguous(), row, col))   # same kernel, same rounding


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [128, 64, 96])
def test_pair_product_one_pass_backward_equals_three_pass(dev, dtype, d):
    """pygho_pair_bwd (one pass over the output gradient on symmetric tuple sets) against the three launches of
    pygho_seg_triple_product it replaces: the node-level gradients bit for bit (same products, same summation order), the table
    gradient to f32 summation-order tolerance and against autograd in f64; the mirror index of the K-hop tuple set is an involution;
    an ASYMMETRIC tuple set (one tuple dropped) is detected and keeps the three-launch path."""
    from pygho_amd import _ops, synth
    hb = synth.make_batch(64, "zinc", seed=9)
    n = hb.num_nodes
    row, col = T(hb.tupleid[0], dev), T(hb.tupleid[1], dev)
    feat = T(hb.tuplefeat.reshape(-1), dev)
    torch.manual_seed(5)
    mk = lambda r: torch.randn(r, d, device=dev).to(dtype)
    left, right, table, w = mk(n), mk(n), mk(16), torch.randn(hb.num_tuples, d, device=dev).to(dtype)

    def grads(flag, row=row, col=col, feat=feat, w=w):
        saved = _ops.USE_PAIR_BWD
        _ops.USE_PAIR_BWD = flag
        try:
            a = [t.clone().requires_grad_(True) for t in (left, right, table)]
            timer = _ops.LaunchTimer()
            with timer:
                out = _ops.pair_product(a[0], a[1], a[2], row, col, feat)
                out.backward(w)
            torch.cuda.synchronize()
            return [t.grad for t in a], set(timer.summary())
        finally:
            _ops.USE_PAIR_BWD = saved

    (gl1, gr1, gt1), tags1 = grads(True)
    (gl3, gr3, gt3), tags3 = grads(False)
    assert any(k.startswith("pair_bwd") for k in tags1) and not any(k.startswith("pair_bwd") for k in tags3)
    assert torch.equal(gl1, gl3), "g_left must be bit-identical to the by-row reduction"
    assert torch.equal(gr1, gr3), "g_right must be bit-identical to the by-column reduction"
    s = float(gt3.float().abs().max())
    torch.testing.assert_close(gt1.float() / s, gt3.float() / s, rtol=0, atol=2.0 ** -7)
    assert float(gt1[4:].abs().max()) == 0.0                                   # features 0..3 only: the other table rows get nothing
    b = [t.double().clone().requires_grad_(True) for t in (left, right, table)]
    (b[0][row] * b[1][col] * b[2][feat]).backward(w.double())
    for got, ref in ((gl1, b[0].grad), (gr1, b[1].grad), (gt1, b[2].grad)):
        sc = float(ref.abs().max())
        torch.testing.assert_close(got.double() / sc, ref / sc, rtol=0, atol=2.0 ** -7)
    r32, c32, f32 = _ops.narrow_i32(row), _ops.narrow_i32(col), _ops.narrow_i32(feat)
    mir = _ops.pair_mirror(r32, c32, f32, n)
    assert mir is not None and torch.equal(mir[mir.long()], torch.arange(hb.num_tuples, device=dev, dtype=torch.int32))
    assert torch.equal(r32[mir.long()], c32) and torch.equal(c32[mir.long()], r32)
    # asymmetric: drop one off-diagonal tuple
    keep = torch.ones(hb.num_tuples, dtype=torch.bool, device=dev)
    keep[int((row != col).nonzero()[0])] = False
    row2, col2, feat2, w2 = row[keep].contiguous(), col[keep].contiguous(), feat[keep].contiguous(), w[keep].contiguous()
    assert _ops.pair_mirror(_ops.narrow_i32(row2), _ops.narrow_i32(col2), _ops.narrow_i32(feat2), n) is None
    (gl, gr, gt), tags = grads(True, row2, col2, feat2, w2)
    assert not any(k.startswith("pair_bwd") for k in tags)
    b = [t.double().clone().requires_grad_(True) for t in (left, right, table)]
    (b[0][row2] * b[1][col2] * b[2][feat2]).backward(w2.double())
    for got, ref in ((gl, b[0].grad), (gr, b[1].grad), (gt, b[2].grad)):
        sc = float(ref.abs().max())
        torch.testing.assert_close(got.double() / sc, ref / sc, rtol=0, atol=2.0 ** -7)


@pytest.mark.parametrize("kind", ["zinc", "i2"])
def test_device_collate_bit_exact(dev, kind):
    """on-device mini-batch collation from the int32 graph store == host block-diagonal collate (hodata/SpData.py:56-112
    increments) for an arbitrary selection with repeats; the collated plan drives the same spspmm result."""
    from pygho_amd import SparseTensor, synth
    from pygho_amd.collate import DeviceGraphStore
    rng = np.random.default_rng(5)
    recs = [synth.make_graph(rng, kind) for _ in range(24 if kind == "zinc" else 10)]
    store = DeviceGraphStore(recs, dev)
    for sel in ([3, 0, 7, 7, 1], list(range(len(recs))), [len(recs) - 1]):
        got = store.collate(sel)
        ref = synth.to_datadict(synth.collate([recs[i] for i in sel]), dev, kind)
        for k, v in ref.items():
            g = got[k]
            if isinstance(v, SparseTensor):
                assert torch.equal(g.indices, v.indices) and torch.equal(g.values, v.values) and tuple(g.shape) == tuple(v.shape), k
            elif torch.is_tensor(v):
                assert v.dtype == g.dtype and torch.equal(g, v), k
            else:
                assert g == v, k
        assert set(got.keys()) == set(ref.keys())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", ["silu", "relu", "none"])
def test_activation_on_load_bit_identical(dev, dtype, act):
    """aggregation with BatchNorm scale / shift + activation applied to one operand as it is loaded == fused BatchNorm+act
    kernel followed by the plain aggregation, bit for bit, on either operand side, with residual and mean scaling."""
    from pygho_amd import _ops, synth
    from pygho_amd._native import check, dtype_code, lib, ptr, stream_ptr
    hb = synth.make_batch(64, "zinc", seed=8)
    acd = T(hb.acd["X___X___1___A___0"], dev)
    nt, ne, d = hb.num_tuples, hb.num_edges, 64
    torch.manual_seed(6)
    Y = (torch.randn(nt, d, device=dev) * 1.3).to(dtype)
    R, G = torch.randn(nt, d, device=dev).to(dtype), torch.randn(nt, d, device=dev).to(dtype)
    A = torch.randn(ne, d, device=dev).to(dtype)
    scale, shift = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev) * 0.3
    H = torch.empty_like(Y)
    check(lib().pygho_bn_act_fwd(ptr(H), ptr(Y), ptr(scale), ptr(shift), nt, d, _ops.ACT_CODE[act], dtype_code(Y), stream_ptr(dev)), "bn_act_fwd")
    plan = _ops.message_plan(acd, nt, nt, ne)
    for aggr in ("sum", "mean"):
        ref = _ops.seg_gmr(nt, H, A, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, aggr, addend=R)
        got = _ops.seg_gmr(nt, Y, A, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, aggr, addend=R, act=(scale, shift, act, 1))
        assert torch.equal(ref, got), aggr
    p, a_g, c_g = plan.by_d()
    inv = plan.fwd.inv_count
    for rs in (None, inv):
        ref = _ops.seg_gmr(ne, G, H, p.seg_ptr, a_g, c_g, "sum", rs)
        got = _ops.seg_gmr(ne, G, Y, p.seg_ptr, a_g, c_g, "sum", rs, act=(scale, shift, act, 2))
        assert torch.equal(ref, got)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [8, 128, 24])
def test_pair_gather_combine(dev, dtype, d):
    """sparse recombination out[t] = (base[t] + row_term[i_t]) + col_term[j_t] with the diagonal tuples' term added / substituted,
    every operand optional, ragged tails (row counts that do not fill the last workgroup)."""
    from pygho_amd import _ops
    gen = torch.Generator().manual_seed(d)
    n, nnz = 37, 1003
    ri = torch.randint(0, n, (nnz,), generator=gen)
    ci = torch.randint(0, n, (nnz,), generator=gen)
    ci[::7] = ri[::7]                                            # plenty of diagonal tuples
    base = torch.randn((nnz, d), generator=gen).to(dtype)
    rt, ct, dg = (torch.randn((n, d), generator=gen).to(dtype) for _ in range(3))
    ri32, ci32 = ri.to(torch.int32).to(dev), ci.to(torch.int32).to(dev)
    if (d * base.element_size()) % 16 != 0:
        assert not _ops.pair_gather_supported(base.to(dev))
        with pytest.raises(RuntimeError, match="16-byte"):
            _ops.pair_gather_combine(base.to(dev), None, None, None, False, ri32, ci32, d, dtype, dev)
        return
    on_diag = (ri == ci)[:, None]
    for use in ((1, 1, 1, 1), (0, 1, 1, 1), (1, 0, 0, 1), (1, 1, 0, 0), (0, 0, 1, 0)):
        for replace in (False, True):
            args = [t.to(dev) if u else None for t, u in zip((base, rt, ct, dg), use)]
            out = _ops.pair_gather_combine(*args, replace, ri32, ci32, d, dtype, dev)
            acc = base.float() if use[0] else torch.zeros((nnz, d))
            if use[1]:
                acc = acc + rt.float()[ri]
            if use[2]:
                acc = acc + ct.float()[ci]
            if use[3]:
                acc = torch.where(on_diag, dg.float()[ri], acc) if replace else torch.where(on_diag, acc + dg.float()[ri], acc)
            assert torch.equal(out.cpu(), acc.to(dtype)), (use, replace)


def test_sparse_pair_views_gradient_one_pass(dev):
    """(diagonal rows, per-i sums, per-j sums) of a sparse 2-D representation and their joint gradient against autograd through
    index_add / indexing on the CPU."""
    from pygho_amd import _ops
    from pygho_amd.backend.SpTensor import indicehash
    gen = torch.Generator().manual_seed(2)
    n, d = 23, 8
    dense_mask = torch.rand((n, n), generator=gen) > 0.6
    dense_mask[5, 5] = False                                     # a node without its diagonal tuple
    dense_mask[torch.arange(0, n, 2), torch.arange(0, n, 2)] = True
    ri, ci = dense_mask.nonzero(as_tuple=True)
    nnz = ri.numel()
    x = torch.randn((nnz, d), generator=gen)
    ind = torch.stack((ri, ci)).to(dev)
    diag_idx = torch.arange(n, device=dev)
    pos = _ops.sorted_match(indicehash(ind), indicehash(diag_idx.reshape(1, -1).expand(2, -1).contiguous()))
    xd = x.to(dev).requires_grad_(True)
    dg, s_r, s_c = _ops.sparse_pair_views(xd, ind[0].contiguous(), ind[1].contiguous(), pos, n)
    ws = [torch.randn((n, d), generator=gen) for _ in range(3)]
    ((dg * ws[0].to(dev)).sum() + (s_r * ws[1].to(dev)).sum() + (s_c * ws[2].to(dev)).sum()).backward()
    xr = x.clone().requires_grad_(True)
    rdg = torch.zeros((n, d)).index_add(0, ri[ri == ci], xr[ri == ci])
    r_r = torch.zeros((n, d)).index_add(0, ri, xr)
    r_c = torch.zeros((n, d)).index_add(0, ci, xr)
    ((rdg * ws[0]).sum() + (r_r * ws[1]).sum() + (r_c * ws[2]).sum()).backward()
    for got, ref in ((dg, rdg), (s_r, r_r), (s_c, r_c)):
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-6, atol=1e-6)


def test_collated_message_plans_equal_sorted_ones(dev):
    """DeviceGraphStore.collate installs the batch's message plan assembled from per-graph groupings (permutations + message
    offsets, per-row counts scanned into CSR pointers): every array equals the plan built from the collated triples by
    sorting, for a key whose operands are (tuples, tuples, edges) and one with (tuples, edges, tuples), repeated / shuffled ids."""
    from pygho_amd import _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    keys = ("X___X___1___A___0", "X___A___1___X___0")
    rng = np.random.default_rng(8)
    recs = [synth.make_graph(rng, "zinc", 3, keys) for _ in range(40)]
    store = DeviceGraphStore(recs, dev)
    ids = torch.tensor([5, 3, 3, 39, 0, 17, 21, 8, 8, 8, 30])
    dd = store.collate(ids)
    for key in keys:
        acd = dd[key + "___acd"]
        roles = synth.parse_key(key)
        n = {"X": dd["X"].nnz, "A": dd["A"].nnz}
        n_out, n_lhs, n_rhs = n[roles[0][0]], n[roles[1][0]], n[roles[3][0]]
        got = _ops.message_plan(acd, n_out, n_lhs, n_rhs)                      # the installed one
        assert got is acd._pygho_plans[("msg", n_out, n_lhs, n_rhs, acd._version)]
        ref = _ops.MessagePlan(acd, n_out, n_lhs, n_rhs)                       # built by sorting
        assert torch.equal(got.fwd.seg_ptr, ref.fwd.seg_ptr) and got.fwd.perm is None and ref.fwd.perm is None
        assert torch.equal(got.c_fwd, ref.c_fwd) and torch.equal(got.d_fwd, ref.d_fwd)
        for a, b in ((got.by_c(), ref.by_c()), (got.by_d(), ref.by_d())):
            assert torch.equal(a[0].seg_ptr, b[0].seg_ptr) and torch.equal(a[0].perm, b[0].perm)
            assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        # the by-edge gradient's scatter plan: collated from the store's per-graph chunks == planned on the collated triples
        sp_got, sp_ref = _ops.scatter_plan(got), _ops.scatter_plan(ref)
        if key == "X___X___1___A___0":
            assert sp_got is not None and sp_got.covers
        if sp_got is not None:                                                # (the other key's blocks -- the graphs' tuple rows -- may exceed 255 rows)
            assert sp_ref is not None and sp_ref.covers == sp_got.covers
            assert sp_got.n_blocks == sp_ref.n_blocks == ids.numel() and sp_got.n_chunks == sp_ref.n_chunks
            assert sp_got.max_edges >= sp_ref.max_edges                       # the store's maximum over ALL its graphs
            for name in ("chunk0", "blk_e", "chunks", "words"):
                assert torch.equal(getattr(sp_got, name), getattr(sp_ref, name)), name


def test_deferred_index_range_check(dev):
    """inside `deferred_index_checks()` the range check of an unsorted grouping does not synchronise on its own: a bad index is
    still reported -- by the next host fetch or when the block ends -- and the plan built meanwhile stays in bounds; outside such a
    block the error is raised at the call."""
    from pygho_amd import _ops
    good = torch.tensor([3, 1, 2, 1, 0], device=dev)
    bad = torch.tensor([3, 9, 2, 1, 0], device=dev)
    with _ops.deferred_index_checks():
        p = _ops.plan_from_keys(good, 4, assume_sorted=False)
    assert p.seg_ptr.tolist() == [0, 1, 3, 4, 5] and p.perm.tolist() == [4, 1, 3, 2, 0]
    with pytest.raises(ValueError, match="out of range"):
        with _ops.deferred_index_checks():
            p = _ops.plan_from_keys(bad, 4, assume_sorted=False)                # no error yet ...
            assert sorted(p.perm.tolist()) == [0, 1, 2, 3, 4] and max(p.seg_ptr.tolist()) <= 5
    _ops.check_deferred_errors()                                   # ... reported once, when the block ended
    with pytest.raises(ValueError, match="out of range"):
        _ops.plan_from_keys(bad, 4, assume_sorted=False)           # outside a block: at the call


@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 256), (torch.float16, 256), (torch.float32, 128), (torch.float32, 256),
                                     (torch.bfloat16, 128)])
@pytest.mark.parametrize("aggr", ["sum", "mean"])
def test_window_kernel_bit_identical(dev, dtype, d, aggr):
    """`pygho_seg_gather_mul_reduce_window` (rhs rows served from an LDS window) against the gather-everything kernel: same
    products and summation order, so forward, residual form and both gradients are bit-identical — on the I2-shape plan (window =
    the edge rows of one or two graphs), on a plan whose rhs indices are spread over all rows (every pass falls back to global
    gathers) and against the oracle in f32."""
    from pygho_amd import _ops, synth
    hb = synth.replicate(synth.make_batch(32, "i2", seed=9), 4)
    acd_np = hb.acd["X___X___2___A___0"]
    nt, ne = hb.num_tuples, hb.num_edges
    assert nt >= 4096 and 2 * ne <= nt
    torch.manual_seed(0)
    xv = torch.randn(nt, d, device=dev).to(dtype)
    av = torch.randn(ne, d, device=dev).to(dtype)
    res = torch.randn(nt, d, device=dev).to(dtype)
    g = torch.randn(nt, d, device=dev).to(dtype)
    rng = np.random.default_rng(3)
    spread = acd_np.copy()
    spread[2] = rng.integers(0, ne, size=spread.shape[1])               # no locality: the window never fits
    saved = (_ops.USE_SEG_WINDOW, _ops.SEG_WINDOW_MIN_ROW_BYTES)
    out = {}
    try:
        _ops.SEG_WINDOW_MIN_ROW_BYTES = 256
        for name, plan_np in (("i2", acd_np), ("spread", spread)):
            for win in (True, False):
                _ops.USE_SEG_WINDOW = win
                acd = T(plan_np, dev)                                   # a new tensor per run: plans are cached on it
                plan = _ops.message_plan(acd, nt, nt, ne)
                x, a = xv.clone().requires_grad_(True), av.clone().requires_grad_(True)
                y = _ops.message_reduce(x, a, acd, nt, nt, ne, aggr)
                y.backward(g)
                y_res = _ops.seg_gmr(nt, xv, av, plan.fwd.seg_ptr, plan.c_fwd, plan.d_fwd, aggr, addend=res)
                out[(name, win)] = (y.detach(), x.grad, a.grad, y_res)
    finally:
        _ops.USE_SEG_WINDOW, _ops.SEG_WINDOW_MIN_ROW_BYTES = saved
    for name in ("i2", "spread"):
        for got, want in zip(out[(name, True)], out[(name, False)]):
            assert torch.equal(got, want), name
    # directly against the oracle on the I2 plan (not only through the gather-everything kernel): f32 bit for bit; 16-bit
    # operands: the oracle on the operands' exact values in f32, the kernel's result within one rounding of the output type,
    # and both gradients (the two transposed plans, also through the window kernel) against the oracle's
    want = O.spspmm_values(N(xv.float()), N(av.float()), acd_np, nt, aggr)
    got = out[("i2", True)][0]
    if dtype == torch.float32:
        assert np.array_equal(N(got), want)
    else:
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        np.testing.assert_allclose(N(got.float()), want, rtol=ulp, atol=ulp * np.abs(want).max() * 2.0 ** -6)
    gx_want, ga_want = O.spspmm_values_grad(N(xv.float()), N(av.float()), acd_np, nt, aggr, N(g.float()))
    tol = dict(rtol=1e-5, atol=1e-4) if dtype == torch.float32 else dict(rtol=2.0 ** -7, atol=2.0 ** -7 * max(np.abs(gx_want).max(), np.abs(ga_want).max()) * 0.25)
    np.testing.assert_allclose(N(out[("i2", True)][1].float()), gx_want, **tol)
    np.testing.assert_allclose(N(out[("i2", True)][2].float()), ga_want, **tol)


@pytest.mark.parametrize("dtype,d", [(torch.bfloat16, 128), (torch.float16, 64)])
@pytest.mark.parametrize("aggr", ["sum", "mean"])
def test_window_kernel_on_the_by_edge_backward_plan(dev, dtype, d, aggr):
    """the gradient of spspmm's second operand on the ZINC 2-tuple plan (segments = edges, BOTH operands tuple-level rows spread
    over their graph) takes the LDS-window kernel with few segments per pass: bit-identical to the gather-everything kernel
    (module switch), and the whole backward against the oracle's gradient."""
    from pygho_amd import _ops, synth
    hb = synth.make_batch(300, "zinc", seed=12)
    key = "X___X___1___A___0"
    acd_np = hb.acd[key]
    nt, ne = hb.num_tuples, hb.num_edges
    assert ne >= 4096 and nt > 2 * ne
    torch.manual_seed(0)
    xv = torch.randn(nt, d, device=dev).to(dtype)
    av = torch.randn(ne, d, device=dev).to(dtype)
    g = torch.randn(nt, d, device=dev).to(dtype)
    saved = _ops.USE_SEG_WINDOW_BY_EDGE
    out = {}
    try:
        for win in (True, False):
            _ops.USE_SEG_WINDOW_BY_EDGE = win
            acd = T(acd_np, dev)
            plan = _ops.message_plan(acd, nt, nt, ne)
            p, a_g, c_g = plan.by_d()
            scale = plan.fwd.inv_count if aggr == "mean" else None
            assert _ops._window_eligible(ne, g, xv, c_g, "sum") == win
            x, a = xv.clone().requires_grad_(True), av.clone().requires_grad_(True)
            _ops.message_reduce(x, a, acd, nt, nt, ne, aggr).backward(g)
            out[win] = (_ops.seg_gmr(ne, g, xv, p.seg_ptr, a_g, c_g, "sum", scale), a.grad, x.grad)
    finally:
        _ops.USE_SEG_WINDOW_BY_EDGE = saved
    for got, want in zip(out[True], out[False]):
        assert torch.equal(got, want)
    assert torch.equal(out[True][0], out[True][1])
    _gx, ga_want = O.spspmm_values_grad(N(xv.float()), N(av.float()), acd_np, nt, aggr, N(g.float()))
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    np.testing.assert_allclose(N(out[True][1].float()), ga_want, rtol=ulp, atol=ulp * np.abs(ga_want).max() * 0.25)


def test_window_kernel_long_and_empty_segments(dev):
    """the LDS-window kernel on a plan with empty segments and segments far longer than the staged index capacity (indices then
    come from global memory), a 64-row rhs (an embedding table: the window is the whole operand), scaled lhs rows and a residual
    addend — bit-identical to the gather-everything kernel."""
    from pygho_amd import _ops
    rng = np.random.default_rng(5)
    n_seg, n_lhs, n_rhs, d = 6000, 10_000, 64, 256
    lens = rng.integers(0, 400, size=n_seg)
    lens[rng.random(n_seg) < 0.2] = 0
    lens[17] = 5000
    ptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    m = int(ptr[-1])
    li = T(rng.integers(0, n_lhs, size=m).astype(np.int32), dev)
    ri = T(rng.integers(0, n_rhs, size=m).astype(np.int32), dev)
    seg_ptr = T(ptr, dev)
    torch.manual_seed(1)
    lhs = torch.randn(n_lhs, d, device=dev).to(torch.bfloat16)
    rhs = torch.randn(n_rhs, d, device=dev).to(torch.bfloat16)
    add = torch.randn(n_seg, d, device=dev).to(torch.bfloat16)
    scale = torch.rand(n_lhs, device=dev)
    saved = _ops.USE_SEG_WINDOW
    out = {}
    try:
        for win in (True, False):
            _ops.USE_SEG_WINDOW = win
            out[win] = (_ops.seg_gmr(n_seg, lhs, rhs, seg_ptr, li, ri, "sum"),
                        _ops.seg_gmr(n_seg, lhs, rhs, seg_ptr, li, ri, "mean", addend=add),
                        _ops.seg_gmr(n_seg, lhs, rhs, seg_ptr, li, ri, "sum", scale, addend=add))
    finally:
        _ops.USE_SEG_WINDOW = saved
    assert _ops._window_eligible(n_seg, lhs, rhs, ri, "sum")
    for got, want in zip(out[True], out[False]):
        assert torch.equal(got, want)
    ref = torch.zeros(n_seg, d, device=dev).index_add_(0, torch.repeat_interleave(torch.arange(n_seg, device=dev), T(lens, dev)),
                                                          lhs.float()[li.long()] * rhs.float()[ri.long()])
    torch.testing.assert_close(out[True][0].float(), ref, rtol=2e-2, atol=2e-1)


def test_debug_index_validation_and_roctx_ranges(dev):
    """PYGHO_DEBUG=1 validates the index arrays of a segment launch before it (the kernels trust them); PYGHO_ROCTX=1 brackets
    every C-ABI launch with a roctx range.  Both are process-wide switches read at import: exercised in a child process."""
    import subprocess, sys, os
    code = r'''
import torch
from pygho_amd import _ops, _native
dev = torch.device("cuda:0")
x = torch.randn(10, 8, device=dev)
ptr = torch.tensor([0, 2, 3], dtype=torch.int32, device=dev)
ok = torch.tensor([0, 9, 5], dtype=torch.int32, device=dev)
out = _ops.seg_gmr(2, x, None, ptr, ok, None, "sum")
assert torch.allclose(out, torch.stack((x[0] + x[9], x[5])))
assert type(_native.lib()).__name__ == "_RangedLib"
bad = torch.tensor([0, 10, 5], dtype=torch.int32, device=dev)
try:
    _ops.seg_gmr(2, x, None, ptr, bad, None, "sum")
except IndexError as e:
    print("caught", e)
else:
    raise SystemExit("out-of-range index not caught")
'''
    env = dict(os.environ, PYGHO_DEBUG="1", PYGHO_ROCTX="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "caught" in r.stdout, r.stderr[-2000:]


def test_xcc_ids_reports_a_valid_xcd_per_workgroup(dev):
    """pygho_xcc_ids (diagnostic): every workgroup reports an XCD in 0..7 and the launch covers all eight; the b % 8 placement the
    segment kernels exploit for L2 locality is a speed assumption, reported by bench.py (`xcd_dispatch`), not asserted here."""
    from pygho_amd._native import check, lib, ptr, stream_ptr
    ids = torch.full((1024,), -1, dtype=torch.int32, device=dev)
    check(lib().pygho_xcc_ids(ptr(ids), 1024, stream_ptr(dev)), "xcc_ids")
    assert int(ids.min()) >= 0 and int(ids.max()) <= 7 and ids.unique().numel() == 8


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,d,n_table", [(1, 8, 1), (777, 64, 4), (100_003, 128, 32), (40_000, 256, 64), (5_000, 100, 7),
                                         (1_200_000, 128, 16), (3_000, 300, 5), (1_000, 33, 5), (2_000, 1, 3), (700, 254, 32),
                                         (900, 256, 33)])
def test_table_grad_matches_index_add(dev, dtype, m, d, n_table):
    """plan-free gradient of a lookup into a small table (csrc/table_grad.hip; autograd of SpTensor.py:476 / the embeddings of
    example/minimal.py:22-34) == index_add in f64 within f32 accumulation error, == the planned path (sort + chunk hierarchy),
    run twice bit-identical; the autograd route of `gather_rows` takes it."""
    from pygho_amd import _ops
    gen = torch.Generator(device=dev).manual_seed(m + d)
    g = torch.randn(m, d, device=dev, generator=gen).to(dtype)
    idx = torch.randint(0, n_table, (m,), device=dev, generator=gen)
    if n_table > 2:
        idx[idx == 1] = 0                                              # an empty table row
    assert _ops.table_grad_ok(g, n_table)
    got = _ops.table_grad(g, idx, n_table)
    assert got.dtype == dtype and torch.equal(got, _ops.table_grad(g, idx, n_table))
    ref = torch.zeros(n_table, d, dtype=torch.float64, device=dev).index_add_(0, idx, g.double())
    mag = torch.zeros(n_table, d, dtype=torch.float64, device=dev).index_add_(0, idx, g.double().abs())
    eps = {torch.float32: 2.0 ** -23, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    # f32 accumulation of <= m terms (error <= log-ish * 2^-24 * sum |g|, bounded loosely) + one rounding to the storage type
    assert bool(((got.double() - ref).abs() <= 64 * 2.0 ** -24 * mag + eps * ref.abs() + 1e-30).all())
    if n_table > 2:
        assert float(got[1].abs().max()) == 0.0
    table = torch.randn(n_table, d, device=dev).to(dtype).requires_grad_(True)
    _ops.gather_rows(table, idx).backward(g)
    assert torch.equal(table.grad, got)
    _ops.USE_TABLE_GRAD = False
    try:
        t2 = table.detach().clone().requires_grad_(True)
        _ops.gather_rows(t2, idx.clone()).backward(g)
    finally:
        _ops.USE_TABLE_GRAD = True
    # (rows that are not a multiple of 16 bytes: the planned path keeps its chunk partial sums in the storage type)
    slack = 4 * eps * mag if (d * g.element_size()) % 16 and dtype != torch.float32 else 0
    assert bool(((t2.grad.double() - got.double()).abs() <= 64 * 2.0 ** -23 * mag + 2 * eps * ref.abs() + slack + 1e-30).all())


def test_table_grad_reports_bad_index(dev):
    """an index outside the table is skipped by the kernel and reported by the deferred range check (next host fetch)"""
    from pygho_amd import _ops
    g = torch.ones(10, 8, device=dev)
    idx = torch.tensor([0, 1, 2, 9, 1, 0, 2, 2, 1, 0], device=dev)
    got = _ops.table_grad(g, idx, 3)
    assert got[:, 0].tolist() == [3.0, 3.0, 3.0]
    with pytest.raises(ValueError, match="out of range"):
        _ops.check_deferred_errors()


def test_collated_batch_needs_no_plan_building(dev):
    """DeviceGraphStore.collate also installs the groupings a model step asks for beyond the message plans -- nodes by graph,
    tuples by root (with their longest segments) and the mirror permutation of the symmetric tuple set -- and sizes its outputs
    from host-side lengths: neither collation nor a full SpModel step on the batch reads anything back from the device or builds a
    plan, and the step's loss / gradients are bit-identical to the same step on a batch whose plans are built the ordinary way."""
    from pygho_amd import _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    from pygho_amd.ngnn import SpModel
    key = "X___X___1___A___0"
    rng = np.random.default_rng(12)
    recs = [synth.make_graph(rng, "zinc", 3, (key,)) for _ in range(48)]
    store = DeviceGraphStore(recs, dev)
    torch.manual_seed(0)
    model = SpModel(1, 2, 64, act_dtype=torch.bfloat16).to(dev)
    ids = torch.tensor([7, 3, 3, 40, 0, 11, 29, 29, 18, 47, 5, 6])

    def step(dd):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(dd)
        loss = torch.nn.functional.l1_loss(dd["y"].unsqueeze(-1), pred.float())
        loss.backward()
        return loss.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]

    step(store.collate(torch.arange(4)))                               # warm-up: workspaces, cast arena
    torch.cuda.synchronize()
    items = []
    orig_item, orig_list = torch.Tensor.item, torch.Tensor.tolist
    torch.Tensor.item = lambda self: (items.append("item") if self.is_cuda else None, orig_item(self))[1]
    torch.Tensor.tolist = lambda self: (items.append("tolist") if self.is_cuda else None, orig_list(self))[1]
    f0 = _ops.FETCHES[0]
    try:
        dd = store.collate(ids)
        loss, grads = step(dd)
    finally:
        torch.Tensor.item, torch.Tensor.tolist = orig_item, orig_list
    assert _ops.FETCHES[0] == f0 and not items, (f"{_ops.FETCHES[0] - f0} planner fetches, {items} inside collate + step")
    # the installed groupings are the ones the planners build
    X = dd["X"]
    row, col, feat = X._row(0), X._row(1), _ops.flat_index(X.values)
    n = int(dd["num_nodes"])
    for keys_t, n_seg, tag in ((dd["batch"], int(dd["num_graphs"]), "scatter"), (row, n, "scatter"), (row, n, "pair-row")):
        got = _ops.cached_plan(keys_t, n_seg, tag)
        ref = _ops.plan_from_keys(keys_t.clone(), n_seg)
        assert got.perm is None and ref.perm is None and torch.equal(got.seg_ptr, ref.seg_ptr)
        assert got.max_len == ref.max_len
    mir = _ops.pair_mirror(_ops.narrow_i32(row), _ops.narrow_i32(col), _ops.narrow_i32(feat), n)
    ref = _ops.pair_mirror(_ops.narrow_i32(row.clone()), _ops.narrow_i32(col.clone()), _ops.narrow_i32(feat.clone()), n)
    assert mir is not None and ref is not None and torch.equal(mir, ref)
    # same step on the same batch with plans built the ordinary way
    ref_dd = synth.to_datadict(synth.collate([recs[i] for i in ids.tolist()]), dev, "zinc")
    loss2, grads2 = step(ref_dd)
    assert torch.equal(loss, loss2)
    for a, b in zip(grads, grads2):
        assert torch.equal(a, b)
    # a tuple set that is not symmetric: the verdict "no mirror" is installed too (no host read either)
    bad = [synth.make_graph(rng, "zinc", 3, (key,)) for _ in range(4)]
    bad[2].tuplefeat = bad[2].tuplefeat.copy()
    r, c = bad[2].tupleid
    off = np.nonzero(r != c)[0][0]
    bad[2].tuplefeat[off] = (bad[2].tuplefeat[off] + 1) % 4
    s2 = DeviceGraphStore(bad, dev)
    f0 = _ops.FETCHES[0]
    d2 = s2.collate([0, 2, 3])
    X2 = d2["X"]
    assert _ops.pair_mirror(_ops.narrow_i32(X2._row(0)), _ops.narrow_i32(X2._row(1)), _ops.narrow_i32(_ops.flat_index(X2.values)),
                            int(d2["num_nodes"])) is None
    assert _ops.pair_mirror(*(lambda d1: (_ops.narrow_i32(d1["X"]._row(0)), _ops.narrow_i32(d1["X"]._row(1)),
                                          _ops.narrow_i32(_ops.flat_index(d1["X"].values)), int(d1["num_nodes"])))(s2.collate([0, 3]))) is not None
    assert _ops.FETCHES[0] == f0
    step(d2)                                                            # the three-launch backward still trains


def test_table_grad_route_does_not_depend_on_history(dev):
    """policy of the small-table gradient (round 5): which route runs is a pure function of the call -- the plan-free kernel unless
    the CALLER built / installed a grouping of the index array AND the pattern is large -- never of how often a pattern was seen
    (round 4 planned a large recurring pattern "after 3 uses", so step 3 and step 4 of a resident-batch run differed bitwise and a
    resumed run diverged from a continuous one).  Eight uses of the same pattern: no plan appears, bit-identical gradients; an
    explicitly built plan switches a large pattern to the planned route, whose result agrees to f32 accumulation error."""
    from pygho_amd import _ops
    n_table, d = 16, 128
    for m in (_ops.TABLE_GRAD_PLAN_ROWS + 5, 4096):
        idx = torch.randint(0, n_table, (m,), device=dev)
        g = torch.randn(m, d, device=dev).to(torch.bfloat16)
        grads = []
        for use in range(8):
            table = torch.randn(n_table, d, device=dev).to(torch.bfloat16).requires_grad_(True)
            _ops.gather_rows(table, idx).backward(g)
            grads.append(table.grad)
            assert ("scatter", n_table, idx._version) not in getattr(idx, "_pygho_plans", {}), (m, use)
        assert all(torch.equal(grads[0], got) for got in grads[1:])
        _ops.cached_plan(idx, n_table, "scatter")                       # the caller's explicit plan
        table = torch.randn(n_table, d, device=dev).to(torch.bfloat16).requires_grad_(True)
        _ops.gather_rows(table, idx).backward(g)
        planned = table.grad
        ref = torch.zeros(n_table, d, dtype=torch.float64, device=dev).index_add_(0, idx, g.double())
        mag = torch.zeros(n_table, d, dtype=torch.float64, device=dev).index_add_(0, idx, g.double().abs())
        for got in (grads[0], planned):
            assert bool(((got.double() - ref).abs() <= 64 * 2.0 ** -24 * mag + 2.0 ** -8 * ref.abs() + 1e-30).all())
        if m < _ops.TABLE_GRAD_PLAN_ROWS:
            assert torch.equal(planned, grads[0])                       # small patterns stay on the plan-free kernel


def test_device_collate_edge_cases(dev):
    """collation of degenerate selections: an empty selection, a single graph, a graph without edges (no messages either) in the
    middle of a batch, ids given as list / numpy / CPU tensor / device tensor -- same arrays as the host collate, plans installed
    for every non-empty family."""
    from pygho_amd import SparseTensor, _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    key = "X___X___1___A___0"
    rng = np.random.default_rng(31)
    recs = [synth.make_graph(rng, "zinc", 3, (key,)) for _ in range(6)]
    lone = recs[2]
    iso = synth.GraphRecord(lone.num_nodes, lone.x, lone.edge_index[:, :0], lone.edge_attr[:0],
                            np.stack([np.arange(lone.num_nodes)] * 2).astype(np.int64), np.zeros(lone.num_nodes, dtype=np.int64),
                            {key: np.zeros((3, 0), dtype=np.int64)}, 0.5)             # isolated nodes: diagonal tuples only
    recs[2] = iso
    store = DeviceGraphStore(recs, dev)
    for sel in ([1, 2, 3], [2], [2, 2], np.asarray([5, 0]), torch.tensor([4, 2, 1]), torch.tensor([3, 3, 0], device=dev)):
        got = store.collate(sel)
        ids = [int(i) for i in (sel.tolist() if hasattr(sel, "tolist") else sel)]
        ref = synth.to_datadict(synth.collate([recs[i] for i in ids]), dev, "zinc")
        for k, v in ref.items():
            g = got[k]
            if isinstance(v, SparseTensor):
                assert torch.equal(g.indices, v.indices) and torch.equal(g.values, v.values), (k, ids)
            elif torch.is_tensor(v):
                assert torch.equal(g, v), (k, ids)
            else:
                assert g == v, (k, ids)
        acd = got[key + "___acd"]
        plan = _ops.message_plan(acd, got["X"].nnz, got["X"].nnz, got["A"].nnz)
        ref_plan = _ops.MessagePlan(ref[key + "___acd"], got["X"].nnz, got["X"].nnz, got["A"].nnz)
        assert torch.equal(plan.fwd.seg_ptr, ref_plan.fwd.seg_ptr)
        if acd.shape[1]:
            assert torch.equal(plan.by_c()[0].perm, ref_plan.by_c()[0].perm) and torch.equal(plan.by_d()[0].seg_ptr, ref_plan.by_d()[0].seg_ptr)
        row = got["X"]._row(0)
        assert torch.equal(_ops.cached_plan(row, int(got["num_nodes"]), "scatter").seg_ptr,
                           _ops.plan_from_keys(row.clone(), int(got["num_nodes"])).seg_ptr)
    empty = store.collate([])
    assert empty["num_graphs"] == 0 and empty["num_nodes"] == 0 and empty["x"].numel() == 0 and empty["X"].nnz == 0
    assert empty[key + "___acd"].shape == (3, 0)


@pytest.mark.parametrize("kind", ["zinc", "i2"])
def test_collated_groupings_on_demand(dev, kind):
    """the groupings of a collated batch's tuples by their other coordinates and of its edges by either endpoint (cross-subgraph
    pooling, unpooling gradients, spmm) are assembled from the store's per-graph parts when an operator asks for them: equal to
    the planner's sort of the collated index rows, without a host read; pooling through them matches pooling through sorted plans."""
    from pygho_amd import _ops, synth
    from pygho_amd.collate import DeviceGraphStore
    rng = np.random.default_rng(17)
    recs = [synth.make_graph(rng, kind) for _ in range(12)]
    store = DeviceGraphStore(recs, dev)
    dd = store.collate([4, 4, 0, 11, 7, 2])
    n, X, A = int(dd["num_nodes"]), dd["X"], dd["A"]
    f0 = _ops.FETCHES[0]
    got = [(_ops.cached_plan(sp._row(dim), n, "scatter"), sp._row(dim)) for sp, dim in
           [(X, k) for k in range(1, X.sparse_dim)] + [(A, 0), (A, 1)]]
    vals = torch.randn(X.nnz, 16, device=dev)
    pooled = _ops.scatter_reduce(vals, X._row(1), n, "mean")
    assert _ops.FETCHES[0] == f0, "assembling the groupings read something back"
    for plan, keys in got:
        ref = _ops.plan_from_keys(keys.clone(), n)
        assert torch.equal(plan.seg_ptr, ref.seg_ptr) and plan.max_len == ref.max_len
        assert (plan.perm is None) == (ref.perm is None) and (plan.perm is None or torch.equal(plan.perm, ref.perm))
    ref_pooled = _ops.scatter_reduce(vals, X._row(1).clone(), n, "mean")
    assert torch.equal(pooled, ref_pooled)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_table_grad_keeps_non_finite_rows_in_their_table_row(dev, dtype):
    """rows of the small-table gradient are accumulated with a 0 / 1 factor while they are finite; a row holding inf / nan must
    reach ONLY its own table row (a factor of 0 would spread nan to all of them)."""
    from pygho_amd import _ops
    m, d, n_table = 5000, 128, 16
    gen = torch.Generator(device=dev).manual_seed(3)
    g = torch.randn(m, d, device=dev, generator=gen).to(dtype)
    idx = torch.randint(0, n_table, (m,), device=dev, generator=gen)
    idx[1234], idx[4000] = 3, 9
    g[1234, 7] = float("inf")
    g[4000, 100] = float("nan")
    got = _ops.table_grad(g, idx, n_table).float()
    assert torch.isinf(got[3, 7]) and torch.isnan(got[9, 100])
    bad = ~torch.isfinite(got)
    bad[3, 7] = bad[9, 100] = False
    assert not bool(bad.any())
    clean = g.clone()
    clean[1234, 7] = clean[4000, 100] = 0
    ref = _ops.table_grad(clean, idx, n_table).float()
    mask = torch.ones_like(got, dtype=torch.bool)
    mask[3, 7] = mask[9, 100] = False
    assert torch.equal(got[mask], ref[mask])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_mean_backward_in_one_pass_has_the_bits_of_the_aten_sequence(dev, dtype):
    """the gradient of a segment mean w.r.t. its rows (`pygho_row_gather_mean`): scale by 1 / max(count, 1) rounded to the row type and
    gather, one launch == torch's (gout * inv.to(dtype))[idx] bit for bit; segments of every length incl. empty ones and one of 1000
    rows; through `torch_scatter_reduce(..., "mean")` the input gradient equals the plain-torch formula"""
    from pygho_amd import _ops
    from pygho_amd.backend.utils import torch_scatter_reduce
    g = torch.Generator(device="cpu").manual_seed(3)
    lens = torch.cat([torch.randint(0, 40, (300,), generator=g), torch.tensor([1000, 0, 1, 3, 7])])
    n_seg = lens.numel()
    idx = torch.repeat_interleave(torch.arange(n_seg), lens)
    idx = idx[torch.randperm(idx.numel(), generator=g)].to(dev)
    seg_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(torch.int32).to(dev)
    gout = torch.randn(n_seg, 128, generator=g).to(dev).to(dtype)
    inv = lens.clamp_min(1).to(torch.float32).reciprocal().to(dev)
    want = (gout * inv.to(dtype).unsqueeze(-1))[idx]
    got = _ops.row_gather_mean(gout, idx.to(torch.int32), seg_ptr)
    assert torch.equal(got, want)
    src = torch.randn(idx.numel(), 128, generator=g).to(dev).to(dtype).requires_grad_(True)
    out = torch_scatter_reduce(0, src, idx, n_seg, "mean")
    out.backward(gout)
    assert torch.equal(src.grad, want)


@pytest.mark.parametrize("n_table,true_rows,capacity", [(16, 33 * 4096, 33 * 4096 + 1), (32, 17 * 2048, 17 * 2048 + 3), (16, 16383, 16640),
                                                        (32, 3000, 3200), (16, 140_000, 150_016)])
def test_small_table_gradient_with_the_row_count_on_the_device_covers_every_row(dev, n_table, true_rows, capacity):
    """`pygho_table_grad_dyn`: the launch is sized for a CAPACITY, the kernel cuts the rows into wavefront shares from the TRUE count --
    the partition of a launch sized for exactly that count (same bits), for which the capacity-sized launch must offer enough
    wavefronts although their number is not monotonic in the row count (it drops where the rows per wavefront step up): true counts
    just below such a step under capacities just above it"""
    from pygho_amd import _ops
    g = torch.Generator(device="cpu").manual_seed(5)
    rows = (torch.randn(capacity, 128, generator=g) * 0.1).to(dev).to(torch.bfloat16)
    idx = torch.randint(0, min(n_table, 28), (capacity,), generator=g).to(dev)
    want = _ops.table_grad(rows[:true_rows].contiguous(), idx[:true_rows].contiguous(), n_table, out_dtype=torch.float32)
    cnt = torch.tensor([true_rows], dtype=torch.int32, device=dev)
    with _ops.row_families({capacity: cnt}):
        got = _ops.table_grad(rows, idx, n_table, out_dtype=torch.float32)
    assert torch.equal(got, want)
    ref = torch.zeros(n_table, 128, dtype=torch.float64, device=dev).index_add_(0, idx[:true_rows], rows[:true_rows].double())
    torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=1e-4)

"""
GPU parity tests of the masked (dense) path: MaskedTensor methods and the MFMA batched contraction against
the golden fixtures generated from the reference (pre-filled inputs, where the reference's behaviour equals
its documented behaviour) and against the numpy oracle / einsum on seeded inputs.
Tolerances: f32 1e-5 (exact-f32 MFMA, different summation order than einsum); bf16/f16 one output rounding.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from pygho_amd import _native
    _native.lib()
    return torch.device("cuda:0")


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def N(t):
    return t.detach().float().cpu().numpy() if t.dtype in (torch.bfloat16, torch.float16) else t.detach().cpu().numpy()


def test_mamamm_golden_forward_backward(dev):
    from pygho_amd import MaskedTensor
    from pygho_amd.backend.Mamamm import mamamm
    g = load_golden("masked_ops.npz")
    Xm, Am, nm = T(g["Xmask"], dev), T(g["Amask"], dev), T(g["nodemask"], dev)
    cases = {"X2A1": ("X", Xm, 2, "A", Am, 1), "A1X1": ("A", Am, 1, "X", Xm, 1), "X2X1": ("X", Xm, 2, "X", Xm, 1)}
    for name, (pn, pm, d1, qn, qm, d2) in cases.items():
        P = T(g[pn], dev).requires_grad_(True)
        Q = T(g[qn], dev).requires_grad_(True)
        out = mamamm(MaskedTensor(P, pm), d1, MaskedTensor(Q, qm), d2, Xm)
        valid = g["Xmask"][..., None]
        np.testing.assert_allclose(N(out.data) * valid, g[f"mamamm_{name}"] * valid, **TOL)
        assert float(N(out.data)[~g["Xmask"]].__abs__().sum()) == 0.0           # documented: masked entries = padvalue
        w = T(g[f"mamamm_{name}_w"] * valid, dev)                                  # gradient only through valid outputs
        (out.data * w).sum().backward()
        # reference gradients were taken with the same weights on ALL outputs; its masked outputs are 0 * inputs
        # only where inputs are zero-filled, so recompute the expectation with the oracle on the masked weights
        Pn, Qn = g[pn], g[qn]
        wm = g[f"mamamm_{name}_w"] * valid
        Pz, Qz = Pn * g[pn + "mask"][..., None], Qn * g[qn + "mask"][..., None]
        if name == "A1X1":      # out[bij] = sum_k P[bki] Q[bkj]
            gP = np.einsum("bijd,bkjd->bkid", wm, Qz)
            gQ = np.einsum("bijd,bkid->bkjd", wm, Pz)
        else:                   # out[bij] = sum_k P[bik] Q[bkj]
            gP = np.einsum("bijd,bkjd->bikd", wm, Qz)
            gQ = np.einsum("bijd,bikd->bkjd", wm, Pz)
        np.testing.assert_allclose(N(P.grad), gP * g[pn + "mask"][..., None], **TOL)
        np.testing.assert_allclose(N(Q.grad), gQ * g[qn + "mask"][..., None], **TOL)
    # node level: A (b,n,n) without dense dims x x (b,n,d)
    out = mamamm(MaskedTensor(T(g["A"][..., 0].copy(), dev), Am), 2, MaskedTensor(T(g["x"], dev), nm), 1, nm)
    np.testing.assert_allclose(N(out.data) * g["nodemask"][..., None], g["mamamm_node"] * g["nodemask"][..., None], **TOL)
    # 3-D representation x adjacency
    m3 = T(g["X3mask"], dev)
    out = mamamm(MaskedTensor(T(g["X3"], dev), m3), 3, MaskedTensor(T(g["A"], dev), Am), 1, m3)
    v3 = g["X3mask"][..., None]
    np.testing.assert_allclose(N(out.data) * v3, g["mamamm_X3A1"] * v3, **TOL)


def test_masked_tensor_methods_golden(dev):
    from pygho_amd import MaskedTensor
    from pygho_amd.backend.MaTensor import filterinf
    g = load_golden("masked_ops.npz")
    X, Xm = T(g["X"], dev), T(g["Xmask"], dev)
    MX = MaskedTensor(X, Xm)
    Mx = MaskedTensor(T(g["x"], dev), T(g["nodemask"], dev))
    for op in ("sum", "mean", "max"):
        for dims in ([1], [2], [1, 2]):
            tag = f"red_{op}_{''.join(map(str, dims))}"
            r = getattr(MX, op)(dims)
            np.testing.assert_allclose(N(r.data), g[tag], **TOL)
            assert np.array_equal(N(r.mask), g[tag + "_mask"])
            rk = getattr(MX, op)(dims, keepdim=True)
            assert rk.data.shape[:3] == tuple(1 if i in dims else s for i, s in enumerate(X.shape[:3]))
    dg = MX.diag([1, 2])
    assert np.array_equal(N(dg.data), g["diag"]) and np.array_equal(N(dg.mask), g["diag_mask"])
    valid = g["Xmask"][..., None]
    assert np.array_equal(N(Mx.unpooling([2], MX).data), g["unpool2"] * valid)      # documented: masked entries read padvalue
    assert np.array_equal(N(Mx.unpooling([1], MX).data), g["unpool1"] * valid)
    assert np.array_equal(N(MaskedTensor(X, Xm, padvalue=float("inf")).fill_masked(1024.)), g["fill1024"])
    inplace = MaskedTensor(X, Xm, padvalue=float("inf"))          # fill_masked_ (MaTensor.py:113-120): mutates the container
    inplace.fill_masked_(1024.)
    assert inplace.padvalue == 1024. and np.array_equal(N(inplace.data), g["fill1024"]) and np.array_equal(N(inplace.mask), g["Xmask"])
    np.testing.assert_allclose(N(MX.diagonalapply(lambda v, f: v * f.unsqueeze(-1)).data), g["diagapply"], **TOL)
    assert np.array_equal(N(MX.catvalue([MX, MX], True).data), g["cat"])
    np.testing.assert_allclose(N(MX.add(MaskedTensor(X * 2, Xm), True).data), g["add_same"], **TOL)
    assert np.array_equal(N(filterinf(T(g["filterinf_in"], dev))), g["filterinf_out"])
    # deviation fixture: garbage at masked positions must not leak (the reference leaks it)
    r = MaskedTensor(T(g["dev_in"], dev), Xm).sum([1])
    np.testing.assert_allclose(N(r.data), g["red_sum_1"], **TOL)
    assert not np.allclose(g["dev_ref_sum1"], g["red_sum_1"])
    # true minimum (the reference calls amax here)
    r = MX.min([2])
    exp, _ = O.ma_reduce(g["X"], g["Xmask"], [2], "min")
    np.testing.assert_allclose(N(r.data), exp, **TOL)


def test_masked_reduce_grads(dev):
    from pygho_amd import MaskedTensor
    rng = np.random.default_rng(0)
    b, n, d = 3, 7, 12
    X = rng.standard_normal((b, n, n, d)).astype(np.float32)
    mask = rng.random((b, n, n)) > 0.4
    mask[0, 2] = False
    for op in ("sum", "mean", "max", "min"):
        for dim in (1, 2):
            xt = T(X, dev).requires_grad_(True)
            out = getattr(MaskedTensor(xt, T(mask, dev)), op)([dim]).data
            w = torch.randn_like(out)
            (out * w).sum().backward()
            xr = torch.from_numpy(X).requires_grad_(True)
            m = torch.from_numpy(mask)[..., None]
            if op == "sum":
                ref = (xr * m).sum(dim)
            elif op == "mean":
                ref = (xr * m).sum(dim) / m.sum(dim).clamp_min(1)
            else:
                fill = float("-inf") if op == "max" else float("inf")
                t = xr.masked_fill(~m, fill)
                ref = t.amax(dim) if op == "max" else t.amin(dim)
                ref = torch.where(torch.isinf(ref), torch.zeros_like(ref), ref)
            np.testing.assert_allclose(N(out), ref.detach().numpy(), **TOL)
            (ref * w.cpu()).sum().backward()
            np.testing.assert_allclose(N(xt.grad), xr.grad.numpy(), **TOL)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(4, 37, 37, 37, 128), (2, 50, 70, 9, 16), (3, 16, 16, 16, 8), (2, 5, 130, 33, 24)])
@pytest.mark.parametrize("layout", [(False, True), (True, True), (False, False), (True, False)])
def test_masked_bmm_vs_einsum(dev, dtype, shape, layout):
    """every storage layout / ragged tile / k-block combination against a float64 einsum."""
    from pygho_amd import _ops
    nb, ni, nk, nj, d = shape
    akf, bkf = layout
    rng = np.random.default_rng(hash((shape, layout)) % (2 ** 31))
    A = rng.standard_normal((nb, ni, nk, d)).astype(np.float32)
    B = rng.standard_normal((nb, nk, nj, d)).astype(np.float32)
    am, bm, om = rng.random((nb, ni, nk)) > 0.3, rng.random((nb, nk, nj)) > 0.3, rng.random((nb, ni, nj)) > 0.2
    At, Bt = T(A, dev, dtype), T(B, dev, dtype)
    Aq, Bq = N(At).astype(np.float64), N(Bt).astype(np.float64)
    exp = np.einsum("bikd,bkjd->bijd", Aq * am[..., None], Bq * bm[..., None]) * om[..., None]
    a_st = At.permute(0, 2, 1, 3).contiguous() if akf else At
    b_st = Bt if bkf else Bt.permute(0, 2, 1, 3).contiguous()
    am_st = T(am, dev).permute(0, 2, 1).contiguous() if akf else T(am, dev)
    bm_st = T(bm, dev) if bkf else T(bm, dev).permute(0, 2, 1).contiguous()
    got = _ops.masked_bmm(a_st, b_st, _ops._mask_u8(am_st), _ops._mask_u8(bm_st), _ops._mask_u8(T(om, dev)),
                          nb, ni, nk, nj, d, akf, bkf)
    eps = {torch.float32: 1e-5, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    scale = np.abs(exp).max()
    np.testing.assert_allclose(N(got), exp, rtol=eps, atol=eps * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(4, 37, 37, 37, 128), (2, 50, 70, 9, 16), (3, 16, 16, 16, 8), (2, 5, 130, 33, 24)])
@pytest.mark.parametrize("layout", [(False, True), (True, True), (False, False), (True, False)])
@pytest.mark.parametrize("sparse_side", ["A", "B", "O"])
def test_masked_bmm_sparse_operand_lists(dev, dtype, shape, layout, sparse_side):
    """one operand with an adjacency-like mask (5 % unmasked, some empty columns): the contraction runs on the neighbour-list kernel
    (pygho_mask_lists + pygho_masked_bmm_lists) in every storage layout, and agrees with a float64 einsum AND with the dense
    matrix-core kernel on the same inputs; NaN planted in the masked slots of both operands never reaches the result."""
    from pygho_amd import _ops
    nb, ni, nk, nj, d = shape
    akf, bkf = layout
    rng = np.random.default_rng(hash((shape, layout, sparse_side)) % (2 ** 31))
    A = rng.standard_normal((nb, ni, nk, d)).astype(np.float32)
    B = rng.standard_normal((nb, nk, nj, d)).astype(np.float32)
    # "O": two dense operands, an adjacency-like OUTPUT mask (the gradient of an adjacency's values): pygho_masked_bmm_outlists
    dens = {"A": (0.05, 0.6, 0.8), "B": (0.6, 0.05, 0.8), "O": (0.6, 0.6, 0.05)}[sparse_side]
    am, bm, om = rng.random((nb, ni, nk)) < dens[0], rng.random((nb, nk, nj)) < dens[1], rng.random((nb, ni, nj)) < dens[2]
    At, Bt = T(A, dev, dtype), T(B, dev, dtype)
    Aq, Bq = N(At).astype(np.float64), N(Bt).astype(np.float64)
    exp = np.einsum("bikd,bkjd->bijd", Aq * am[..., None], Bq * bm[..., None]) * om[..., None]
    nan = torch.full((), float("nan"), dtype=dtype, device=dev)
    At = torch.where(T(am, dev)[..., None], At, nan)
    Bt = torch.where(T(bm, dev)[..., None], Bt, nan)
    a_st = At.permute(0, 2, 1, 3).contiguous() if akf else At
    b_st = Bt if bkf else Bt.permute(0, 2, 1, 3).contiguous()
    am_st = T(am, dev).permute(0, 2, 1).contiguous() if akf else T(am, dev)
    bm_st = T(bm, dev) if bkf else T(bm, dev).permute(0, 2, 1).contiguous()
    masks = (_ops._mask_u8(am_st), _ops._mask_u8(bm_st), _ops._mask_u8(T(om, dev)))
    supported = (d * At.element_size()) % 16 == 0
    assert min(_ops._mask_density(m) for m in masks) <= _ops.BMM_LIST_DENSITY
    _ops.USE_BMM_BLOCKS = False       # where the multi-block matrix-core kernel is eligible it is preferred over the lists
    try:
        got = _ops.masked_bmm(a_st, b_st, *masks, nb, ni, nk, nj, d, akf, bkf)
    finally:
        _ops.USE_BMM_BLOCKS = True
    if supported:
        sparse_mask = masks["ABO".index(sparse_side)]
        assert getattr(sparse_mask, "_pygho_lists", None), "the neighbour-list kernel did not run"
    eps = {torch.float32: 1e-5, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    scale = max(np.abs(exp).max(), 1e-3)
    np.testing.assert_allclose(N(got), exp, rtol=eps, atol=eps * scale)
    _ops.USE_BMM_LISTS = False
    try:
        dense = _ops.masked_bmm(a_st, b_st, *masks, nb, ni, nk, nj, d, akf, bkf)      # multi-block / 16x16x32 matrix-core kernel
    finally:
        _ops.USE_BMM_LISTS = True
    np.testing.assert_allclose(N(got), N(dense), rtol=eps, atol=eps * scale)
    np.testing.assert_allclose(N(dense), exp, rtol=eps, atol=eps * scale)                # NaN-planted masked slots never reach it either


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 37, 37, 37, 128), (2, 9, 64, 23, 256), (5, 1, 1, 1, 128), (2, 40, 3, 8, 384), (2, 33, 61, 70, 128)])
@pytest.mark.parametrize("layout", [(False, True), (True, False)])
def test_masked_bmm_multiblock_kernel_shapes(dev, dtype, shape, layout):
    """the multi-block matrix-core kernel (csrc/masked_bmm_blocks.h: rows of whole 256-B multiples, k <= 64): ragged tiles in
    both output dims, k = 1 / 3 / 61 / 64, two and three channel groups, padded-batch extents with an empty and a full batch
    element, NaN planted in every masked operand slot, against a float64 einsum; asymmetric operands (a transposed or
    lane-permuted result cannot pass)."""
    from pygho_amd import _ops
    nb, ni, nk, nj, d = shape
    akf, bkf = layout
    rng = np.random.default_rng(hash((shape, layout)) % (2 ** 31))
    A = (rng.standard_normal((nb, ni, nk, d)) + np.arange(d) * 0.01).astype(np.float32)
    B = (rng.standard_normal((nb, nk, nj, d)) - np.arange(nj)[None, None, :, None] * 0.02).astype(np.float32)
    am, bm, om = rng.random((nb, ni, nk)) > 0.3, rng.random((nb, nk, nj)) > 0.3, rng.random((nb, ni, nj)) > 0.2
    if nb > 2:                       # padded-batch structure: one empty element, one clipped to a corner
        am[0], bm[0], om[0] = False, False, False
        am[1, ni // 2:], bm[1, :, nj // 2:], om[1, ni // 2:], om[1, :, nj // 2:] = False, False, False, False
    At, Bt = T(A, dev, dtype), T(B, dev, dtype)
    Aq, Bq = N(At).astype(np.float64), N(Bt).astype(np.float64)
    exp = np.einsum("bikd,bkjd->bijd", Aq * am[..., None], Bq * bm[..., None]) * om[..., None]
    nan = torch.full((), float("nan"), dtype=dtype, device=dev)
    At = torch.where(T(am, dev)[..., None], At, nan)
    Bt = torch.where(T(bm, dev)[..., None], Bt, nan)
    a_st = At.permute(0, 2, 1, 3).contiguous() if akf else At
    b_st = Bt if bkf else Bt.permute(0, 2, 1, 3).contiguous()
    am_st = T(am, dev).permute(0, 2, 1).contiguous() if akf else T(am, dev)
    bm_st = T(bm, dev) if bkf else T(bm, dev).permute(0, 2, 1).contiguous()
    masks = (_ops._mask_u8(am_st), _ops._mask_u8(bm_st), _ops._mask_u8(T(om, dev)))
    assert (d * At.element_size()) % 256 == 0 and nk <= 64
    eps = {torch.float32: 1e-5, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    scale = max(np.abs(exp).max(), 1e-3)
    for use_ext in (True, False):
        _ops.USE_BMM_EXTENTS = use_ext
        try:
            got = _ops.masked_bmm(a_st, b_st, *masks, nb, ni, nk, nj, d, akf, bkf)
            nomask = _ops.masked_bmm(torch.nan_to_num(a_st), torch.nan_to_num(b_st), None, None, None, nb, ni, nk, nj, d, akf, bkf)
        finally:
            _ops.USE_BMM_EXTENTS = True
        np.testing.assert_allclose(N(got), exp, rtol=eps, atol=eps * scale)
        np.testing.assert_allclose(N(nomask) * om[..., None], exp, rtol=eps, atol=eps * scale)


def test_masked_bmm_baseline_size_properties(dev):
    """BASELINE config 3's size -- (1024, 37, 37, 128) bf16, a padded ZINC-shape batch: node-pair masks on both operands and the
    output (the PPGN contraction) and an adjacency mask on the second operand -- where a float64 einsum of the whole batch is 11 GB:
    (i) a batch slice of 16 elements against the float64 einsum; (ii) scaling an operand by a power of two scales the result bit
    for bit; (iii) batch elements are independent: the first 64 elements computed alone equal the first 64 of the full launch bit
    for bit; (iv) masked output slots are exact zeros; (v) two runs are bit-identical."""
    from pygho_amd import _ops, synth
    dn = synth.make_dense_batch(256, seed=2, hidden=128, nmax=37)
    rep = 4
    t = lambda a, dt=None: (torch.from_numpy(a).to(dev).repeat((rep,) + (1,) * (a.ndim - 1)).to(dt) if dt
                            else torch.from_numpy(a).to(dev).repeat((rep,) + (1,) * (a.ndim - 1)))
    X, Xm = t(dn["X"], torch.bfloat16), t(dn["Xmask"])
    Am = t(dn["Amask"])
    nb, n, d = X.shape[0], X.shape[1], X.shape[3]
    assert (nb, n, d) == (1024, 37, 128)
    torch.manual_seed(0)
    Y = torch.randn_like(X)
    mx, ma = _ops._mask_u8(Xm), _ops._mask_u8(Am)
    for name, bmask, bmask_b in (("X Y (pair masks)", mx, Xm), ("X A (adjacency mask)", ma, Am)):
        f = lambda a, b, nbb=nb: _ops.masked_bmm(a[:nbb].contiguous(), b[:nbb].contiguous(), mx[:nbb].contiguous(), bmask[:nbb].contiguous(),
                                                 mx[:nbb].contiguous(), nbb, n, n, n, d, False, True)
        out = f(X, Y)
        assert torch.equal(out, f(X, Y)), name                                                     # (v)
        assert torch.equal(f(X * 2, Y), out * 2) and torch.equal(f(X, Y * 0.25), out * 0.25), name   # (ii)
        assert torch.equal(f(X, Y, 64), out[:64]), name                                            # (iii)
        assert float(out[~Xm.bool()].abs().max()) == 0.0, name                                     # (iv)
        sl = slice(100, 116)
        exp = torch.einsum("bikd,bkjd->bijd", X[sl].double() * Xm[sl].bool()[..., None], Y[sl].double() * bmask_b[sl].bool()[..., None]) \
            * Xm[sl].bool()[..., None]
        scale = float(exp.abs().max())
        torch.testing.assert_close(out[sl].double() / scale, exp / scale, rtol=0, atol=2.0 ** -8, msg=name)   # (i)


def test_mamamm_baseline_size_elementwise_vs_f64_einsum(dev):
    """BASELINE config 3's size, EVERY output element of the whole (1024, 37, 37, 128) bf16 batch: mamamm(X, 2, Y, 1) (pair masks on
    both operands) and mamamm(X, 2, A, 1) (adjacency-masked second operand) through the pygho API against the reference's einsum
    (pygho/backend/Mamamm.py:45-47) evaluated in float64 on the device in batch slices of 64.  Bar: one bf16 rounding of an
    f32-accumulated sum of exact products -- 2^-9 relative (2^-8 allowed) plus 1e-5 of the largest magnitude for the accumulation
    order of the matrix-core instruction; masked output slots exactly zero."""
    from pygho_amd import MaskedTensor, synth
    from pygho_amd.backend.Mamamm import mamamm
    dn = synth.make_dense_batch(256, seed=2, hidden=128, nmax=37)
    rep = 4
    t = lambda a, dt=None: (torch.from_numpy(a).to(dev).repeat((rep,) + (1,) * (a.ndim - 1)).to(dt) if dt
                            else torch.from_numpy(a).to(dev).repeat((rep,) + (1,) * (a.ndim - 1)))
    Xraw, Xm, Araw, Am = t(dn["X"], torch.bfloat16), t(dn["Xmask"]), t(dn["A"], torch.bfloat16), t(dn["Amask"])
    assert tuple(Xraw.shape) == (1024, 37, 37, 128)
    torch.manual_seed(0)
    Yraw = torch.randn_like(Xraw) * Xm.unsqueeze(-1).to(torch.bfloat16)
    X = MaskedTensor(Xraw, Xm, 0.0, True)
    for name, Braw, Bm in (("X Y", Yraw, Xm), ("X A", Araw, Am)):
        out = mamamm(X, 2, MaskedTensor(Braw, Bm, 0.0, True), 1, Xm)
        got = out.data
        assert float(got[~Xm.bool()].float().abs().max()) == 0.0, name
        worst = 0.0
        for lo in range(0, got.shape[0], 64):
            sl = slice(lo, lo + 64)
            exp = torch.einsum("bikd,bkjd->bijd", Xraw[sl].double() * Xm[sl].bool()[..., None], Braw[sl].double() * Bm[sl].bool()[..., None]) \
                * Xm[sl].bool()[..., None]
            scale = float(exp.abs().max())
            err = (got[sl].double() - exp).abs() - 2.0 ** -8 * exp.abs()
            worst = max(worst, float(err.max()) / max(scale, 1e-30))
        assert worst <= 1e-5, f"{name}: an element is off by more than one bf16 rounding + 1e-5 of the scale ({worst:.3e})"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", [(False, True), (True, True), (False, False), (True, False)])
def test_masked_bmm_padded_batch_extents(dev, dtype, layout):
    """padded-batch masks (node-mask outer products of ragged sizes, one empty batch element, one full one; k spanning two staging
    blocks): the matrix-core kernel clipped to the per-element extents (pygho_mask_extents + pygho_masked_bmm_clipped) gives the
    same result as the unclipped kernel bit for bit, and both agree with a float64 einsum."""
    from pygho_amd import _ops
    nb, ni, nk, nj, d = 6, 50, 70, 41, 16
    akf, bkf = layout
    rng = np.random.default_rng(5)
    sizes = [(0, 0, 0), (50, 70, 41), (17, 33, 9), (49, 1, 40), (3, 65, 41), (20, 20, 20)]
    am = np.zeros((nb, ni, nk), bool); bm = np.zeros((nb, nk, nj), bool); om = np.zeros((nb, ni, nj), bool)
    for b, (a, k, j) in enumerate(sizes):
        am[b, :a, :k] = rng.random((a, k)) > 0.2
        bm[b, :k, :j] = rng.random((k, j)) > 0.2
        om[b, :a, :j] = rng.random((a, j)) > 0.1
    A = rng.standard_normal((nb, ni, nk, d)).astype(np.float32)
    B = rng.standard_normal((nb, nk, nj, d)).astype(np.float32)
    At, Bt = T(A, dev, dtype), T(B, dev, dtype)
    Aq, Bq = N(At).astype(np.float64), N(Bt).astype(np.float64)
    exp = np.einsum("bikd,bkjd->bijd", Aq * am[..., None], Bq * bm[..., None]) * om[..., None]
    a_st = At.permute(0, 2, 1, 3).contiguous() if akf else At
    b_st = Bt if bkf else Bt.permute(0, 2, 1, 3).contiguous()
    am_st = T(am, dev).permute(0, 2, 1).contiguous() if akf else T(am, dev)
    bm_st = T(bm, dev) if bkf else T(bm, dev).permute(0, 2, 1).contiguous()
    masks = (_ops._mask_u8(am_st), _ops._mask_u8(bm_st), _ops._mask_u8(T(om, dev)))
    ext = _ops._mask_extents(*masks, nb, ni, nk, nj, akf, bkf)
    want = np.array([[min(a, a2), min(k, k2), min(j, j2)] for (a, k, j), (a2, k2, j2) in
                     zip([(am[b].any(1).nonzero()[0].max(initial=-1) + 1, am[b].any(0).nonzero()[0].max(initial=-1) + 1, nj) for b in range(nb)],
                         [(ni, bm[b].any(1).nonzero()[0].max(initial=-1) + 1, bm[b].any(0).nonzero()[0].max(initial=-1) + 1) for b in range(nb)])])
    want[:, 0] = np.minimum(want[:, 0], [om[b].any(1).nonzero()[0].max(initial=-1) + 1 for b in range(nb)])
    want[:, 2] = np.minimum(want[:, 2], [om[b].any(0).nonzero()[0].max(initial=-1) + 1 for b in range(nb)])
    assert np.array_equal(N(ext), want)
    got = _ops.masked_bmm(a_st, b_st, *masks, nb, ni, nk, nj, d, akf, bkf)
    _ops.USE_BMM_EXTENTS = False
    try:
        plain = _ops.masked_bmm(a_st, b_st, *masks, nb, ni, nk, nj, d, akf, bkf)
    finally:
        _ops.USE_BMM_EXTENTS = True
    assert torch.equal(got, plain)
    eps = {torch.float32: 1e-5, torch.bfloat16: 2.0 ** -8}[dtype]
    np.testing.assert_allclose(N(got), exp, rtol=eps, atol=eps * np.abs(exp).max())


def test_masked_bmm_transpose_detecting(dev):
    """A = I with an asymmetric B: a swapped output layout cannot pass."""
    from pygho_amd import _ops
    nb, n, d = 2, 20, 8
    A = torch.eye(n, device=dev).reshape(1, n, n, 1).expand(nb, n, n, d).contiguous()
    B = (torch.arange(n, device=dev).reshape(n, 1) * 100 + torch.arange(n, device=dev).reshape(1, n)).float()
    B = B.reshape(1, n, n, 1) + torch.arange(d, device=dev).float().reshape(1, 1, 1, d) * 0.25
    B = B.expand(nb, n, n, d).contiguous()
    got = _ops.masked_bmm(A, B, None, None, None, nb, n, n, n, d, False, True)
    assert torch.equal(got, B)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 7, 5, 8), (2, 37, 37, 128), (5, 3, 9, 12), (2, 6, 4, 6), (1, 70, 2, 2048)])
def test_masked_elementwise_kernels_all_row_sizes(dev, dtype, shape):
    """fill / single-dim reductions (+ gradients) / broadcast at row sizes that take the 16-byte-per-lane kernels
    (row bytes % 16 == 0), the element-wise kernels (odd rows, > 256 chunks) and ragged workgroup tails.  Masked slots hold
    NaN: nothing of them may reach a result (the vector forms never fetch them)."""
    from pygho_amd import _ops
    b, n1, n2, d = shape
    gen = torch.Generator().manual_seed(b * 1000 + d)
    x = torch.randn(shape, generator=gen).to(dtype)
    mask = torch.rand((b, n1, n2), generator=gen) > 0.45
    mask[0, 1] = False
    mask[-1, :, 0] = False
    xd = torch.where(mask[..., None], x, torch.full_like(x, float("nan"))).to(dev)
    md = mask.to(dev)
    x32 = torch.where(mask[..., None], x.float(), torch.zeros(()))
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)

    filled = _ops.masked_fill(xd, md, 3.0)
    exp = torch.where(mask[..., None], x, torch.full_like(x, 3.0))
    assert torch.equal(filled.cpu(), exp)

    for dim in (1, 2):
        cnt = mask.sum(dim)
        for aggr in ("sum", "mean", "max", "min"):
            xg = xd.clone().requires_grad_(True)
            out, om = _ops.masked_reduce(xg, md, dim, aggr)
            assert torch.equal(om.cpu(), cnt > 0)
            if aggr in ("sum", "mean"):
                # sequential f32 accumulation in index order, as the kernels do
                acc = torch.zeros(out.shape, dtype=torch.float32)
                for k in range(shape[dim]):
                    acc = acc + x32.select(dim, k)
                if aggr == "mean":
                    acc = torch.where(cnt[..., None] > 0, acc / cnt.clamp_min(1)[..., None].float(), torch.zeros(()))
                assert torch.equal(out.detach().cpu(), acc.to(dtype))
            else:
                fill = float("-inf") if aggr == "max" else float("inf")
                t = x.float().masked_fill(~mask[..., None], fill)
                ref = t.amax(dim) if aggr == "max" else t.amin(dim)
                ref = torch.where(torch.isinf(ref), torch.zeros(()), ref)
                assert torch.equal(out.detach().cpu().float(), ref)
            if aggr in ("sum", "mean"):
                w = torch.randn(out.shape, generator=gen).to(dtype)
                out.backward(w.to(dev))
                g = w.float().unsqueeze(dim)
                if aggr == "mean":
                    g = g / cnt.clamp_min(1)[..., None].float().unsqueeze(dim)
                gexp = torch.where(mask[..., None], g.expand(shape), torch.zeros(())).to(dtype)
                assert torch.equal(xg.grad.cpu(), gexp)

    # broadcast of a (b, n2, d) node tensor along dim 1 and of a (b, n1, d) tensor along dim 2
    for dim, src_shape in ((1, (b, n2, d)), (2, (b, n1, d))):
        src = torch.randn(src_shape, generator=gen).to(dtype)
        out = _ops.masked_broadcast(src.to(dev), md, dim, -2.0, 2)
        exp = torch.where(mask[..., None], src.unsqueeze(dim).expand(shape), torch.full((), -2.0, dtype=dtype))
        assert torch.equal(out.cpu(), exp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 7, 7, 8), (2, 37, 37, 128), (4, 5, 9, 12), (2, 9, 4, 64)])
def test_masked_pair_combine(dev, dtype, shape):
    """out = mask ? ((base + row_term[b,i]) + col_term[b,j]) (diagonal: + / replaced by diag_term[b,i]) : 0, every operand
    optional, square and rectangular tuple grids, NaN in the masked slots of base (never fetched)."""
    from pygho_amd import _ops
    b, n1, n2, d = shape
    nd = min(n1, n2)
    gen = torch.Generator().manual_seed(n1 * 100 + d)
    mask = torch.rand((b, n1, n2), generator=gen) > 0.4
    mask[0, 0] = False
    base = torch.randn(shape, generator=gen).to(dtype)
    rt = torch.randn((b, n1, d), generator=gen).to(dtype)
    ct = torch.randn((b, n2, d), generator=gen).to(dtype)
    dg = torch.randn((b, nd, d), generator=gen).to(dtype)
    eye = torch.eye(n1, n2, dtype=torch.bool)[None, :, :, None]
    dg_full = torch.zeros(shape)
    idx = torch.arange(nd)
    dg_full[:, idx, idx] = dg.float()
    if (d * base.element_size()) % 16 != 0:                      # no 16-byte form: refused loudly, callers keep the separate passes
        assert not _ops.pair_combine_supported(base.to(dev))
        with pytest.raises(RuntimeError, match="16-byte"):
            _ops.masked_pair_combine(base.to(dev), None, None, None, False, mask.to(dev), shape, dtype, dev)
        return
    based = torch.where(mask[..., None], base, torch.full_like(base, float("nan"))).to(dev)
    for use in ((1, 1, 1, 1), (0, 1, 1, 1), (1, 0, 0, 0), (1, 1, 0, 1), (0, 0, 1, 0), (1, 1, 1, 0)):
        for replace in (False, True):
            args = [based if use[0] else None, rt.to(dev) if use[1] else None, ct.to(dev) if use[2] else None,
                    dg.to(dev) if use[3] else None]
            out = _ops.masked_pair_combine(*args, replace, mask.to(dev), shape, dtype, dev)
            acc = base.float() if use[0] else torch.zeros(shape)
            if use[1]:
                acc = acc + rt.float()[:, :, None, :]
            if use[2]:
                acc = acc + ct.float()[:, None, :, :]
            if use[3]:
                acc = torch.where(eye, dg_full, acc) if replace else torch.where(eye, acc + dg_full, acc)
            exp = torch.where(mask[..., None], acc, torch.zeros(())).to(dtype)
            assert torch.equal(out.cpu(), exp), (use, replace)
    # no mask = all valid
    out = _ops.masked_pair_combine(base.to(dev), rt.to(dev), None, None, False, None, shape, dtype, dev)
    assert torch.equal(out.cpu(), (base.float() + rt.float()[:, :, None, :]).to(dtype))


def test_pair_views_gradient_one_pass(dev):
    """(diagonal rows, sum over dim 1, sum over dim 2) of a padded representation and their joint gradient against autograd
    through the plain torch expressions."""
    from pygho_amd import _ops
    gen = torch.Generator().manual_seed(5)
    b, n, d = 3, 6, 8
    mask = torch.rand((b, n, n), generator=gen) > 0.35
    x = torch.randn((b, n, n, d), generator=gen)
    xd = x.to(dev).requires_grad_(True)
    dg, s1, s2 = _ops.pair_views(xd, mask.to(dev))
    ws = [torch.randn(t.shape, generator=gen) for t in (dg, s1, s2)]
    (dg * ws[0].to(dev)).sum().add((s1 * ws[1].to(dev)).sum()).add((s2 * ws[2].to(dev)).sum()).backward()
    xr = x.clone().requires_grad_(True)
    xm = xr * mask[..., None]
    rdg = torch.diagonal(xm, 0, 1, 2).movedim(-1, 1)
    r1, r2 = xm.sum(1), xm.sum(2)
    ((rdg * ws[0]).sum() + (r1 * ws[1]).sum() + (r2 * ws[2]).sum()).backward()
    for got, ref in ((dg, rdg), (s1, r1), (s2, r2)):
        np.testing.assert_allclose(N(got), ref.detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(N(xd.grad), xr.grad.numpy(), rtol=1e-6, atol=1e-6)


def test_spmamm_vs_reference_fixture(dev):
    """spmamm in the configuration the reference itself can run (scalar adjacency values / value-less A, no dense dims, pre-filled
    MaskedTensor, explicit output mask): HIP path against outputs of the reference (tests/golden/spmamm.npz)."""
    from conftest import load_golden
    from pygho_amd import MaskedTensor, SparseTensor
    from pygho_amd.backend.Spmamm import spmamm
    g = load_golden("spmamm.npz")
    shape = [int(v) for v in g["shapeA"]]
    for dim1 in (1, 2):
        for dim2 in (1, 2):
            tag = f"dim1_{dim1}_dim2_{dim2}"
            B = MaskedTensor(T(g["B_" + tag], dev), T(g["Bmask_" + tag], dev), 0.0, True)
            om = T(g["omask_" + tag], dev)
            for av, key in ((T(g["Aval"], dev), "sum_"), (None, "sum_novalue_")):
                out = spmamm(SparseTensor(T(g["ind"], dev), av, shape, True), dim1, B, dim2, om, "sum")
                assert torch.equal(out.mask, om)
                np.testing.assert_allclose(N(out.raw), g[key + tag], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("aggr", ["sum", "max", "min"])
@pytest.mark.parametrize("dim1", [1, 2])
def test_spmamm_vs_oracle_and_einsum(dev, aggr, dim1):
    """sparse (b, n, m, d) adjacency x masked representation (reference Spmamm.py:12-68, implemented to its documented semantics)
    against the oracle; the sum form also against an einsum, and its gradients wrt both value sets against autograd on the CPU."""
    from pygho_amd import MaskedTensor, SparseTensor
    from pygho_amd.backend.Spmamm import spmamm
    rng = np.random.default_rng(3 + dim1)
    b, n, m, l, d = 4, 6, 6, 5, 8
    Adense = rng.standard_normal((b, n, m, d)).astype(np.float32)
    Amask = rng.random((b, n, m)) > 0.6
    Amask[1, 2] = False                                           # a target without any message
    ind = np.stack(np.nonzero(Amask))
    Aval = Adense[ind[0], ind[1], ind[2]]
    cases = [((b, m, l, d), 1), ((b, l, m, d), 2)] if aggr == "sum" else [((b, m, d), 1)]
    for bshape, dim2 in cases:
        B = rng.standard_normal(bshape).astype(np.float32)
        Bmask = rng.random(bshape[:-1]) > 0.3
        av = T(Aval, dev).requires_grad_(True)
        bv = T(B, dev).requires_grad_(True)
        out = spmamm(SparseTensor(T(ind, dev), av, [b, n, m, d], True), dim1, MaskedTensor(bv, T(Bmask, dev)), dim2, None, aggr)
        exp = O.spmamm(ind, Aval, (b, n, m, d), dim1, B, Bmask, dim2, aggr)
        np.testing.assert_allclose(N(out.raw), exp, rtol=1e-5, atol=1e-5)
        if aggr != "sum":
            continue
        at, bt = torch.from_numpy(Adense).requires_grad_(True), torch.from_numpy(B).requires_grad_(True)
        am, bm = torch.from_numpy(Amask)[..., None], torch.from_numpy(Bmask)[..., None]
        spec = {(1, 1): "bknd,bkld->bnld", (2, 1): "bnkd,bkld->bnld", (1, 2): "bknd,blkd->blnd", (2, 2): "bnkd,blkd->blnd"}[(dim1, dim2)]
        ref = torch.einsum(spec, at * am, bt * bm)
        np.testing.assert_allclose(N(out.raw), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
        w = torch.randn(ref.shape, generator=torch.Generator().manual_seed(0))
        out.raw.backward(w.to(dev))
        ref.backward(w)
        np.testing.assert_allclose(N(av.grad), at.grad.numpy()[ind[0], ind[1], ind[2]], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(N(bv.grad) * Bmask[..., None], bt.grad.numpy() * Bmask[..., None], rtol=1e-5, atol=1e-5)


def test_new_kernels_empty_and_degenerate_inputs(dev):
    """zero-sized batches / tuple sets / contractions and fully masked inputs through the dense-path kernels added in round 1."""
    from pygho_amd import MaskedTensor, _ops
    from pygho_amd.backend.Mamamm import mamamm
    from pygho_amd.hodata import to_dense_adj, to_dense_tuplefeat, to_dense_x
    dt = torch.float32
    out = _ops.masked_pair_combine(torch.empty((0, 3, 3, 8), device=dev), None, None, None, False, None, (0, 3, 3, 8), dt, dev)
    assert out.shape == (0, 3, 3, 8)
    e32 = torch.empty(0, dtype=torch.int32, device=dev)
    out = _ops.pair_gather_combine(torch.empty((0, 8), device=dev), torch.ones((4, 8), device=dev), None, None, False, e32, e32, 8, dt, dev)
    assert out.shape == (0, 8)
    # a batch whose adjacency has no edge at all: neighbour lists are empty, the product is exactly zero
    x = torch.randn(3, 5, 5, 8, device=dev)
    xm = torch.ones(3, 5, 5, dtype=torch.bool, device=dev)
    a = torch.randn(3, 5, 5, 8, device=dev)
    am = torch.zeros(3, 5, 5, dtype=torch.bool, device=dev)
    got = mamamm(MaskedTensor(x, xm), 2, MaskedTensor(a, am), 1, xm)
    assert torch.count_nonzero(got.data) == 0
    # every output masked
    got = mamamm(MaskedTensor(x, xm), 2, MaskedTensor(a, xm), 1, am)
    assert torch.count_nonzero(got.data) == 0
    # builders: a batch of empty graphs next to one real graph, and an edge-less batch
    ptr = torch.tensor([0, 0, 3, 3], device=dev)
    mt = to_dense_x(torch.arange(3, device=dev).float().reshape(3, 1), ptr)
    assert mt.mask.cpu().tolist() == [[False] * 3, [True] * 3, [False] * 3] and mt.data[1].flatten().cpu().tolist() == [0., 1., 2.]
    shape = torch.tensor([[0, 0], [2, 2], [0, 0]], device=dev)
    mt = to_dense_tuplefeat(torch.arange(4, device=dev), shape, torch.tensor([0, 0, 4, 4], device=dev))
    assert mt.mask.sum().item() == 4 and mt.data[1].cpu().tolist() == [[0, 1], [2, 3]]
    none = torch.empty((2, 0), dtype=torch.int64, device=dev)
    mt = to_dense_adj(none, torch.empty(0, dtype=torch.int64, device=dev), torch.empty((0, 4), device=dev), 3, 2, -1.0)
    assert not mt.mask.any() and torch.all(mt.data == -1.0) and tuple(mt.shape) == (2, 3, 3, 4)


@pytest.mark.parametrize("shape,md,dims", [((2, 5, 5, 5, 8), 4, [1, 2, 3]), ((2, 4, 6, 4, 4, 8), 5, [1, 3, 4]), ((3, 4, 4, 4, 4), 5, [1, 2, 3, 4])])
def test_masked_diag_over_more_than_two_dims(dev, shape, md, dims):
    """`MaskedTensor.diag` over > 2 masked dims (reference MaTensor.py:208-223 raises for every such input -- its loop keeps the
    original dim numbers after the first diagonal; round 4 raised NotImplementedError): the documented intent, against the oracle's
    element-wise restatement; values, mask, fill state and the gradient (the diagonal's positions receive it, nothing else)."""
    from oracle import np_oracle as O
    from pygho_amd import MaskedTensor
    rng = np.random.default_rng(4)
    data = rng.standard_normal(shape).astype(np.float32)
    mask = rng.random(shape[:md]) > 0.3
    x = torch.from_numpy(data).to(dev).requires_grad_(True)
    X = MaskedTensor(x, torch.from_numpy(mask).to(dev), 0.0, False)
    out = X.diag(dims)
    d, m = O.ma_diag(data, mask, dims)
    assert np.array_equal(out.mask.cpu().numpy(), m)
    got = out.fill_masked(0.).detach().cpu().numpy()
    want = d * m.reshape(m.shape + (1,) * (d.ndim - m.ndim))
    assert np.array_equal(got, want)
    out.fill_masked(0.).sum().backward()
    gexp = np.zeros_like(data)
    n = shape[dims[0]]
    for i in range(n):
        sel = tuple(i if k in dims else slice(None) for k in range(md))
        gexp[sel] = mask[sel].reshape(mask[sel].shape + (1,) * (data.ndim - md))
    assert np.array_equal(x.grad.cpu().numpy(), gexp)


def test_lds_tiled_contraction_variant_returns_the_same_bits():
    """round 6: the 16 x 16 workgroup-tile kernel (`csrc/masked_bmm_tiled.h`, opt-in through PYGHO_BMM_TILED=1 -- it measured slower than the
    multi-block kernel) stays correct: in a child process with the switch on, the contraction over ragged shapes, all
    mask combinations and both operand layouts equals the default kernel's result bit for bit (the k order per output is the same)."""
    import os
    import subprocess
    import sys
    import tempfile
    if not torch.cuda.is_available():
        pytest.skip("needs the ROCm device")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import sys, torch
sys.path.insert(0, %r)
from pygho_amd import _ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
outs = []
for (nb, ni, nk, nj, d, dt) in ((3, 37, 37, 37, 128, torch.bfloat16), (2, 16, 5, 33, 128, torch.float16), (4, 9, 64, 20, 256, torch.bfloat16)):
    for akf in (False, True):
        for bkf in (True, False):
            A = torch.randn((nb, nk, ni, d) if akf else (nb, ni, nk, d), device=dev).to(dt)
            B = torch.randn((nb, nk, nj, d) if bkf else (nb, nj, nk, d), device=dev).to(dt)
            am = (torch.rand(A.shape[:3], device=dev) < 0.6).to(torch.uint8)
            bm = (torch.rand(B.shape[:3], device=dev) < 0.5).to(torch.uint8)
            om = (torch.rand((nb, ni, nj), device=dev) < 0.7).to(torch.uint8)
            for masks in ((am, bm, om), (None, bm, None), (am, None, om), (None, None, None)):
                outs.append(_ops.masked_bmm(A, B, masks[0], masks[1], masks[2], nb, ni, nk, nj, d, akf, bkf).cpu())
torch.save(outs, sys.argv[1])
''' % repo
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for tiled in ("0", "1"):
            path = os.path.join(tmp, f"out{tiled}.pt")
            env = dict(os.environ, PYGHO_BMM_TILED=tiled)
            r = subprocess.run([sys.executable, "-c", script, path], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-3000:]
            res[tiled] = torch.load(path)
        assert len(res["0"]) == len(res["1"]) == 48
        for a, b in zip(res["0"], res["1"]):
            assert torch.equal(a, b)

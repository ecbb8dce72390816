"""GPU parity, op level: every C-ABI entry point against the oracle (torch-CPU float64 for the conv family — the same
operators the reference calls — and oracle/ops_np.py for the rest) on identical seeded inputs.

Stated tolerance (SURVEY.md §8c noise floor): the MFMA path is a k-ordered fp32 fma chain, so for a length-K dot product
of O(1) terms we allow |err| <= 2e-6 * sqrt(K) * max|terms| + 1e-6 (random-walk bound with ample margin); elementwise
ops must agree to 2e-6 relative.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops_np as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pcg():
    import pcgan_amd
    pcgan_amd.load()
    return pcgan_amd


def dev():
    return torch.device("cuda:0")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


# B, Cin, Cout, H, W, k, s, p
CONV_CASES = [
    # DCGAN-64 discriminator / generator layers at full width, small batch
    (4, 64, 128, 32, 32, 4, 2, 1),     # D2 / G4 adjoint
    (4, 128, 256, 16, 16, 4, 2, 1),    # D3 / G3 adjoint
    (4, 256, 512, 8, 8, 4, 2, 1),      # D4 / G2 adjoint
    (8, 1, 64, 64, 64, 4, 2, 1),       # D1 (thin Cin) / G5 adjoint
    (8, 512, 1, 4, 4, 4, 1, 0),        # D5 (thin Cout; a full-window layer: dot / outer-product kernels)
    (37, 512, 1, 4, 4, 4, 1, 0), (19, 64, 1, 4, 4, 4, 1, 0), (5, 128, 1, 5, 5, 5, 1, 0),   # ragged sample counts, a row that is no multiple of 256 float4
    (8, 8192, 100, 1, 1, 1, 1, 0),     # G1 adjoint as a 1x1 conv (K tail: Cout=100)
    # counteRGAN shapes
    (3, 64, 64, 28, 28, 3, 1, 1),      # resblock conv
    (3, 64, 128, 14, 14, 3, 2, 1),     # D conv 14->7
    (3, 128, 256, 7, 7, 3, 2, 1),      # D conv 7->4 (odd extent: unequal phases)
    (3, 3, 64, 28, 28, 3, 1, 1),       # conv_in (thin Cin=3)
    (3, 64, 1, 28, 28, 3, 1, 1),       # conv_out (thin Cout)
    (3, 2, 64, 28, 28, 3, 2, 1),       # D entry (thin Cin=2)
    # WGAN-GP shapes (conditional_gan/mnist/mnist_wgan_conditional.py:51-108)
    (2, 1, 256, 28, 28, 3, 2, 0),      # critic conv1 (thin Cin, no padding, 28 -> 13)
    (2, 256, 512, 13, 13, 3, 2, 0),    # critic conv2 13 -> 6
    (2, 512, 1024, 6, 6, 3, 2, 0),     # critic conv3 6 -> 2
    (3, 8192, 1024, 1, 1, 1, 1, 0),    # critic Linear 8192 -> 1024 as a 1x1 conv
    (2, 1024, 1024, 4, 4, 4, 1, 0),    # G ConvT(1024,1024,4,1,0) adjoint
    (2, 512, 1024, 7, 7, 3, 2, 1),     # G ConvT(1024,512,3,2,1) adjoint (4 -> 7)
    (2, 256, 512, 14, 14, 4, 2, 1),    # G ConvT(512,256,4,2,1) adjoint
    (2, 1, 256, 28, 28, 4, 2, 1),      # G ConvT(256,1,4,2,1) adjoint (thin)
    (3, 4, 8, 13, 13, 3, 2, 0), (3, 8, 16, 6, 6, 3, 2, 0), (3, 1, 4, 28, 28, 3, 2, 0),   # reduced-width golden critic
    (3, 8, 16, 7, 7, 3, 2, 1), (3, 4, 8, 14, 14, 4, 2, 1), (3, 1, 4, 28, 28, 4, 2, 1), (3, 16, 16, 4, 4, 4, 1, 0),  # reduced G
    # ragged / tiny: partial tiles in M, N and K
    (1, 4, 4, 5, 5, 3, 1, 1),
    (2, 36, 20, 6, 10, 3, 2, 1),
    (5, 8, 8, 64, 64, 4, 2, 1),        # reduced-width golden nets (g_hidden=8)
]


def _conv_ref(B, Cin, Cout, H, W, k, s, p, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    return x, w, b, y, dy


def _tol(K, scale):
    return 2e-6 * math.sqrt(K) * scale + 1e-6


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,s,p", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(pcg, B, Cin, Cout, H, W, k, s, p):
    ops = pcg.ops
    x, w, b, y, dy = _conv_ref(B, Cin, Cout, H, W, k, s, p, seed=B * 131 + Cin)
    g = ops.conv_geom(B, H, W, Cin, Cout, k, k, s, p)
    xd = nhwc(x.detach()).float().to(dev())
    wd = nhwc(w.detach()).float().to(dev())          # OHWI
    bd = b.float().to(dev())
    dyd = nhwc(dy).float().to(dev())

    yd = ops.conv2d_fwd(g, xd, wd, bd)
    err = (yd.cpu().double() - nhwc(y.detach())).abs().max().item()
    assert err <= _tol(Cin * k * k, 4.0), f"fwd max err {err}"

    dxd = ops.conv2d_dgrad(g, dyd, wd)
    err = (dxd.cpu().double() - nhwc(x.grad)).abs().max().item()
    assert err <= _tol(Cout * k * k, 4.0), f"dgrad max err {err}"

    dwd = torch.full((Cout, k, k, Cin), 0.5, dtype=torch.float32, device=dev())
    ops.conv2d_wgrad(g, xd, dyd, dwd, accumulate=False)
    ref = nhwc(w.grad)
    K = B * y.shape[2] * y.shape[3]
    err = (dwd.cpu().double() - ref).abs().max().item()
    assert err <= _tol(K, 8.0), f"wgrad max err {err}"
    # accumulate: second call adds into dw (mnist_dcgan.py:153,161 accumulate .grad over two backward calls)
    ops.conv2d_wgrad(g, xd, dyd, dwd, accumulate=True)
    err = (dwd.cpu().double() - 2 * ref).abs().max().item()
    assert err <= 2 * _tol(K, 8.0), f"wgrad accumulate max err {err}"


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,s,p", [(4, 64, 128, 32, 32, 4, 2, 1), (3, 64, 64, 28, 28, 3, 1, 1), (8, 1, 64, 64, 64, 4, 2, 1),
                                                   (8, 512, 1, 4, 4, 4, 1, 0), (3, 2, 64, 28, 28, 3, 2, 1), (3, 64, 1, 28, 28, 3, 1, 1)])
@pytest.mark.parametrize("act,slope", [(O.ACT_LRELU, 0.2), (O.ACT_TANH, 0.0), (O.ACT_SIGMOID, 0.0), (O.ACT_RELU, 0.0)])
def test_conv_fused_activation_equals_conv_then_activation(pcg, B, Cin, Cout, H, W, k, s, p, act, slope):
    """pcg_conv2d_{fwd,dgrad}_act == the plain op followed by pcg_act_fwd, bit for bit (same arithmetic, one pass less)."""
    ops = pcg.ops
    g = ops.conv_geom(B, H, W, Cin, Cout, k, k, s, p)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, W, Cin, generator=gen).to(dev()); w = (torch.randn(Cout, k, k, Cin, generator=gen) * 0.1).to(dev())
    dy = torch.randn(B, g.OH, g.OW, Cout, generator=gen).to(dev())
    b_out, b_in = torch.randn(Cout, generator=gen).to(dev()), torch.randn(Cin, generator=gen).to(dev())
    assert torch.equal(ops.conv2d_fwd(g, x, w, b_out, act=act, slope=slope), ops.act_fwd(ops.conv2d_fwd(g, x, w, b_out), act, slope))
    assert torch.equal(ops.conv2d_dgrad(g, dy, w, b_in, act=act, slope=slope), ops.act_fwd(ops.conv2d_dgrad(g, dy, w, b_in), act, slope))


@pytest.mark.parametrize("B", [8, 37])
@pytest.mark.parametrize("act,slope", [(O.ACT_LRELU, 0.2), (O.ACT_RELU, 0.0), (O.ACT_TANH, 0.0)])
def test_full_window_grad_input_with_activation(pcg, B, act, slope):
    """The outer-product grad-input of a full-window Cout = 1 layer (no bias) with the activation in the write == plain + pcg_act_fwd."""
    ops = pcg.ops
    g = ops.conv_geom(B, 4, 4, 512, 1, 4, 4, 1, 0)
    gen = torch.Generator().manual_seed(11)
    w = (torch.randn(1, 4, 4, 512, generator=gen) * 0.1).to(dev()); dy = torch.randn(B, 1, 1, 1, generator=gen).to(dev())
    plain = ops.conv2d_dgrad(g, dy, w)
    assert torch.equal(plain, (dy.reshape(B, 1, 1, 1) * w).reshape(B, 4, 4, 512))          # one multiply per element: exact
    assert torch.equal(ops.conv2d_dgrad(g, dy, w, None, act=act, slope=slope), ops.act_fwd(plain, act, slope))


@pytest.mark.parametrize("B,groups", [(32, 1), (64, 2), (48, 3)])
@pytest.mark.parametrize("act,slope", [(O.ACT_LRELU, 0.2), (O.ACT_RELU, 0.0)])
def test_full_window_grad_input_through_batchnorm_backward(pcg, B, groups, act, slope):
    """pcg_conv2d_dgrad_bnbwd_full == conv2d_dgrad followed by the BatchNorm + activation backward, group by group (the column sums are
    fp64 in both; only their order differs)."""
    ops = pcg.ops
    C, Bg = 512, B // groups
    g = ops.conv_geom(B, 4, 4, C, 1, 4, 4, 1, 0)
    assert ops.full_dgrad_bn_bwd_ok(g, groups) and not ops.full_dgrad_bn_bwd_ok(ops.conv_geom(B + 1, 4, 4, C, 1, 4, 4, 1, 0), 1)
    gen = torch.Generator().manual_seed(5 + B)
    z = (torch.randn(B, 4, 4, C, generator=gen) * 1.3 + 0.2).to(dev()); w = (torch.randn(1, 4, 4, C, generator=gen) * 0.1).to(dev())
    dy = torch.randn(B, 1, 1, 1, generator=gen).to(dev())
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=gen)).to(dev()), (0.1 * torch.randn(C, generator=gen)).to(dev())
    zg = z.reshape(groups, Bg * 16, C)
    mean = zg.mean(1).contiguous(); invstd = (1.0 / torch.sqrt(zg.var(1, unbiased=False) + 1e-5)).contiguous()
    dg_ref, db_ref = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    ref = []
    for k in range(groups):
        gk = ops.conv_geom(Bg, 4, 4, C, 1, 4, 4, 1, 0)
        d = ops.conv2d_dgrad(gk, dy[k * Bg:(k + 1) * Bg].contiguous(), w)
        ref.append(ops.bn_act_bwd(d, z[k * Bg:(k + 1) * Bg].contiguous(), None, C, mean[k].contiguous(), invstd[k].contiguous(), gamma, act, slope,
                                  dg_ref, db_ref, k > 0, beta=beta))
    ref = torch.cat(ref)
    dg, db = torch.full((C,), 7.0, device=dev()), torch.full((C,), -7.0, device=dev())
    out = ops.full_dgrad_bn_bwd(g, dy, w, z, mean, invstd, gamma, beta, act, slope, dg, db, False, groups=groups)
    scale = ref.abs().max().item()
    assert (out - ref).abs().max().item() <= 2e-6 * scale
    np.testing.assert_allclose(dg.cpu().numpy(), dg_ref.cpu().numpy(), rtol=2e-5, atol=2e-6 * dg_ref.abs().max().item())
    np.testing.assert_allclose(db.cpu().numpy(), db_ref.cpu().numpy(), rtol=2e-5, atol=2e-6 * db_ref.abs().max().item())
    assert torch.equal(ops.full_dgrad_bn_bwd(g, dy, w, z, mean, invstd, gamma, beta, act, slope, None, None, False, groups=groups), out)   # no parameter gradients wanted
    # accumulate: a second call adds into dgamma / dbeta
    ops.full_dgrad_bn_bwd(g, dy, w, z, mean, invstd, gamma, beta, act, slope, dg, db, True, groups=groups)
    np.testing.assert_allclose(dg.cpu().numpy(), 2 * dg_ref.cpu().numpy(), rtol=2e-5, atol=4e-6 * dg_ref.abs().max().item())


@pytest.mark.parametrize("B", [4, 64])
@pytest.mark.parametrize("transposed", [True, False])
def test_bnsum_epilogue_without_an_addend(pcg, B, transposed):
    """conv2d_dgrad_add(addend=None, bnsum=...) — the head of a skip chain: the plain grad-input with the next BatchNorm's backward
    column sums — == the same call with an all-zero addend, bit for bit (result and partial rows), in both epilogue forms (small and
    large launches take different kernels)."""
    ops = pcg.ops
    C, H = 64, 28
    g = ops.conv_geom(B, H, H, C, C, 3, 3, 1, 1)
    gen = torch.Generator().manual_seed(B)
    dy = torch.randn(B, H, H, C, generator=gen).to(dev()); w = (torch.randn(C, 3, 3, C, generator=gen) * 0.05).to(dev())
    z = (torch.randn(B, H, H, C, generator=gen) * 1.2 + 0.1).to(dev())
    mean, invstd = ops.bn_train_stats(z, C, 1e-5, 0.1)
    ga, wa = (ops.adjoint_geom(g), ops.conv_weight_adjoint(w)) if transposed else (g, w)
    d0, p0, n0 = ops.conv2d_dgrad_add(ga, dy, wa, torch.zeros(B, H, H, C, device=dev()), bnsum=(z, mean, invstd, 0.1), transposed=transposed)
    out = torch.full((B, H, H, C), float("nan"), device=dev())        # whatever the output buffer held must not leak into the result
    d1, p1, n1 = ops.conv2d_dgrad_add(ga, dy, wa, None, out=out, bnsum=(z, mean, invstd, 0.1), transposed=transposed)
    assert n0 == n1 and torch.equal(d0, d1)
    rows = 2 * C * n0
    assert torch.equal(p0.view(torch.float64)[:rows], p1.view(torch.float64)[:rows])
    ref = ops.conv2d_dgrad(g, dy, w)
    assert (d1 - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_deferred_slab_reductions_are_bit_identical(pcg):
    """ops.slab_reductions_deferred(): the weight gradients of a sweep reduced in one launch == reduced per call, bit for bit; two sums
    into the same gradient stay in call order; a split-K FORWARD inside the block is not deferred."""
    ops = pcg.ops
    lib = pcg._lib.load()
    gen = torch.Generator().manual_seed(21)
    cases = [(16, 128, 256, 16, 4, 2, 1), (16, 64, 128, 32, 4, 2, 1), (16, 1, 64, 64, 4, 2, 1), (32, 512, 1, 4, 4, 1, 0), (8, 64, 64, 28, 3, 1, 1)]
    data = []
    for B, Cin, Cout, H, k, s, p in cases:
        g = ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)
        x = torch.randn(B, H, H, Cin, generator=gen).to(dev()); dy = torch.randn(B, g.OH, g.OW, Cout, generator=gen).to(dev())
        data.append((g, x, dy, (Cout, k, k, Cin)))
    ref = []
    for g, x, dy, shp in data:
        dw = torch.full(shp, 0.25, device=dev())
        ops.conv2d_wgrad(g, x, dy, dw, False)
        ops.conv2d_wgrad(g, x, dy, dw, True)              # accumulate on top
        ref.append(dw)
    assert lib.pcg_slab_defer_pending() == -1
    out = [torch.full(shp, 0.25, device=dev()) for _, _, _, shp in data]
    with ops.slab_reductions_deferred():
        for (g, x, dy, _), dw in zip(data, out):
            ops.conv2d_wgrad(g, x, dy, dw, False)
        assert lib.pcg_slab_defer_pending() >= 3              # (stream-K weight gradients have no slabs to defer)
        for (g, x, dy, _), dw in zip(data, out):
            ops.conv2d_wgrad(g, x, dy, dw, True)              # same dw again: the first batch is launched before this one is recorded
    assert lib.pcg_slab_defer_pending() == -1
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    check = pcg._lib.check
    check(lib.pcg_slab_defer_begin(None), "begin")
    try:
        with pytest.raises(pcg.PcgError, match="already deferring"):
            check(lib.pcg_slab_defer_begin(None), "begin")
    finally:
        check(lib.pcg_slab_defer_flush(None), "flush")


def test_conv_rejects_bad_geometry(pcg):
    ops = pcg.ops
    g = ops.conv_geom(2, 8, 8, 8, 8, 4, 4, 2, 1)
    g.OH = 5  # inconsistent
    x = torch.zeros(2, 8, 8, 8, device=dev()); w = torch.zeros(8, 4, 4, 8, device=dev())
    with pytest.raises(pcg.PcgError, match="inconsistent"):
        ops.conv2d_fwd(g, x, w, None, out=torch.zeros(2, 5, 5, 8, device=dev()))
    with pytest.raises(pcg.PcgError, match="GPU"):
        ops.act_fwd(torch.zeros(4), ops.ACT_RELU)


@pytest.mark.parametrize("rows,C", [(4 * 16 * 16, 128), (4 * 8 * 8, 256), (6 * 28 * 28, 64), (37, 10), (1000, 512), (5, 8)])
@pytest.mark.parametrize("act,slope", [(O.ACT_RELU, 0.0), (O.ACT_LRELU, 0.2)])
def test_batchnorm_fwd_bwd(pcg, rows, C, act, slope):
    ops = pcg.ops
    rng = np.random.default_rng(rows + C)
    x = (rng.standard_normal((rows, C)) * 1.5 + 0.3).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C)).astype(np.float32)
    dy = rng.standard_normal((rows, C)).astype(np.float32)
    rm0, rv0 = rng.standard_normal(C).astype(np.float32), (1 + rng.random(C)).astype(np.float32)
    mean, invstd, nrm, nrv = O.bn_train_stats(x, 1e-5, 0.1, rm0, rv0)
    y = O.bn_apply_act(x, mean, invstd, gamma, beta, act, slope)
    dx, dg, db = O.bn_act_bwd(dy, x, y, mean, invstd, gamma, act, slope)

    d = dev()
    xd, gd, bd, dyd = (torch.from_numpy(a).to(d) for a in (x, gamma, beta, dy))
    rm, rv = torch.from_numpy(rm0).to(d), torch.from_numpy(rv0).to(d)
    nbt = torch.zeros(1, dtype=torch.int64, device=d)
    md, isd = ops.bn_train_stats(xd, C, 1e-5, 0.1, rm, rv, nbt)
    np.testing.assert_allclose(md.cpu().numpy(), mean, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(isd.cpu().numpy(), invstd, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rm.cpu().numpy(), nrm, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), nrv, rtol=1e-5, atol=1e-6)
    assert nbt.item() == 1
    yd = ops.bn_apply_act(xd, C, md, isd, gd, bd, act, slope)
    np.testing.assert_allclose(yd.cpu().numpy(), y, rtol=2e-5, atol=2e-6)
    dgd = torch.full((C,), 1.0, device=d); dbd = torch.full((C,), -1.0, device=d)
    dxd = ops.bn_act_bwd(dyd, xd, yd, C, md, isd, gd, act, slope, dgd, dbd, accumulate=True)
    scale = math.sqrt(rows)
    np.testing.assert_allclose(dgd.cpu().numpy() - 1.0, dg, rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(dbd.cpu().numpy() + 1.0, db, rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(dxd.cpu().numpy(), dx, rtol=1e-4, atol=2e-5)
    # eval mode: running statistics, invstd = rsqrt(var + eps) inside the kernel
    ye = ops.bn_apply_act(xd, C, rm, rv, gd, bd, act, slope, var_eps=1e-5)
    ref = O.bn_apply_act(x, nrm, 1.0 / np.sqrt(nrv + 1e-5), gamma, beta, act, slope)
    np.testing.assert_allclose(ye.cpu().numpy(), ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("act,slope", [(O.ACT_RELU, 0.0), (O.ACT_LRELU, 0.2), (O.ACT_TANH, 0.0), (O.ACT_SIGMOID, 0.0)])
@pytest.mark.parametrize("n", [4096 * 64, 1001])
def test_activations(pcg, act, slope, n):
    ops = pcg.ops
    rng = np.random.default_rng(n + act)
    x = (rng.standard_normal(n) * 2).astype(np.float32)
    dy = rng.standard_normal(n).astype(np.float32)
    y = O.act_fwd(x, act, slope)
    xd, dyd = torch.from_numpy(x).to(dev()), torch.from_numpy(dy).to(dev())
    yd = ops.act_fwd(xd, act, slope)
    np.testing.assert_allclose(yd.cpu().numpy(), y, rtol=2e-6, atol=2e-7)
    dxd = ops.act_bwd(dyd, yd, act, slope)
    np.testing.assert_allclose(dxd.cpu().numpy(), dy * O.act_grad_from_out(yd.cpu().numpy(), act, slope), rtol=2e-6, atol=2e-7)
    # in place (nn.ReLU(True) / LeakyReLU(inplace=True) in the reference)
    ops.act_fwd(xd, act, slope, out=xd)
    assert torch.equal(xd, yd)


@pytest.mark.parametrize("n", [512, 1000, 7])
def test_bce_losses(pcg, n):
    ops = pcg.ops
    rng = np.random.default_rng(n)
    p = rng.random(n).astype(np.float32)
    p[0] = 0.0; p[-1] = 1.0  # saturated: the -100 clamp and the 1e-12 denominator clamp
    z = (rng.standard_normal(n) * 5).astype(np.float32)
    for t in (0.0, 1.0):
        l, gr = O.bce(p, t)
        ld, gd = ops.bce_fwd_bwd(torch.from_numpy(p).to(dev()), None, t)
        np.testing.assert_allclose(ld.item(), l, rtol=2e-6)
        np.testing.assert_allclose(gd.cpu().numpy(), gr, rtol=2e-5, atol=1e-9)
        l, gr = O.bce_with_logits(z, t)
        ld, gd = ops.bce_logits_fwd_bwd(torch.from_numpy(z).to(dev()), t)
        np.testing.assert_allclose(ld.item(), l, rtol=2e-6)
        # sigmoid(z) - t cancels in fp32 when |z| is large: absolute floor of one fp32 ulp of 1.0, over n
        np.testing.assert_allclose(gd.cpu().numpy(), gr, rtol=2e-5, atol=1.2e-7 / n)
    tt = (rng.random(n) > 0.5).astype(np.float32)
    l, gr = O.bce(p, tt)
    ld, gd = ops.bce_fwd_bwd(torch.from_numpy(p).to(dev()), torch.from_numpy(tt).to(dev()), 0.0)
    np.testing.assert_allclose(ld.item(), l, rtol=2e-6)


@pytest.mark.parametrize("betas,wd,decoupled", [((0.5, 0.999), 0.0, False), ((0.9, 0.999), 0.0, False), ((0.0, 0.9), 0.01, True)])
@pytest.mark.parametrize("n", [4096, 1003])
def test_adam(pcg, betas, wd, decoupled, n):
    ops = pcg.ops
    rng = np.random.default_rng(n)
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n); v = np.zeros(n); pr = p.astype(np.float64)
    d = dev()
    pd = torch.from_numpy(p.copy()).to(d); md = torch.zeros(n, device=d); vd = torch.zeros(n, device=d)
    pd2 = pd.clone(); md2 = torch.zeros(n, device=d); vd2 = torch.zeros(n, device=d)
    step_dev = torch.zeros(1, dtype=torch.int64, device=d); hyper = torch.zeros(12, device=d)
    for step in range(1, 5):
        g = rng.standard_normal(n).astype(np.float32)
        gd = torch.from_numpy(g).to(d)
        pr, m, v = O.adam_step(pr, g, m, v, step, 2e-4, betas[0], betas[1], 1e-8, wd, decoupled)
        ops.adam_step(pd, gd, md, vd, 2e-4, betas[0], betas[1], 1e-8, wd, decoupled, step)
        ops.adam_step_capturable(pd2, gd, md2, vd2, 2e-4, betas[0], betas[1], 1e-8, wd, decoupled, step_dev, hyper)
    np.testing.assert_allclose(pd.cpu().numpy(), pr, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(md.cpu().numpy(), m, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(vd.cpu().numpy(), v, rtol=1e-5, atol=1e-9)
    assert step_dev.item() == 4
    assert torch.equal(pd, pd2) and torch.equal(md, md2) and torch.equal(vd, vd2)


def test_fill_sumsq_colsum(pcg):
    ops = pcg.ops
    d = dev()
    t = torch.empty(1003, device=d)
    ops.fill(t, 0.25)
    assert torch.equal(t.cpu(), torch.full((1003,), 0.25))
    out = torch.zeros(1, device=d)
    ops.sumsq(t, out)
    np.testing.assert_allclose(out.item(), 1003 * 0.0625, rtol=1e-6)
    rng = np.random.default_rng(0)
    for rows, C in [(3 * 28 * 28, 64), (100, 1), (777, 10)]:
        a = rng.standard_normal((rows, C)).astype(np.float32)
        db = torch.ones(C, device=d)
        ops.colsum(rows, C, torch.from_numpy(a).to(d), db, accumulate=True)
        np.testing.assert_allclose(db.cpu().numpy() - 1.0, a.astype(np.float64).sum(0), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,s,p", [(2, 32, 512, 9, 9, 3, 2, 1), (3, 16, 512, 6, 6, 3, 2, 0), (2, 8, 640, 11, 7, 3, 2, 1)])
def test_dgrad_gemm_col2im_path_equals_phase_kernel(pcg, B, Cin, Cout, H, W, k, s, p):
    """Kernel sizes that are not a multiple of the stride (k3 s2: the WGAN-GP critic) take the 'one GEMM + col2im' grad-input when
    the caller provides pcg_conv2d_dgrad_workspace_bytes of scratch, and the sub-pixel-phase kernel without it.  Same MACs, taps
    summed per output pixel in a different association: both agree with each other and with torch's conv_transpose2d
    (2e-6 * sqrt(K) * scale, the tolerance of the other conv tests), with a ConvTranspose bias and a fused ReLU on top."""
    import ctypes
    from pcgan_amd import _lib
    ops = pcg.ops
    g = ops.conv_geom(B, H, W, Cin, Cout, k, k, s, p)
    gen = torch.Generator().manual_seed(B * 100 + Cin)
    w = torch.randn(Cout, k, k, Cin, generator=gen) * 0.1
    dy = torch.randn(B, g.OH, g.OW, Cout, generator=gen)
    b_in = torch.randn(Cin, generator=gen)
    lib = _lib.load()
    wd, dyd, bd = w.to(dev()), dy.to(dev()), b_in.to(dev())
    ops.tune("dgrad_gemm", 1)            # (r03: geometries where the phase form has clearly fewer MACs otherwise keep the phase form)
    try:
        assert lib.pcg_conv2d_dgrad_workspace_bytes(ctypes.byref(g)) == B * g.OH * g.OW * k * k * Cin * 4
        got_gemm = ops.conv2d_dgrad(g, dyd, wd, bd, act=O.ACT_RELU)                   # workspace given -> GEMM + col2im
    finally:
        ops.tune("dgrad_gemm", -1)
    got_phase = torch.empty_like(got_gemm)
    _lib.check(lib.pcg_conv2d_dgrad_act(ctypes.byref(g), ops._p(dyd), ops._p(wd), ops._p(bd), O.ACT_RELU, 0.0, ops._p(got_phase), None, 0,
                                        ops._stream()), "pcg_conv2d_dgrad_act")          # no workspace -> phase kernel
    ref = torch.nn.functional.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b_in.double(), stride=s,
                                               padding=p, output_padding=(H + 2 * p - k) % s if H == W else 0)
    if ref.shape[2:] != (H, W):   # output_padding per axis
        ref = torch.nn.functional.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b_in.double(), stride=s,
                                                   padding=p, output_padding=((H + 2 * p - k) % s, (W + 2 * p - k) % s))
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    tol = _tol(Cout * k * k, 4.0)
    assert (got_gemm.cpu().double() - ref).abs().max().item() <= tol
    assert (got_phase.cpu().double() - ref).abs().max().item() <= tol
    assert (got_gemm - got_phase).abs().max().item() <= tol


def test_large_partial_row_counts_two_level_finalize(pcg):
    """>= 4096 conv-epilogue partial rows (one per 64 output rows) take the two-level fp64 finalize.  At M = 64*64*64 = 262144 rows
    (4096 partial rows): BatchNorm statistics out of the conv epilogue vs the separate colreduce pass over the conv output, and the
    fused backward (mask + column sums in the grad-input epilogue, then pcg_bn_bwd_partial) vs grad-input + pcg_bn_act_bwd_premask."""
    ops = pcg.ops
    B, C, H = 64, 64, 64
    g = ops.conv_geom(B, H, H, C, C, 3, 3, 1, 1)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, H, C, generator=gen).to(dev())
    w = (torch.randn(C, 3, 3, C, generator=gen) * 0.05).to(dev())
    rm, rv, nbt = torch.zeros(C, device=dev()), torch.ones(C, device=dev()), torch.zeros((), dtype=torch.int64, device=dev())
    z, mean, invstd = ops.conv_bn_train(g, x, w, None, False, 1e-5, 0.1, rm, rv, nbt)
    z2 = ops.conv2d_fwd(g, x, w, None)
    assert torch.equal(z, z2)
    rm2, rv2, nbt2 = torch.zeros(C, device=dev()), torch.ones(C, device=dev()), torch.zeros((), dtype=torch.int64, device=dev())
    mean2, invstd2 = ops.bn_train_stats(z2, C, 1e-5, 0.1, rm2, rv2, nbt2)
    zd = z.double()
    mref = zd.mean(dim=(0, 1, 2)); iref = 1.0 / torch.sqrt(zd.var(dim=(0, 1, 2), unbiased=False) + 1e-5)
    for got_m, got_i in ((mean, invstd), (mean2, invstd2)):
        assert (got_m.double() - mref).abs().max().item() <= 1e-6 and ((got_i.double() - iref).abs() / iref).max().item() <= 1e-6
    assert torch.allclose(rm, rm2, rtol=0, atol=1e-7) and torch.allclose(rv, rv2, rtol=1e-6, atol=0) and int(nbt) == 1
    # backward: the layer above is the same conv; its grad-input carries the LeakyReLU mask and the column sums of this BatchNorm
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).to(dev()), (torch.randn(C, generator=gen) * 0.1).to(dev())
    dz_up = torch.randn(B, H, H, C, generator=gen).to(dev())
    res = ops.conv_bwd_data_fused(g, dz_up, w, False, O.ACT_LRELU, 0.2, z_below=z, bn=(mean, invstd, gamma, beta))
    assert res is not None and res[2] >= 4096
    dg, db = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    dz_f = ops.bn_bwd_partial(res[0], z, C, mean, invstd, gamma, res[1], res[2], dg, db, False)
    d_plain = ops.conv2d_dgrad(g, dz_up, w)
    dg2, db2 = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    dz_s = ops.bn_act_bwd(d_plain, z, None, C, mean, invstd, gamma, O.ACT_LRELU, 0.2, dg2, db2, False, beta=beta)
    for a, b_, what in ((dg, dg2, "dgamma"), (db, db2, "dbeta"), (dz_f, dz_s, "dz")):
        den = max(b_.double().norm().item(), 1e-30)
        assert ((a.double() - b_.double()).norm().item() / den) <= 2e-6, what


# ---- input transforms: BatchNorm + ReLU / LeakyReLU of the producing layer applied inside the consumer's gathers -----------------
XF_CASES = [
    # B, Cin, Cout, H, W, k, s, p
    (4, 64, 128, 32, 32, 4, 2, 1),     # DCGAN D2->D3 shape family (128x128 tile)
    (4, 128, 64, 16, 16, 4, 2, 1),     # N = 64 tile
    (3, 64, 64, 28, 28, 3, 1, 1),      # 3x3 s1 (padding on all sides: padded taps must stay zero AFTER the transform)
    (2, 36, 20, 6, 10, 3, 2, 1),       # ragged: K tail (Cin = 36 -> second k-tile has 4 live channels), partial M / N tiles
    (3, 128, 256, 7, 7, 3, 2, 1),      # odd extent: unequal sub-pixel phases in the grad-input kernel
    (2, 8192, 1024, 1, 1, 1, 1, 0),    # split-K forward (long K, few tiles)
]


@pytest.mark.parametrize("act,slope", [(2, 0.2), (1, 0.0), (0, 0.0)])
@pytest.mark.parametrize("B,Cin,Cout,H,W,k,s,p", XF_CASES)
def test_input_transform_equals_bn_apply_then_conv(pcg, B, Cin, Cout, H, W, k, s, p, act, slope):
    """pcg_conv2d_{fwd,dgrad,wgrad}_xf read act(z*scale + shift) inside the gather.  scale / shift come from the statistics
    finalize with the same expression pcg_bn_apply_act evaluates, so every result is BIT-identical to materialising
    a = act(bn(z)) first and running the plain kernel on it."""
    ops = pcg.ops
    g = torch.Generator(device="cuda:0").manual_seed(B * 1000 + Cin + k)
    geom = ops.conv_geom(B, H, W, Cin, Cout, k, k, s, p)
    w = torch.randn(Cout, k, k, Cin, generator=g, device="cuda:0") / math.sqrt(Cin * k * k)
    # (1) the transformed operand is x (Conv2d forward + weight gradient)
    zx = torch.randn(B, H, W, Cin, generator=g, device="cuda:0") * 1.5 + 0.3
    gam, bet = torch.rand(Cin, generator=g, device="cuda:0") + 0.5, torch.randn(Cin, generator=g, device="cuda:0") * 0.3
    mean, invstd, coef = ops.bn_train_stats(zx, Cin, 1e-5, 0.1, gamma=gam, beta=bet)
    a = ops.bn_apply_act(zx, Cin, mean, invstd, gam, bet, act, slope)
    xf = ops.InputXform(coef, act, slope)
    dy = torch.randn(B, geom.OH, geom.OW, Cout, generator=g, device="cuda:0")
    assert torch.equal(ops.conv2d_fwd(geom, zx, w, xf=xf), ops.conv2d_fwd(geom, a, w))
    assert torch.equal(ops.conv2d_fwd(geom, zx, w, act=2, slope=0.2, xf=xf), ops.conv2d_fwd(geom, a, w, act=2, slope=0.2))
    dw1, dw2 = torch.empty_like(w), torch.empty_like(w)
    ops.conv2d_wgrad(geom, zx, dy, dw1, False, xf_x=xf)
    ops.conv2d_wgrad(geom, a, dy, dw2, False)
    assert torch.equal(dw1, dw2)
    # fused statistics + coefficient output of the consumer itself
    gam2, bet2 = torch.rand(Cout, generator=g, device="cuda:0") + 0.5, torch.randn(Cout, generator=g, device="cuda:0") * 0.3
    r1 = ops.conv_bn_train(geom, zx, w, None, False, 1e-5, 0.1, None, None, None, xf=xf, gamma=gam2, beta=bet2)
    r2 = ops.conv_bn_train(geom, a, w, None, False, 1e-5, 0.1, None, None, None)
    assert len(r1) == 4 and all(torch.equal(u, v) for u, v in zip(r1[:3], r2))
    sc = gam2 * r1[2]
    assert torch.equal(r1[3][:Cout], sc) and torch.allclose(r1[3][Cout:], bet2 - r1[1] * sc, rtol=1e-6, atol=1e-7)
    # (2) the transformed operand is dy (ConvTranspose2d forward = grad-input kernel, and its weight gradient)
    zy = torch.randn(B, geom.OH, geom.OW, Cout, generator=g, device="cuda:0") * 0.7 - 0.2
    mean, invstd, coef = ops.bn_train_stats(zy, Cout, 1e-5, 0.1, gamma=gam2, beta=bet2)
    ay = ops.bn_apply_act(zy, Cout, mean, invstd, gam2, bet2, act, slope)
    xfy = ops.InputXform(coef, act, slope)
    assert torch.equal(ops.conv2d_dgrad(geom, zy, w, xf=xfy), ops.conv2d_dgrad(geom, ay, w))
    xg = torch.randn(B, H, W, Cin, generator=g, device="cuda:0")
    ops.conv2d_wgrad(geom, xg, zy, dw1, False, xf_dy=xfy)
    ops.conv2d_wgrad(geom, xg, ay, dw2, False)
    assert torch.equal(dw1, dw2)
    if s <= 2:
        r1 = ops.conv_bn_train(geom, zy, w, None, True, 1e-5, 0.1, None, None, None, xf=xfy, gamma=gam, beta=bet)
        r2 = ops.conv_bn_train(geom, ay, w, None, True, 1e-5, 0.1, None, None, None)
        assert all(torch.equal(u, v) for u, v in zip(r1[:3], r2))


@pytest.mark.parametrize("B,H,C", [(8, 64, 64), (3, 28, 64), (4, 28, 256)])
@pytest.mark.parametrize("act,slope", [(O.ACT_RELU, 0.0), (O.ACT_LRELU, 0.2)])
def test_input_transform_on_the_one_channel_transposed_layer(pcg, B, H, C, act, slope):
    """pcg_conv2d_dgrad_xf / pcg_conv2d_wgrad_xf(xf_dy) on a Cin = 1, Cout = 64 k geometry (pcg_conv2d_xf_thin_ok: DCGAN's and WGAN-GP's
    last ConvTranspose2d) == the plain calls on bn_apply_act(z): bit for bit."""
    ops = pcg.ops
    g = ops.conv_geom(B, H, H, 1, C, 4, 4, 2, 1)
    assert ops.xform_thin_ok(g) and not ops.xform_thin_ok(ops.conv_geom(B, H, H, 1, 32, 4, 4, 2, 1))
    gen = torch.Generator().manual_seed(B + H + C)
    z = (torch.randn(B, g.OH, g.OW, C, generator=gen) * 1.5 + 0.3).to(dev()); w = (torch.randn(C, 4, 4, 1, generator=gen) * 0.1).to(dev())
    dimg = torch.randn(B, H, H, 1, generator=gen).to(dev())
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=gen)).to(dev()), (0.1 * torch.randn(C, generator=gen)).to(dev())
    mean, invstd, coef = ops.bn_train_stats(z, C, 1e-5, 0.1, gamma=gamma, beta=beta)
    a = ops.bn_apply_act(z, C, mean, invstd, gamma, beta, act, slope)
    xf = ops.InputXform(coef, act, slope)
    assert torch.equal(ops.conv2d_dgrad(g, z, w, None, act=O.ACT_TANH, xf=xf), ops.conv2d_dgrad(g, a, w, None, act=O.ACT_TANH))
    dw1, dw2 = torch.zeros(C, 4, 4, 1, device=dev()), torch.zeros(C, 4, 4, 1, device=dev())
    ops.conv2d_wgrad(g, dimg, z, dw1, False, xf_dy=xf)
    ops.conv2d_wgrad(g, dimg, a, dw2, False)
    assert torch.equal(dw1, dw2)


@pytest.mark.parametrize("B,groups", [(32, 1), (64, 2), (48, 3)])
@pytest.mark.parametrize("act,slope", [(O.ACT_LRELU, 0.2), (O.ACT_RELU, 0.0)])
def test_full_window_layer_reads_the_pre_batchnorm_tensor(pcg, B, groups, act, slope):
    """ops.BnInput: the full-window one-channel convolution's forward and weight gradient on (z, batch statistics per group) == the plain
    calls on bn_apply_act(z) of every group: bit for bit."""
    ops = pcg.ops
    C, Bg = 512, B // groups
    g = ops.conv_geom(B, 4, 4, C, 1, 4, 4, 1, 0)
    assert ops.bnin_full_ok(g, groups) and not ops.bnin_full_ok(ops.conv_geom(B + 8, 4, 4, C, 1, 4, 4, 1, 0), 1)
    gen = torch.Generator().manual_seed(9 + B)
    z = (torch.randn(B, 4, 4, C, generator=gen) * 1.3 + 0.2).to(dev()); w = (torch.randn(1, 4, 4, C, generator=gen) * 0.05).to(dev())
    bias = torch.randn(1, generator=gen).to(dev()); dy = torch.randn(B, 1, 1, 1, generator=gen).to(dev())
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=gen)).to(dev()), (0.1 * torch.randn(C, generator=gen)).to(dev())
    zg = z.reshape(groups, Bg * 16, C)
    mean = zg.mean(1).contiguous(); invstd = (1.0 / torch.sqrt(zg.var(1, unbiased=False) + 1e-5)).contiguous()
    a = torch.cat([ops.bn_apply_act(z[k * Bg:(k + 1) * Bg].contiguous(), C, mean[k].contiguous(), invstd[k].contiguous(), gamma, beta, act, slope)
                   for k in range(groups)])
    bi = ops.BnInput(mean, invstd, gamma, beta, act, slope, groups)
    assert torch.equal(ops.conv2d_fwd(g, z, w, bias, act=O.ACT_SIGMOID, xf=bi), ops.conv2d_fwd(g, a, w, bias, act=O.ACT_SIGMOID))
    dw1, dw2 = torch.full((1, 4, 4, C), 0.5, device=dev()), torch.full((1, 4, 4, C), 0.5, device=dev())
    ops.conv2d_wgrad(g, z, dy, dw1, True, xf_x=bi)
    ops.conv2d_wgrad(g, a, dy, dw2, True)
    assert torch.equal(dw1, dw2)


def test_input_transform_rejected_on_thin_layers(pcg):
    ops = pcg.ops
    geom = ops.conv_geom(2, 8, 8, 64, 1, 4, 4, 2, 1)
    coef = torch.ones(128, device="cuda:0")
    with pytest.raises(pcg.PcgError, match="MFMA path"):
        ops.conv2d_fwd(geom, torch.zeros(2, 8, 8, 64, device="cuda:0"), torch.zeros(1, 4, 4, 64, device="cuda:0"),
                       xf=ops.InputXform(coef, 1, 0.0))


@pytest.mark.parametrize("rows,C,act", [(64 * 28 * 28, 64, 0), (16 * 16 * 16, 128, 2), (1000, 36, 2), (4 * 7 * 7, 20, 0)])
def test_bn_backward_with_fused_bias_colsum(pcg, rows, C, act):
    """pcg_bn_act_bwd_db / pcg_bn_bwd_partial_db: dx is what the plain calls return, bit for bit, and dcol (+)= the column sums of
    that dx (fp64 accumulation, so equal to a float64 reduction of dx to fp32 rounding) — fast path (C/4 a power of two) and the
    generic fallback."""
    ops = pcg.ops
    g = torch.Generator(device="cuda:0").manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g, device="cuda:0") * 1.3 + 0.2
    dy = torch.randn(rows, C, generator=g, device="cuda:0") + 0.05
    gam, bet = torch.rand(C, generator=g, device="cuda:0") + 0.5, torch.randn(C, generator=g, device="cuda:0") * 0.2
    mean, invstd = ops.bn_train_stats(x, C, 1e-5, 0.1)
    dg1, db1, dg2, db2 = (torch.zeros(C, device="cuda:0") for _ in range(4))
    ref = ops.bn_act_bwd(dy, x, None, C, mean, invstd, gam, act, 0.2, dg1, db1, False, beta=bet, dy_scale=0.5)
    dcol = torch.full((C,), 3.0, device="cuda:0")
    got = ops.bn_act_bwd(dy, x, None, C, mean, invstd, gam, act, 0.2, dg2, db2, False, beta=bet, dy_scale=0.5, dcol=dcol, accumulate_col=True)
    assert torch.equal(got, ref) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
    want = ref.double().sum(0)
    np.testing.assert_allclose((dcol - 3.0).cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-6 * float(ref.abs().sum(0).max()) + 1e-6)
    dcol2 = torch.empty(C, device="cuda:0")
    ops.bn_act_bwd(dy, x, None, C, mean, invstd, gam, act, 0.2, dg2, db2, False, beta=bet, dy_scale=0.5, dcol=dcol2, accumulate_col=False)
    np.testing.assert_allclose(dcol2.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-6 * float(ref.abs().sum(0).max()) + 1e-6)


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,p", [(3, 64, 64, 28, 28, 3, 1), (2, 32, 48, 9, 7, 3, 1), (2, 64, 32, 8, 8, 1, 0), (2, 16, 24, 10, 10, 5, 2)])
def test_stride1_grad_input_as_forward_conv_with_adjoint_weight(pcg, B, Cin, Cout, H, W, k, p):
    """pcg_conv_weight_adjoint + the forward kernel on dy (geometry {OHxOW, Cout -> IHxIW, Cin, pad K-1-p}) is the grad-input of a
    stride-1 convolution: compared with the grad-input kernel (same products, another summation order) and with float64."""
    ops = pcg.ops
    g = torch.Generator(device="cuda:0").manual_seed(B + Cin + k)
    geom = ops.conv_geom(B, H, W, Cin, Cout, k, k, 1, p)
    w = torch.randn(Cout, k, k, Cin, generator=g, device="cuda:0") / math.sqrt(Cout * k * k)
    dy = torch.randn(B, geom.OH, geom.OW, Cout, generator=g, device="cuda:0")
    wa = ops.conv_weight_adjoint(w)
    assert torch.equal(wa, w.flip(1, 2).permute(3, 1, 2, 0).contiguous())
    ga = ops.adjoint_geom(geom)
    assert (ga.OH, ga.OW, ga.Cout) == (H, W, Cin)
    got = ops.conv2d_fwd(ga, dy, wa)
    ref = ops.conv2d_dgrad(geom, dy, w)
    wt = w.permute(0, 3, 1, 2).double().cpu()
    truth = torch.nn.functional.conv_transpose2d(dy.permute(0, 3, 1, 2).double().cpu(), wt, stride=1, padding=p).permute(0, 2, 3, 1)
    tol = _tol(Cout * k * k, 1.0 / math.sqrt(Cout * k * k)) if "_tol" in globals() else 1e-4
    assert float((got.cpu().double() - truth).abs().max()) <= max(tol, 2e-5)
    assert float((got - ref).abs().max()) <= 2e-5
    addend = torch.randn(B, H, W, Cin, generator=g, device="cuda:0")
    assert torch.equal(ops.conv2d_dgrad_add(ga, dy, wa, addend, transposed=True), got + addend) or \
        float((ops.conv2d_dgrad_add(ga, dy, wa, addend, transposed=True) - (got + addend)).abs().max()) <= 1e-6

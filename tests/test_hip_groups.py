"""GPU parity of the GROUPED discriminator pass (r04): the D step's two netD passes (mnist_dcgan.py:151 on the real batch, :159 on
fake.detach()) run as ONE pass over 2B images — `SequentialConvNet.forward_groups`, include/pcgan_hip.h "grouped batches".

What must hold against two separate passes (the statement-by-statement order of the reference):
  * every forward tensor, both losses, every BatchNorm statistic, the running statistics after the step and the BatchNorm
    parameter gradients are BIT-IDENTICAL (same tiles, same per-64-row partial sums, same fixed finalize order, groups added in
    pass order) wherever the two forms reduce over the same partial rows;
  * the convolution weight gradients differ only by the ORDER of the sum over pixels (one K loop over 2B images instead of two
    results added into .grad): stated tolerance rel-L2 <= 2e-6, max <= 1e-5 of the tensor's max;
  * against the fp32 / fp64 oracle the grouped step meets the same bounds as the ungrouped one.
"""
import numpy as np
import pytest
import torch

from oracle import dcgan_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pcg():
    import pcgan_amd
    from pcgan_amd import dcgan  # noqa: F401
    pcgan_amd.load()
    return pcgan_amd


def _rel_l2(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _nets(pcg, cfg, seed=1):
    D = pcg.dcgan
    refG, refD = R.build(cfg, seed=seed)
    netG, netD = D.Generator(cfg), D.Discriminator(cfg)
    netG.load_state_dict(refG.state_dict()); netD.load_state_dict(refD.state_dict())
    return netG.to(DEV), netD.to(DEV), refG, refD


@pytest.mark.parametrize("B,C,HW,groups", [(256, 128, 16, 2), (512, 256, 8, 2), (128, 64, 32, 2), (96, 128, 16, 3)])
def test_grouped_conv_bn_matches_separate_passes(pcg, B, C, HW, groups):
    """pcg_conv2d_fwd_bn_g + pcg_bn_apply_act_g against `groups` calls of pcg_conv2d_fwd_bn + pcg_bn_apply_act on the slices."""
    ops = pcg.ops
    torch.manual_seed(0)
    Cin = C // 2
    x = torch.randn(groups * B, 2 * HW, 2 * HW, Cin, device=DEV)
    x[B:] = x[B:] * 1.7 + 0.3                                  # the groups have different statistics
    w = torch.randn(C, 4, 4, Cin, device=DEV) * 0.05
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    g_all = ops.conv_geom(groups * B, 2 * HW, 2 * HW, Cin, C, 4, 4, 2, 1)
    g_one = ops.conv_geom(B, 2 * HW, 2 * HW, Cin, C, 4, 4, 2, 1)
    assert ops.group_fwd_ok(g_all, groups)
    rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    z, mean, invstd = ops.conv_bn_train_g(g_all, x, w, None, 1e-5, 0.1, rm, rv, nbt, groups)
    y = ops.bn_apply_act_g(z, C, mean, invstd, gamma, beta, pcg.ops.ACT_LRELU, 0.2, groups)
    rm2, rv2, nbt2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    for k in range(groups):
        zk, mk, ik = ops.conv_bn_train(g_one, x[k * B:(k + 1) * B].contiguous(), w, None, False, 1e-5, 0.1, rm2, rv2, nbt2)
        yk = ops.bn_apply_act(zk, C, mk, ik, gamma, beta, pcg.ops.ACT_LRELU, 0.2)
        if groups == 2:
            assert torch.equal(z[k * B:(k + 1) * B], zk), f"group {k}: conv output differs"
            # same 64-row partial sums, same finalize order whenever neither form takes the two-level finalize: bit-identical
            assert torch.equal(mean[k], mk) and torch.equal(invstd[k], ik), f"group {k}: statistics differ"
            assert torch.equal(y[k * B:(k + 1) * B], yk), f"group {k}: activated output differs"
        else:
            # three groups of 96 images: 576 tiles — the grouped launch takes the stream-K form (another cut of the K sum) where
            # the 192-tile launches of the separate passes do not: equal up to the order of the sum
            K = 16 * Cin
            tol = 16 * 2.0 ** -24 * K * float(x.abs().mean() * w.abs().mean()) * 4 + 1e-6
            assert float((z[k * B:(k + 1) * B] - zk).abs().max()) <= tol
            np.testing.assert_allclose(mean[k].cpu().numpy(), mk.cpu().numpy(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(invstd[k].cpu().numpy(), ik.cpu().numpy(), rtol=1e-5)
    if groups == 2:
        assert torch.equal(rm, rm2) and torch.equal(rv, rv2)
    assert int(nbt.item()) == groups == int(nbt2.item())
    # and against a float64 evaluation of the statistics of the kernel's own output
    for k in range(groups):
        zz = z[k * B:(k + 1) * B].double().reshape(-1, C)
        np.testing.assert_allclose(mean[k].cpu().numpy(), zz.mean(0).cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(invstd[k].cpu().numpy(), (1.0 / torch.sqrt(zz.var(0, unbiased=False) + 1e-5)).cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize("B,Cin,Cout,H", [(256, 64, 128, 32), (512, 256, 512, 8)])
def test_grouped_bn_backward_matches_separate_passes(pcg, B, Cin, Cout, H):
    """Grad-input with the fused BatchNorm-backward epilogue on two groups + pcg_bn_bwd_partial_g, and the unfused
    pcg_bn_act_bwd_premask_g, against the ungrouped calls on each half."""
    ops = pcg.ops
    G = 2
    torch.manual_seed(1)
    g_all = ops.conv_geom(G * B, H, H, Cin, Cout, 4, 4, 2, 1)
    g_one = ops.conv_geom(B, H, H, Cin, Cout, 4, 4, 2, 1)
    assert ops.group_dgrad_ok(g_all, G)
    w = torch.randn(Cout, 4, 4, Cin, device=DEV) * 0.05
    dy = torch.randn(G * B, H // 2, H // 2, Cout, device=DEV)
    zl = torch.randn(G * B, H, H, Cin, device=DEV)
    zl[B:] = zl[B:] * 0.6 - 0.2
    gamma, beta = torch.rand(Cin, device=DEV) + 0.5, torch.randn(Cin, device=DEV) * 0.1
    mean = torch.stack([zl[k * B:(k + 1) * B].reshape(-1, Cin).mean(0) for k in range(G)]).contiguous()
    invstd = torch.stack([1.0 / torch.sqrt(zl[k * B:(k + 1) * B].reshape(-1, Cin).var(0, unbiased=False) + 1e-5) for k in range(G)]).contiguous()
    dg, db = torch.zeros(Cin, device=DEV), torch.zeros(Cin, device=DEV)
    dm, partial, nparts, nph = ops.conv_bwd_data_fused_g(g_all, dy, w, ops.ACT_LRELU, 0.2, zl, (mean, invstd, gamma, beta), G)
    dz = ops.bn_bwd_partial_g(dm, zl, Cin, mean, invstd, gamma, partial, nparts, nph, dg, db, False, G)
    dg2, db2 = torch.zeros(Cin, device=DEV), torch.zeros(Cin, device=DEV)
    for k in range(G):
        sl = slice(k * B, (k + 1) * B)
        r = ops.conv_bwd_data_fused(g_one, dy[sl].contiguous(), w, False, ops.ACT_LRELU, 0.2, z_below=zl[sl].contiguous(),
                                    bn=(mean[k].contiguous(), invstd[k].contiguous(), gamma, beta))
        assert r is not None
        dzk = ops.bn_bwd_partial(r[0], zl[sl].contiguous(), Cin, mean[k].contiguous(), invstd[k].contiguous(), gamma, r[1], r[2], dg2, db2, k > 0)
        assert torch.equal(dm[sl], r[0]), f"group {k}: masked grad-input differs"
        # the ungrouped finalize may take its two-level form (>= 4096 partial rows) where the grouped one sums directly: fp64 sums
        # in another order, 1e-7 at most
        assert _rel_l2(dz[sl], dzk) <= 1e-6, f"group {k}: BatchNorm backward differs"
    assert _rel_l2(dg, dg2) <= 1e-6 and _rel_l2(db, db2) <= 1e-6
    # unfused form (behind a thin layer): premask_g vs the ungrouped premask call per half
    d = torch.randn(G * B, H, H, Cin, device=DEV)
    dg, db = torch.zeros(Cin, device=DEV), torch.zeros(Cin, device=DEV)
    out = ops.bn_act_bwd_g(d, zl, Cin, mean, invstd, gamma, beta, ops.ACT_LRELU, 0.2, dg, db, False, G)
    dg2, db2 = torch.zeros(Cin, device=DEV), torch.zeros(Cin, device=DEV)
    for k in range(G):
        sl = slice(k * B, (k + 1) * B)
        ok = ops.bn_act_bwd(d[sl].contiguous(), zl[sl].contiguous(), None, Cin, mean[k].contiguous(), invstd[k].contiguous(), gamma,
                            ops.ACT_LRELU, 0.2, dg2, db2, k > 0, beta=beta)
        assert torch.equal(out[sl], ok), f"group {k}: unfused BatchNorm backward differs"
    assert torch.equal(dg, dg2) and torch.equal(db, db2)


def test_bce_pair_carries_the_bits_of_two_losses(pcg):
    ops = pcg.ops
    torch.manual_seed(2)
    n = 512
    p = torch.rand(2 * n, device=DEV).clamp(1e-4, 1 - 1e-4)
    loss, _ = ops.bce_pair(p, n, 1.0, 0.0)
    l0, d0 = ops.bce_fwd_bwd(p[:n].contiguous(), None, 1.0)
    l1, d1 = ops.bce_fwd_bwd(p[n:].contiguous(), None, 0.0)
    assert torch.equal(loss[0:1], l0) and torch.equal(loss[1:2], l1) and torch.equal(loss[2], (l0 + l1)[0])
    one = torch.ones(1, device=DEV)
    _, dp = ops.bce_pair(p, n, 1.0, 0.0, need_loss=False, need_grad=True, cotangents=(None, None, one))
    assert torch.equal(dp[:n], d0) and torch.equal(dp[n:], d1)
    # the oracle's loss
    ref = torch.nn.functional.binary_cross_entropy(p[:n].cpu(), torch.ones(n)) + torch.nn.functional.binary_cross_entropy(p[n:].cpu(), torch.zeros(n))
    np.testing.assert_allclose(loss[2].item(), ref.item(), rtol=2e-6)


@pytest.mark.parametrize("cfg,B,exact", [({"g_hidden": 16, "d_hidden": 16, "z_dim": 32}, 32, True), (None, 64, False), (None, 512, True)])
def test_paired_d_step_equals_two_passes(pcg, cfg, B, exact):
    """dcgan.train_step(pair=True) against pair=False with lr = 0 (every kernel runs, the weights stay put — Adam's first update is
    sign-like): losses, D outputs, BatchNorm running statistics bit-identical; BatchNorm parameter gradients
    bit-identical; conv weight gradients within the stated sum-order tolerance; G's gradients bit-identical (the G step is the same
    code either way and D's weights did not move).  exact=False: a small batch at full width, where B and 2B images take different
    launch forms (stream-K / K-slices cut the K sum differently): everything within the sum-order tolerance instead.  B = 512 at
    full width is the bench configuration."""
    D = pcg.dcgan
    c = dict(cfg or {}, lr=0.0)
    netG, netD, _, _ = _nets(pcg, c)
    netG2, netD2, _, _ = _nets(pcg, c)
    if not exact:
        # another cut of a K sum moves a pre-activation that lies within rounding of zero across the LeakyReLU kink: ONE such flip
        # is ~1e-3 of a gradient tensor — not an error of either form.  With smooth activations (same kernels,
        # same mask code paths) the two forms must agree to rounding.
        for net in (netD, netD2):
            for m in net.modules():
                if isinstance(m, torch.nn.LeakyReLU):
                    m.negative_slope = 0.99
    assert netD.supports_groups((B, 1, 64, 64), 2)
    crit, optD, optG = D.make_optimizers(netG, netD, c)
    crit2, optD2, optG2 = D.make_optimizers(netG2, netD2, c)
    real, noise = R.synthetic_batch(B, seed=3, config=c)
    real, noise = real.to(DEV), noise.to(DEV)
    o1 = D.train_step(netG, netD, crit, optD, optG, real, noise, c, pair=True)
    o2 = D.train_step(netG2, netD2, crit2, optD2, optG2, real, noise, c, pair=False)
    def same(a, b, what, l2tol=2e-5, mxtol=2e-4):
        if exact:
            assert torch.equal(a, b), what
        elif a.dtype.is_floating_point:
            l2 = _rel_l2(a, b)
            mx = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            assert l2 <= l2tol and mx <= mxtol, f"{what}: rel-L2 {l2:.2e}, max/absmax {mx:.2e}"
        else:
            assert torch.equal(a, b), what
    for k in ("errD_real", "errD_fake", "errG", "out_real", "out_fake", "out_g"):
        same(o1[k], o2[k], k)
    for (n, b1), (_, b2) in zip(netD.named_buffers(), netD2.named_buffers()):
        same(b1, b2, f"D buffer {n}")
    for (n, b1), (_, b2) in zip(netG.named_buffers(), netG2.named_buffers()):
        same(b1, b2, f"G buffer {n}")
    for (n, p1), (_, p2) in zip(netD.named_parameters(), netD2.named_parameters()):
        if p1.dim() == 1:
            same(p1.grad, p2.grad, f"D BatchNorm gradient {n}", 1e-4, 1e-3)
        else:
            l2 = _rel_l2(p1.grad, p2.grad)
            mx = float((p1.grad - p2.grad).abs().max() / p2.grad.abs().max())
            lim = (2e-6, 1e-5) if exact else (1e-4, 1e-3)
            assert l2 <= lim[0] and mx <= lim[1], f"D weight gradient {n}: rel-L2 {l2:.2e}, max/absmax {mx:.2e}"
    for (n, p1), (_, p2) in zip(netG.named_parameters(), netG2.named_parameters()):
        same(p1.grad, p2.grad, f"G gradient {n}", 1e-4, 1e-3)


def test_paired_d_step_against_the_oracle(pcg):
    """The grouped D step against the fp32 oracle (oracle/dcgan_ref.py) at the smoke width: losses and D's gradients, lr = 0."""
    D = pcg.dcgan
    c = {"g_hidden": 16, "d_hidden": 16, "z_dim": 32, "lr": 0.0}
    netG, netD, refG, refD = _nets(pcg, c)
    crit, optD, optG = D.make_optimizers(netG, netD, c)
    rcrit, roptD, roptG = R.make_optimizers(refG, refD, c)
    real, noise = R.synthetic_batch(32, seed=5, config=c)
    out = D.train_step(netG, netD, crit, optD, optG, real.to(DEV), noise.to(DEV), c, pair=True, skip_dead_d_wgrad=False)
    ref = R.dcgan_step(refG, refD, rcrit, roptD, roptG, real, noise, c)
    for k in ("errD_real", "errD_fake", "errG"):
        np.testing.assert_allclose(out[k].item(), ref[k], rtol=2e-5, atol=1e-6, err_msg=k)
    for (n, p), (_, q) in zip(netG.named_parameters(), refG.named_parameters()):
        l2 = _rel_l2(p.grad.cpu(), q.grad)
        assert l2 <= 2e-3, f"G grad {n}: rel-L2 {l2:.2e}"          # batch-32 BatchNorm backward: fp32 noise


def test_paired_step_graph_replay_is_bit_identical_to_eager(pcg):
    D = pcg.dcgan
    from pcgan_amd.nn import GraphedStep
    c = {"g_hidden": 16, "d_hidden": 16, "z_dim": 32}
    netG, netD, _, _ = _nets(pcg, c)
    netG2, netD2, _, _ = _nets(pcg, c)
    crit, optD, optG = D.make_optimizers(netG, netD, c)
    crit2, optD2, optG2 = D.make_optimizers(netG2, netD2, c)
    batches = [R.synthetic_batch(32, seed=10 + i, config=c) for i in range(3)]
    s_real, s_noise = batches[0][0].to(DEV).clone(), batches[0][1].to(DEV).clone()
    gs = GraphedStep(lambda: D.train_step(netG, netD, crit, optD, optG, s_real, s_noise, c), {"real": s_real, "noise": s_noise},
                     [netG, netD], [optD, optG])
    for real, noise in batches:
        gs.load(real=real.to(DEV), noise=noise.to(DEV))
        o1 = gs.replay()
        o2 = D.train_step(netG2, netD2, crit2, optD2, optG2, real.to(DEV), noise.to(DEV), c)
        for k in ("errD_real", "errD_fake", "errG"):
            assert torch.equal(o1[k], o2[k]), k
    assert torch.equal(netD.flat_params, netD2.flat_params) and torch.equal(netG.flat_params, netG2.flat_params)


def test_forward_groups_refuses_what_it_cannot_run(pcg):
    D = pcg.dcgan
    netG, netD, _, _ = _nets(pcg, {"g_hidden": 16, "d_hidden": 16, "z_dim": 32})
    assert not netD.supports_groups((3, 1, 64, 64), 2)           # 3 * 16 * 16 rows per group: not whole 128-row tiles
    assert not netG.supports_groups((8, 32, 1, 1), 2)            # transposed convolutions
    with pytest.raises(pcg.PcgError, match="not eligible"):
        netD.forward_groups([torch.zeros(3, 1, 64, 64, device=DEV)] * 2)
    x = torch.zeros(32, 1, 64, 64, device=DEV, requires_grad=True)
    out = netD.forward_groups([x, x])
    with pytest.raises(Exception, match="inputs are not implemented"):
        out.sum().backward()


@pytest.mark.parametrize("B", [8, 512])
def test_thin_grad_input_through_batchnorm_backward_without_being_written(pcg, B):
    """pcg_conv2d_fwd_bnbwd_thin (DCGAN: G5's grad-input pushed through G4's BatchNorm + ReLU backward, the gradient itself never
    written) against the chain it replaces: conv2d_fwd (thin) -> bn_act_bwd.  Same d per element (same expand code), same mask
    expression; the two column sums are added in another fixed order (fp64): dz / dgamma / dbeta to 1e-6, and against float64."""
    ops = pcg.ops
    torch.manual_seed(4)
    C = 64
    g = ops.conv_geom(B, 64, 64, 1, C, 4, 4, 2, 1)             # adjoint geometry of ConvTranspose2d(64, 1, 4, 2, 1): x = image side
    assert ops.thin_fwd_bn_bwd_ok(g)
    dimg = torch.randn(B, 64, 64, 1, device=DEV)
    w = torch.randn(C, 4, 4, 1, device=DEV) * 0.05
    z = torch.randn(B, 32, 32, C, device=DEV) * 1.3 + 0.2
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    mean = z.reshape(-1, C).mean(0).contiguous()
    invstd = (1.0 / torch.sqrt(z.reshape(-1, C).var(0, unbiased=False) + 1e-5)).contiguous()
    for acc in (False, True):
        dg0, db0 = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        dg, db = dg0.clone(), db0.clone()
        dz = ops.thin_fwd_bn_bwd(g, dimg, w, z, mean, invstd, gamma, beta, ops.ACT_RELU, 0.0, dg, db, acc)
        dg2, db2 = dg0.clone(), db0.clone()
        d = ops.conv2d_fwd(g, dimg, w)
        dz2 = ops.bn_act_bwd(d, z, None, C, mean, invstd, gamma, ops.ACT_RELU, 0.0, dg2, db2, acc, beta=beta)
        assert _rel_l2(dz, dz2) <= 1e-6 and float((dz - dz2).abs().max()) <= 1e-5 * float(dz2.abs().max())
        assert _rel_l2(dg, dg2) <= 1e-6 and _rel_l2(db, db2) <= 1e-6
    # float64 evaluation of the whole chain
    x64 = dimg.double().permute(0, 3, 1, 2).cpu()
    d64 = torch.nn.functional.conv2d(x64, w.double().permute(0, 3, 1, 2).cpu(), stride=2, padding=1).permute(0, 2, 3, 1)
    z64, m64, i64 = z.double().cpu(), mean.double().cpu(), invstd.double().cpu()
    pre = z64 * (gamma.double().cpu() * i64) + (beta.double().cpu() - m64 * gamma.double().cpu() * i64)
    dm = d64 * (pre > 0)
    xh = (z64 - m64) * i64
    n = dm.reshape(-1, C).shape[0]
    want = gamma.double().cpu() * i64 * (dm - dm.reshape(-1, C).sum(0) / n - xh * (dm * xh).reshape(-1, C).sum(0) / n)
    assert _rel_l2(dz.cpu(), want) <= 2e-4      # (batch 512: 5e-5 — the two means are differences of sums over 524288 rows)


def test_weight_gradients_beside_the_batchnorm_backward_are_the_same_numbers(pcg):
    """SequentialConvNet.wgrad_overlap = "bn" (experiment: a layer's weight gradient on a side stream beside the next layer's
    BatchNorm-backward passes) changes the order of launches across two streams, not one number: eager and graph-replayed steps are
    bit-identical to the one-stream schedule."""
    from pcgan_amd.nn import GraphedStep, SequentialConvNet
    D = pcg.dcgan
    c = {"g_hidden": 16, "d_hidden": 16, "z_dim": 32}
    batches = [R.synthetic_batch(32, seed=20 + i, config=c) for i in range(3)]
    res = {}
    for mode in (None, "bn", "bn-graph"):
        SequentialConvNet.wgrad_overlap = "bn" if mode else None
        netG, netD, _, _ = _nets(pcg, c)
        crit, optD, optG = D.make_optimizers(netG, netD, c)
        if mode == "bn-graph":
            s_real, s_noise = batches[0][0].to(DEV).clone(), batches[0][1].to(DEV).clone()
            gs = GraphedStep(lambda: D.train_step(netG, netD, crit, optD, optG, s_real, s_noise, c), {"real": s_real, "noise": s_noise},
                             [netG, netD], [optD, optG])
        for real, noise in batches:
            if mode == "bn-graph":
                gs.load(real=real.to(DEV), noise=noise.to(DEV))
                o = gs.replay()
            else:
                o = D.train_step(netG, netD, crit, optD, optG, real.to(DEV), noise.to(DEV), c)
        torch.cuda.synchronize()
        res[mode] = ([o[k].item() for k in ("errD_real", "errD_fake", "errG")], netG.flat_params.clone(), netD.flat_params.clone())
    SequentialConvNet.wgrad_overlap = None
    for mode in ("bn", "bn-graph"):
        assert res[mode][0] == res[None][0], mode
        assert torch.equal(res[mode][1], res[None][1]) and torch.equal(res[mode][2], res[None][2]), mode


def test_deferred_slab_reductions_in_the_step_are_the_same_numbers(pcg):
    """SequentialConvNet.defer_slab_reductions (opt-in: one slab-reduction launch per backward sweep, pcg_slab_defer_*): eager and
    graph-replayed steps are bit-identical to the per-layer reductions, paired and two-pass D step."""
    from pcgan_amd.nn import GraphedStep, SequentialConvNet
    D = pcg.dcgan
    c = {"g_hidden": 16, "d_hidden": 16, "z_dim": 32}
    batches = [R.synthetic_batch(32, seed=40 + i, config=c) for i in range(3)]
    for pair in (True, False):
        res = {}
        for mode in (None, "defer", "defer-graph"):
            SequentialConvNet.defer_slab_reductions = mode is not None
            netG, netD, _, _ = _nets(pcg, c)
            crit, optD, optG = D.make_optimizers(netG, netD, c)
            if mode == "defer-graph":
                s_real, s_noise = batches[0][0].to(DEV).clone(), batches[0][1].to(DEV).clone()
                gs = GraphedStep(lambda: D.train_step(netG, netD, crit, optD, optG, s_real, s_noise, c, pair=pair),
                                 {"real": s_real, "noise": s_noise}, [netG, netD], [optD, optG])
            for real, noise in batches:
                if mode == "defer-graph":
                    gs.load(real=real.to(DEV), noise=noise.to(DEV))
                    o = gs.replay()
                else:
                    o = D.train_step(netG, netD, crit, optD, optG, real.to(DEV), noise.to(DEV), c, pair=pair)
            torch.cuda.synchronize()
            res[mode] = ([o[k].item() for k in ("errD_real", "errD_fake", "errG")], netG.flat_params.clone(), netD.flat_params.clone())
        SequentialConvNet.defer_slab_reductions = False
        assert pcg._lib.load().pcg_slab_defer_pending() == -1
        for mode in ("defer", "defer-graph"):
            assert res[mode][0] == res[None][0], (pair, mode)
            assert torch.equal(res[mode][1], res[None][1]) and torch.equal(res[mode][2], res[None][2]), (pair, mode)

"""GPU parity, conditional WGAN-GP (conditional_gan/mnist/mnist_wgan_conditional.py): the drop-ins in pcgan_amd.wgan against
  (a) three loop iterations of the reference's own classes and loop body at reduced width (tests/golden/wgan_ref_small.npz),
  (b) the oracle restatement (oracle/wgan_ref.py) evaluated live in float64 (truth) and float32 (noise floor) — including the
      gradient penalty's second-order terms, which torch autograd derives there and pcg_instnorm_bwd_bwd + the conv family
      compute here.
Permutes / interpolation are exact or 1-ulp; floating point within the tolerances stated in each assert."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import wgan_ref as WR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pcg():
    import pcgan_amd
    from pcgan_amd import wgan  # noqa: F401
    return pcgan_amd


def _dev(t):
    return t.to(DEV).contiguous()


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _close(a, b, rtol, atol, msg=""):
    np.testing.assert_allclose(a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy(), rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("B,C,H,W", [(3, 4, 13, 13), (2, 8, 6, 6), (5, 16, 2, 2), (2, 256, 13, 13), (3, 1024, 2, 2), (2, 100, 3, 5),
                                     (256, 1024, 2, 2), (256, 512, 6, 6), (64, 256, 13, 13)])     # the last three: config 3's bench shapes
def test_instance_norm_forward_backward_and_backward_of_backward(pcg, B, C, H, W):
    """pcg_instnorm_{fwd,bwd,bwd_bwd} against torch autograd in float64: y; dx, dgamma, dbeta for a cotangent dy; and, with a
    second cotangent r on dx (create_graph=True), the three cotangents reaching dy, x and gamma."""
    ops = pcg.ops
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = (torch.randn(B, C, H, W, generator=g, dtype=torch.float64) * 1.7 + 0.3).requires_grad_(True)
    gamma = (torch.randn(C, generator=g, dtype=torch.float64) * 0.5 + 1).requires_grad_(True)
    beta = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    dy = torch.randn(B, C, H, W, generator=g, dtype=torch.float64).requires_grad_(True)
    r = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    n = F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5)
    y = F.leaky_relu(n, 0.2)
    dx, dgam, dbet = torch.autograd.grad(n, [x, gamma, beta], dy, create_graph=True)
    ddy, ez, dg2 = torch.autograd.grad(dx, [dy, x, gamma], r)
    HW = H * W
    xd, dyd, rd = _dev(_nhwc(x.detach()).float()), _dev(_nhwc(dy.detach()).float()), _dev(_nhwc(r).float())
    gd, bd = _dev(gamma.detach().float()), _dev(beta.detach().float())
    yd, mean, invstd = ops.instnorm_fwd(xd, B, HW, C, gd, bd, 1e-5, pcg._lib.ACT_LRELU, 0.2)
    _close(yd, _nhwc(y.detach()), 2e-5, 2e-5)
    dxd, dgp, dbp = ops.instnorm_bwd(dyd, xd, B, HW, C, mean, invstd, gd)
    sc = float(dx.detach().abs().max())
    _close(dxd, _nhwc(dx.detach()), 1e-4, 2e-5 * sc, "dx")
    _close(dgp.sum(0), dgam.detach(), 1e-4, 2e-5 * float(dgam.detach().abs().max()) + 1e-5, "dgamma")
    _close(dbp.sum(0), dbet.detach(), 1e-4, 2e-5 * float(dbet.detach().abs().max()) + 1e-5, "dbeta")
    ddyd, ezd, dg2p = ops.instnorm_bwd_bwd(rd, dyd, xd, B, HW, C, mean, invstd, gd)
    _close(ddyd, _nhwc(ddy), 1e-4, 2e-5 * float(ddy.abs().max()), "ddy")
    _close(ezd, _nhwc(ez), 2e-4, 5e-5 * float(ez.abs().max()), "ez")
    _close(dg2p.sum(0), dg2, 2e-4, 5e-5 * float(dg2.abs().max()) + 1e-5, "dgamma (second order)")
    # optional outputs
    only_ez = ops.instnorm_bwd_bwd(rd, dyd, xd, B, HW, C, mean, invstd, gd, need_ddy=False, need_gamma=False)
    assert only_ez[0] is None and only_ez[2] is None and torch.equal(only_ez[1], ezd)


@pytest.mark.parametrize("B,C,H,W", [(3, 8, 13, 13), (5, 64, 6, 6), (256, 1024, 2, 2), (64, 256, 13, 13), (2, 6, 3, 5)])
def test_fused_stage_backward_equals_the_chain_it_replaces(pcg, B, C, H, W):
    """pcg_instnorm_bwd_fused (LeakyReLU' -> InstanceNorm' -> + addend, per-sample partials of dgamma / dbeta / the conv bias gradient),
    pcg_rowsum3 and the masked ddy of pcg_instnorm_bwd_bwd_act against float64 autograd of the chain
    z -> InstanceNorm -> LeakyReLU (mnist_wgan_conditional.py:88-95) — what the critic's backward sweeps launch per stage."""
    ops = pcg.ops
    g = torch.Generator().manual_seed(B * 77 + C)
    z = (torch.randn(B, C, H, W, generator=g, dtype=torch.float64) * 1.3 + 0.2).requires_grad_(True)
    gamma = (torch.randn(C, generator=g, dtype=torch.float64) * 0.5 + 1).requires_grad_(True)
    beta = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    d = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)          # cotangent on the activation's output
    add = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    r = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    y = F.leaky_relu(F.instance_norm(z, weight=gamma, bias=beta, eps=1e-5), 0.2)
    dz, dgam, dbet = torch.autograd.grad(y, [z, gamma, beta], d)
    want_dx = dz + add
    HW = H * W
    zd, dd, addd, rd = (_dev(_nhwc(t.detach()).float()) for t in (z, d, add, r))
    gd, bd = _dev(gamma.detach().float()), _dev(beta.detach().float())
    yd, mean, invstd = ops.instnorm_fwd(zd, B, HW, C, gd, bd, 1e-5, pcg._lib.ACT_LRELU, 0.2)
    dx, dn, dgp, dbp, dsp = ops.instnorm_bwd_fused(dd, zd, B, HW, C, mean, invstd, gd, act_y=yd, slope=0.2, keep_dn=True, addend=addd,
                                                   need_params=True, need_dxsum=True)
    sc = float(want_dx.abs().max())
    _close(dx, _nhwc(want_dx), 1e-4, 2e-5 * sc, "dx + addend")
    mask = torch.where(_nhwc(y.detach()) > 0, 1.0, 0.2)
    _close(dn, _nhwc(d) * mask, 1e-6, 1e-6, "dn")
    # the same launch without dn / addend / partials gives the same dz
    dx2, dn2, _, _, _ = ops.instnorm_bwd_fused(dd, zd, B, HW, C, mean, invstd, gd, act_y=yd, slope=0.2, need_params=False)
    assert dn2 is None
    _close(dx2, _nhwc(dz), 1e-4, 2e-5 * float(dz.abs().max()), "dz")
    # one launch for the three vector gradients, with and without accumulation
    out = [torch.full((C,), 0.5, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)]
    ops.rowsum3([(dgp, out[0], True), (dbp, out[1], False), (dsp, out[2], False)], B, C)
    _close(out[0], dgam + 0.5, 1e-4, 2e-5 * float(dgam.abs().max()) + 1e-5, "dgamma (+=)")
    _close(out[1], dbet, 1e-4, 2e-5 * float(dbet.abs().max()) + 1e-5, "dbeta")
    _close(out[2], want_dx.sum((0, 2, 3)), 1e-4, 3e-5 * float(want_dx.abs().sum((0, 2, 3)).max()) + 1e-5, "sum of the stored dx (conv bias gradient)")
    # masked ddy
    ddy_plain, ez, dg2 = ops.instnorm_bwd_bwd(rd, dn, zd, B, HW, C, mean, invstd, gd)
    ddy_masked, ez2, _ = ops.instnorm_bwd_bwd(rd, dn, zd, B, HW, C, mean, invstd, gd, act_y=yd, slope=0.2)
    assert torch.equal(ez, ez2)
    assert torch.equal(ddy_masked, ops.act_bwd(ddy_plain, yd, pcg._lib.ACT_LRELU, 0.2))


def test_flatten_interpolate_gradient_penalty(pcg):
    ops, W = pcg.ops, pcg.wgan
    g = torch.Generator().manual_seed(5)
    B, C, H, Wd = 4, 16, 2, 3
    a = torch.randn(B, C, H, Wd, generator=g)
    flat = ops.nhwc_to_nchw_flat(_dev(_nhwc(a)), B, H * Wd, C).view(B, -1)
    assert torch.equal(flat.cpu(), a.flatten(1))                                            # nn.Flatten of the NCHW tensor: exact
    back = ops.nhwc_to_nchw_flat(flat, B, H * Wd, C, inverse=True).view(B, H, Wd, C)
    assert torch.equal(back.cpu(), _nhwc(a))
    real, fake, alpha = torch.randn(B, 1, 28, 28, generator=g), torch.randn(B, 1, 28, 28, generator=g), torch.rand(B, 1, generator=g)
    ref = alpha.view(-1, 1, 1, 1) * real + ((1. - alpha.view(-1, 1, 1, 1)) * fake)
    _close(ops.interpolate(_dev(alpha), _dev(real), _dev(fake)), ref, 0, 1.2e-7 * float(ref.abs().max()))   # fma vs mul+add: 1 ulp
    grads = (torch.randn(B, 1, 28, 28, generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    pen = 10.0 * ((grads.view(B, -1).norm(dim=1) - 1.) ** 2).mean()
    (pen * 0.7).backward()
    gd = _dev(grads.detach().float()).requires_grad_(True)
    p = W.gradient_penalty(gd, 10.0)
    (p * 0.7).backward()
    _close(p, pen.detach(), 1e-5, 1e-6)
    _close(gd.grad, grads.grad, 1e-4, 1e-6 * float(grads.grad.abs().max()))


# ---------------------------------------------------------------------------------------------------------------------
def _build_pair(pcg, hp_kwargs, state=None, seed=1):
    W = pcg.wgan
    hp = W.Hyperparameter(**hp_kwargs)
    ohp = WR.Hyperparameter(**{k: v for k, v in hp_kwargs.items()})
    oc, og = WR.build(ohp, seed=seed)
    if state is not None:
        oc.load_state_dict(state[0]); og.load_state_dict(state[1])
    critic, generator = W.Critic(hp), W.Generator(hp)
    assert list(critic.state_dict()) == list(oc.state_dict()) and list(generator.state_dict()) == list(og.state_dict())
    critic.load_state_dict(oc.state_dict()); generator.load_state_dict(og.state_dict())
    return hp, ohp, critic.to(DEV), generator.to(DEV), oc, og


@pytest.mark.parametrize("batched", [False, True])
def test_golden_reference_loop_iterations(pcg, golden_dir, batched):
    """Three iterations of the reference's loop body (:133-168, reduced width): image gradients, losses, every gradient after
    the first critic+generator update, critic gradients after the last critic update, all parameters and buffers at the end."""
    W = pcg.wgan
    gold = dict(np.load(os.path.join(golden_dir, "wgan_ref_small.npz")))
    w, steps, batch = int(gold["meta.width"]), int(gold["meta.steps"]), int(gold["meta.batch"])
    kw = dict(critic_size=w, generator_size=w, critic_hidden_size=w, batchsize=batch)
    state = ({k[7:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.C.")},
             {k[7:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.G.")})
    hp, _, critic, generator, _, _ = _build_pair(pcg, kw, state)
    c_opt, g_opt = W.make_optimizers(critic, generator)
    eye = torch.eye(hp.num_classes, device=DEV)
    for k in range(steps):
        real, labels = _dev(torch.from_numpy(gold[f"step{k}.real"])), _dev(torch.from_numpy(gold[f"step{k}.labels"]))
        out = W.critic_step(critic, generator, c_opt, hp, real, eye[labels], _dev(torch.from_numpy(gold[f"step{k}.noise"])),
                            _dev(torch.from_numpy(gold[f"step{k}.alpha"])), batched=batched)
        gg = gold[f"step{k}.gradients"]
        np.testing.assert_allclose(out["gradients"].detach().cpu().numpy(), gg, rtol=2e-4, atol=3e-5 * np.abs(gg).max(), err_msg=f"gradients {k}")
        for name in ("critic_loss", "gradient_penalty", "loss_real"):
            # after the first update the weights carry AdamW's (beta1 = 0: sign-like) amplification of fp32 gradient noise
            rt, at = (5e-5, 2e-6) if k == 0 else (2e-3, 2e-4)
            np.testing.assert_allclose(out[name].item(), gold[f"step{k}.{name}"], rtol=rt, atol=at, err_msg=f"step{k}.{name}")
        if k % hp.n_critic == 0:
            go = W.generator_step(critic, generator, g_opt, eye[_dev(torch.from_numpy(gold[f"step{k}.fake_idx"]))],
                                  _dev(torch.from_numpy(gold[f"step{k}.noise_g"])), skip_dead_critic_wgrad=False)
            # evaluated with the critic AFTER its AdamW step: same amplification of gradient noise as above
            np.testing.assert_allclose(go["generator_loss"].item(), gold[f"step{k}.generator_loss"], rtol=2e-3, atol=2e-4)
        if k == 0:
            for net, tag in ((critic, "C"), (generator, "G")):
                scale = max(float(np.abs(gold[f"step0.grad.{tag}.{n}"]).max()) for n, _ in net.named_parameters())
                for n, p in net.named_parameters():
                    ref = gold[f"step0.grad.{tag}.{n}"]
                    # floor relative to the net's largest gradient: conv biases in front of InstanceNorm / BatchNorm have a
                    # true gradient of exactly 0 and hold summation noise in any fp32 implementation
                    np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=3e-4, atol=3e-5 * np.abs(ref).max() + 5e-6 * scale,
                                               err_msg=f"step0 grad {tag}.{n}")
    scale = max(float(np.abs(gold[f"final.grad.C.{n}"]).max()) for n, _ in critic.named_parameters())
    for n, p in critic.named_parameters():
        ref = gold[f"final.grad.C.{n}"]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=5e-4, atol=5e-5 * np.abs(ref).max() + 5e-6 * scale, err_msg=f"final grad {n}")
    for net, tag in ((critic, "C"), (generator, "G")):
        dead = {n for n, _ in net.named_parameters() if np.abs(gold[f"step0.grad.{tag}.{n}"]).max() < 1e-4}
        for k_, v in net.state_dict().items():
            ref = gold[f"final.{tag}.{k_}"]
            if k_ in dead:      # zero-gradient biases: AdamW(beta1=0) turns the noise sign into a +-lr move per step
                assert np.abs(v.cpu().numpy() - ref).max() <= 2.2 * 1e-4 * steps, k_
                continue
            atol = 1e-4 if k_.endswith("running_mean") else 3e-5
            d = np.abs(v.cpu().numpy().astype(np.float64) - ref)
            bad = d > atol + 2e-4 * np.abs(ref)
            # AdamW with beta1 = 0 moves every entry by ~lr * sign(g) per step: an entry whose gradient is at noise level may
            # take the other sign — allow a few such entries, each bounded by the total possible move
            assert bad.sum() <= max(2, 0.01 * d.size) and d.max() <= 2.2 * 1e-4 * steps, (f"final.{tag}.{k_}", int(bad.sum()), float(d.max()))


@pytest.mark.parametrize("width,batch,batched", [(64, 8, False), (64, 8, True), (1024, 32, True)])
def test_critic_and_generator_steps_vs_oracle_float64(pcg, width, batch, batched):
    """Wider nets (every channel count a multiple of 16, MFMA kernels engaged), one critic step + one generator step: losses,
    image gradients and every parameter gradient against the float64 oracle; tolerance = max(1e-4 of the tensor's scale,
    3x the float32 oracle's own distance from float64, 2e-6 of the net's largest gradient).
    (1024, 32) is the reference's FULL width (mnist_wgan_conditional.py:20-31) — r04: the composition of the stream-K launches, the
    unequal-phase grad-input, the register-resident InstanceNorm at C = 1024 and the backward of the backward, checked end to end
    against the oracle (losses, every gradient incl. rel-L2, the weights after AdamW); ~2 s of CPU work per precision."""
    W = pcg.wgan
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    kw = dict(critic_size=width, generator_size=width, critic_hidden_size=width, batchsize=batch)
    hp, ohp, critic, generator, oc, og = _build_pair(pcg, kw)
    x, y, z, alpha, y2, z2 = WR.synthetic_batch(ohp, batch, seed=4, dtype=torch.float64)
    res = {}
    for dt in (torch.float64, torch.float32):
        c, g = WR.Critic(ohp), WR.Generator(ohp)
        c.load_state_dict(oc.state_dict()); g.load_state_dict(og.state_dict())
        c, g = c.to(dt), g.to(dt)
        c_opt, g_opt = WR.make_optimizers(c, g)
        o1 = WR.critic_step(c, g, c_opt, ohp, x.to(dt), y.to(dt), z.to(dt), alpha.to(dt))
        cg = {n: p.grad.double().clone() for n, p in c.named_parameters()}
        o2 = WR.generator_step(c, g, g_opt, y2.to(dt), z2.to(dt))
        gg = {n: p.grad.double().clone() for n, p in g.named_parameters()}
        res[dt] = (o1, o2, cg, gg, {k_: v.detach().clone() for k_, v in c.state_dict().items()},
                   {k_: v.detach().clone() for k_, v in g.state_dict().items()})
    c_opt, g_opt = W.make_optimizers(critic, generator)
    out = W.critic_step(critic, generator, c_opt, hp, _dev(x.float()), _dev(y.float()), _dev(z.float()), _dev(alpha.float()), batched=batched)
    mine_cg = {n: p.grad.detach().cpu().double().clone() for n, p in critic.named_parameters()}
    out2 = W.generator_step(critic, generator, g_opt, _dev(y2.float()), _dev(z2.float()))
    mine_gg = {n: p.grad.detach().cpu().double().clone() for n, p in generator.named_parameters()}
    (o1_64, o2_64, cg64, gg64), (o1_32, o2_32, cg32, gg32) = res[torch.float64][:4], res[torch.float32][:4]
    for k in ("critic_loss", "loss_real", "loss_fake", "gradient_penalty"):
        tol = max(2e-5 + 2e-5 * abs(o1_64[k]), 3 * abs(o1_32[k] - o1_64[k]))
        assert abs(out[k].item() - o1_64[k]) <= tol, (k, out[k].item(), o1_64[k], tol)
    tol = max(2e-5 + 2e-5 * abs(o2_64["generator_loss"]), 3 * abs(o2_32["generator_loss"] - o2_64["generator_loss"]))
    assert abs(out2["generator_loss"].item() - o2_64["generator_loss"]) <= tol

    def check(mine, truth, noise32, label, net_scale):
        scale = float(truth.abs().max())
        tol = max(1e-4 * scale, 3 * float((noise32 - truth).abs().max()), 2e-6 * net_scale)
        err = float((mine - truth).abs().max())
        assert err <= tol, (label, err, tol, scale)
    check(out["gradients"].detach().cpu().double(), o1_64["gradients"], o1_32["gradients"].double(), "image gradients", 0.0)
    cs, gs = max(float(v.abs().max()) for v in cg64.values()), max(float(v.abs().max()) for v in gg64.values())

    def check_l2(mine, truth, noise32, label, net_scale):     # rel-L2 of the whole tensor (a tile in the wrong place shows here)
        den = float(truth.norm())
        if den < 1e-3 * net_scale * truth.numel() ** 0.5:   # zero-gradient parameters (biases in front of a normalisation): noise
            return
        l2, l2r = float((mine - truth).norm()) / den, float((noise32 - truth).norm()) / den
        assert l2 <= max(1e-4, 3 * l2r), (label, l2, l2r)
    for n in cg64:
        check(mine_cg[n], cg64[n], cg32[n], f"critic grad {n}", cs)
        check_l2(mine_cg[n], cg64[n], cg32[n], f"critic grad {n} (rel-L2)", cs)
    for n in gg64:
        check(mine_gg[n], gg64[n], gg32[n], f"generator grad {n}", gs)
        check_l2(mine_gg[n], gg64[n], gg32[n], f"generator grad {n} (rel-L2)", gs)
    # the weights after the AdamW steps (:118-119: lr 1e-4, beta1 = 0 — every entry moves by ~lr * sign(g)): against the fp32
    # oracle's weights; an entry whose gradient is at noise level may take the other sign, bounded by the possible move
    lr = 1e-4
    for net, onet, tag, g64_, sc_ in ((critic, res[torch.float32][4], "critic", cg64, cs), (generator, res[torch.float32][5], "generator", gg64, gs)):
        for k_, v in net.state_dict().items():
            ref = onet[k_]
            if k_ in g64_ and float(g64_[k_].abs().max()) < 1e-5 * sc_:     # zero-gradient bias: the noise's sign decides a +-lr move
                assert float((v.detach().cpu().double() - ref.double()).abs().max()) <= 2.2 * lr, (tag, k_)
                continue
            if not ref.dtype.is_floating_point:
                assert int(v) == int(ref), k_
                continue
            d = (v.detach().cpu().double() - ref.double()).abs()
            tol = 2e-6 + 2e-5 * ref.double().abs()
            bad = d > tol
            if k_ in g64_:
                # with beta1 = 0 the move of an entry is lr * g / (|g| + eps): an entry whose gradient is at the noise level of its
                # tensor (or near Adam's eps) moves by a noise-decided fraction of lr — only entries with a well-determined
                # gradient are counted; every entry is bounded by the possible move
                g_ = g64_[k_].abs()
                bad = bad & (g_ > max(1e-6, 1e-3 * float(g_.max())))
            nbad = int(bad.sum())
            assert float(d.max()) <= 2.2 * lr + 1e-4 * float(ref.abs().max()) and nbad <= max(4, 0.02 * d.numel()), (tag, k_, float(d.max()), nbad, d.numel())


def test_reference_style_penalty_expression_is_a_drop_in(pcg):
    """The reference writes the penalty with plain tensor ops on the autograd.grad result (:149-150).  That code, unchanged,
    must drive the same backward-of-backward as the fused gradient_penalty op: identical parameter gradients."""
    W = pcg.wgan
    kw = dict(critic_size=16, generator_size=16, critic_hidden_size=16, batchsize=5)
    hp, ohp, critic, generator, _, _ = _build_pair(pcg, kw)
    x, y, z, alpha, _, _ = WR.synthetic_batch(ohp, 5, seed=8)
    x, y = _dev(x), _dev(y)
    grads = []
    for style in ("reference", "fused"):
        critic.zero_grad()
        interp = (x * 0.5).requires_grad_(True)
        d = critic(interp, y)
        gradients = torch.autograd.grad(d, interp, torch.ones((5, 1), device=DEV), create_graph=True, only_inputs=True)[0]
        if style == "reference":
            pen = hp.gp_lambda * ((gradients.view(5, -1).norm(dim=1) - 1.) ** 2).mean()
        else:
            pen = W.gradient_penalty(gradients, hp.gp_lambda)
        pen.backward()
        grads.append({n: p.grad.clone() for n, p in critic.named_parameters()})
        assert interp.grad is None or True
    scale = max(float(v.abs().max()) for v in grads[0].values())
    for n in grads[0]:
        _close(grads[0][n], grads[1][n], 1e-4, 2e-6 * scale, n)


def test_train_loop_on_device_draws(pcg):
    W, ops = pcg.wgan, pcg.ops
    hp = W.Hyperparameter(critic_size=16, generator_size=16, critic_hidden_size=16, batchsize=8, n_critic=2)
    critic, generator = W.build(torch.device(DEV), hp, seed=1)
    data = [(torch.rand(8, 1, 28, 28) * 2 - 1, torch.randint(0, 10, (8,))) for _ in range(4)]
    before = {k: v.clone() for k, v in generator.state_dict().items()}
    hist = W.train(critic, generator, data, hp, torch.device(DEV), rng=ops.DeviceRNG(3), epochs=2)
    assert len(hist) == 2 and all(np.isfinite(v) for e in hist for v in e)
    assert any(not torch.equal(before[k], v) for k, v in generator.state_dict().items())
    a = ops.DeviceRNG(9).rand((4096,), torch.device(DEV))
    assert 0.0 <= float(a.min()) and float(a.max()) < 1.0 and abs(float(a.mean()) - 0.5) < 0.02


def test_graphed_updates_equal_the_eager_updates_bitwise(pcg):
    """wgan.GraphedSteps (both updates replayed from HIP graphs) against critic_step / generator_step launched kernel by kernel:
    same parameters, optimizer state and logged losses after every update of three loop iterations; then train(graphed=True)
    against train(graphed=False) on the same batches and the same device draws, ragged last batch included."""
    W, ops = pcg.wgan, pcg.ops
    dev = torch.device(DEV)
    hp = W.Hyperparameter(critic_size=32, generator_size=32, critic_hidden_size=32, batchsize=16, n_critic=2)
    B = hp.batchsize

    def fresh():
        critic, generator = W.build(dev, hp, seed=5)
        return (critic, generator) + W.make_optimizers(critic, generator)

    ce, ge, coe, goe = fresh()
    cg, gg, cog, gog = fresh()
    gs = W.GraphedSteps(cg, gg, cog, gog, hp, B, dev)
    for m_e, m_g in ((ce, cg), (ge, gg)):                         # building the graphs does not advance training
        for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
            assert torch.equal(a, b), k
    rng = ops.DeviceRNG(11)
    for it in range(3):
        x = rng.rand((B, 1, 28, 28), dev).mul_(2.0).sub_(1.0)
        lab = ops.onehot(rng.randint(0, 10, B, dev), 10)
        noise, alpha = rng.randn((B, hp.latent_size), dev), rng.rand((B, 1), dev)
        oe = W.critic_step(ce, ge, coe, hp, x, lab, noise, alpha)
        og = gs.critic_step(x, lab, noise, alpha)
        for k in ("critic_loss", "loss_real", "loss_fake", "gradient_penalty", "gradients", "fake_image"):
            assert torch.equal(oe[k], og[k]), (it, k)
        fake, noise = ops.onehot(rng.randint(0, 10, B, dev), 10), rng.randn((B, hp.latent_size), dev)
        le = W.generator_step(ce, ge, goe, fake, noise)["generator_loss"]
        lg = gs.generator_step(fake, noise)["generator_loss"]
        assert torch.equal(le, lg), it
        for m_e, m_g in ((ce, cg), (ge, gg)):
            for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
                assert torch.equal(a, b), (it, k)
    with pytest.raises(pcg.PcgError):
        gs.critic_step(x[:5], lab[:5], noise[:5], alpha[:5])
    # the loop
    data = [(torch.rand(n, 1, 28, 28) * 2 - 1, torch.randint(0, 10, (n,))) for n in (B, B, B, 5)]
    c1, g1 = W.build(dev, hp, seed=2)
    c2, g2 = W.build(dev, hp, seed=2)
    h1 = W.train(c1, g1, data, hp, dev, rng=ops.DeviceRNG(3), epochs=2)
    h2 = W.train(c2, g2, data, hp, dev, rng=ops.DeviceRNG(3), epochs=2, graphed=True)
    assert h1 == h2
    for (k, a), (_, b) in zip(g1.state_dict().items(), g2.state_dict().items()):
        assert torch.equal(a, b), k

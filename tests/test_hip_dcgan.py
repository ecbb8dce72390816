"""GPU parity, network level: the DCGAN drop-ins (pcgan_amd.dcgan) against
  (a) golden vectors produced by the reference's own classes and loop body (tests/golden/dcgan_ref_small.npz), and
  (b) the oracle restatement (oracle/dcgan_ref.py, torch CPU fp32) run live at the reference's full width.
Tolerances are the stated fp32 ones of SURVEY.md §8c: forward 1e-5 rel, gradients 1e-4 rel (L2) / 1e-3 of the tensor's
max (element-wise tail), weights after k Adam steps 1e-4 + the Adam sign-noise floor (see test_oracle_golden.py).
"""
import os

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
    return pcgan_amd


def _load_sd(module, sd):
    module.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _check_grads(named_params, ref_named, what):
    for (n, p), (n2, q) in zip(named_params, ref_named):
        assert n == n2
        got, ref = p.grad.detach().cpu().numpy(), q if isinstance(q, np.ndarray) else q.grad.detach().numpy()
        assert got.shape == ref.shape, (n, got.shape, ref.shape)
        l2 = _rel_l2(got, ref)
        mx = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert l2 <= 1e-4 and mx <= 1e-3, f"{what} grad {n}: rel-L2 {l2:.2e}, max/absmax {mx:.2e}"


def test_golden_reference_trajectory(pcg, golden_dir):
    D = pcg.dcgan
    gold = dict(np.load(os.path.join(golden_dir, "dcgan_ref_small.npz")))
    cfg = {"g_hidden": int(gold["meta.g_hidden"]), "d_hidden": int(gold["meta.d_hidden"]), "z_dim": int(gold["meta.z_dim"])}
    netG, netD = D.Generator(cfg), D.Discriminator(cfg)
    # identical state_dict surface as the reference classes
    assert {f"init.G.{k}" for k in netG.state_dict()} == {k for k in gold if k.startswith("init.G.")}
    assert {f"init.D.{k}" for k in netD.state_dict()} == {k for k in gold if k.startswith("init.D.")}
    sdG = {k[7:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.G.")}
    sdD = {k[7:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.D.")}
    _load_sd(netG, sdG); _load_sd(netD, sdD)
    netG.to(DEV); netD.to(DEV)

    # forward, train mode then eval mode (running stats after exactly one train-mode pass, as in make_golden.py)
    z, real = torch.from_numpy(gold["fwd.z"]).to(DEV), torch.from_numpy(gold["fwd.real"]).to(DEV)
    with torch.no_grad():
        np.testing.assert_allclose(netG(z).cpu().numpy(), gold["fwd.G_out"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(netD(real).cpu().numpy(), gold["fwd.D_out"], rtol=1e-5, atol=2e-6)
        netG.eval(); netD.eval()
        np.testing.assert_allclose(netG(z).cpu().numpy(), gold["fwd.G_out_eval"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(netD(real).cpu().numpy(), gold["fwd.D_out_eval"], rtol=1e-5, atol=2e-6)
    netG.train(); netD.train()
    _load_sd(netG, {k: v.to(DEV) for k, v in sdG.items()}); _load_sd(netD, {k: v.to(DEV) for k, v in sdD.items()})

    crit, optD, optG = D.make_optimizers(netG, netD, cfg)
    for k in range(int(gold["meta.steps"])):
        real = torch.from_numpy(gold[f"step{k}.real"]).to(DEV)
        noise = torch.from_numpy(gold[f"step{k}.noise"]).to(DEV)
        out = D.train_step(netG, netD, crit, optD, optG, real, noise, cfg, skip_dead_d_wgrad=False)
        got = {"errD_real": out["errD_real"].item(), "errD_fake": out["errD_fake"].item(), "errG": out["errG"].item(),
               "D_x": out["out_real"].mean().item(), "D_G_z1": out["out_fake"].mean().item(), "D_G_z2": out["out_g"].mean().item()}
        for name, val in got.items():
            np.testing.assert_allclose(val, gold[f"step{k}.{name}"], rtol=2e-5, atol=1e-6, err_msg=f"step {k} {name}")
    for key, v in netG.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), gold[f"final.G.{key}"], rtol=1e-4, atol=5e-6, err_msg=key)
    for key, v in netD.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), gold[f"final.D.{key}"], rtol=1e-4, atol=5e-6, err_msg=key)
    _check_grads(list(netG.named_parameters()), [(n, gold[f"final.G.grad.{n}"]) for n, _ in netG.named_parameters()], "G")
    # with skip_dead_d_wgrad=False D's .grad holds D-step + G-step gradients exactly like the reference's autograd
    _check_grads(list(netD.named_parameters()), [(n, gold[f"final.D.grad.{n}"]) for n, _ in netD.named_parameters()], "D")


@pytest.mark.parametrize("batch,skip", [(8, True), (6, False)])
def test_full_width_step_vs_oracle(pcg, batch, skip):
    """Reference widths (g_hidden = d_hidden = 64, z = 100): two training steps against the CPU oracle."""
    D = pcg.dcgan
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    refG, refD = R.build(None, seed=1)
    netG, netD = D.Generator(), D.Discriminator()
    _load_sd(netG, refG.state_dict()); _load_sd(netD, refD.state_dict())
    netG.to(DEV); netD.to(DEV)
    rcrit, roptD, roptG = R.make_optimizers(refG, refD)
    crit, optD, optG = D.make_optimizers(netG, netD)
    for step in range(2):
        real, noise = R.synthetic_batch(batch, seed=10 + step)
        ref = R.dcgan_step(refG, refD, rcrit, roptD, roptG, real, noise)
        out = D.train_step(netG, netD, crit, optD, optG, real.to(DEV), noise.to(DEV), skip_dead_d_wgrad=skip)
        for name in ("errD_real", "errD_fake", "errG"):
            np.testing.assert_allclose(out[name].item(), ref[name], rtol=2e-5, atol=1e-6, err_msg=f"step {step} {name}")
        _check_grads(list(netG.named_parameters()), list(refG.named_parameters()), f"G step {step}")
        if not skip:
            _check_grads(list(netD.named_parameters()), list(refD.named_parameters()), f"D step {step}")
    # one fused launch per net: all parameters of a net are one contiguous segment
    assert optD.num_segments() == 1 and optG.num_segments() == 1
    for (k, v), (k2, r) in zip(netG.state_dict().items(), refG.state_dict().items()):
        assert k == k2
        np.testing.assert_allclose(v.cpu().numpy(), r.numpy(), rtol=1e-4, atol=5e-6, err_msg=f"G {k}")
    for (k, v), (k2, r) in zip(netD.state_dict().items(), refD.state_dict().items()):
        assert k == k2
        np.testing.assert_allclose(v.cpu().numpy(), r.numpy(), rtol=1e-4, atol=5e-6, err_msg=f"D {k}")


def test_skip_dead_d_wgrad_changes_nothing_observable(pcg):
    """Skipping D's weight gradients in the G step (discarded by netD.zero_grad() at :147) must leave parameters,
    losses and G gradients bit-identical."""
    D = pcg.dcgan
    cfg = {"g_hidden": 16, "d_hidden": 16, "z_dim": 32}
    refG, refD = R.build(cfg, seed=3)
    res = []
    for skip in (True, False):
        netG, netD = D.Generator(cfg), D.Discriminator(cfg)
        _load_sd(netG, refG.state_dict()); _load_sd(netD, refD.state_dict())
        netG.to(DEV); netD.to(DEV)
        crit, optD, optG = D.make_optimizers(netG, netD, cfg)
        for step in range(2):
            real, noise = R.synthetic_batch(16, seed=step, config=cfg)
            out = D.train_step(netG, netD, crit, optD, optG, real.to(DEV), noise.to(DEV), cfg, skip_dead_d_wgrad=skip)
        res.append((netG.flat_params.clone(), netD.flat_params.clone(), netG.flat_grads.clone(), out["errG"].clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


def test_module_surface(pcg):
    """What a user of the reference classes relies on: parameters(), state_dict round trip, .apply(weights_init),
    zero_grad, detach, no_grad inference, error on CPU use."""
    D = pcg.dcgan
    cfg = {"g_hidden": 8, "d_hidden": 8, "z_dim": 16}
    torch.manual_seed(0)
    netG = D.Generator(cfg)
    netG.apply(D.weights_init)
    ref = R.Generator(cfg)
    assert [tuple(p.shape) for p in netG.parameters()] == [tuple(p.shape) for p in ref.parameters()]
    with pytest.raises(pcg.PcgError, match="no CPU path"):
        netG(torch.zeros(2, 16, 1, 1))
    netG.to(DEV)
    z = torch.randn(4, 16, 1, 1, device=DEV)
    y = netG(z)
    assert y.shape == (4, 1, 64, 64) and y.requires_grad
    sd = {k: v.clone() for k, v in netG.state_dict().items()}
    y.sum().backward()
    g0 = netG.main[0].weight.grad.clone()
    assert g0.shape == netG.main[0].weight.shape and float(g0.abs().sum()) > 0
    netG.zero_grad()
    assert float(netG.flat_grads.abs().sum()) == 0.0 and netG.main[0].weight.grad is not None
    with torch.no_grad():
        assert not netG(z).requires_grad
    net2 = D.Generator(cfg).to(DEV)
    net2.load_state_dict(sd)
    net2.train()
    # same weights + same running stats -> same output (running stats were updated once by the first forward)
    netG.load_state_dict(sd)
    assert torch.equal(net2(z), netG(z))

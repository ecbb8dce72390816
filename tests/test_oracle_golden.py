"""CPU: the DCGAN restatement (oracle/dcgan_ref.py) against vectors produced by the reference's own code
(tests/golden/make_golden.py lifts Generator/Discriminator/weights_init and the loop body out of mnist_dcgan.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import dcgan_ref as R

torch.set_num_threads(4)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "dcgan_ref_small.npz")))


def _cfg(gold):
    return {"g_hidden": int(gold["meta.g_hidden"]), "d_hidden": int(gold["meta.d_hidden"]), "z_dim": int(gold["meta.z_dim"])}


def _load(module, gold, prefix):
    sd = {k[len(prefix) + 1:]: torch.from_numpy(v.copy()) for k, v in gold.items()
          if k.startswith(prefix + ".") and not k.startswith(prefix + ".grad.")}
    module.load_state_dict(sd, strict=True)


def test_state_dict_keys_match_reference(gold):
    cfg = _cfg(gold)
    netG, netD = R.Generator(cfg), R.Discriminator(cfg)
    assert {f"init.G.{k}" for k in netG.state_dict()} == {k for k in gold if k.startswith("init.G.")}
    assert {f"init.D.{k}" for k in netD.state_dict()} == {k for k in gold if k.startswith("init.D.")}


def test_weights_init_reproduces_reference_seed(gold):
    cfg = _cfg(gold)
    netG, netD = R.build(cfg, seed=1)  # mnist_dcgan.py:33,119-122
    for k, v in netG.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), gold[f"init.G.{k}"])
    for k, v in netD.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), gold[f"init.D.{k}"])


def test_forward_train_and_eval(gold):
    cfg = _cfg(gold)
    netG, netD = R.Generator(cfg), R.Discriminator(cfg)
    _load(netG, gold, "init.G"); _load(netD, gold, "init.D")
    z, real = torch.from_numpy(gold["fwd.z"]), torch.from_numpy(gold["fwd.real"])
    np.testing.assert_allclose(netG(z).detach().numpy(), gold["fwd.G_out"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(netD(real).detach().numpy(), gold["fwd.D_out"], rtol=1e-6, atol=1e-7)
    _load(netG, gold, "init.G"); _load(netD, gold, "init.D")  # reset running stats touched by the train-mode pass
    # the golden eval outputs were produced after ONE train-mode forward on the same inputs
    netG(z); netD(real)
    netG.eval(); netD.eval()
    np.testing.assert_allclose(netG(z).detach().numpy(), gold["fwd.G_out_eval"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(netD(real).detach().numpy(), gold["fwd.D_out_eval"], rtol=1e-6, atol=1e-7)


def test_training_steps_match_reference_loop_body(gold):
    cfg = _cfg(gold)
    netG, netD = R.Generator(cfg), R.Discriminator(cfg)
    _load(netG, gold, "init.G"); _load(netD, gold, "init.D")
    crit, optD, optG = R.make_optimizers(netG, netD, cfg)
    for k in range(int(gold["meta.steps"])):
        real, noise = torch.from_numpy(gold[f"step{k}.real"]), torch.from_numpy(gold[f"step{k}.noise"])
        out = R.dcgan_step(netG, netD, crit, optD, optG, real, noise, cfg)
        for name, val in out.items():
            np.testing.assert_allclose(val, gold[f"step{k}.{name}"], rtol=1e-6, atol=1e-7, err_msg=f"step {k} {name}")
    # Adam normalises each gradient by its own magnitude, so an fp32 reduction-order difference in a near-zero
    # gradient moves a weight by up to ~lr (2e-4) * O(1e-2) after 3 steps: tolerances reflect that noise floor
    # (golden made with 8 CPU threads, this test runs with 4).
    for k, v in netG.state_dict().items():
        np.testing.assert_allclose(v.numpy(), gold[f"final.G.{k}"], rtol=1e-4, atol=5e-6, err_msg=k)
    for k, v in netD.state_dict().items():
        np.testing.assert_allclose(v.numpy(), gold[f"final.D.{k}"], rtol=1e-4, atol=5e-6, err_msg=k)
    for n, p in netG.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), gold[f"final.G.grad.{n}"], rtol=1e-4, atol=1e-6, err_msg=n)
    for n, p in netD.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), gold[f"final.D.grad.{n}"], rtol=1e-4, atol=1e-6, err_msg=n)


def test_outer_loop_matches_reference(golden_dir):
    """mnist_dcgan.py:129-198 (lifted and run unmodified by make_golden.make_dcgan_loop): epochs x batches, running loss averages,
    the train-mode viz forward at iteration 0 and at the very end (it moves the generator's BatchNorm running statistics)."""
    gold = dict(np.load(os.path.join(golden_dir, "dcgan_loop_small.npz")))
    cfg = dict(_cfg(gold), batch_size=int(gold["meta.batch"]), epochs=int(gold["meta.epochs"]))
    netG, netD = R.Generator(cfg), R.Discriminator(cfg)
    _load(netG, gold, "init.G"); _load(netD, gold, "init.D")
    data = [(torch.from_numpy(gold[f"data.{k}"]),) for k in range(int(gold["meta.nbatches"]))]
    torch.manual_seed(int(gold["meta.loop_seed"]))
    g_losses, d_losses, imgs, iters = R.dcgan_train(data, cfg, netG, netD)
    assert iters == int(gold["iters"]) == 6 and len(imgs) == 2
    np.testing.assert_allclose(g_losses, gold["epoch_G_losses"], rtol=1e-5)
    np.testing.assert_allclose(d_losses, gold["epoch_D_losses"], rtol=1e-5)
    for k, img in enumerate(imgs):
        np.testing.assert_allclose(img.numpy(), gold[f"img.{k}"], rtol=1e-4, atol=2e-5)
    assert int(netG.state_dict()["main.1.num_batches_tracked"]) == int(gold["final.G.main.1.num_batches_tracked"]) == 6 + 2
    for k, v in netG.state_dict().items():
        np.testing.assert_allclose(v.numpy(), gold[f"final.G.{k}"], rtol=1e-4, atol=1e-5, err_msg=k)
    for k, v in netD.state_dict().items():
        np.testing.assert_allclose(v.numpy(), gold[f"final.D.{k}"], rtol=1e-4, atol=1e-5, err_msg=k)


# ---- CounteRGAN/mnist: oracle/countergan_ref.py against the reference's own modules + train_countergan --------------
from oracle import countergan_ref as CR  # noqa: E402


def _digest(t, nsamples=64):
    a = np.asarray(t.detach().cpu().numpy(), dtype=np.float64).ravel()
    idx = np.linspace(0, a.size - 1, num=min(nsamples, a.size)).astype(np.int64)
    return np.concatenate([[a.sum(), np.abs(a).sum(), (a * a).sum()], a[idx]])


@pytest.fixture(scope="module")
def cgold(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "countergan_ref_b4.npz")))


def test_countergan_init_and_forward(cgold):
    G, D, C = CR.build(seed=int(cgold["meta.seed"]))
    for tag, net in (("G", G), ("D", D), ("C", C)):
        keys = [k for k in cgold if k.startswith(f"init.{tag}.")]
        assert [f"init.{tag}.{k}" for k in net.state_dict()] == keys      # same names, same order as the reference
        for k, v in net.state_dict().items():
            np.testing.assert_array_equal(_digest(v.float()), cgold[f"init.{tag}.{k}"], err_msg=f"{tag}.{k}")
    x, y, t, m = (torch.from_numpy(cgold[f"in.{n}"]) for n in ("x", "y", "target_y", "mask"))
    raw, masked = G(x, t, m)
    np.testing.assert_allclose(raw.detach().numpy(), cgold["fwd.raw"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(masked.detach().numpy(), cgold["fwd.masked"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(D(x, y).detach().numpy(), cgold["fwd.d_logits"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(C(x).detach().numpy(), cgold["fwd.c_logits"], rtol=1e-5, atol=1e-7)


def test_countergan_step_matches_reference_train_loop(cgold):
    G, D, C = CR.build(seed=int(cgold["meta.seed"]))
    opt_g, opt_d, bce, ce = CR.make_optimizers(G, D)
    x, y = torch.from_numpy(cgold["in.x"]), torch.from_numpy(cgold["in.y"])
    t, m = torch.from_numpy(cgold["step.target_y"]), torch.from_numpy(cgold["step.mask"])
    out = CR.countergan_step(G, D, C, opt_g, opt_d, bce, ce, x, y, t, m)
    log = str(cgold["step.log"])
    # the reference prints D(real), D(fake) with 3 decimals, g_adv / g_cls with 4, reg with 6 (trainer.py:135-137)
    assert f"D(real)={out['d_real_p']:.3f}" in log and f"D(fake)={out['d_fake_p']:.3f}" in log, log
    assert f"g_adv={out['g_adv']:.4f}" in log and f"g_cls={out['g_cls']:.4f}" in log and f"reg={out['reg_l1']:.6f}" in log, log
    assert f"G: {out['g_loss']:.4f}, D: {out['d_loss']:.4f}" in log, log
    for tag, net in (("G", G), ("D", D)):
        for n, p in net.named_parameters():
            np.testing.assert_allclose(_digest(p.grad), cgold[f"grad.{tag}.{n}"], rtol=2e-4, atol=1e-7, err_msg=f"grad {tag}.{n}")
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(_digest(v.float()), cgold[f"final.{tag}.{k}"], rtol=1e-4, atol=5e-6, err_msg=f"final {tag}.{k}")


def test_countergan_train_loop_tail_matches_reference(golden_dir):
    """The reference's whole train_countergan, 2 epochs x 3 batches: epoch means, G_grad / D_grad (D's = D-step gradients + the
    generator step's critic weight gradients), the residual_mean field, the saved generator."""
    import re
    g = dict(np.load(os.path.join(golden_dir, "countergan_loop_b4.npz")))
    G, D, C = CR.build(seed=int(g["meta.seed"]))
    E, S = int(g["meta.epochs"]), int(g["meta.nbatches"])
    batches = [(torch.from_numpy(g["data.x"][i]), torch.from_numpy(g["data.y"][i])) for i in range(S)]
    t = [[torch.from_numpy(g["it.target_y"][e, i]) for i in range(S)] for e in range(E)]
    m = [[torch.from_numpy(g["it.mask"][e, i]) for i in range(S)] for e in range(E)]
    hist, gG, gD, first = CR.train_countergan(G, D, C, batches, t, m, E)
    np.testing.assert_allclose(gG, g["epoch.G_grad"], rtol=2e-4)
    np.testing.assert_allclose(gD, g["epoch.D_grad"], rtol=2e-4)
    log = str(g["log"])
    for e in range(E):
        assert f"[GAN] Epoch {e + 1}/{E} | G: {hist[e][0]:.4f}, D: {hist[e][1]:.4f}, G_cls: {hist[e][2]:.4f}, G_grad: {gG[e]:.4f}, D_grad: {gD[e]:.4f}" in log, log
        assert f"reg={first[e]['reg_l1']:.6f}, residual_mean={first[e]['reg_l1']:.4f}" in log, log
    for k, v in G.state_dict().items():
        if re.fullmatch(r"resblocks\.\d+\.conv[12]\.bias", k):
            # a conv bias in front of a BatchNorm has an exactly-zero gradient; what arrives is fp32 noise whose sign Adam turns
            # into a +-lr step per iteration: only bound the move (strided samples; the sums are sums of such noise)
            assert np.abs(_digest(v.float())[3:] - g[f"saved.G.{k}"][3:]).max() <= E * S * 2.2 * CR.Config.g_lr, k
            continue
        np.testing.assert_allclose(_digest(v.float()), g[f"saved.G.{k}"], rtol=1e-4, atol=5e-6, err_msg=f"saved {k}")


# ---- simple_gan/moons (BASELINE config 1): oracle/moons_ref.py against the reference's own train_gan -----------------
from oracle import moons_ref as MR  # noqa: E402


def test_moons_epoch_matches_reference_train_gan(golden_dir):
    gold = dict(np.load(os.path.join(golden_dir, "moons_ref.npz")))
    G, D = MR.build_generator(32, 128), MR.build_discriminator(128)
    G.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.G.")})
    D.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.D.")})
    optG, optD = MR.make_optimizers(G, D)
    X, z = torch.from_numpy(gold["X_shuffled"]), torch.from_numpy(gold["z"])
    totD = totG = 0.0
    for i, real in enumerate(X.split(50)):
        lD, lG = MR.moons_step(G, D, optG, optD, real, z[2 * i], z[2 * i + 1])
        totD += lD; totG += lG
    np.testing.assert_allclose(totD, float(gold["loss_D_total"]), rtol=1e-6)
    np.testing.assert_allclose(totG, float(gold["loss_G_total"]), rtol=1e-6)
    for tag, net in (("G", G), ("D", D)):
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.numpy(), gold[f"final.{tag}.{k}"], rtol=1e-5, atol=1e-6, err_msg=f"{tag}.{k}")


def test_countergan_trained_checkpoint_eval_forward(golden_dir):
    """Real-weight anchor: the reference's shipped generator.pt in eval mode (BatchNorm running statistics)."""
    gold = dict(np.load(os.path.join(golden_dir, "countergan_trained_eval.npz")))
    G = CR.ResidualGenerator()
    G.load_state_dict(torch.load(os.path.join(golden_dir, "countergan_generator_trained.pt"), map_location="cpu", weights_only=True))
    G.eval()
    with torch.no_grad():
        raw, masked = G(torch.from_numpy(gold["x"]), torch.from_numpy(gold["target"]), torch.from_numpy(gold["mask"]))
    np.testing.assert_allclose(raw.numpy(), gold["raw"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(masked.numpy(), gold["masked"], rtol=1e-5, atol=1e-6)


# ---- house_sales_kc_usa (tabular CounteRGAN): oracle/house_ref.py against ONE batch of the reference's own train_countergan
from oracle import house_ref as HR  # noqa: E402


@pytest.fixture(scope="module")
def hgold(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "house_ref_b64.npz")))


def _house_setup(hgold, dtype=torch.float32):
    G, D, clf = HR.build(seed=0)                      # same construction order as make_golden.py: clf, G (seed 0)
    for k, v in clf.state_dict().items():
        np.testing.assert_array_equal(_digest(v.float()), hgold[f"init.C.{k}"], err_msg=f"C.{k}")
    assert [f"init.G.{k}" for k in G.state_dict()] == [k for k in hgold if k.startswith("init.G.")]
    G.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in hgold.items() if k.startswith("init.G.")})
    assert [f"init.D.{k}" for k in D.state_dict()] == [k for k in hgold if k.startswith("init.D.")]   # weight_orig / _u / _v keys
    D.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in hgold.items() if k.startswith("init.D.")})
    x, y, t, m = (torch.from_numpy(hgold[f"in.{n}"]) for n in ("x", "y", "target_y", "mask"))
    gumbel = {int(k[7:]): torch.from_numpy(v) for k, v in hgold.items() if k.startswith("gumbel.")}
    return G, D, clf, x, y, t, m, gumbel


def test_house_step_matches_reference_train_loop(hgold):
    import re
    G, D, clf, x, y, t, m, gumbel = _house_setup(hgold)
    assert bool((t != y).all()) and float(m[:, HR.CONFIG["immutable_idx"]].abs().sum()) == 0.0   # trainer.py:248-255
    opt_G, opt_D = HR.make_optimizers(G, D)
    out = HR.house_step(G, D, clf, opt_G, opt_D, x, y, t, m, gumbel, HR.cat_norm_maps())
    log = str(hgold["log"])

    def logged(pat):
        return float(re.search(pat, log).group(1))
    assert abs(out["d_real_p"] - logged(r"D\(real\)=([0-9.]+)")) <= 6e-4 and abs(out["d_fake_p"] - logged(r"D\(fake\)=([0-9.]+)")) <= 6e-4
    assert abs(out["g_adv"] - logged(r"g_adv=(-?[0-9.]+)")) <= 7e-5 and abs(out["g_cls"] - logged(r"g_cls=([0-9.]+)")) <= 7e-5
    assert abs(out["reg"] - logged(r"reg=([0-9.]+)")) <= 2e-6 and abs(out["mask_pen"] - logged(r"mask_pen=([0-9.]+)")) <= 7e-6
    assert abs(out["D_loss"] - logged(r"\] D: (-?[0-9.]+)")) <= 7e-5 and abs(out["G_loss"] - logged(r", G: (-?[0-9.]+)")) <= 7e-5
    for n, p in G.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), hgold[f"grad.G.{n}"], rtol=2e-4, atol=2e-6, err_msg=f"grad {n}")
    for k, v in G.state_dict().items():
        gk = f"grad.G.{k}"
        if gk in hgold and np.abs(hgold[gk]).max() < 1e-6:
            # a Linear bias in front of BatchNorm1d has an exactly-zero gradient; the stored one is fp32 noise and Adam
            # turns its sign into a +-lr step: only bound the move
            assert np.abs(v.numpy() - hgold[f"final.G.{k}"]).max() <= 2.2 * HR.CONFIG["lr_G"], k
            continue
        np.testing.assert_allclose(v.numpy(), hgold[f"final.G.{k}"], rtol=1e-4, atol=2e-5, err_msg=f"final {k}")


def test_house_train_loop_matches_reference_trainer(golden_dir):
    """The WHOLE of the reference's train_countergan (2 epochs x 3 batches of 64 out of 209 rows, scaler-derived category values,
    the four diagnostics, epoch means, grad norms, the saved generator) against the oracle's restatement on the recorded draws."""
    g = dict(np.load(os.path.join(golden_dir, "house_loop.npz")))
    G, D, clf = HR.build(seed=0)
    for k, v in clf.state_dict().items():
        np.testing.assert_array_equal(_digest(v.float()), g[f"init.C.{k}"], err_msg=f"C.{k}")
    G.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("init.G.")})
    D.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("init.D.")})
    assert [int(f) for f in g["head_order"]] == list(HR.CONFIG["categorical_info"])
    norm = HR.cat_norm_maps_scaler(g["scaler.data_min"], g["scaler.data_max"], {f: g[f"raw_values.{f}"] for f in HR.CONFIG["categorical_info"]})
    X, y = torch.from_numpy(g["data.X"]), torch.from_numpy(g["data.y"])
    E, S = int(g["meta.epochs"]), int(g["meta.nbatches"])
    rows = [[torch.from_numpy(g["it.rows"][e, i]) for i in range(S)] for e in range(E)]
    # drop_last: every epoch uses 3 x 64 distinct rows of the 209
    assert all(len(set(np.concatenate([r.numpy() for r in rows[e]]).tolist())) == S * 64 for e in range(E))
    t = [[torch.from_numpy(g["it.target_y"][e, i]) for i in range(S)] for e in range(E)]
    m = [[torch.from_numpy(g["it.mask"][e, i]) for i in range(S)] for e in range(E)]
    gm = [[torch.from_numpy(g["it.gumbel"][e, i]) for i in range(S)] for e in range(E)]
    it, gG, gD = HR.train_countergan(G, D, clf, X, y, rows, t, m, gm, norm)
    tol = {"d_loss": 2e-5, "g_loss": 2e-4, "pred_gain": 1e-6, "sparsity": 2e-3, "l2_reg": 2e-4, "class_flip_rate": 0.02}
    for k, a in tol.items():
        np.testing.assert_allclose(np.array(it[k]), g[f"it.{k}"], rtol=0, atol=a, err_msg=k)
    np.testing.assert_allclose(gG, g["epoch.G_grad"], rtol=2e-3)
    np.testing.assert_allclose(gD, g["epoch.D_grad"], rtol=2e-3)
    for k, v in G.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(g[f"saved.G.{k}"])
            continue
        # six Adam steps; biases in front of a BatchNorm carry sign noise (+-lr per step)
        assert np.abs(v.numpy() - g[f"saved.G.{k}"]).max() <= 6 * 2.2 * HR.CONFIG["lr_G"], k


def test_house_trained_checkpoints_eval_forward(golden_dir):
    """The generator / classifier checkpoints the reference ships, through the oracle restatement in eval mode with
    hard Gumbel-softmax, against the reference modules' own outputs (tests/golden/make_golden.py: make_house_trained)."""
    gold = dict(np.load(os.path.join(golden_dir, "house_trained_eval.npz")))
    G, _, clf = HR.build(seed=0)
    G.load_state_dict(torch.load(os.path.join(golden_dir, "house_generator_trained.pt"), map_location="cpu", weights_only=True))
    clf.load_state_dict(torch.load(os.path.join(golden_dir, "house_classifier_trained.pt"), map_location="cpu", weights_only=True))
    G.eval(); clf.eval()
    x, t, m = torch.from_numpy(gold["x"]), torch.from_numpy(gold["target_y"]), torch.from_numpy(gold["mask"])
    gumbel = {int(k[7:]): torch.from_numpy(v) for k, v in gold.items() if k.startswith("gumbel.")}
    with torch.no_grad():
        cont, logits, samples = G(x, torch.nn.functional.one_hot(t, 4).float(), m, gumbel, temperature=0.5, hard=True)
        cl = clf(x)
    np.testing.assert_allclose(cont.numpy(), gold["cont"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cl.numpy(), gold["clf_logits"], rtol=1e-5, atol=1e-5)
    for f in gumbel:
        np.testing.assert_allclose(logits[f].numpy(), gold[f"logits.{f}"], rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(samples[f].numpy(), gold[f"samples.{f}"])          # one-hot: exact


# ---- conditional WGAN-GP: oracle/wgan_ref.py against the reference's own classes + loop body at reduced width -------------
from oracle import wgan_ref as WR  # noqa: E402


def test_wgan_gp_steps_match_reference_loop(golden_dir):
    gold = dict(np.load(os.path.join(golden_dir, "wgan_ref_small.npz")))
    w, steps = int(gold["meta.width"]), int(gold["meta.steps"])
    hp = WR.Hyperparameter(critic_size=w, generator_size=w, critic_hidden_size=w, batchsize=int(gold["meta.batch"]))
    critic, generator = WR.build(hp, seed=1)
    for k, v in critic.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), gold[f"init.C.{k}"], err_msg=k)       # same construction order, same seed (:13,:116)
    for k, v in generator.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), gold[f"init.G.{k}"], err_msg=k)
    c_opt, g_opt = WR.make_optimizers(critic, generator)
    eye = torch.eye(hp.num_classes)
    for k in range(steps):
        real, labels = torch.from_numpy(gold[f"step{k}.real"]), torch.from_numpy(gold[f"step{k}.labels"])
        out = WR.critic_step(critic, generator, c_opt, hp, real, eye[labels], torch.from_numpy(gold[f"step{k}.noise"]),
                             torch.from_numpy(gold[f"step{k}.alpha"]))
        gg = gold[f"step{k}.gradients"]       # steps after the first start from weights that already differ by Adam's fp32 noise
        np.testing.assert_allclose(out["gradients"].numpy(), gg, rtol=1e-4, atol=2e-5 * np.abs(gg).max())
        for name in ("critic_loss", "gradient_penalty", "loss_real"):
            np.testing.assert_allclose(out[name], gold[f"step{k}.{name}"], rtol=2e-5, atol=1e-6, err_msg=f"step{k}.{name}")
        if k % hp.n_critic == 0:
            g = WR.generator_step(critic, generator, g_opt, eye[torch.from_numpy(gold[f"step{k}.fake_idx"])],
                                  torch.from_numpy(gold[f"step{k}.noise_g"]))
            np.testing.assert_allclose(g["generator_loss"], gold[f"step{k}.generator_loss"], rtol=2e-5, atol=1e-6)
        if k == 0:
            for net, tag in ((critic, "C"), (generator, "G")):
                scale = max(float(np.abs(gold[f"step0.grad.{tag}.{n}"]).max()) for n, _ in net.named_parameters())
                for n, p in net.named_parameters():
                    ref = gold[f"step0.grad.{tag}.{n}"]
                    # floor relative to the net's largest gradient: a conv bias in front of InstanceNorm / BatchNorm has a true
                    # gradient of exactly 0, and what any fp32 run stores there is summation noise
                    np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max() + 3e-6 * scale, err_msg=f"{tag}.{n}")
    scale = max(float(np.abs(gold[f"final.grad.C.{n}"]).max()) for n, _ in critic.named_parameters())
    for n, p in critic.named_parameters():
        ref = gold[f"final.grad.C.{n}"]
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max() + 3e-6 * scale, err_msg=n)
    for net, tag in ((critic, "C"), (generator, "G")):
        dead = {n for n, _ in net.named_parameters() if np.abs(gold.get(f"step0.grad.{tag}.{n}", np.ones(1))).max() < 1e-4}
        for k_, v in net.state_dict().items():
            if k_ in dead:      # zero-gradient biases: AdamW(beta1=0) turns the noise sign into a +-lr move per step
                assert np.abs(v.numpy() - gold[f"final.{tag}.{k_}"]).max() <= 2.2 * 1e-4 * steps, k_
                continue
            # a BatchNorm running_mean inherits the +-lr wander of the dead conv bias in front of it
            atol = 1e-4 if k_.endswith("running_mean") else 2e-5
            np.testing.assert_allclose(v.numpy(), gold[f"final.{tag}.{k_}"], rtol=1e-4, atol=atol, err_msg=f"final.{tag}.{k_}")


# ---- evaluation path: oracle restatements against the reference's lifted eval functions -----------
def _house_eval_setup(golden_dir):
    gold = dict(np.load(os.path.join(golden_dir, "house_eval.npz")))
    G, _, clf = HR.build(seed=0)
    G.load_state_dict(torch.load(os.path.join(golden_dir, "house_generator_trained.pt"), map_location="cpu", weights_only=True))
    clf.load_state_dict(torch.load(os.path.join(golden_dir, "house_classifier_trained.pt"), map_location="cpu", weights_only=True))
    G.eval(); clf.eval()
    lo, hi = gold["scaler.data_min"], gold["scaler.data_max"]
    norm = {f: torch.tensor((gold[f"raw_values.{f}"] - lo[f]) / ((hi[f] - lo[f]) + 1e-12), dtype=torch.float32)
            for f in HR.CONFIG["categorical_info"]}                                            # eval_utils.py:56-67
    return gold, G, clf, norm


def test_house_eval_metrics_match_reference(golden_dir):
    gold, G, clf, norm = _house_eval_setup(golden_dir)
    n = int(gold["meta.exact_rows"])
    X, y = torch.from_numpy(gold["X_test"][:n]), gold["y_test"][:n]
    cfs = []
    with torch.no_grad():
        for t in range(4):
            x = X[torch.from_numpy(y != t)]
            gumbel = {f: torch.from_numpy(gold[f"exact.gumbel.{t}.{f}"]) for f in HR.CONFIG["categorical_info"]}
            m = HR.metrics_one_target(G, clf, x, t, gumbel, norm)
            np.testing.assert_allclose([m["class_flip"], m["prediction_gain"], m["avg_actionability"]], gold["exact.metrics"][t],
                                       rtol=1e-5, atol=1e-6, err_msg=f"target {t}")
            cfs.append(m["x_cf"])
    np.testing.assert_allclose(torch.cat(cfs).numpy(), gold["exact.x_cf"], rtol=1e-5, atol=1e-6)
    # the reference's own full run reproduces the metrics file it ships (hard Gumbel draws differ: statistical agreement)
    assert np.abs(gold["full.metrics"] - gold["shipped.metrics"]).max() < 0.01


def test_countergan_eval_metrics_match_reference(golden_dir):
    gold = dict(np.load(os.path.join(golden_dir, "countergan_eval.npz")))
    G, _, _ = CR.build(seed=0)
    G.load_state_dict(torch.load(os.path.join(golden_dir, "countergan_generator_trained.pt"), map_location="cpu", weights_only=True))
    torch.manual_seed(3)
    C = CR.CNNClassifier()
    for k, v in C.state_dict().items():
        np.testing.assert_array_equal(_digest(v), gold[f"C.{k}"], err_msg=k)
    m, (x_vis, x_cf_vis) = CR.evaluate_counterfactuals(G, C, torch.from_numpy(gold["x"]), torch.from_numpy(gold["y_true"]),
                                                       torch.from_numpy(gold["y_target"]))
    np.testing.assert_allclose([m["class_flip_rate"], m["prediction_gain"], m["actionability"]], gold["metrics"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x_cf_vis.numpy(), gold["x_cf_vis"], rtol=1e-5, atol=1e-6)


# ---- classifier pre-training: oracle steps against the reference's own train_classifier runs ------
def test_mnist_classifier_pretraining_matches_reference(golden_dir):
    import re
    gold = dict(np.load(os.path.join(golden_dir, "classifier_pretrain_mnist.npz")))
    torch.manual_seed(5)
    C = CR.CNNClassifier()
    for k, v in C.state_dict().items():
        np.testing.assert_array_equal(_digest(v), gold[f"init.{k}"], err_msg=k)
    opt = torch.optim.Adam(C.parameters(), lr=1e-3)                                          # trainer.py:9
    for i in range(2):
        CR.classifier_train_step(C, opt, torch.from_numpy(gold[f"x{i}"]), torch.from_numpy(gold[f"y{i}"]),
                                 (torch.from_numpy(gold[f"mask{2 * i}"]), torch.from_numpy(gold[f"mask{2 * i + 1}"])))
    for k, v in C.state_dict().items():
        np.testing.assert_allclose(_digest(v), gold[f"final.{k}"], rtol=2e-4, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(C.state_dict()["fc.4.weight"].numpy(), gold["final.fc.4.weight.full"], rtol=1e-4, atol=2e-5)
    C.eval()
    with torch.no_grad():
        acc = (C(torch.from_numpy(gold["x2"])).argmax(1) == torch.from_numpy(gold["y2"])).float().mean().item()
    assert abs(acc - float(re.search(r"Val Acc: ([0-9.]+)", str(gold["log"])).group(1))) < 1e-4


def _house_clf_gold(golden_dir):
    gold = dict(np.load(os.path.join(golden_dir, "classifier_pretrain_house.npz")))
    rows = np.concatenate([gold[f"step{i}.rows"] for i in range(int(gold["meta.steps"]))])
    cw = HR.balanced_class_weights(gold["y"][rows], 4)                                       # weights from the training split (:53)
    return gold, torch.tensor(cw, dtype=torch.float32)


def house_clf_dead_entries(gold, cw):
    """Parameter entries whose true gradient is (near) zero at some step — e.g. the bias of a first-layer unit whose
    pre-activations keep one sign over the whole batch: LeakyReLU is then linear there and the BatchNorm1d that follows cancels
    a shift exactly.  fp32 implementations hold summation noise in those entries and Adam turns its sign into +-lr steps, so
    they can only be bounded, not compared.  Found with the oracle in float64."""
    clf = HR.NNClassifier(17, 4)
    clf.load_state_dict({k[5:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.")})
    clf = clf.double()
    opt = torch.optim.AdamW(clf.parameters(), lr=1e-3, weight_decay=1e-4)
    X, y = torch.from_numpy(gold["X"]).double(), torch.from_numpy(gold["y"])
    dead = {n: torch.zeros_like(p, dtype=torch.bool) for n, p in clf.named_parameters()}
    for i in range(int(gold["meta.steps"])):
        r = torch.from_numpy(gold[f"step{i}.rows"])
        HR.classifier_train_step(clf, opt, cw.double(), X[r], y[r], [torch.from_numpy(gold[f"step{i}.mask{j}"]) for j in range(3)])
        for n, p in clf.named_parameters():
            dead[n] |= p.grad.abs() < 1e-6 * p.grad.abs().max()
    return {n: d.numpy() for n, d in dead.items()}


def adam_close(got, ref, lr, steps, what, dead=None, rtol=2e-4, atol=2e-5):
    """Parameters after a few Adam steps: equal within rtol/atol; `dead` entries (see house_clf_dead_entries) and at most
    max(2, 1 %) further entries (gradients that merely come close to zero) are only bounded by the total possible move."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    bad = d > atol + rtol * np.abs(ref)
    if dead is not None:
        bad &= ~dead
    assert bad.sum() <= max(2, 0.01 * d.size) and d.max() <= 2.2 * lr * steps + atol, (what, int(bad.sum()), float(d.max()))


def test_house_classifier_pretraining_matches_reference(golden_dir):
    gold, cw = _house_clf_gold(golden_dir)
    dead = house_clf_dead_entries(gold, cw)
    clf = HR.NNClassifier(17, 4)
    clf.load_state_dict({k[5:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.")})
    opt = torch.optim.AdamW(clf.parameters(), lr=1e-3, weight_decay=1e-4)                    # trainer.py:58
    X, y = torch.from_numpy(gold["X"]), torch.from_numpy(gold["y"])
    for i in range(int(gold["meta.steps"])):
        r = torch.from_numpy(gold[f"step{i}.rows"])
        HR.classifier_train_step(clf, opt, cw, X[r], y[r], [torch.from_numpy(gold[f"step{i}.mask{j}"]) for j in range(3)])
    for k, v in clf.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            # statistics of activations whose bias entries wandered by +-lr: bounded by that wander
            np.testing.assert_allclose(v.numpy(), gold[f"final.{k}"], rtol=1e-3, atol=1e-3, err_msg=k)
            continue
        adam_close(v.numpy(), gold[f"final.{k}"], lr=1e-3, steps=3, what=k, dead=dead.get(k))

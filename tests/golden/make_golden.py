#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own code on the CPU (this container only: /root/reference
does not travel to the GPU box; the .npz files written next to this script do).

DCGAN (`dconv_gan/mnist/mnist_dcgan.py`) cannot be imported (torchvision, /mnt/data, trains at import), so the
pure-torch pieces are lifted out of its syntax tree and executed unmodified:
  * `weights_init`, `Generator`, `Discriminator`                 (lines 63-116)
  * the construction of nets / loss / optimizers                  (lines 119-127)
  * the body of the inner training loop up to optimizerG.step()   (lines 147-175)
Nothing of the reference's text is stored in this repo — only the numbers it produces.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/dcgan_ref_small.npz)
"""
import ast
import os
import sys

import numpy as np
import torch

REF = os.environ.get("PCG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _lift_dcgan(config):
    path = os.path.join(REF, "dconv_gan/mnist/mnist_dcgan.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    defs, setup, body = [], [], []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in ("weights_init", "Generator", "Discriminator"):
            defs.append(node)
        elif isinstance(node, (ast.Assign, ast.Expr)) and 119 <= node.lineno <= 127:
            setup.append(node)  # netG/netD creation + .apply(weights_init), criterion, optimizerD/G
        elif isinstance(node, ast.For) and node.lineno == 140:
            inner = [n for n in node.body if isinstance(n, ast.For)]
            assert len(inner) == 1 and inner[0].lineno == 143, "reference layout changed"
            body = [n for n in inner[0].body if 147 <= n.lineno <= 175]
    assert len(defs) == 3 and len(setup) == 7 and body and body[-1].lineno == 175, (len(defs), len(setup), len(body))
    ns = {"torch": torch, "nn": torch.nn, "optim": torch.optim, "config": config, "device": torch.device("cpu")}
    exec(compile(ast.Module(body=defs, type_ignores=[]), path, "exec"), ns)
    setup_code = compile(ast.Module(body=setup, type_ignores=[]), path, "exec")
    step_code = compile(ast.Module(body=body, type_ignores=[]), path, "exec")
    return ns, setup_code, step_code


def _sd(prefix, module, out):
    for k, v in module.state_dict().items():
        out[f"{prefix}.{k}"] = v.detach().cpu().numpy().copy()


def make_dcgan_small(path, g_hidden=8, d_hidden=8, z_dim=16, batch=4, steps=3):
    config = {"image_channel": 1, "z_dim": z_dim, "g_hidden": g_hidden, "d_hidden": d_hidden, "real_label": 1.0,
              "fake_label": 0.0, "lr": 2e-4, "seed": 1, "batch_size": batch}
    ns, setup_code, step_code = _lift_dcgan(config)
    torch.manual_seed(config["seed"])  # mnist_dcgan.py:33
    exec(setup_code, ns)
    netG, netD = ns["netG"], ns["netD"]
    out = {"meta.g_hidden": np.int64(g_hidden), "meta.d_hidden": np.int64(d_hidden), "meta.z_dim": np.int64(z_dim),
           "meta.batch": np.int64(batch), "meta.steps": np.int64(steps)}
    _sd("init.G", netG, out)
    _sd("init.D", netD, out)

    # single forward passes at init (train mode, but on copies so running stats of the trained nets are untouched)
    import copy
    gen = torch.Generator().manual_seed(7)
    real0 = torch.rand(batch, 1, 64, 64, generator=gen) * 2 - 1
    z0 = torch.randn(batch, z_dim, 1, 1, generator=gen)
    g0, d0 = copy.deepcopy(netG), copy.deepcopy(netD)
    out["fwd.real"], out["fwd.z"] = real0.numpy(), z0.numpy()
    out["fwd.G_out"] = g0(z0).detach().numpy()
    out["fwd.D_out"] = d0(real0).detach().numpy()
    g0.eval(); d0.eval()
    out["fwd.G_out_eval"] = g0(z0).detach().numpy()
    out["fwd.D_out_eval"] = d0(real0).detach().numpy()

    for k in range(steps):
        gen = torch.Generator().manual_seed(100 + k)
        real = torch.rand(batch, 1, 64, 64, generator=gen) * 2 - 1
        # the loop body draws its noise from the global RNG (:156); replay that draw to record it
        torch.manual_seed(1000 + k)
        noise = torch.randn(batch, z_dim, 1, 1)
        torch.manual_seed(1000 + k)
        ns["data"] = (real,)
        exec(step_code, ns)
        out[f"step{k}.real"], out[f"step{k}.noise"] = real.numpy(), noise.numpy()
        for name in ("errD_real", "errD_fake", "errD", "errG"):
            out[f"step{k}.{name}"] = np.float32(ns[name].item())
        for name in ("D_x", "D_G_z1", "D_G_z2"):
            out[f"step{k}.{name}"] = np.float32(ns[name])
        assert torch.equal(ns["noise"], noise)
    _sd("final.G", netG, out)
    _sd("final.D", netD, out)
    for n, p in netG.named_parameters():
        out[f"final.G.grad.{n}"] = p.grad.detach().numpy().copy()
    for n, p in netD.named_parameters():   # = D-step grads + the G-step's D wgrad? no: zeroed at :147, so last D-step + G-step
        out[f"final.D.grad.{n}"] = p.grad.detach().numpy().copy()
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


def make_dcgan_loop(path, g_hidden=8, d_hidden=8, z_dim=16, batch=4, epochs=2, nbatches=3):
    """The reference's OUTER loop (mnist_dcgan.py:129-198: viz noise, epochs x batches, running losses, the train-mode viz forward
    every 500 iterations and at the end) lifted from the syntax tree and executed unmodified on a seeded synthetic `dataloader`.
    `vutils.make_grid` (torchvision, image-grid plotting, :190) is outside the numerical path: a pass-through keeps the raw batch."""
    import types
    config = {"image_channel": 1, "z_dim": z_dim, "g_hidden": g_hidden, "d_hidden": d_hidden, "real_label": 1.0,
              "fake_label": 0.0, "lr": 2e-4, "seed": 1, "batch_size": batch, "epochs": epochs}
    ns, setup_code, _ = _lift_dcgan(config)
    src = os.path.join(REF, "dconv_gan/mnist/mnist_dcgan.py")
    with open(src) as f:
        tree = ast.parse(f.read(), filename=src)
    loop = [n for n in tree.body if 129 <= n.lineno <= 198]
    assert loop and isinstance(loop[-1], ast.For) and loop[-1].lineno == 140 and loop[-1].end_lineno == 198, "reference layout changed"
    loop_code = compile(ast.Module(body=loop, type_ignores=[]), src, "exec")
    torch.manual_seed(config["seed"])                      # :33
    exec(setup_code, ns)                                   # nets + init + optimizers, :119-127
    out = {"meta.g_hidden": np.int64(g_hidden), "meta.d_hidden": np.int64(d_hidden), "meta.z_dim": np.int64(z_dim),
           "meta.batch": np.int64(batch), "meta.epochs": np.int64(epochs), "meta.nbatches": np.int64(nbatches)}
    _sd("init.G", ns["netG"], out)
    _sd("init.D", ns["netD"], out)
    gen = torch.Generator().manual_seed(4242)
    data = [(torch.rand(batch, 1, 64, 64, generator=gen) * 2 - 1,) for _ in range(nbatches)]
    for k, (r,) in enumerate(data):
        out[f"data.{k}"] = r.numpy()
    ns["dataloader"] = data
    ns["vutils"] = types.SimpleNamespace(make_grid=lambda t, **kw: t)
    torch.manual_seed(777)                                 # the loop's global-RNG draws: viz_noise (:130), then one noise per iteration (:156)
    out["meta.loop_seed"] = np.int64(777)
    exec(loop_code, ns)
    out["epoch_G_losses"] = np.asarray(ns["epoch_G_losses"], np.float64)
    out["epoch_D_losses"] = np.asarray(ns["epoch_D_losses"], np.float64)
    out["iters"] = np.int64(ns["iters"])
    out["viz_noise"] = ns["viz_noise"].numpy()
    for k, img in enumerate(ns["img_list"]):
        out[f"img.{k}"] = img.numpy()
    _sd("final.G", ns["netG"], out)
    _sd("final.D", ns["netD"], out)
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


def tensor_digest(t, nsamples=64):
    """[sum, sum|.|, sum of squares] in float64 + `nsamples` strided samples: a compact pin for a large tensor."""
    a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float64).ravel()
    idx = np.linspace(0, a.size - 1, num=min(nsamples, a.size)).astype(np.int64)
    return np.concatenate([[a.sum(), np.abs(a).sum(), (a * a).sum()], a[idx]])


def make_countergan(path, batch=4, seed=0):
    """CounteRGAN/mnist: the reference's own modules (models/*.py, trainer.py) imported from the mounted checkout.
    Nets are seeded (torch.manual_seed(seed); classifier, generator, discriminator constructed in main.py's order,
    :18-20) rather than stored: 3.2 M parameters would not be a small fixture.  Stored: inputs, forward outputs, the
    scalars of one training step (trainer.py:96-123 executed through the reference's train_countergan), and digests
    (sums + strided samples) of every initial parameter, every gradient and every updated parameter."""
    mdir = os.path.join(REF, "conditional_counteRGAN/mnist")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    sys.path.insert(0, mdir)
    import importlib
    for name in ("config", "models", "models.generator", "models.discriminator", "models.classifier", "trainer"):
        sys.modules.pop(name, None)
    cfgmod = importlib.import_module("config")
    gen_mod = importlib.import_module("models.generator")
    dis_mod = importlib.import_module("models.discriminator")
    cls_mod = importlib.import_module("models.classifier")
    trainer = importlib.import_module("trainer")
    cfg = cfgmod.Config()

    torch.manual_seed(seed)
    classifier = cls_mod.CNNClassifier(num_classes=cfg.num_classes)
    generator = gen_mod.ResidualGenerator(img_shape=cfg.img_shape, num_classes=cfg.num_classes)
    discriminator = dis_mod.Discriminator(img_shape=cfg.img_shape, num_classes=cfg.num_classes)
    classifier.eval()
    for p_ in classifier.parameters():
        p_.requires_grad = False

    out = {"meta.batch": np.int64(batch), "meta.seed": np.int64(seed)}
    for tag, net in (("G", generator), ("D", discriminator), ("C", classifier)):
        for k, v in net.state_dict().items():
            out[f"init.{tag}.{k}"] = tensor_digest(v.float())

    g = torch.Generator().manual_seed(123)
    x = torch.rand(batch, 1, 28, 28, generator=g) * 2 - 1
    y = torch.randint(0, 10, (batch,), generator=g)
    target_y = torch.randint(0, 10, (batch,), generator=g)
    torch.manual_seed(77)
    mask = trainer.build_mask(x, cfg.patch_size, "cpu", cfg.num_modifiable_patches)
    out.update({"in.x": x.numpy(), "in.y": y.numpy(), "in.target_y": target_y.numpy(), "in.mask": mask.numpy()})
    assert float(mask.sum()) == batch * cfg.num_modifiable_patches * cfg.patch_size ** 2

    # forward-only vectors (train mode, on copies)
    import copy
    g0, d0 = copy.deepcopy(generator), copy.deepcopy(discriminator)
    raw, masked = g0(x, target_y, mask)
    out["fwd.raw"], out["fwd.masked"] = raw.detach().numpy(), masked.detach().numpy()
    out["fwd.d_logits"] = d0(x, y).detach().numpy()
    out["fwd.c_logits"] = classifier(x).detach().numpy()

    # one iteration through the reference's own train_countergan: a one-batch loader, one epoch; the loop draws
    # target_y (:94) and the mask (:95) from the global RNG -> replay the draws to record them
    cfg.num_epochs_gan = 1
    cfg.save_dir = "/tmp/pcg_golden_out"
    cfg.generator_path = os.path.join(cfg.save_dir, "generator.pt")
    os.makedirs(cfg.save_dir, exist_ok=True)
    torch.manual_seed(999)
    t_rec = torch.randint(0, cfg.num_classes, (batch,))
    m_rec = trainer.build_mask(x, cfg.patch_size, "cpu", cfg.num_modifiable_patches)
    torch.manual_seed(999)
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trainer.train_countergan(generator, discriminator, classifier, [(x, y)], cfg, "cpu")
    out["step.target_y"], out["step.mask"] = t_rec.numpy(), m_rec.numpy()
    out["step.log"] = np.array(buf.getvalue())
    for tag, net in (("G", generator), ("D", discriminator)):
        for k, v in net.state_dict().items():
            out[f"final.{tag}.{k}"] = tensor_digest(v.float())
        for n, p_ in net.named_parameters():
            out[f"grad.{tag}.{n}"] = tensor_digest(p_.grad)
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")
    print(buf.getvalue().strip().splitlines()[0])


def make_countergan_loop(path, batch=4, nbatches=3, epochs=2, seed=0):
    """CounteRGAN/mnist: the reference's whole train_countergan (trainer.py:76-163) over `epochs` x `nbatches` batches — per-epoch
    means, the per-epoch G_grad / D_grad print (:142-147, the one place grad_norm :41-42 is used; D's .grad there = its D-step
    gradients + the generator step's critic weight gradients, no zeroing between :111 and :122), residual_mean in the batch log
    line (:137), the generator checkpoint (:162).  Nets seeded as in make_countergan; recorded: the batches, the draws each
    iteration made (captured at the generator's input), grad_norm's return values, the log, digests of the saved generator."""
    mdir = os.path.join(REF, "conditional_counteRGAN/mnist")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    sys.path.insert(0, mdir)
    import contextlib, importlib, io
    for name in list(sys.modules):
        if name in ("config", "trainer", "data_utils") or name == "models" or name.startswith("models."):
            sys.modules.pop(name)
    cfgmod = importlib.import_module("config")
    gen_mod = importlib.import_module("models.generator")
    dis_mod = importlib.import_module("models.discriminator")
    cls_mod = importlib.import_module("models.classifier")
    trainer = importlib.import_module("trainer")
    cfg = cfgmod.Config()
    torch.manual_seed(seed)
    classifier = cls_mod.CNNClassifier(num_classes=cfg.num_classes)
    generator = gen_mod.ResidualGenerator(img_shape=cfg.img_shape, num_classes=cfg.num_classes)
    discriminator = dis_mod.Discriminator(img_shape=cfg.img_shape, num_classes=cfg.num_classes)
    classifier.eval()
    for p_ in classifier.parameters():
        p_.requires_grad = False
    out = {"meta.batch": np.int64(batch), "meta.seed": np.int64(seed), "meta.epochs": np.int64(epochs), "meta.nbatches": np.int64(nbatches)}
    for tag, net in (("G", generator), ("D", discriminator), ("C", classifier)):
        for k, v in net.state_dict().items():
            out[f"init.{tag}.{k}"] = tensor_digest(v.float())
    g = torch.Generator().manual_seed(321)
    loader = [(torch.rand(batch, 1, 28, 28, generator=g) * 2 - 1, torch.randint(0, 10, (batch,), generator=g)) for _ in range(nbatches)]
    out["data.x"] = torch.stack([b[0] for b in loader]).numpy()
    out["data.y"] = torch.stack([b[1] for b in loader]).numpy()
    cfg.num_epochs_gan = epochs
    cfg.save_dir = "/tmp/pcg_golden_out_loop"
    cfg.generator_path = os.path.join(cfg.save_dir, "generator.pt")
    os.makedirs(cfg.save_dir, exist_ok=True)
    calls, norms = [], []
    h = generator.register_forward_pre_hook(lambda mod, args: calls.append((args[1].clone(), args[2].clone())))
    real_gn = trainer.grad_norm

    def rec_gn(params):
        v = real_gn(params)
        norms.append(v)
        return v
    trainer.grad_norm = rec_gn
    torch.manual_seed(555)
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            trainer.train_countergan(generator, discriminator, classifier, loader, cfg, "cpu")
    finally:
        trainer.grad_norm = real_gn
        h.remove()
    assert len(calls) == epochs * nbatches and len(norms) == 2 * epochs
    out["it.target_y"] = torch.stack([c[0] for c in calls]).reshape(epochs, nbatches, batch).numpy()
    out["it.mask"] = torch.stack([c[1] for c in calls]).reshape(epochs, nbatches, batch, 1, 28, 28).numpy()
    out["epoch.G_grad"], out["epoch.D_grad"] = np.array(norms[0::2]), np.array(norms[1::2])
    out["log"] = np.array(buf.getvalue())
    saved = torch.load(cfg.generator_path, map_location="cpu", weights_only=True)
    out["saved.keys"] = np.array(list(saved.keys()))
    for k, v in saved.items():
        out[f"saved.G.{k}"] = tensor_digest(v.float())
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")
    print("\n".join(l for l in buf.getvalue().splitlines() if l.startswith("[GAN]")))


def make_moons(path, n=100, seed=5):
    """simple_gan/moons/make_moons_gan.py: build_generator / build_discriminator / train_gan lifted from the syntax tree
    (the module trains and plots at import).  One epoch over `n` seeded 2-D points (2 batches of 50); train_gan draws its
    noise from the global torch RNG (:64,:79) and shuffles with numpy (:57): both are seeded and the draws replayed."""
    path_src = os.path.join(REF, "simple_gan/moons/make_moons_gan.py")
    with open(path_src) as f:
        tree = ast.parse(f.read(), filename=path_src)
    defs = [n_ for n_ in tree.body if isinstance(n_, ast.FunctionDef) and n_.name in ("build_generator", "build_discriminator", "train_gan")]
    assert len(defs) == 3
    ns = {"torch": torch, "nn": torch.nn, "np": np, "device": torch.device("cpu")}
    exec(compile(ast.Module(body=defs, type_ignores=[]), path_src, "exec"), ns)
    cfg = {"z_dim": 32, "hidden_dim": 128, "batch_size": 50, "lr": 1e-3, "epochs": 1}
    torch.manual_seed(seed)
    G = ns["build_generator"](cfg["z_dim"], cfg["hidden_dim"])
    D = ns["build_discriminator"](cfg["hidden_dim"])
    out = {}
    for tag, net in (("G", G), ("D", D)):
        for k, v in net.state_dict().items():
            out[f"init.{tag}.{k}"] = v.numpy().copy()
    rs = np.random.RandomState(seed)
    X = rs.standard_normal((n, 2)).astype(np.float64)
    np.random.seed(seed + 1)
    Xs = X.copy(); np.random.shuffle(Xs)             # replay of :57
    torch.manual_seed(seed + 2)
    zs = [torch.randn(cfg["batch_size"], cfg["z_dim"]) for _ in range(2 * (n // cfg["batch_size"]))]   # replay of :64,:79
    np.random.seed(seed + 1); torch.manual_seed(seed + 2)
    lossD, lossG = ns["train_gan"](X, G, D, cfg)
    out["X_shuffled"] = Xs.astype(np.float32)
    out["z"] = torch.stack(zs).numpy()
    out["loss_D_total"], out["loss_G_total"] = np.float64(lossD[0]), np.float64(lossG[0])
    for tag, net in (("G", G), ("D", D)):
        for k, v in net.state_dict().items():
            out[f"final.{tag}.{k}"] = v.numpy().copy()
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.3f} MB; loss_D {lossD[0]:.5f} loss_G {lossG[0]:.5f}")


def make_countergan_trained(path_npz, path_pt, batch=8):
    """Real-weight anchor: the generator checkpoint the reference ships (conditional_counteRGAN/mnist/results/generator.pt,
    8,440 training steps — SURVEY.md §6) loaded into the reference's own ResidualGenerator, eval mode (BatchNorm running
    statistics), forward on seeded inputs.  The checkpoint is data: it is copied next to the vectors as a fixture."""
    import shutil
    mdir = os.path.join(REF, "conditional_counteRGAN/mnist")
    sys.path.insert(0, mdir)
    import importlib
    gen_mod = importlib.import_module("models.generator")
    src = os.path.join(mdir, "results/generator.pt")
    sd = torch.load(src, map_location="cpu", weights_only=True)
    G = gen_mod.ResidualGenerator()
    G.load_state_dict(sd)
    G.eval()
    g = torch.Generator().manual_seed(2024)
    x = torch.rand(batch, 1, 28, 28, generator=g) * 2 - 1
    t = torch.randint(0, 10, (batch,), generator=g)
    pm = torch.zeros(batch, 16)
    for b in range(batch):
        pm[b, torch.randperm(16, generator=g)[:10]] = 1.0
    mask = torch.nn.functional.interpolate(pm.view(batch, 1, 4, 4), size=(28, 28), mode="nearest")
    with torch.no_grad():
        raw, masked = G(x, t, mask)
        x_cf = torch.clamp(x + masked, -1.0, 1.0)
    np.savez_compressed(path_npz, x=x.numpy(), target=t.numpy(), mask=mask.numpy(), raw=raw.numpy(), masked=masked.numpy(),
                        x_cf=x_cf.numpy())
    shutil.copyfile(src, path_pt)
    os.chmod(path_pt, 0o644)
    print(f"wrote {path_npz} and {path_pt} ({os.path.getsize(path_pt) / 1e6:.2f} MB); |raw| max {raw.abs().max().item():.4f}")


def make_house(path, bs=64):
    """conditional_counteRGAN/house_sales_kc_usa: ONE batch through the reference's own train_countergan (trainer.py:186-378,
    epochs=1, a dataset of exactly one batch).  The loop draws everything from the global torch RNG after
    torch.manual_seed(config['seed']); a forward pre-hook on the generator captures what it was called with (the shuffled
    batch, the target one-hots, the feature mask) and the RNG state, from which the Gumbel noise of F.gumbel_softmax
    (generator.py:90) is replayed; the discriminator, which train_countergan creates internally right after seeding, is
    re-created the same way to record its initial state."""
    import contextlib, importlib, io
    mdir = os.path.join(REF, "conditional_counteRGAN/house_sales_kc_usa")
    scratch = "/tmp/pcg_golden_house"
    os.makedirs(scratch, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(scratch)                      # config.py creates results/ in the cwd at import (config.py:4-11)
    try:
        sys.path.insert(0, mdir)
        for name in list(sys.modules):
            if name in ("config", "trainer", "data_utils") or name == "models" or name.startswith("models."):
                sys.modules.pop(name)
        cfg = importlib.import_module("config").config
        gen_mod = importlib.import_module("models.generator")
        dis_mod = importlib.import_module("models.discriminator")
        clf_mod = importlib.import_module("models.nn_classifier")
        trainer = importlib.import_module("trainer")
        cfg.update({"cuda": "cpu", "epochs": 1, "batch_size": bs, "scaler": None, "out_dir": scratch,
                    "generator_path": os.path.join(scratch, "gen.pt")})
        torch.manual_seed(0)
        clf = clf_mod.NNClassifier(cfg["input_dim"], output_dim=cfg["num_classes"])
        G = gen_mod.ResidualGenerator(cfg["input_dim"], cfg["hidden_dim"], cfg["num_classes"], continuous_idx=cfg["continuous_idx"],
                                      categorical_info={k: {"n": v["n"], "raw_values": v["raw_values"]} for k, v in cfg["categorical_info"].items()},
                                      tau=cfg["gumbel_tau"])
        clf.eval()
        for p_ in clf.parameters():
            p_.requires_grad = False
        out = {"meta.bs": np.int64(bs)}
        for k, v in G.state_dict().items():
            out[f"init.G.{k}"] = v.numpy().copy()
        for k, v in clf.state_dict().items():
            out[f"init.C.{k}"] = tensor_digest(v.float())
        rs = np.random.RandomState(11)
        X = rs.random_sample((bs, cfg["input_dim"])).astype(np.float32)
        y = np.arange(bs) % cfg["num_classes"]
        rs.shuffle(y)
        cap = {}

        def pre_hook(mod, args, kwargs):
            cap["x"], cap["target_onehot"] = args[0].clone(), args[1].clone()
            cap["mask"], cap["temperature"], cap["hard"] = kwargs["mask"].clone(), kwargs["temperature"], kwargs["hard"]
            cap["rng"] = torch.get_rng_state()
        h = G.register_forward_pre_hook(pre_hook, with_kwargs=True)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            trainer.train_countergan(G, cfg, X, y, clf)
        h.remove()
        assert cap["hard"] is False and abs(cap["temperature"] - 0.5) < 1e-12
        # replay: discriminator init (train_countergan: torch.manual_seed(42) then Discriminator(...), :188,:227)
        torch.manual_seed(cfg["seed"])
        D0 = dis_mod.Discriminator(cfg["input_dim"], cfg["hidden_dim"], cfg["num_classes"])
        for k, v in D0.state_dict().items():
            out[f"init.D.{k}"] = v.numpy().copy()
        # replay: Gumbel noise, in the order of the heads (generator.py:86-90)
        torch.set_rng_state(cap["rng"])
        for idx_str, head in G.fc_cat_logits.items():
            out[f"gumbel.{idx_str}"] = (-torch.empty(bs, head.out_features).exponential_().log()).numpy()
        # the DataLoader shuffled the rows: recover the permutation by matching rows
        xb = cap["x"].numpy()
        perm = np.array([int(np.argmin(np.abs(X - r).sum(1))) for r in xb])
        assert np.array_equal(X[perm], xb)
        out.update({"in.x": xb, "in.y": y[perm].astype(np.int64), "in.target_y": cap["target_onehot"].argmax(1).numpy(),
                    "in.mask": cap["mask"].numpy(), "log": np.array(buf.getvalue())})
        for k, v in G.state_dict().items():
            out[f"final.G.{k}"] = v.numpy().copy()
        for n_, p_ in G.named_parameters():
            out[f"grad.G.{n_}"] = p_.grad.numpy().copy()
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")
        print(buf.getvalue().strip().splitlines()[0])
    finally:
        os.chdir(cwd)


def make_house_loop(path, bs=64, nbatches=3, epochs=2, extra_rows=17):
    """conditional_counteRGAN/house_sales_kc_usa: the WHOLE of the reference's train_countergan (trainer.py:186-378) — seeding,
    DataLoader(shuffle=True, drop_last=True) over nbatches*bs + extra_rows rows (the remainder is dropped every epoch), the
    Discriminator built inside, cat_norm_maps from config['scaler'], `epochs` x `nbatches` iterations, the four per-iteration
    diagnostics (:318-343), the epoch summaries with grad_norm (:357-366), torch.save of the generator (:377).  Recorded: per
    iteration what the generator was called with (shuffled rows, target one-hots, feature mask) and the RNG state from which the
    Gumbel noise is replayed; the lists the trainer averages at each epoch end (np.mean is wrapped for the run: full precision
    instead of the printed four digits); grad_norm's return values; the log; the saved generator."""
    import contextlib, importlib, io
    mdir = os.path.join(REF, "conditional_counteRGAN/house_sales_kc_usa")
    scratch = "/tmp/pcg_golden_house_loop"
    os.makedirs(scratch, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        sys.path.insert(0, mdir)
        for name in list(sys.modules):
            if name in ("config", "trainer", "data_utils") or name == "models" or name.startswith("models."):
                sys.modules.pop(name)
        cfg = importlib.import_module("config").config
        gen_mod = importlib.import_module("models.generator")
        dis_mod = importlib.import_module("models.discriminator")
        clf_mod = importlib.import_module("models.nn_classifier")
        trainer = importlib.import_module("trainer")

        class Scaler:                      # the two attributes train_countergan reads of the fitted MinMaxScaler (:207-209)
            pass
        rs = np.random.RandomState(23)
        n_rows = nbatches * bs + extra_rows
        sc = Scaler()
        sc.data_min_ = np.zeros(cfg["input_dim"])
        sc.data_max_ = np.ones(cfg["input_dim"])
        for f, info in cfg["categorical_info"].items():       # categorical columns span their raw values, like the real data
            sc.data_min_[f], sc.data_max_[f] = float(min(info["raw_values"])), float(max(info["raw_values"]))
        gpath = os.path.join(scratch, "gen_loop.pt")
        cfg.update({"cuda": "cpu", "epochs": epochs, "batch_size": bs, "scaler": sc, "out_dir": scratch, "generator_path": gpath})
        torch.manual_seed(0)
        clf = clf_mod.NNClassifier(cfg["input_dim"], output_dim=cfg["num_classes"])
        G = gen_mod.ResidualGenerator(cfg["input_dim"], cfg["hidden_dim"], cfg["num_classes"], continuous_idx=cfg["continuous_idx"],
                                      categorical_info={k: {"n": v["n"], "raw_values": v["raw_values"]} for k, v in cfg["categorical_info"].items()},
                                      tau=cfg["gumbel_tau"])
        clf.eval()
        for p_ in clf.parameters():
            p_.requires_grad = False
        out = {"meta.bs": np.int64(bs), "meta.epochs": np.int64(epochs), "meta.nbatches": np.int64(nbatches), "meta.seed": np.int64(cfg["seed"]),
               "scaler.data_min": sc.data_min_.copy(), "scaler.data_max": sc.data_max_.copy()}
        for f, info in cfg["categorical_info"].items():
            out[f"raw_values.{f}"] = np.asarray(info["raw_values"], dtype=np.float64)
        for k, v in G.state_dict().items():
            out[f"init.G.{k}"] = v.numpy().copy()
        for k, v in clf.state_dict().items():
            out[f"init.C.{k}"] = tensor_digest(v.float())
        X = rs.random_sample((n_rows, cfg["input_dim"])).astype(np.float32)
        for f, info in cfg["categorical_info"].items():       # categorical columns hold scaled category values
            raw = np.asarray(info["raw_values"], dtype=float)
            X[:, f] = ((rs.choice(raw, n_rows) - sc.data_min_[f]) / (sc.data_max_[f] - sc.data_min_[f])).astype(np.float32)
        y = np.arange(n_rows) % cfg["num_classes"]
        rs.shuffle(y)
        out["data.X"], out["data.y"] = X.copy(), y.astype(np.int64)
        calls, means, norms = [], [], []

        def pre_hook(mod, args, kwargs):
            calls.append({"x": args[0].clone(), "t": args[1].clone(), "mask": kwargs["mask"].clone(), "rng": torch.get_rng_state()})
            assert kwargs["hard"] is False and abs(kwargs["temperature"] - 0.5) < 1e-12
        h = G.register_forward_pre_hook(pre_hook, with_kwargs=True)
        real_mean, real_gn = np.mean, trainer.grad_norm

        def rec_mean(a, *args, **kw):
            if isinstance(a, list) and not args and not kw:
                means.append(np.array(a, dtype=np.float64))
            return real_mean(a, *args, **kw)

        def rec_gn(params):
            v = real_gn(params)
            norms.append(v)
            return v
        buf = io.StringIO()
        np.mean, trainer.grad_norm = rec_mean, rec_gn
        try:
            with contextlib.redirect_stdout(buf):
                trainer.train_countergan(G, cfg, X, y, clf)
        finally:
            np.mean, trainer.grad_norm = real_mean, real_gn
        h.remove()
        assert len(calls) == epochs * nbatches and len(means) == 6 * epochs and len(norms) == 2 * epochs
        torch.manual_seed(cfg["seed"])                        # :188 then :227
        D0 = dis_mod.Discriminator(cfg["input_dim"], cfg["hidden_dim"], cfg["num_classes"])
        for k, v in D0.state_dict().items():
            out[f"init.D.{k}"] = v.numpy().copy()
        T = sum(hd.out_features for hd in G.fc_cat_logits.values())
        xs, ts, ms, gs, perms = [], [], [], [], []
        for c in calls:
            torch.set_rng_state(c["rng"])
            gs.append(torch.cat([-torch.empty(bs, hd.out_features).exponential_().log() for hd in G.fc_cat_logits.values()], 1).numpy())
            xb = c["x"].numpy()
            rows = np.array([int(np.argmin(np.abs(X - r).sum(1))) for r in xb])
            assert np.array_equal(X[rows], xb)
            perms.append(rows); ts.append(c["t"].argmax(1).numpy()); ms.append(c["mask"].numpy())
        assert gs[0].shape == (bs, T)
        out["it.rows"] = np.stack(perms).reshape(epochs, nbatches, bs).astype(np.int64)
        out["it.target_y"] = np.stack(ts).reshape(epochs, nbatches, bs).astype(np.int64)
        out["it.mask"] = np.stack(ms).reshape(epochs, nbatches, bs, -1).astype(np.float32)
        out["it.gumbel"] = np.stack(gs).reshape(epochs, nbatches, bs, T).astype(np.float32)
        out["head_order"] = np.array([int(k) for k in G.fc_cat_logits.keys()], dtype=np.int64)
        names = ("d_loss", "g_loss", "pred_gain", "sparsity", "l2_reg", "class_flip_rate")        # the order of the np.mean calls (:350-355)
        for i, nme in enumerate(names):
            out[f"it.{nme}"] = np.stack([means[6 * e + i] for e in range(epochs)])
        out["epoch.G_grad"] = np.array(norms[0::2]); out["epoch.D_grad"] = np.array(norms[1::2])
        out["log"] = np.array(buf.getvalue())
        saved = torch.load(gpath, map_location="cpu", weights_only=True)
        for k, v in saved.items():
            out[f"saved.G.{k}"] = v.numpy().copy()
            assert torch.equal(v, G.state_dict()[k])
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")
        print("\n".join(l for l in buf.getvalue().splitlines() if l.startswith("[1/") or l.startswith("[2/")))
    finally:
        os.chdir(cwd)


def make_house_trained(path_npz, path_g, path_c, bs=32):
    """Real-weight anchor for the tabular path: the generator and classifier checkpoints the reference ships
    (house_sales_kc_usa/generator_model.pt, clf_model.pt) loaded into the reference's own modules, eval mode, with
    hard=True Gumbel-softmax as eval_utils.py:76-77 calls it; the noise F.gumbel_softmax drew is replayed from the RNG
    state captured just before the call.  The checkpoints are data and are copied next to the vectors."""
    import importlib, shutil
    mdir = os.path.join(REF, "conditional_counteRGAN/house_sales_kc_usa")
    scratch = "/tmp/pcg_golden_house"
    os.makedirs(scratch, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        sys.path.insert(0, mdir)
        for name in list(sys.modules):
            if name in ("config", "trainer", "data_utils") or name == "models" or name.startswith("models."):
                sys.modules.pop(name)
        cfg = importlib.import_module("config").config
        gen_mod = importlib.import_module("models.generator")
        clf_mod = importlib.import_module("models.nn_classifier")
        G = gen_mod.ResidualGenerator(cfg["input_dim"], cfg["hidden_dim"], cfg["num_classes"], continuous_idx=cfg["continuous_idx"],
                                      categorical_info={k: {"n": v["n"], "raw_values": v["raw_values"]} for k, v in cfg["categorical_info"].items()},
                                      tau=cfg["gumbel_tau"])
        clf = clf_mod.NNClassifier(cfg["input_dim"], output_dim=cfg["num_classes"])
        G.load_state_dict(torch.load(os.path.join(mdir, "generator_model.pt"), map_location="cpu", weights_only=True))
        clf.load_state_dict(torch.load(os.path.join(mdir, "clf_model.pt"), map_location="cpu", weights_only=True))
        G.eval(); clf.eval()
        g = torch.Generator().manual_seed(77)
        x = torch.rand(bs, cfg["input_dim"], generator=g)
        t = torch.randint(0, cfg["num_classes"], (bs,), generator=g)
        mask = torch.ones(bs, cfg["input_dim"])
        mask[:, cfg["immutable_idx"]] = 0.0                                   # eval_utils.py:48-50
        onehot = torch.nn.functional.one_hot(t, cfg["num_classes"]).float()
        torch.manual_seed(5)
        state = torch.get_rng_state()
        with torch.no_grad():
            cont, cat_logits, cat_samples = G(x, onehot, mask=mask, temperature=cfg["gumbel_tau"], hard=True)
            clf_logits = clf(x)
        torch.set_rng_state(state)
        out = {"x": x.numpy(), "target_y": t.numpy(), "mask": mask.numpy(), "cont": cont.numpy(), "clf_logits": clf_logits.numpy()}
        for idx_str, head in G.fc_cat_logits.items():
            out[f"gumbel.{idx_str}"] = (-torch.empty(bs, head.out_features).exponential_().log()).numpy()
            out[f"logits.{idx_str}"] = cat_logits[int(idx_str)].numpy()
            out[f"samples.{idx_str}"] = cat_samples[int(idx_str)].numpy()
        np.savez_compressed(path_npz, **out)
        for src, dst in (("generator_model.pt", path_g), ("clf_model.pt", path_c)):
            shutil.copyfile(os.path.join(mdir, src), dst)
            os.chmod(dst, 0o644)
        print(f"wrote {path_npz}, {path_g}, {path_c}; |cont| max {cont.abs().max().item():.4f}")
    finally:
        os.chdir(cwd)


def make_wgan_small(path, width=16, batch=6, steps=3):
    """conditional_gan/mnist/mnist_wgan_conditional.py cannot be imported (torchvision, torch.cuda.get_device_name, dataset and
    training at import), so — as for DCGAN — its pure-torch pieces are lifted out of the syntax tree and executed unmodified,
    with every "cuda" string constant turned into "cpu":
      * Hyperparameter :21-31 + `hp = Hyperparameter()` :34, Generator :51-78, Critic :80-108
      * nets / optimizers / all_labels / grad_tensor                      :116-126
      * the body of the inner loop up to generator_optimizer.step()        :133-168
    at reduced width (critic_size = generator_size = critic_hidden_size = `width`).  The loop draws noise / alpha / labels
    from the global RNG; each step is seeded and the draws are replayed in the same order to record them."""
    path_src = os.path.join(REF, "conditional_gan/mnist/mnist_wgan_conditional.py")
    with open(path_src) as f:
        tree = ast.parse(f.read(), filename=path_src)

    class _Cpu(ast.NodeTransformer):
        def visit_Constant(self, node):
            return ast.copy_location(ast.Constant("cpu"), node) if node.value == "cuda" else node
    tree = ast.fix_missing_locations(_Cpu().visit(tree))
    hp_nodes, defs, setup, body = [], [], [], []
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "Hyperparameter" or (isinstance(node, ast.Assign) and node.lineno == 34):
            hp_nodes.append(node)
        elif isinstance(node, ast.ClassDef) and node.name in ("Generator", "Critic"):
            defs.append(node)
        elif isinstance(node, ast.Assign) and node.lineno in (116, 118, 119, 123, 126):
            setup.append(node)
        elif isinstance(node, ast.For) and node.lineno == 129:
            inner = [n for n in node.body if isinstance(n, ast.For)]
            assert len(inner) == 1 and inner[0].lineno == 132, "reference layout changed"
            body = [n for n in inner[0].body if 133 <= n.lineno <= 168]
    assert len(hp_nodes) == 2 and len(defs) == 2 and len(setup) == 5 and body and body[-1].end_lineno == 168
    from dataclasses import dataclass
    ns = {"torch": torch, "nn": torch.nn, "optim": torch.optim, "autograd": torch.autograd, "dataclass": dataclass}
    exec(compile(ast.Module(body=hp_nodes, type_ignores=[]), path_src, "exec"), ns)
    hp = ns["hp"]
    hp.critic_size = hp.generator_size = hp.critic_hidden_size = width
    hp.batchsize = batch
    exec(compile(ast.Module(body=defs, type_ignores=[]), path_src, "exec"), ns)
    torch.manual_seed(1)                                                   # :13
    exec(compile(ast.Module(body=setup, type_ignores=[]), path_src, "exec"), ns)
    critic, generator = ns["critic"], ns["generator"]
    step_code = compile(ast.Module(body=body, type_ignores=[]), path_src, "exec")
    out = {"meta.width": np.int64(width), "meta.batch": np.int64(batch), "meta.steps": np.int64(steps), "meta.n_critic": np.int64(hp.n_critic)}
    _sd("init.C", critic, out)
    _sd("init.G", generator, out)
    for k in range(steps):
        gen = torch.Generator().manual_seed(300 + k)
        real = torch.rand(batch, 1, 28, 28, generator=gen) * 2 - 1
        labels = torch.randint(0, hp.num_classes, (batch,), generator=gen)
        torch.manual_seed(2000 + k)                                        # replay of the draws, in the loop body's order
        noise = torch.randn((batch, hp.latent_size)); alpha = torch.rand((batch, 1))
        g_step = k % hp.n_critic == 0
        if g_step:
            fake_idx = torch.randint(hp.num_classes, size=[batch]); noise_g = torch.randn((batch, hp.latent_size))
        torch.manual_seed(2000 + k)
        ns["data"], ns["batch_idx"] = (real, labels), k
        exec(step_code, ns)
        assert torch.equal(ns["alpha"], alpha) and torch.equal(ns["noise"], noise_g if g_step else noise)
        out.update({f"step{k}.real": real.numpy(), f"step{k}.labels": labels.numpy(), f"step{k}.noise": noise.numpy(),
                    f"step{k}.alpha": alpha.numpy(), f"step{k}.critic_loss": np.float32(ns["critic_loss"].item()),
                    f"step{k}.gradient_penalty": np.float32(ns["gradient_penalty"].item()),
                    f"step{k}.loss_real": np.float32(ns["critic_loss_real"].item()),
                    f"step{k}.gradients": ns["gradients"].detach().numpy().copy()})
        if g_step:
            out.update({f"step{k}.fake_idx": fake_idx.numpy(), f"step{k}.noise_g": noise_g.numpy(),
                        f"step{k}.generator_loss": np.float32(ns["generator_loss"].item())})
        if k == 0:       # gradients right after the first critic+generator update (critic .grad = critic-step grads + the G step's)
            for n_, p_ in critic.named_parameters():
                out[f"step0.grad.C.{n_}"] = p_.grad.detach().numpy().copy()
            for n_, p_ in generator.named_parameters():
                out[f"step0.grad.G.{n_}"] = p_.grad.detach().numpy().copy()
    _sd("final.C", critic, out)
    _sd("final.G", generator, out)
    for n_, p_ in critic.named_parameters():
        out[f"final.grad.C.{n_}"] = p_.grad.detach().numpy().copy()      # last step had no G update: pure critic-step gradients
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


def _lift_functions(path, names, ns):
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(defs) == len(names), (path, [d.name for d in defs])
    exec(compile(ast.Module(body=defs, type_ignores=[]), path, "exec"), ns)
    return ns


def make_house_eval(path, exact_rows=512):
    """SURVEY.md section 8f item 2 — the counterfactual evaluation path of house_sales_kc_usa: `build_counterfactuals`
    (eval_utils.py:25-181) and `compute_metrics_per_target` (:185-289) are lifted from the syntax tree (the module imports
    seaborn, which is absent) and run with the reference's own modules, the checkpoints it ships and its own preprocessing of
    the dataset it ships (data_utils.load_and_preprocess: quartile labels, MinMax).  Stored: the scaled test split (data), the
    scaler range and category tables, the metrics of a full run (statistical anchor, next to the shipped
    results/countergan_metrics.csv) and an exact case — the first `exact_rows` rows as one batch per target class, with the
    hard Gumbel-softmax draws replayed from the RNG states captured at each generator call."""
    import contextlib, importlib, io
    import pandas as pd
    from typing import Optional, Tuple
    import torch.nn.functional as F
    from torch.utils.data import DataLoader, TensorDataset
    mdir = os.path.join(REF, "conditional_counteRGAN/house_sales_kc_usa")
    scratch = "/tmp/pcg_golden_house"
    os.makedirs(scratch, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        sys.path.insert(0, mdir)
        for name in list(sys.modules):
            if name in ("config", "trainer", "data_utils") or name == "models" or name.startswith("models."):
                sys.modules.pop(name)
        cfg = importlib.import_module("config").config
        gen_mod = importlib.import_module("models.generator")
        clf_mod = importlib.import_module("models.nn_classifier")
        data_utils = importlib.import_module("data_utils")
        with contextlib.redirect_stdout(io.StringIO()):
            _, X_test, _, y_test = data_utils.load_and_preprocess(os.path.join(mdir, "kc_house_data.csv"), cfg)
        cfg["cuda"] = "cpu"
        ns = {"torch": torch, "np": np, "pd": pd, "F": F, "TensorDataset": TensorDataset, "DataLoader": DataLoader, "Tuple": Tuple,
              "Optional": Optional}
        _lift_functions(os.path.join(mdir, "eval_utils.py"), ("build_counterfactuals", "compute_metrics_per_target"), ns)
        G = gen_mod.ResidualGenerator(cfg["input_dim"], cfg["hidden_dim"], cfg["num_classes"], continuous_idx=cfg["continuous_idx"],
                                      categorical_info={k: {"n": v["n"], "raw_values": v["raw_values"]} for k, v in cfg["categorical_info"].items()},
                                      tau=cfg["gumbel_tau"])
        clf = clf_mod.NNClassifier(cfg["input_dim"], output_dim=cfg["num_classes"])
        G.load_state_dict(torch.load(os.path.join(mdir, "generator_model.pt"), map_location="cpu", weights_only=True))
        clf.load_state_dict(torch.load(os.path.join(mdir, "clf_model.pt"), map_location="cpu", weights_only=True))
        G.eval(); clf.eval()
        out = {"X_test": X_test.astype(np.float32), "y_test": y_test.astype(np.int64),
               "scaler.data_min": np.asarray(cfg["scaler"].data_min_, np.float64), "scaler.data_max": np.asarray(cfg["scaler"].data_max_, np.float64),
               "meta.batch_size": np.int64(cfg["batch_size"]), "meta.exact_rows": np.int64(exact_rows)}
        for f, info in cfg["categorical_info"].items():
            out[f"raw_values.{f}"] = np.asarray(info["raw_values"], np.float64)
        shipped = pd.read_csv(os.path.join(mdir, "results/countergan_metrics.csv"))
        out["shipped.metrics"] = shipped[["class_flip", "prediction_gain", "avg_actionability"]].to_numpy(np.float64)
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            df, _, _ = ns["compute_metrics_per_target"](G, clf, X_test, y_test, cfg)
        out["full.metrics"] = df[["class_flip", "prediction_gain", "avg_actionability"]].to_numpy(np.float64)
        # exact case
        states = []
        h = G.register_forward_pre_hook(lambda m, a: states.append(torch.get_rng_state()))
        cfg2 = dict(cfg, batch_size=exact_rows)
        torch.manual_seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            df2, orig_vis, cf_vis = ns["compute_metrics_per_target"](G, clf, X_test[:exact_rows], y_test[:exact_rows], cfg2, max_vis=10 ** 9)
        h.remove()
        assert len(states) == cfg["num_classes"]
        out["exact.metrics"] = df2[["class_flip", "prediction_gain", "avg_actionability"]].to_numpy(np.float64)
        out["exact.x_cf"] = cf_vis.astype(np.float32)                       # concatenated over the target classes
        for t, st in enumerate(states):
            bs = int((y_test[:exact_rows] != t).sum())
            torch.set_rng_state(st)
            for idx_str, head in G.fc_cat_logits.items():
                out[f"exact.gumbel.{t}.{idx_str}"] = (-torch.empty(bs, head.out_features).exponential_().log()).numpy()
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB; full-run metrics\n{out['full.metrics']}\nshipped\n{out['shipped.metrics']}")
    finally:
        os.chdir(cwd)


def make_countergan_eval(path, batch=16):
    """SURVEY.md section 8f item 2, mnist: `evaluate_counterfactuals` (conditional_counteRGAN/mnist/eval_utils.py:46-79), lifted
    (the module imports seaborn), with the reference's ResidualGenerator + the generator checkpoint it ships and a seeded
    CNNClassifier (the trained best_classifier.pt is not in the repository)."""
    import importlib
    import torch.nn.functional as F
    mdir = os.path.join(REF, "conditional_counteRGAN/mnist")
    sys.path.insert(0, mdir)
    for name in list(sys.modules):
        if name in ("config", "trainer") or name == "models" or name.startswith("models."):
            sys.modules.pop(name)
    gen_mod = importlib.import_module("models.generator")
    clf_mod = importlib.import_module("models.classifier")
    ns = {"torch": torch, "F": F}
    _lift_functions(os.path.join(mdir, "eval_utils.py"), ("evaluate_counterfactuals",), ns)
    G = gen_mod.ResidualGenerator()
    G.load_state_dict(torch.load(os.path.join(mdir, "results/generator.pt"), map_location="cpu", weights_only=True))
    torch.manual_seed(3)
    C = clf_mod.CNNClassifier()
    g = torch.Generator().manual_seed(21)
    x = torch.rand(batch, 1, 28, 28, generator=g) * 2 - 1
    y_true = torch.randint(0, 10, (batch,), generator=g)
    y_target = torch.randint(0, 10, (batch,), generator=g)
    metrics, (x_vis, x_cf_vis) = ns["evaluate_counterfactuals"](G, C, x, y_true, y_target, "cpu")
    out = {"x": x.numpy(), "y_true": y_true.numpy(), "y_target": y_target.numpy(), "x_cf_vis": x_cf_vis.numpy(),
           "metrics": np.array([metrics["class_flip_rate"], metrics["prediction_gain"], metrics["actionability"]], np.float64)}
    for k, v in C.state_dict().items():          # seeded init (torch.manual_seed(3)): digests only, the test re-creates it
        out[f"C.{k}"] = tensor_digest(v)
    np.savez_compressed(path, **out)
    print(f"wrote {path}: metrics {out['metrics']}")


def _dropout_recorder(module, store):
    """Forward pre-hook factory: remember the RNG state right before a Dropout module draws its noise (training calls only)."""
    def hook(mod, args):
        if mod.training:
            store.append((mod, tuple(args[0].shape), torch.get_rng_state()))
    return module.register_forward_pre_hook(hook)


def _replay_dropout(store):
    """[torch] nn.Dropout: noise = empty_like(x).bernoulli_(1-p); nn.Dropout2d: noise = empty([B,C,1,1]).bernoulli_(1-p)."""
    keep = torch.get_rng_state()
    masks = []
    for mod, shape, state in store:
        torch.set_rng_state(state)
        nshape = (shape[0], shape[1], 1, 1) if isinstance(mod, torch.nn.Dropout2d) else shape
        masks.append(torch.empty(nshape).bernoulli_(1 - mod.p).reshape(shape[0], -1).numpy().copy())
    torch.set_rng_state(keep)
    return masks


def make_house_preprocess(path_csv, path_npz, nrows=3000):
    """SURVEY.md section 8f item 4 — house_sales_kc_usa/data_utils.py:load_and_preprocess run UNMODIFIED (imported) on the first
    `nrows` data rows of the dataset the reference ships (kc_house_data.csv; the subset is committed as a data fixture, the
    full file does not travel).  Stored: both scaled splits, both label vectors, the quartile bin edges and the scaler range."""
    import contextlib, importlib, io
    mdir = os.path.join(REF, "conditional_counteRGAN/house_sales_kc_usa")
    with open(os.path.join(mdir, "kc_house_data.csv")) as f:
        lines = f.readlines()
    with open(path_csv, "w") as f:
        f.writelines(lines[: nrows + 1])
    scratch = "/tmp/pcg_golden_house"
    os.makedirs(scratch, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        sys.path.insert(0, mdir)
        for name in list(sys.modules):
            if name in ("config", "data_utils"):
                sys.modules.pop(name)
        data_utils = importlib.import_module("data_utils")
        cfg = {}
        with contextlib.redirect_stdout(io.StringIO()):
            Xtr, Xte, ytr, yte = data_utils.load_and_preprocess(path_csv, cfg)
        np.savez_compressed(path_npz, X_train=np.asarray(Xtr, np.float64), X_test=np.asarray(Xte, np.float64),
                            y_train=np.asarray(ytr, np.int64), y_test=np.asarray(yte, np.int64), bins=np.asarray(cfg["bins"], np.float64),
                            data_min=np.asarray(cfg["scaler"].data_min_, np.float64), data_max=np.asarray(cfg["scaler"].data_max_, np.float64))
        print(f"wrote {path_npz}: train {Xtr.shape}, test {Xte.shape}, bins {cfg['bins']}")
    finally:
        os.chdir(cwd)
        sys.path.remove(mdir)


def make_classifier_pretrain(path_mnist, path_house):
    """SURVEY.md section 8f item 3 — the classifier pre-training loops, run by the reference's own functions:
    conditional_counteRGAN/mnist/trainer.py:train_classifier (:8-39) on two seeded batches + one validation batch, and
    house_sales_kc_usa/trainer.py:train_classifier (:18-180) for one epoch of three batches (its NNClassifier is created
    inside the function: the class it sees is wrapped to attach the hooks).  Dropout draws are replayed from captured RNG
    states; the batches the shuffling DataLoader produced are captured at the model input."""
    import contextlib, importlib, io, types
    # ---- mnist
    mdir = os.path.join(REF, "conditional_counteRGAN/mnist")
    sys.path.insert(0, mdir)
    for name in list(sys.modules):
        if name in ("config", "trainer", "data_utils") or name == "models" or name.startswith("models."):
            sys.modules.pop(name)
    clf_mod = importlib.import_module("models.classifier")
    trainer = importlib.import_module("trainer")
    torch.manual_seed(5)
    C = clf_mod.CNNClassifier()
    out = {}
    for k, v in C.state_dict().items():
        out[f"init.{k}"] = tensor_digest(v)
    g = torch.Generator().manual_seed(31)
    batches = [(torch.rand(8, 1, 28, 28, generator=g) * 2 - 1, torch.randint(0, 10, (8,), generator=g)) for _ in range(3)]
    store = []
    hooks = [_dropout_recorder(m, store) for m in C.modules() if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d))]
    cfg = types.SimpleNamespace(cls_lr=1e-3, num_epochs_clf=1, classifier_path="/tmp/pcg_golden_clf.pt")
    buf = io.StringIO()
    torch.manual_seed(77)
    with contextlib.redirect_stdout(buf):
        trainer.train_classifier(C, batches[:2], batches[2:], cfg, "cpu")
    for h in hooks:
        h.remove()
    masks = _replay_dropout(store)
    assert len(masks) == 4
    for i, (x, y) in enumerate(batches):
        out[f"x{i}"], out[f"y{i}"] = x.numpy(), y.numpy()
    for i, m in enumerate(masks):
        out[f"mask{i}"] = m                                   # step0: Dropout2d [8,128], Dropout [8,256]; step1: same
    out["log"] = np.array(buf.getvalue())
    for k, v in C.state_dict().items():
        out[f"final.{k}"] = tensor_digest(v)
    out["final.fc.4.weight.full"] = C.state_dict()["fc.4.weight"].numpy().copy()
    out["final.conv.0.weight.full"] = C.state_dict()["conv.0.weight"].numpy().copy()
    np.savez_compressed(path_mnist, **out)
    print(f"wrote {path_mnist}; {buf.getvalue().strip().splitlines()[0]}")
    # ---- house
    hdir = os.path.join(REF, "conditional_counteRGAN/house_sales_kc_usa")
    scratch = "/tmp/pcg_golden_house"
    os.makedirs(scratch, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        sys.path.remove(mdir)
        sys.path.insert(0, hdir)
        for name in list(sys.modules):
            if name in ("config", "trainer", "data_utils") or name == "models" or name.startswith("models."):
                sys.modules.pop(name)
        cfg = importlib.import_module("config").config
        trainer = importlib.import_module("trainer")
        base_cls = trainer.NNClassifier
        store, inputs, made = [], [], []

        class Hooked(base_cls):
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                made.append({kk: vv.clone() for kk, vv in self.state_dict().items()})
                for m in self.modules():
                    if isinstance(m, torch.nn.Dropout):
                        _dropout_recorder(m, store)
                self.register_forward_pre_hook(lambda mod, args: inputs.append((mod.training, args[0].clone())))
        trainer.NNClassifier = Hooked
        # the reference passes verbose=True to ReduceLROnPlateau (:59), a logging-only argument that torch >= 2.7 no longer
        # accepts (the reference pins torch 2.4.1): drop it for the duration of the call
        _RLP = torch.optim.lr_scheduler.ReduceLROnPlateau

        class _RLPCompat(_RLP):
            def __init__(self, *a, verbose=None, **k):
                super().__init__(*a, **k)
        torch.optim.lr_scheduler.ReduceLROnPlateau = _RLPCompat
        rs = np.random.RandomState(3)
        X = rs.random_sample((160, cfg["input_dim"])).astype(np.float64)
        y = np.arange(160) % 4
        rs.shuffle(y)
        y[:10] = 0                                                             # a little imbalance: non-trivial class weights
        cfg.update({"cuda": "cpu", "clf_epochs": 1, "batch_size": 48, "out_dir": scratch, "clf_model_path": os.path.join(scratch, "clf.pt")})
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            model = trainer.train_classifier(X, X[:8], y, y[:8], None, cfg)
        trainer.NNClassifier = base_cls
        torch.optim.lr_scheduler.ReduceLROnPlateau = _RLP
        masks = _replay_dropout(store)
        train_inputs = [t for tr, t in inputs if tr]
        val_inputs = [t for tr, t in inputs if not tr]
        assert len(train_inputs) == 3 and len(masks) == 9
        out = {"X": X.astype(np.float32), "y": y.astype(np.int64), "log": np.array(buf.getvalue()), "meta.steps": np.int64(3)}
        for k, v in made[0].items():
            out[f"init.{k}"] = v.numpy().copy()
        Xf = torch.tensor(X, dtype=torch.float32)
        for i, t in enumerate(train_inputs):
            rows = np.array([int(torch.nonzero((Xf == r).all(1))[0]) for r in t])
            out[f"step{i}.rows"] = rows
            for j in range(3):
                out[f"step{i}.mask{j}"] = masks[3 * i + j]
        vrows = np.concatenate([np.array([int(torch.nonzero((Xf == r).all(1))[0]) for r in t]) for t in val_inputs])
        out["val.rows"] = vrows
        for k, v in model.state_dict().items():
            out[f"final.{k}"] = v.numpy().copy()
        np.savez_compressed(path_house, **out)
        print(f"wrote {path_house}; {buf.getvalue().strip().splitlines()[0]}")
    finally:
        os.chdir(cwd)


def make_mnist_resize(path, n=24):
    """SURVEY.md section 8f item 4 — the DCGAN input transform (mnist_dcgan.py:42-46): transforms.Resize(64) on the PIL image
    ([torchvision] F.resize of a PIL image = Image.resize(size, BILINEAR)), ToTensor ([torchvision] uint8 -> float32 .div(255)),
    Normalize((0.5,), (0.5,)) (sub_(mean).div_(std)).  torchvision itself is absent here; Pillow — the library that does the
    arithmetic — is present, and the two tensor steps are restated with the torch calls torchvision makes."""
    from PIL import Image
    rs = np.random.RandomState(12)
    imgs = (rs.rand(n, 28, 28) * 255).astype(np.uint8)
    yy, xx = np.mgrid[0:28, 0:28]
    for i in range(n // 2):                       # half of them digit-like: smooth blobs with saturated cores and black background
        cy, cx, r = rs.uniform(8, 20), rs.uniform(8, 20), rs.uniform(3, 8)
        imgs[i] = np.clip(320 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r)) - 30, 0, 255).astype(np.uint8)
    resized = np.stack([np.asarray(Image.fromarray(im, mode="L").resize((64, 64), Image.BILINEAR)) for im in imgs])
    t = torch.from_numpy(resized.copy()).to(torch.float32).div(255)
    t = t.sub_(0.5).div_(0.5)
    np.savez_compressed(path, images=imgs, resized_u8=resized, out=t.numpy().reshape(n, 1, 64, 64))
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.0f} KB")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit(f"{REF} not found — golden vectors can only be regenerated where the reference is mounted")
    only = sys.argv[1:]
    if only == ["countergan_loop"]:
        make_countergan_loop(os.path.join(HERE, "countergan_loop_b4.npz"))
        sys.exit(0)
    if only == ["house_loop"]:
        make_house_loop(os.path.join(HERE, "house_loop.npz"))
        sys.exit(0)
    if only == ["dcgan_loop"]:          # regenerate one fixture without touching the others
        make_dcgan_loop(os.path.join(HERE, "dcgan_loop_small.npz"))
        sys.exit(0)
    make_dcgan_small(os.path.join(HERE, "dcgan_ref_small.npz"))
    make_dcgan_loop(os.path.join(HERE, "dcgan_loop_small.npz"))
    make_countergan(os.path.join(HERE, "countergan_ref_b4.npz"))
    make_countergan_loop(os.path.join(HERE, "countergan_loop_b4.npz"))
    make_moons(os.path.join(HERE, "moons_ref.npz"))
    make_countergan_trained(os.path.join(HERE, "countergan_trained_eval.npz"), os.path.join(HERE, "countergan_generator_trained.pt"))
    make_house(os.path.join(HERE, "house_ref_b64.npz"))
    make_house_trained(os.path.join(HERE, "house_trained_eval.npz"), os.path.join(HERE, "house_generator_trained.pt"),
                       os.path.join(HERE, "house_classifier_trained.pt"))
    make_wgan_small(os.path.join(HERE, "wgan_ref_small.npz"))
    make_house_eval(os.path.join(HERE, "house_eval.npz"))
    make_house_loop(os.path.join(HERE, "house_loop.npz"))
    make_countergan_eval(os.path.join(HERE, "countergan_eval.npz"))
    make_classifier_pretrain(os.path.join(HERE, "classifier_pretrain_mnist.npz"), os.path.join(HERE, "classifier_pretrain_house.npz"))
    make_mnist_resize(os.path.join(HERE, "mnist_resize.npz"))
    make_house_preprocess(os.path.join(HERE, "kc_house_head3000.csv"), os.path.join(HERE, "house_preprocess.npz"))

"""GPU parity, simple_gan/moons (BASELINE config 1): one epoch of the reference's own train_gan (golden) replayed
through the HIP kernels — Linear layers as 1x1 convolutions (MFMA and thin paths), ReLU, Sigmoid, BCE, Adam."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_moons_epoch_vs_reference_golden(golden_dir):
    import pcgan_amd  # noqa: F401
    from pcgan_amd import moons as M
    gold = dict(np.load(os.path.join(golden_dir, "moons_ref.npz")))
    G, D = M.build_generator(32, 128), M.build_discriminator(128)
    assert [f"init.G.{k}" for k in G.state_dict()] == [k for k in gold if k.startswith("init.G.")]   # main.0.weight ...
    G.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.G.")})
    D.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.D.")})
    G.to(DEV); D.to(DEV)
    optG, optD = M.make_optimizers(G, D)
    X, z = torch.from_numpy(gold["X_shuffled"]).to(DEV), torch.from_numpy(gold["z"]).to(DEV)
    totD = totG = 0.0
    for i, real in enumerate(X.split(50)):
        lD, lG = M.train_step(G, D, optG, optD, real.contiguous(), z[2 * i].contiguous(), z[2 * i + 1].contiguous())
        totD += lD.item(); totG += lG.item()
    np.testing.assert_allclose(totD, float(gold["loss_D_total"]), rtol=2e-5)
    np.testing.assert_allclose(totG, float(gold["loss_G_total"]), rtol=2e-5)
    for tag, net in (("G", G), ("D", D)):
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), gold[f"final.{tag}.{k}"], rtol=1e-4, atol=2e-5, err_msg=f"{tag}.{k}")


def test_moons_batch256_vs_oracle():
    """BASELINE config 1 names batch 256 (the golden above is the reference's own default, 50): eight batches of 256 rows of
    make_moons-shaped data through the HIP step and through the oracle restatement (pinned to the golden by
    tests/test_oracle_golden.py), same weights, same noise.  Tolerances: losses 2e-5, weights 1e-4 after 8 Adam steps."""
    import pcgan_amd  # noqa: F401
    from oracle import moons_ref as R
    from pcgan_amd import moons as M
    torch.manual_seed(11)
    rG, rD = R.build_generator(32, 128), R.build_discriminator(128)
    G, D = M.build_generator(32, 128), M.build_discriminator(128)
    G.load_state_dict(rG.state_dict()); D.load_state_dict(rD.state_dict())
    G.to(DEV); D.to(DEV)
    roptG, roptD = R.make_optimizers(rG, rD)
    optG, optD = M.make_optimizers(G, D)
    g = torch.Generator().manual_seed(12)
    t = torch.rand(2048, generator=g) * 3.14159265
    upper = torch.rand(2048, generator=g) < 0.5
    X = torch.where(upper[:, None], torch.stack([t.cos(), t.sin()], 1), torch.stack([1 - t.cos(), 0.5 - t.sin()], 1))
    X = X + 0.05 * torch.randn(2048, 2, generator=g)
    for i, real in enumerate(X.split(256)):
        zd, zg = torch.randn(256, 32, generator=g), torch.randn(256, 32, generator=g)
        wD, wG = R.moons_step(rG, rD, roptG, roptD, real, zd, zg)
        lD, lG = M.train_step(G, D, optG, optD, real.to(DEV).contiguous(), zd.to(DEV), zg.to(DEV))
        np.testing.assert_allclose(lD.item(), wD, rtol=2e-5, err_msg=f"loss_D, batch {i}")
        np.testing.assert_allclose(lG.item(), wG, rtol=2e-5, err_msg=f"loss_G, batch {i}")
    for ours, ref, tag in ((G, rG, "G"), (D, rD, "D")):
        for (k, v), (_, w) in zip(ours.state_dict().items(), ref.state_dict().items()):
            np.testing.assert_allclose(v.cpu().numpy(), w.detach().numpy(), rtol=1e-4, atol=2e-5, err_msg=f"{tag}.{k}")

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

"""GPU: the DCGAN input transform on the device (pcgan_amd.data.ResizeNormalize -> pcg_resize8_normalize) against Pillow +
the torchvision tensor arithmetic (tests/golden/mnist_resize.npz): bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_resize_normalize_is_bit_exact_against_pillow(golden_dir):
    import pcgan_amd  # noqa: F401
    from pcgan_amd import data
    gold = np.load(os.path.join(golden_dir, "mnist_resize.npz"))
    tf = data.ResizeNormalize((28, 28), (64, 64), 0.5, 0.5, device="cuda:0")
    out = tf(torch.from_numpy(gold["images"]).to("cuda:0"))
    assert out.shape == (gold["images"].shape[0], 1, 64, 64)
    assert np.array_equal(out.cpu().numpy(), gold["out"])
    # a full training batch keeps the per-image result (one block per image)
    big = torch.from_numpy(np.tile(gold["images"], (22, 1, 1))[:512]).to("cuda:0")
    out2 = tf(big)
    assert torch.equal(out2[:24], out) and torch.equal(out2[480:504], out)


def test_checkpoints_written_from_the_hip_modules_load_into_the_reference_modules(tmp_path):
    """torch.save(model.state_dict()) as the reference's loops do (mnist/trainer.py:159, house trainer.py:365): after training
    steps on the GPU (parameters are views of a flat, channels_last buffer) the file loads into the plain PyTorch modules of
    the oracle with identical values, keys and shapes."""
    import pcgan_amd  # noqa: F401
    from pcgan_amd import countergan as K, wgan as W
    from oracle import countergan_ref as CR, wgan_ref as WR
    dev = torch.device("cuda:0")
    G = K.ResidualGenerator().to(dev)
    x = torch.rand(4, 1, 28, 28, device=dev) * 2 - 1
    t = torch.randint(0, 10, (4,), device=dev)
    m = torch.ones(4, 1, 28, 28, device=dev)
    raw, masked = G(x, t, m)
    (raw.sum() + masked.sum()).backward()                    # forces the flat layout + one update of BatchNorm running statistics
    path = tmp_path / "g.pt"
    torch.save(G.state_dict(), path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    refG, _, _ = CR.build(seed=0)
    refG.load_state_dict(sd)
    for k, v in G.state_dict().items():
        assert torch.equal(refG.state_dict()[k], v.cpu()), k
    hp = W.Hyperparameter(critic_size=16, generator_size=16, critic_hidden_size=16)
    critic, _ = W.build(dev, hp)
    critic(torch.rand(2, 1, 28, 28, device=dev), torch.eye(10, device=dev)[:2]).sum().backward()
    torch.save(critic.state_dict(), tmp_path / "c.pt")
    refC = WR.Critic(WR.Hyperparameter(critic_size=16, generator_size=16, critic_hidden_size=16))
    refC.load_state_dict(torch.load(tmp_path / "c.pt", map_location="cpu", weights_only=True))
    for k, v in critic.state_dict().items():
        assert torch.equal(refC.state_dict()[k], v.cpu()), k

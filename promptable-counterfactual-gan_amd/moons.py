"""Host-side mirror of `simple_gan/moons/make_moons_gan.py` (BASELINE config 1) on the HIP kernels: the two MLPs are
nn.Sequential stacks of nn.Linear / ReLU / Sigmoid run by SequentialConvNet (a Linear is a 1x1 convolution on a
[B,1,1,F] activation, its [out,in] weight already OHWI), the log losses (:70,:83) are the BCE kernel.

    build_generator :33-38, build_discriminator :40-46, train_gan batch body :62-87 -> train_step
"""
import torch.nn as nn

from .dcgan import _labels
from .nn import BCELoss, HipSequential
from .optim import Adam

config = {"n_samples": 2000, "z_dim": 32, "hidden_dim": 128, "batch_size": 50, "lr": 1e-3, "epochs": 500}   # :9-17


def build_generator(z_dim, hidden_dim):
    return HipSequential(nn.Linear(z_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, 2))


def build_discriminator(hidden_dim):
    return HipSequential(nn.Linear(2, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, 1), nn.Sigmoid())


def make_optimizers(generator, discriminator, cfg=config):
    return Adam(generator.parameters(), lr=cfg["lr"]), Adam(discriminator.parameters(), lr=cfg["lr"])          # :50-51


_bce = BCELoss()


def train_step(generator, discriminator, optimizer_G, optimizer_D, real_batch, z_d, z_g):
    """One batch of train_gan (:62-87); the two noise draws (:64,:79) are passed in.
    -mean(log D_real + log(1 - D_fake)) = BCE(D_real, 1) + BCE(D_fake, 0) and -mean(log D_fake) = BCE(D_fake, 1); the BCE
    kernel clamps log at -100 like torch's BCELoss, the reference's bare torch.log gives inf at D in {0,1}."""
    n = real_batch.shape[0]
    dev = real_batch.device
    ones, zeros = _labels.get(n, 1.0, dev), _labels.get(n, 0.0, dev)
    fake_batch = generator(z_d)                                                   # :65
    D_real = discriminator(real_batch)                                            # :67
    D_fake = discriminator(fake_batch)                                            # :68
    loss_D = _bce(D_real.view(-1), ones) + _bce(D_fake.view(-1), zeros)           # :70
    optimizer_D.zero_grad()                                                       # :72
    loss_D.backward()                                                             # :73
    optimizer_D.step()                                                            # :74
    fake_batch = generator(z_g)                                                   # :80
    D_fake = discriminator(fake_batch)                                            # :81
    loss_G = _bce(D_fake.view(-1), ones)                                          # :83
    optimizer_G.zero_grad()                                                       # :85
    loss_G.backward()                                                             # :86
    optimizer_G.step()                                                            # :87
    return loss_D, loss_G

"""ORACLE — test infrastructure only.  CPU restatement of `simple_gan/moons/make_moons_gan.py` (BASELINE config 1, the
CPU "plumbing" case): build_generator :33-38, build_discriminator :40-46, the body of train_gan's batch loop :62-87.
The script itself trains and plots at import (:128-137); tests/golden/make_golden.py lifts the three functions out of
its syntax tree and runs them to pin this restatement."""
import torch
import torch.nn as nn

CONFIG = {"z_dim": 32, "hidden_dim": 128, "batch_size": 50, "lr": 1e-3}   # make_moons_gan.py:9-17


def build_generator(z_dim, hidden_dim):
    return nn.Sequential(nn.Linear(z_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, 2))                      # :33-38


def build_discriminator(hidden_dim):
    return nn.Sequential(nn.Linear(2, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, 1), nn.Sigmoid())           # :40-46


def make_optimizers(generator, discriminator, config=CONFIG):
    return (torch.optim.Adam(generator.parameters(), lr=config["lr"]),                                            # :50
            torch.optim.Adam(discriminator.parameters(), lr=config["lr"]))                                        # :51


def moons_step(generator, discriminator, optimizer_G, optimizer_D, real_batch, z_d, z_g):
    """One batch of train_gan (:62-87) with the two noise draws (:64, :79) supplied."""
    fake_batch = generator(z_d)                                             # :65
    D_real = discriminator(real_batch)                                      # :67
    D_fake = discriminator(fake_batch)                                      # :68
    loss_D = -torch.mean(torch.log(D_real) + torch.log(1 - D_fake))         # :70
    optimizer_D.zero_grad()                                                 # :72
    loss_D.backward()                                                       # :73
    optimizer_D.step()                                                      # :74
    fake_batch = generator(z_g)                                             # :80
    D_fake = discriminator(fake_batch)                                      # :81
    loss_G = -torch.mean(torch.log(D_fake))                                 # :83
    optimizer_G.zero_grad()                                                 # :85
    loss_G.backward()                                                       # :86
    optimizer_G.step()                                                      # :87
    return loss_D.item(), loss_G.item()

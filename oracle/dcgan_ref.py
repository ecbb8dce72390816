"""ORACLE — test infrastructure only (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product path).

CPU restatement of the reference's DCGAN hot path, `dconv_gan/mnist/mnist_dcgan.py` (the script itself cannot be
imported: it needs torchvision, opens /mnt/data and trains at import time — SURVEY.md §8c).  Every FLOP of that file
is a torch.nn / torch.optim call, so the restatement calls the same PyTorch operators on the CPU (fp32); the op-level
arithmetic behind them is restated independently in `oracle/ops_np.py`.

Pinning: `tests/golden/make_golden.py` extracts the reference's own `weights_init` / `Generator` / `Discriminator`
definitions and the body of its training loop (lines 147-175) from the reference file with `ast` and runs THEM on
seeded synthetic batches; `tests/test_oracle_golden.py` checks this restatement against those vectors.
"""
import torch
import torch.nn as nn
import torch.optim as optim

# mnist_dcgan.py:15-30 (only the keys the nets and the step read)
DEFAULT_CONFIG = {
    "image_channel": 1,
    "z_dim": 100,
    "g_hidden": 64,
    "d_hidden": 64,
    "real_label": 1.0,
    "fake_label": 0.0,
    "lr": 2e-4,
}


def weights_init(m):
    """mnist_dcgan.py:63-69 — N(0, 0.02) on every *Conv* weight, BN weight ~ N(1, 0.02), BN bias = 0."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm") != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class Generator(nn.Module):
    """mnist_dcgan.py:72-93 — z [B, z_dim, 1, 1] -> image [B, C, 64, 64]."""

    def __init__(self, config=None):
        super().__init__()
        c = dict(DEFAULT_CONFIG, **(config or {}))
        g, z, ch = c["g_hidden"], c["z_dim"], c["image_channel"]
        self.main = nn.Sequential(
            nn.ConvTranspose2d(z, g * 8, 4, 1, 0, bias=False),      # :76
            nn.BatchNorm2d(g * 8), nn.ReLU(True),                    # :77-78
            nn.ConvTranspose2d(g * 8, g * 4, 4, 2, 1, bias=False),  # :79
            nn.BatchNorm2d(g * 4), nn.ReLU(True),
            nn.ConvTranspose2d(g * 4, g * 2, 4, 2, 1, bias=False),  # :82
            nn.BatchNorm2d(g * 2), nn.ReLU(True),
            nn.ConvTranspose2d(g * 2, g, 4, 2, 1, bias=False),      # :85
            nn.BatchNorm2d(g), nn.ReLU(True),
            nn.ConvTranspose2d(g, ch, 4, 2, 1, bias=False),         # :88
            nn.Tanh(),                                               # :89
        )

    def forward(self, input):
        return self.main(input)


class Discriminator(nn.Module):
    """mnist_dcgan.py:96-116 — image [B, C, 64, 64] -> probability [B]."""

    def __init__(self, config=None):
        super().__init__()
        c = dict(DEFAULT_CONFIG, **(config or {}))
        d, ch = c["d_hidden"], c["image_channel"]
        self.main = nn.Sequential(
            nn.Conv2d(ch, d, 4, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),                           # :100-101
            nn.Conv2d(d, d * 2, 4, 2, 1, bias=False), nn.BatchNorm2d(d * 2), nn.LeakyReLU(0.2, inplace=True),  # :102-104
            nn.Conv2d(d * 2, d * 4, 4, 2, 1, bias=False), nn.BatchNorm2d(d * 4), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d * 4, d * 8, 4, 2, 1, bias=False), nn.BatchNorm2d(d * 8), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d * 8, 1, 4, 1, 0, bias=False), nn.Sigmoid(),                                           # :111-112
        )

    def forward(self, input):
        return self.main(input).view(-1, 1).squeeze(1)  # :116


def make_optimizers(netG, netD, config=None):
    """mnist_dcgan.py:125-127."""
    c = dict(DEFAULT_CONFIG, **(config or {}))
    criterion = nn.BCELoss()
    optimizerD = optim.Adam(netD.parameters(), lr=c["lr"], betas=(0.5, 0.999))
    optimizerG = optim.Adam(netG.parameters(), lr=c["lr"], betas=(0.5, 0.999))
    return criterion, optimizerD, optimizerG


def dcgan_step(netG, netD, criterion, optimizerD, optimizerG, real, noise, config=None):
    """One iteration of the reference loop body, mnist_dcgan.py:147-175, with the batch (`real`, :148) and the noise
    (:156, drawn from the device RNG in the reference) supplied by the caller so that runs are comparable.
    Returns the scalars the reference logs (:154,162-163,172-174) as Python floats."""
    c = dict(DEFAULT_CONFIG, **(config or {}))
    # (1) Update D network
    netD.zero_grad()                                                                   # :147
    b_size = real.size(0)
    # :150 (dtype=torch.float there; real.dtype here so the same restatement can be evaluated in float64 as "truth")
    label = torch.full((b_size,), c["real_label"], dtype=real.dtype, device=real.device)
    output = netD(real)                                                                # :151
    errD_real = criterion(output, label)                                               # :152
    errD_real.backward()                                                               # :153
    D_x = output.mean().item()                                                         # :154
    fake = netG(noise)                                                                 # :157
    label.fill_(c["fake_label"])                                                       # :158
    output = netD(fake.detach())                                                       # :159
    errD_fake = criterion(output, label)                                               # :160
    errD_fake.backward()                                                               # :161
    D_G_z1 = output.mean().item()                                                      # :162
    errD = errD_real + errD_fake                                                       # :163
    optimizerD.step()                                                                  # :164
    # (2) Update G network
    netG.zero_grad()                                                                   # :169
    label.fill_(c["real_label"])                                                       # :170
    output = netD(fake)                                                                # :171
    errG = criterion(output, label)                                                    # :172
    errG.backward()                                                                    # :173
    D_G_z2 = output.mean().item()                                                      # :174
    optimizerG.step()                                                                  # :175
    return {
        "errD_real": errD_real.item(), "errD_fake": errD_fake.item(), "errD": errD.item(), "errG": errG.item(),
        "D_x": D_x, "D_G_z1": D_G_z1, "D_G_z2": D_G_z2,
    }


def dcgan_train(dataloader, config, netG, netD, log=None):
    """The outer loop, mnist_dcgan.py:125-198, without the plots: optimizers (:125-127), the fixed viz noise (:130, drawn from
    the global generator BEFORE the first iteration), per iteration the loop body (noise drawn from the global generator, :156),
    the running sums of errG / errD (:184-185), and — every 500 iterations and at the very end (:187) — a forward of netG on
    viz_noise under no_grad with the net still in training mode (:188-189), which updates its BatchNorm running statistics.
    Returns (epoch_G_losses, epoch_D_losses, img_list of raw generated batches, iters)."""
    c = dict(DEFAULT_CONFIG, **(config or {}))
    criterion, optimizerD, optimizerG = make_optimizers(netG, netD, c)
    viz_noise = torch.randn(c["batch_size"], c["z_dim"], 1, 1)                          # :130
    img_list, epoch_G_losses, epoch_D_losses = [], [], []
    iters = 0
    for epoch in range(c["epochs"]):                                                   # :140
        running_G_loss = running_D_loss = 0.0
        for i, data in enumerate(dataloader):                                          # :143
            real = data[0]
            noise = torch.randn(real.size(0), c["z_dim"], 1, 1)                        # :156
            out = dcgan_step(netG, netD, criterion, optimizerD, optimizerG, real, noise, c)
            if log is not None and i % 200 == 0:                                       # :178-181
                log(f"[{epoch}/{c['epochs']}][{i}/{len(dataloader)}] Loss_D: {out['errD']:.4f} Loss_G: {out['errG']:.4f}")
            running_G_loss += out["errG"]                                              # :184
            running_D_loss += out["errD"]                                              # :185
            if iters % 500 == 0 or (epoch == c["epochs"] - 1 and i == len(dataloader) - 1):   # :187
                with torch.no_grad():
                    img_list.append(netG(viz_noise).detach().cpu())                   # :188-189
            iters += 1
        epoch_G_losses.append(running_G_loss / len(dataloader))                        # :195-198
        epoch_D_losses.append(running_D_loss / len(dataloader))
    return epoch_G_losses, epoch_D_losses, img_list, iters


def synthetic_batch(batch, seed, config=None, dtype=torch.float32):
    """Seeded MNIST-shaped synthetic batch: real ~ U[-1,1) [B,C,64,64], z ~ N(0,1) [B,z,1,1]."""
    c = dict(DEFAULT_CONFIG, **(config or {}))
    g = torch.Generator().manual_seed(seed)
    real = torch.rand(batch, c["image_channel"], 64, 64, generator=g, dtype=dtype) * 2 - 1
    noise = torch.randn(batch, c["z_dim"], 1, 1, generator=g, dtype=dtype)
    return real, noise


def build(config=None, seed=1):
    """Nets initialised as the reference does (seed :33, init :119-122), on the CPU."""
    torch.manual_seed(seed)
    netG = Generator(config)
    netG.apply(weights_init)
    netD = Discriminator(config)
    netD.apply(weights_init)
    return netG, netD

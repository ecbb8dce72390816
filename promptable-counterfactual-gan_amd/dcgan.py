"""Host-side mirror of `dconv_gan/mnist/mnist_dcgan.py` on the HIP kernels: same `config` keys, same class names,
same `nn.Sequential` layer list (so `state_dict()` keys are `main.0.weight` ... `main.12.weight` exactly as in the
reference), same loss / optimizer construction and the same D-step / G-step sequence.

    reference (mnist_dcgan.py)                      here
    --------------------------------------------    ------------------------------------------------
    weights_init               :63-69               weights_init (unchanged semantics)
    Generator / Discriminator  :72-116              Generator / Discriminator (SequentialConvNet)
    criterion, optimizerD/G    :125-127             make_optimizers -> pcgan_amd BCELoss, Adam
    loop body                  :147-175             train_step (no host sync inside; scalars stay on device)
    outer loop                 :129-198             train (epochs x batches, logging, the train-mode viz forward :187-191)
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import nn as _nn
from .nn import BCELoss, SequentialConvNet
from .optim import Adam

# mnist_dcgan.py:15-30
config = {
    "cuda": True,
    "batch_size": 128,
    "image_channel": 1,
    "z_dim": 100,
    "g_hidden": 64,
    "d_hidden": 64,
    "x_dim": 64,
    "epochs": 20,
    "real_label": 1.0,
    "fake_label": 0.0,
    "lr": 2e-4,
    "seed": 1,
}


def _cfg(c):
    return dict(config, **(c or {}))


def weights_init(m):
    """mnist_dcgan.py:63-69 (class-name substring match: 'Conv' also hits ConvTranspose2d)."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1 and hasattr(m, "weight") and isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm") != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class Generator(SequentialConvNet):
    """mnist_dcgan.py:72-93."""

    def __init__(self, cfg=None):
        c = _cfg(cfg)
        g, z, ch = c["g_hidden"], c["z_dim"], c["image_channel"]
        super().__init__(nn.Sequential(
            nn.ConvTranspose2d(z, g * 8, 4, 1, 0, bias=False), nn.BatchNorm2d(g * 8), nn.ReLU(True),
            nn.ConvTranspose2d(g * 8, g * 4, 4, 2, 1, bias=False), nn.BatchNorm2d(g * 4), nn.ReLU(True),
            nn.ConvTranspose2d(g * 4, g * 2, 4, 2, 1, bias=False), nn.BatchNorm2d(g * 2), nn.ReLU(True),
            nn.ConvTranspose2d(g * 2, g, 4, 2, 1, bias=False), nn.BatchNorm2d(g), nn.ReLU(True),
            nn.ConvTranspose2d(g, ch, 4, 2, 1, bias=False), nn.Tanh(),
        ))


class Discriminator(SequentialConvNet):
    """mnist_dcgan.py:96-116."""

    def __init__(self, cfg=None):
        c = _cfg(cfg)
        d, ch = c["d_hidden"], c["image_channel"]
        super().__init__(nn.Sequential(
            nn.Conv2d(ch, d, 4, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d, d * 2, 4, 2, 1, bias=False), nn.BatchNorm2d(d * 2), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d * 2, d * 4, 4, 2, 1, bias=False), nn.BatchNorm2d(d * 4), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d * 4, d * 8, 4, 2, 1, bias=False), nn.BatchNorm2d(d * 8), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d * 8, 1, 4, 1, 0, bias=False), nn.Sigmoid(),
        ))

    def forward(self, input):
        return super().forward(input).view(-1, 1).squeeze(1)  # :116

    def forward_groups(self, inputs):
        """netD on several independent batches in one pass (nn.SequentialConvNet.forward_groups): [G*B] probabilities."""
        return super().forward_groups(inputs).view(-1, 1).squeeze(1)


def make_optimizers(netG, netD, cfg=None):
    """mnist_dcgan.py:125-127."""
    c = _cfg(cfg)
    criterion = BCELoss()
    optimizerD = Adam(netD.parameters(), lr=c["lr"], betas=(0.5, 0.999))
    optimizerG = Adam(netG.parameters(), lr=c["lr"], betas=(0.5, 0.999))
    return criterion, optimizerD, optimizerG


class _Labels:
    """The reference refills one `label` tensor per pass (:150,158,170); two constant tensors per batch size do
    the same job without a fill launch per pass."""

    def __init__(self):
        self._cache = {}

    def get(self, n, value, device):
        key = (n, float(value), device)
        t = self._cache.get(key)
        if t is None:
            t = ops.fill(torch.empty(n, dtype=torch.float32, device=device), value)
            self._cache[key] = t
        return t


_labels = _Labels()


def train_step(netG, netD, criterion, optimizerD, optimizerG, real, noise, cfg=None, dp=None, skip_dead_d_wgrad=True, pair=True):
    """One iteration of the reference loop body (mnist_dcgan.py:147-175) for a batch `real` already on the GPU and a
    noise tensor `noise` ([B, z_dim, 1, 1]; the reference draws it at :156).

    Returns the loss tensors (errD_real, errD_fake, errG) and the three D outputs, all still on the device — the
    reference's `.item()` calls (:154,162,174) are host syncs for logging and belong to the caller.

    skip_dead_d_wgrad: in the G step the reference's autograd also computes D's weight gradients, which
    `netD.zero_grad()` (:147) throws away at the next iteration.  With True they are not computed (D's parameters are
    marked requires_grad=False around that pass); parameters, losses and G gradients are unchanged.
    dp: optional `parallel.GradSync` — averages D's / G's flat gradient bucket across ranks before each Adam step.
    pair: run the D step's two discriminator passes (:151 on `real`, :159 on `fake.detach()`) as ONE pass over 2B images
    (`netD.forward_groups`, include/pcgan_hip.h "grouped batches") when the net and the batch allow it: the generator's forward
    (:157, independent of D) moves in front, `errD_real.backward(); errD_fake.backward()` (:153,161 — two accumulations into
    .grad) becomes the backward of errD = errD_real + errD_fake (:163), BatchNorm keeps per-pass statistics and two running-
    statistics updates.  Same numbers except for the order of the sum inside D's weight gradients (tests/test_hip_groups.py).
    pair=False is the statement-by-statement order.
    """
    c = _cfg(cfg)
    b_size = real.size(0)
    dev = real.device
    pair = (pair and isinstance(criterion, BCELoss) and not (dp is not None and getattr(dp, "sync_bn", False))
            and hasattr(netD, "supports_groups") and netD.supports_groups(real.shape, 2))
    # (1) Update D network
    if pair:
        netD.drop_grads()                                         # :147 — every D parameter gets a gradient below: the first writer overwrites
        if dp is not None:
            dp.wait(netG)                                         # previous iteration's G all-reduce + Adam(G) done
        fake = netG(noise)                                        # :157
        out_pair = netD.forward_groups([real, fake.detach()])    # :151 + :159
        errD_real, errD_fake, errD = _nn.bce_pair(out_pair, c["real_label"], c["fake_label"])   # :152, :160, :163
        _nn.backward(errD)                                        # :153 + :161
        out_real, out_fake = out_pair[:b_size], out_pair[b_size:]
    else:
        netD.zero_grad()                                          # :147
        label = _labels.get(b_size, c["real_label"], dev)        # :150
        out_real = netD(real)                                     # :151
        errD_real = criterion(out_real, label)                    # :152
        _nn.backward(errD_real)                                   # :153
        if dp is not None:
            dp.wait(netG)                                         # previous iteration's G all-reduce + Adam(G) done
        fake = netG(noise)                                        # :157
        label = _labels.get(b_size, c["fake_label"], dev)        # :158
        out_fake = netD(fake.detach())                            # :159
        errD_fake = criterion(out_fake, label)                    # :160
        _nn.backward(errD_fake)                                   # :161
    if dp is not None:
        dp.sync_now(netD)
    optimizerD.step()                                             # :164
    # (2) Update G network
    if pair:
        netG.drop_grads()                                         # :169 — every G parameter gets a gradient in this backward
    else:
        netG.zero_grad()                                          # :169
    label = _labels.get(b_size, c["real_label"], dev)            # :170
    if skip_dead_d_wgrad:
        for p in netD.parameters():
            p.requires_grad_(False)
    try:
        out_g = netD(fake)                                        # :171
        errG = criterion(out_g, label)                            # :172
        _nn.backward(errG)                                        # :173
    finally:
        if skip_dead_d_wgrad:
            for p in netD.parameters():
                p.requires_grad_(True)
    if dp is not None:
        dp.sync_then(netG, optimizerG.step)                       # overlapped with the next D(real) pass
    else:
        optimizerG.step()                                         # :175
    return {"errD_real": errD_real, "errD_fake": errD_fake, "errG": errG, "out_real": out_real, "out_fake": out_fake,
            "out_g": out_g}


def build(cfg=None, device="cuda", seed=None):
    """Nets built and initialised as the reference does (:119-122)."""
    c = _cfg(cfg)
    if seed is not None:
        torch.manual_seed(seed)
    netG = Generator(c).to(device)
    netG.apply(weights_init)
    netD = Discriminator(c).to(device)
    netD.apply(weights_init)
    return netG, netD


def train(dataloader, cfg=None, netG=None, netD=None, device="cuda", log=print, device_rng=None, dp=None):
    """The reference's training script from the fixed viz noise to the per-epoch loss averages (mnist_dcgan.py:129-198) with
    its plotting left out: `dataloader` yields (images, ...) batches (:143,148), per iteration one `train_step`, every 200
    iterations the log line (:178-181), every 500 iterations and at the very end a forward of the generator on the fixed
    `viz_noise` under no_grad (:187-191).  That viz forward runs with the generator in TRAINING mode in the reference, so it
    updates the BatchNorm running statistics — it is part of the numerical path and is kept; the image-grid rendering
    (torchvision.make_grid, :190) is plotting and is not: `img_list` holds the raw generated batches on the CPU.

    Noise: drawn like the reference does (`torch.randn` from torch's global generator, here on the CPU and copied to the device:
    seed-for-seed the draws of the reference's CPU path), or on the GPU from `device_rng` (an ops.DeviceRNG).
    Host syncs: the reference calls .item() six times per iteration; here the loss tensors are read once per epoch (and at the log
    lines), summed in the same order, so the averages are the same numbers.
    Returns a dict: netG, netD, epoch_G_losses, epoch_D_losses, img_list, iters."""
    c = _cfg(cfg)
    dev = torch.device(device)
    if netG is None or netD is None:
        netG, netD = build(c, device=dev)                                     # :119-122
    criterion, optimizerD, optimizerG = make_optimizers(netG, netD, c)        # :125-127

    def randn(b):
        if device_rng is not None:
            return device_rng.randn((b, c["z_dim"], 1, 1), dev)
        return torch.randn(b, c["z_dim"], 1, 1).to(dev)

    viz_noise = randn(c["batch_size"])                                        # :130
    img_list, epoch_G_losses, epoch_D_losses = [], [], []
    iters = 0
    nbatches = len(dataloader)
    log("Starting Training Loop...")                                          # :139
    for epoch in range(c["epochs"]):                                          # :140
        pending = []                                                          # (errD_real, errD_fake, errG) device scalars of the epoch
        for i, data in enumerate(dataloader):                                 # :143
            real = data[0].to(dev)                                            # :148
            noise = randn(real.size(0))                                       # :156
            out = train_step(netG, netD, criterion, optimizerD, optimizerG, real, noise, c, dp=dp)   # :147-175
            pending.append((out["errD_real"].detach(), out["errD_fake"].detach(), out["errG"].detach()))   # no grad_fn kept: see countergan.train_countergan
            if i % 200 == 0:                                                  # :178-181
                errD = float(np.float32(out["errD_real"].item()) + np.float32(out["errD_fake"].item()))   # :163
                log(f"[{epoch}/{c['epochs']}][{i}/{nbatches}] Loss_D: {errD:.4f} Loss_G: {out['errG'].item():.4f} "
                    f"D(x): {ops.mean_fwd(out['out_real'].detach().contiguous()).item():.4f} "
                    f"D(G(z)): {ops.mean_fwd(out['out_fake'].detach().contiguous()).item():.4f} / "
                    f"{ops.mean_fwd(out['out_g'].detach().contiguous()).item():.4f}")
            if iters % 500 == 0 or (epoch == c["epochs"] - 1 and i == nbatches - 1):   # :187
                if dp is not None:
                    dp.wait(netG)                                             # Adam(G) of this iteration runs on the side stream
                with torch.no_grad():
                    img_list.append(netG(viz_noise).detach().cpu())           # :188-189 (train mode: updates running stats)
            iters += 1                                                        # :193
        running_G_loss = running_D_loss = 0.0                                 # :141-142, :184-185 (same order of additions)
        for e_real, e_fake, e_g in pending:
            running_G_loss += e_g.item()
            running_D_loss += float(np.float32(e_real.item()) + np.float32(e_fake.item()))   # errD is an fp32 sum (:163)
        epoch_G_losses.append(running_G_loss / nbatches)                      # :195-198
        epoch_D_losses.append(running_D_loss / nbatches)
    if dp is not None:
        dp.wait_all()
    return {"netG": netG, "netD": netD, "epoch_G_losses": epoch_G_losses, "epoch_D_losses": epoch_D_losses, "img_list": img_list,
            "iters": iters}

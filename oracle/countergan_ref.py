"""ORACLE — test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline); never the product path.

CPU restatement of the reference's CounteRGAN/mnist hot path:
    conditional_counteRGAN/mnist/config.py            (hyper-parameters, :3-28)
    conditional_counteRGAN/mnist/models/generator.py  (_ResBlock :5-22, ResidualGenerator :25-86)
    conditional_counteRGAN/mnist/models/discriminator.py (:5-38)
    conditional_counteRGAN/mnist/models/classifier.py (:4-28)
    conditional_counteRGAN/mnist/trainer.py           (build_mask :45-72, loop body of train_countergan :89-123)
on the same PyTorch operators the reference calls.  Unlike the DCGAN script these modules ARE importable, so
tests/golden/make_golden.py imports them directly and tests/test_oracle_golden.py pins this restatement to their
outputs (forward, losses, every gradient, parameters after Adam).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim


class Config:
    """config.py:3-28 (the fields the training step reads)."""
    batch_size = 128
    d_lr = 1e-5
    g_lr = 5e-5
    lambda_adv = 1.0
    lambda_cls = 1.0
    lambda_reg = 2.5
    lambda_mask = 2.0
    patch_size = 7
    num_modifiable_patches = 10
    img_shape = (1, 28, 28)
    num_classes = 10


class _ResBlock(nn.Module):
    """generator.py:5-22 — x + 0.1 * bn2(conv2(act(bn1(conv1(x))))); convs 3x3 p1 with bias; shared LeakyReLU(0.2)."""

    def __init__(self, channels, activation):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(channels)
        self.act = activation
        self.conv2 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm2d(channels)

    def forward(self, x):
        out = self.act(self.bn1(self.conv1(x)))        # :17
        out = self.bn2(self.conv2(out))                # :18
        return x + 0.1 * out                           # :20


class ResidualGenerator(nn.Module):
    """generator.py:25-86."""

    def __init__(self, img_shape=(1, 28, 28), num_classes=10, base_ch=64, n_resblocks=6, residual_scaling=0.1):
        super().__init__()
        C, H, W = img_shape
        self.embed = nn.Embedding(num_classes, H * W)                          # :36
        self.conv_in = nn.Conv2d(C + 2, base_ch, kernel_size=3, padding=1)     # :39
        self.act = nn.LeakyReLU(0.2, inplace=True)                             # :40
        self.resblocks = nn.Sequential(*[_ResBlock(base_ch, self.act) for _ in range(n_resblocks)])  # :43-46
        self.conv_mid = nn.Conv2d(base_ch, base_ch, kernel_size=3, padding=1)  # :49
        self.conv_out = nn.Conv2d(base_ch, 1, kernel_size=3, padding=1)        # :50
        self.residual_scaling = residual_scaling
        self._init_weights()

    def _init_weights(self):
        """generator.py:58-69."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0.2)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, mean=0.0, std=0.01)

    def forward(self, x, target, mask=None):
        B, C, H, W = x.shape
        y_map = self.embed(target).view(B, 1, H, W).to(x.dtype)               # :73
        inp = torch.cat([x, y_map, mask], dim=1)                               # :74
        h = self.act(self.conv_in(inp))                                        # :76
        h = self.resblocks(h)                                                  # :77
        h = self.act(self.conv_mid(h))                                         # :78
        raw_residual = self.conv_out(h) * self.residual_scaling               # :80
        masked_residual = raw_residual * mask                                  # :82
        return raw_residual, masked_residual


class Discriminator(nn.Module):
    """discriminator.py:5-38."""

    def __init__(self, img_shape=(1, 28, 28), num_classes=10):
        super().__init__()
        C, H, W = img_shape
        self.cond_embed = nn.Embedding(num_classes, H * W)
        self.img_channel = 2
        self.d_hidden = 64
        d = self.d_hidden
        self.main = nn.Sequential(
            nn.Conv2d(2, d, 3, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d, d * 2, 3, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d * 2, d * 4, 3, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d * 4, d * 4, 3, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
            nn.AdaptiveAvgPool2d(1),
        )
        self.flatten = nn.Flatten()
        self.adv_head = nn.Linear(d * 4, 1)

    def forward(self, x, cond_idx):
        B, C, H, W = x.shape
        cond_map = self.cond_embed(cond_idx).view(B, 1, H, W)                 # :35
        z = self.main(torch.cat([x, cond_map], dim=1))                         # :36
        return self.adv_head(self.flatten(z))                                  # :37-38


class CNNClassifier(nn.Module):
    """classifier.py:4-28."""

    def __init__(self, num_classes=10):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(1, 32, 3, 1, 1), nn.ReLU(),
            nn.Conv2d(32, 64, 3, 2, 1), nn.ReLU(),
            nn.Conv2d(64, 128, 3, 2, 1), nn.ReLU(),
            nn.Dropout2d(0.25),
        )
        self.fc = nn.Sequential(nn.Flatten(), nn.Linear(128 * 7 * 7, 256), nn.ReLU(), nn.Dropout(0.5), nn.Linear(256, num_classes))

    def forward(self, x):
        return self.fc(self.conv(x))


def build_mask(x, patch_size, device, num_modifiable_patches=None):
    """trainer.py:45-72 (per-sample randperm of the patch grid, nearest upsample)."""
    bs, c, h, w = x.shape
    nph, npw = h // patch_size, w // patch_size
    total = nph * npw
    patch_mask = torch.zeros((bs, 1, nph, npw), device=device)
    if num_modifiable_patches is None or num_modifiable_patches >= total:
        patch_mask = torch.randint(0, 2, patch_mask.shape, device=device).float()
    else:
        for b in range(bs):
            idx = torch.randperm(total, device=device)[:num_modifiable_patches]
            patch_mask.view(bs, -1)[b, idx] = 1.0
    return F.interpolate(patch_mask, size=(h, w), mode="nearest").repeat(1, c, 1, 1)


def make_optimizers(generator, discriminator, cfg=Config):
    """trainer.py:77-80."""
    opt_g = optim.Adam(generator.parameters(), lr=cfg.g_lr)
    opt_d = optim.Adam(discriminator.parameters(), lr=cfg.d_lr)
    return opt_g, opt_d, nn.BCEWithLogitsLoss(), nn.CrossEntropyLoss()


def countergan_step(generator, discriminator, classifier, opt_g, opt_d, bce, ce, x, y, target_y, mask, cfg=Config):
    """One iteration of train_countergan's loop body (trainer.py:89-123) with the random draws (`target_y` :94,
    `mask` :95) supplied by the caller.  Returns the logged scalars as Python floats."""
    raw_residual, masked_residual = generator(x, target_y, mask)              # :96
    x_cf = torch.clamp(x + masked_residual, -1.0, 1.0)                        # :97
    mask_penalty_pre = torch.mean(torch.abs(raw_residual * (1.0 - mask)))     # :99
    # Discriminator update
    opt_d.zero_grad()                                                          # :102
    d_real_logits = discriminator(x, y)                                       # :103
    d_fake_logits = discriminator(x_cf.detach(), target_y)                    # :104
    d_loss = bce(d_real_logits, torch.ones_like(d_real_logits)) + bce(d_fake_logits, torch.zeros_like(d_fake_logits))  # :106-107
    d_loss.backward()                                                          # :111
    opt_d.step()                                                               # :112
    # Generator update
    opt_g.zero_grad()                                                          # :115
    g_fake_logits = discriminator(x_cf, target_y)                             # :116
    g_adv = bce(g_fake_logits, torch.ones_like(g_fake_logits))                # :117
    g_cls = ce(classifier(x_cf), target_y)                                    # :118
    reg_l1 = torch.abs(masked_residual).mean()                                # :119
    g_loss = cfg.lambda_adv * g_adv + cfg.lambda_cls * g_cls + cfg.lambda_reg * reg_l1 + cfg.lambda_mask * mask_penalty_pre  # :121
    g_loss.backward()                                                          # :122
    opt_g.step()                                                               # :123
    return {"d_loss": d_loss.item(), "g_loss": g_loss.item(), "g_adv": g_adv.item(), "g_cls": g_cls.item(),
            "reg_l1": reg_l1.item(), "mask_pen": mask_penalty_pre.item(),
            "d_real_p": torch.sigmoid(d_real_logits).mean().item(), "d_fake_p": torch.sigmoid(d_fake_logits).mean().item()}


def grad_norm(parameters):
    """trainer.py:41-42."""
    return torch.sqrt(sum((p.grad.data.norm() ** 2) for p in parameters if p.grad is not None)).item()


def train_countergan(generator, discriminator, classifier, batches, target_y, mask, epochs, cfg=Config):
    """trainer.py:76-147 with the draws supplied (target_y[e][i], mask[e][i]): per-epoch means of g_loss / d_loss / g_cls
    (:139-141) and grad_norm of both nets at the epoch's end (:142-143).  Returns (history, G_grad list, D_grad list, per-epoch
    first-batch log scalars)."""
    opt_g, opt_d, bce, ce = make_optimizers(generator, discriminator, cfg)
    hist, gG, gD, first = [], [], [], []
    for e in range(epochs):
        g_epoch = d_epoch = cls_epoch = 0.0
        for i, (x, y) in enumerate(batches):
            out = countergan_step(generator, discriminator, classifier, opt_g, opt_d, bce, ce, x, y, target_y[e][i], mask[e][i], cfg)
            g_epoch += out["g_loss"]; d_epoch += out["d_loss"]; cls_epoch += out["g_cls"]
            if i == 0:
                first.append(out)
        n = len(batches)
        hist.append((g_epoch / n, d_epoch / n, cls_epoch / n))
        gG.append(grad_norm(generator.parameters())); gD.append(grad_norm(discriminator.parameters()))
    return hist, gG, gD, first


def synthetic_batch(batch, seed, cfg=Config, dtype=torch.float32):
    """x ~ U[-1,1) [B,1,28,28]; y, target_y ~ U{0..9}; mask = 10 of 16 7x7 patches per sample."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 1, 28, 28, generator=g, dtype=dtype) * 2 - 1
    y = torch.randint(0, cfg.num_classes, (batch,), generator=g)
    target_y = torch.randint(0, cfg.num_classes, (batch,), generator=g)
    grid = 28 // cfg.patch_size
    pm = torch.zeros(batch, grid * grid, dtype=dtype)
    for b in range(batch):
        pm[b, torch.randperm(grid * grid, generator=g)[:cfg.num_modifiable_patches]] = 1.0
    mask = F.interpolate(pm.view(batch, 1, grid, grid), size=(28, 28), mode="nearest")
    return x, y, target_y, mask


def build(seed=0):
    """Generator / discriminator / frozen classifier as main.py:18-33 builds them (classifier random-init here: the
    trained checkpoint `best_classifier.pt` is not shipped with the reference — SURVEY.md §8c)."""
    torch.manual_seed(seed)
    classifier = CNNClassifier(Config.num_classes)
    generator = ResidualGenerator(Config.img_shape, Config.num_classes)
    discriminator = Discriminator(Config.img_shape, Config.num_classes)
    classifier.eval()                                   # main.py:30
    for p in classifier.parameters():                   # main.py:31-33
        p.requires_grad = False
    return generator, discriminator, classifier


def evaluate_counterfactuals(generator, classifier, x, y_true, y_target):
    """eval_utils.py:46-79 (eval mode, all-ones mask, clamp to [-1,1]; flip rate, prediction gain, actionability)."""
    import torch.nn.functional as F
    classifier.eval(); generator.eval()
    with torch.no_grad():
        residual = generator(x, y_target, torch.ones_like(x))[1]
        x_cf = torch.clamp(x + residual, -1.0, 1.0)
        logits = classifier(x_cf)
        probs = F.softmax(logits, dim=1)
    ar = torch.arange(len(y_target))
    return {"class_flip_rate": (logits.argmax(1) == y_target).float().mean().item(),
            "prediction_gain": (probs[ar, y_target] - probs[ar, y_true]).mean().item(),
            "actionability": torch.abs(x_cf - x).mean().item()}, ((x + 1.0) / 2.0, (x_cf + 1.0) / 2.0)


def classifier_forward_train(C, x, masks):
    """CNNClassifier.forward in training mode (classifier.py:22-28) with the Dropout2d(0.25) / Dropout(0.5) noise supplied:
    masks = (m2d [B,128], m1 [B,256]) of {0,1}; [torch] dropout multiplies by mask / (1 - p)."""
    m2d, m1 = masks
    h = x
    for m in C.conv:
        h = h * (m2d[:, :, None, None].to(h.dtype) / (1 - m.p)) if isinstance(m, nn.Dropout2d) else m(h)
    for m in C.fc:
        h = h * (m1.to(h.dtype) / (1 - m.p)) if isinstance(m, nn.Dropout) else m(h)
    return h


def classifier_train_step(C, optimizer, x, y, masks):
    """One iteration of train_classifier's inner loop, trainer.py:16-20."""
    import torch.nn.functional as F
    C.train()
    optimizer.zero_grad()
    loss = F.cross_entropy(classifier_forward_train(C, x, masks), y)
    loss.backward()
    optimizer.step()
    return loss.item()

"""ORACLE — test infrastructure only.  CPU restatement of the reference's tabular "prompted counterfactual" CounteRGAN,
`conditional_counteRGAN/house_sales_kc_usa/`:
    config.py                  :13-82   (dimensions, categorical heads, loss weights)
    models/generator.py        FiLM :6-16, ResidualBlock :19-35, ResidualGenerator :38-92
    models/discriminator.py    :5-20    (four spectral-norm Linears + LeakyReLU(0.2))
    models/nn_classifier.py    :4-32    (frozen MLP classifier, eval mode)
    trainer.py                 loop body of train_countergan :241-316, cat_norm_maps :205-223, the per-iteration diagnostics
                               :318-343, epoch means + grad_norm :349-366 (train_countergan below)
on the same PyTorch operators.  The reference modules are importable: tests/golden/make_golden.py runs ONE batch through
the reference's own train_countergan and records the random draws it made (targets, feature mask, Gumbel noise), so this
restatement is pinned on the identical draws.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.utils.spectral_norm as spectral_norm
import torch.optim as optim

# config.py:13-82 (the fields the step reads)
FEATURES = ["bedrooms", "bathrooms", "sqft_living", "sqft_lot", "floors", "waterfront", "view", "condition", "grade",
            "sqft_above", "sqft_basement", "yr_built", "yr_renovated", "lat", "long", "sqft_living15", "sqft_lot15"]
CONFIG = {
    "input_dim": 17, "num_classes": 4, "hidden_dim": 32, "lr_G": 1e-3, "lr_D": 1e-3,
    "lambda_cls": 2.0, "lambda_reg": 1.0, "lambda_mask": 1.0, "gumbel_tau": 0.5,
    "immutable_idx": [FEATURES.index(f) for f in ("lat", "long", "yr_built", "yr_renovated")],
    "categorical_info": {FEATURES.index("bedrooms"): 9, FEATURES.index("bathrooms"): 30, FEATURES.index("floors"): 6,
                         FEATURES.index("waterfront"): 2, FEATURES.index("view"): 5, FEATURES.index("condition"): 5,
                         FEATURES.index("grade"): 13},
}
CONFIG["continuous_idx"] = [i for i in range(17) if i not in CONFIG["categorical_info"]]


class FiLM(nn.Module):
    def __init__(self, hidden_dim, cond_dim):
        super().__init__()
        self.gamma = nn.Linear(cond_dim, hidden_dim)
        self.beta = nn.Linear(cond_dim, hidden_dim)

    def forward(self, h, cond):
        return self.gamma(cond) * h + self.beta(cond)            # generator.py:13-16


class ResidualBlock(nn.Module):
    def __init__(self, hidden_dim, cond_dim):
        super().__init__()
        self.fc1 = nn.Linear(hidden_dim, hidden_dim)
        self.bn1 = nn.BatchNorm1d(hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.bn2 = nn.BatchNorm1d(hidden_dim)
        self.film = FiLM(hidden_dim, cond_dim)

    def forward(self, h, cond):
        out = F.relu(self.film(self.bn1(self.fc1(h)), cond))      # :28-30
        out = self.film(self.bn2(self.fc2(out)), cond)            # :31-33 (the SAME FiLM module, applied twice)
        return h + out                                            # :34


class ResidualGenerator(nn.Module):
    """generator.py:38-92, with the Gumbel noise of F.gumbel_softmax (:90) supplied: `gumbel[idx]` per categorical head."""

    def __init__(self, input_dim, hidden_dim, num_classes, continuous_idx, categorical_info, n_blocks=5, residual_scaling=0.1, tau=0.5):
        super().__init__()
        self.continuous_idx = list(continuous_idx)
        self.categorical_info = dict(categorical_info)
        self.cond_dim = input_dim + num_classes
        self.fc_in = nn.Linear(input_dim + self.cond_dim, hidden_dim)
        self.blocks = nn.ModuleList([ResidualBlock(hidden_dim, self.cond_dim) for _ in range(n_blocks)])
        self.fc_cont = nn.Linear(hidden_dim, len(self.continuous_idx))
        self.fc_cat_logits = nn.ModuleDict({str(i): nn.Linear(hidden_dim, n) for i, n in self.categorical_info.items()})
        self.residual_scaling = residual_scaling
        self.tau = tau

    def forward(self, x, target_onehot, mask, gumbel, temperature=None, hard=False):
        cond = torch.cat([target_onehot, mask], dim=1)            # :73
        h = F.relu(self.fc_in(torch.cat([x, cond], dim=1)))       # :74-75
        for b in self.blocks:
            h = b(h, cond)
        cont_residual = self.fc_cont(h) * self.residual_scaling   # :81
        tau = self.tau if temperature is None else float(temperature)
        cat_logits, cat_samples = {}, {}
        for idx_str, head in self.fc_cat_logits.items():
            logits = head(h)
            cat_logits[int(idx_str)] = logits
            soft = ((logits + gumbel[int(idx_str)]) / tau).softmax(-1)                       # [torch] F.gumbel_softmax
            if hard:                                                                         # straight-through one-hot (eval_utils.py:77)
                onehot = torch.zeros_like(soft).scatter_(-1, soft.max(-1, keepdim=True)[1], 1.0)
                soft = onehot - soft.detach() + soft
            cat_samples[int(idx_str)] = soft
        return cont_residual, cat_logits, cat_samples


class Discriminator(nn.Module):
    """discriminator.py:5-20."""

    def __init__(self, input_dim, hidden_dim, num_classes):
        super().__init__()
        self.net = nn.Sequential(
            spectral_norm(nn.Linear(input_dim + num_classes, hidden_dim)), nn.LeakyReLU(0.2, inplace=True),
            spectral_norm(nn.Linear(hidden_dim, hidden_dim * 2)), nn.LeakyReLU(0.2, inplace=True),
            spectral_norm(nn.Linear(hidden_dim * 2, hidden_dim * 4)), nn.LeakyReLU(0.2, inplace=True),
            spectral_norm(nn.Linear(hidden_dim * 4, 1)))

    def forward(self, x, target_onehot):
        return self.net(torch.cat([x, target_onehot], dim=1))


class NNClassifier(nn.Module):
    """nn_classifier.py:4-32."""

    def __init__(self, input_dim, output_dim=4):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(input_dim, 256), nn.LeakyReLU(0.1), nn.BatchNorm1d(256), nn.Dropout(0.3),
            nn.Linear(256, 256), nn.LeakyReLU(0.1), nn.BatchNorm1d(256), nn.Dropout(0.2),
            nn.Linear(256, 128), nn.LeakyReLU(0.1), nn.BatchNorm1d(128), nn.Dropout(0.1),
            nn.Linear(128, 64), nn.LeakyReLU(0.1), nn.BatchNorm1d(64),
            nn.Linear(64, output_dim))

    def forward(self, x):
        return self.net(x)


def cat_norm_maps(config=CONFIG):
    """trainer.py:218-223 (the no-scaler fallback: category k of n -> k / (n - 1))."""
    return {f: torch.arange(n, dtype=torch.float32) / max(1.0, n - 1) for f, n in config["categorical_info"].items()}


def make_optimizers(G, D, config=CONFIG):
    return optim.Adam(G.parameters(), lr=config["lr_G"]), optim.Adam(D.parameters(), lr=config["lr_D"])   # :230-231


def house_step(G, D, clf, opt_G, opt_D, x, y, target_y, mask, gumbel, norm_maps, config=CONFIG):
    """One iteration of the loop body, trainer.py:241-316, with the draws (target :248-249, mask :253-255, Gumbel noise
    inside G :259-261) supplied."""
    nc = config["num_classes"]
    bs, d_dim = x.shape
    target_onehot = F.one_hot(target_y, nc).to(x.dtype)                                      # :250
    cont_residual, cat_logits, cat_samples = G(x, target_onehot, mask, gumbel, temperature=config["gumbel_tau"])  # :259-261
    residual_full = torch.zeros((bs, d_dim), dtype=cont_residual.dtype)                      # :266
    for i, f in enumerate(config["continuous_idx"]):
        residual_full[:, f] = cont_residual[:, i]                                            # :269-270
    for f, sample in cat_samples.items():
        residual_full[:, f] = sample.matmul(norm_maps[f].to(x.dtype)) - x[:, f]             # :274-279
    masked_residual = residual_full * mask                                                   # :281
    x_cf = x + masked_residual                                                               # :282
    mask_penalty_pre = torch.mean(torch.abs(residual_full * (1.0 - mask)))                   # :287
    D_real = D(x, F.one_hot(y, nc).to(x.dtype))                                              # :290
    D_fake = D(x_cf.detach(), target_onehot)                                                 # :291
    D_loss = -D_real.mean() + D_fake.mean()                                                  # :292
    opt_D.zero_grad(); D_loss.backward(); opt_D.step()                                       # :293-295
    D_fake_forG = D(x_cf, target_onehot)                                                     # :298
    G_adv = -D_fake_forG.mean()                                                              # :299
    G_cls = F.cross_entropy(clf(x_cf), target_y)                                             # :301-302
    G_reg = torch.mean(torch.norm(masked_residual, p=1, dim=1))                              # :305
    G_loss = G_adv + config["lambda_cls"] * G_cls + config["lambda_reg"] * G_reg + config["lambda_mask"] * mask_penalty_pre  # :307-312
    opt_G.zero_grad(); G_loss.backward(); opt_G.step()                                       # :314-316
    return {"D_loss": D_loss.item(), "G_loss": G_loss.item(), "g_adv": G_adv.item(), "g_cls": G_cls.item(),
            "reg": G_reg.item(), "mask_pen": mask_penalty_pre.item(),
            "d_real_p": torch.sigmoid(D_real).mean().item(), "d_fake_p": torch.sigmoid(D_fake_forG).mean().item()}


def cat_norm_maps_scaler(data_min, data_max, raw_values):
    """trainer.py:205-216: normalised category values from the fitted MinMaxScaler's data_min_ / data_max_."""
    out = {}
    for f, raw in raw_values.items():
        rng_ = float(data_max[f]) - float(data_min[f])
        out[f] = torch.tensor((np.asarray(raw, dtype=float) - float(data_min[f])) / (rng_ + 1e-12), dtype=torch.float32)
    return out


def grad_norm(params):
    """trainer.py:182-183: the SUM of the parameters' gradient norms."""
    return sum(p.grad.norm().item() for p in params if p.grad is not None)


def train_countergan(G, D, clf, X, y, rows, target_y, mask, gumbel, norm_maps, config=CONFIG):
    """trainer.py:237-366 with everything random supplied: rows[e][i] (the DataLoader's batch), target_y / mask / gumbel[e][i]
    (packed [B, T] in head order).  Per iteration the step (:241-316, house_step) and the diagnostics (:318-343); per epoch the
    means (:349-355) and grad_norm of G and D (:362).  Returns the per-iteration lists and the per-epoch grad norms."""
    opt_G, opt_D = make_optimizers(G, D, config)
    heads = list(config["categorical_info"].items())
    names = ("d_loss", "g_loss", "pred_gain", "sparsity", "l2_reg", "class_flip_rate")
    it = {k: [] for k in names}
    gG, gD = [], []
    nc = config["num_classes"]
    for e in range(len(rows)):
        for k in names:
            it[k].append([])
        for i in range(len(rows[e])):
            x, yb = X[rows[e][i]], y[rows[e][i]]
            t, m = target_y[e][i], mask[e][i]
            gd, off = {}, 0
            for f, n in heads:
                gd[f] = gumbel[e][i][:, off:off + n]
                off += n
            # --- the step, keeping what the diagnostics read (house_step restated with its intermediates exposed)
            bs, d_dim = x.shape
            target_onehot = F.one_hot(t, nc).to(x.dtype)
            cont_residual, _, cat_samples = G(x, target_onehot, m, gd, temperature=config["gumbel_tau"])
            residual_full = torch.zeros((bs, d_dim), dtype=cont_residual.dtype)
            for j, f in enumerate(config["continuous_idx"]):
                residual_full[:, f] = cont_residual[:, j]
            for f, sample in cat_samples.items():
                residual_full[:, f] = sample.matmul(norm_maps[f].to(x.dtype)) - x[:, f]
            masked_residual = residual_full * m
            x_cf = x + masked_residual
            mask_penalty_pre = torch.mean(torch.abs(residual_full * (1.0 - m)))
            D_real = D(x, F.one_hot(yb, nc).to(x.dtype))
            D_fake = D(x_cf.detach(), target_onehot)
            D_loss = -D_real.mean() + D_fake.mean()
            opt_D.zero_grad(); D_loss.backward(); opt_D.step()
            D_fake_forG = D(x_cf, target_onehot)
            G_adv = -D_fake_forG.mean()
            clf_preds = clf(x_cf)
            G_cls = F.cross_entropy(clf_preds, t)
            G_reg = torch.mean(torch.norm(masked_residual, p=1, dim=1))
            G_loss = G_adv + config["lambda_cls"] * G_cls + config["lambda_reg"] * G_reg + config["lambda_mask"] * mask_penalty_pre
            opt_G.zero_grad(); G_loss.backward(); opt_G.step()
            with torch.no_grad():                                                               # :318-343
                probs_orig, probs_cf = F.softmax(clf(x), dim=1), F.softmax(clf_preds, dim=1)
                ar = torch.arange(bs)
                it["pred_gain"][-1].append((probs_cf[ar, t] - probs_orig[ar, t]).mean().item())
                it["sparsity"][-1].append(1.0 - (torch.abs(masked_residual) > 1e-3).float().mean().item())
                it["l2_reg"][-1].append(torch.mean(torch.norm(masked_residual, p=2, dim=1)).item())
                it["class_flip_rate"][-1].append((torch.argmax(clf_preds, dim=1) == t).float().mean().item())
            it["d_loss"][-1].append(D_loss.item()); it["g_loss"][-1].append(G_loss.item())
        gG.append(grad_norm(G.parameters())); gD.append(grad_norm(D.parameters()))              # :362
    return it, gG, gD


def build_counterfactuals(G, x, target_onehot, gumbel, norm_maps, config=CONFIG):
    """eval_utils.py:25-181 for this generator's (cont_residual, cat_logits, cat_samples) signature: immutable mask :48-50,
    hard Gumbel-softmax :76-77, residual assembly :127-171, mask :174-177, clamp to [0,1] :180."""
    mask = torch.ones_like(x)
    mask[:, config["immutable_idx"]] = 0.0
    cont_residual, _, cat_samples = G(x, target_onehot, mask, gumbel, temperature=config["gumbel_tau"], hard=True)
    residual_full = torch.zeros_like(x)
    for i, f in enumerate(config["continuous_idx"]):
        residual_full[:, f] = cont_residual[:, i]
    for f, sample in cat_samples.items():
        residual_full[:, f] = sample.matmul(norm_maps[f].to(x.dtype)) - x[:, f]
    masked_residual = residual_full * mask
    return masked_residual, torch.clamp(x + masked_residual, 0.0, 1.0)


def metrics_one_target(G, clf, x, target, gumbel, norm_maps, config=CONFIG):
    """One (batch, target class) step of compute_metrics_per_target, eval_utils.py:233-262, for rows whose class != target."""
    nc = config["num_classes"]
    target_vec = torch.full((x.shape[0],), target, dtype=torch.long)
    masked_residual, _ = build_counterfactuals(G, x, F.one_hot(target_vec, nc).to(x.dtype), gumbel, norm_maps, config)
    x_cf = x + masked_residual                                                                # :243
    probs_orig, logits_cf = F.softmax(clf(x), dim=1), clf(x_cf)
    probs_cf = F.softmax(logits_cf, dim=1)
    ar = torch.arange(x.shape[0])
    return {"class_flip": (logits_cf.argmax(1) == target_vec).float().mean().item(),
            "prediction_gain": (probs_cf[ar, target_vec] - probs_orig[ar, target_vec]).mean().item(),
            "avg_actionability": masked_residual.abs().mean().item(), "x_cf": x_cf}


def build(seed=0, config=CONFIG):
    torch.manual_seed(seed)
    clf = NNClassifier(config["input_dim"], config["num_classes"])
    G = ResidualGenerator(config["input_dim"], config["hidden_dim"], config["num_classes"], config["continuous_idx"],
                          config["categorical_info"], tau=config["gumbel_tau"])
    D = Discriminator(config["input_dim"], config["hidden_dim"], config["num_classes"])
    clf.eval()
    for p in clf.parameters():
        p.requires_grad = False
    return G, D, clf


def synthetic_batch(batch, seed, config=CONFIG, dtype=torch.float32):
    """x ~ U[0,1) [B,17]; y ~ U{0..3}; target != y; Bernoulli(0.5) mask with immutable columns zeroed;
    Gumbel(0,1) noise per categorical head."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, config["input_dim"], generator=g, dtype=dtype)
    y = torch.randint(0, config["num_classes"], (batch,), generator=g)
    t = torch.randint(0, config["num_classes"], (batch,), generator=g)
    t = torch.where(t == y, (t + 1) % config["num_classes"], t)
    mask = torch.randint(0, 2, (batch, config["input_dim"]), generator=g).to(dtype)
    mask[:, config["immutable_idx"]] = 0.0
    gumbel = {f: -torch.empty(batch, n, dtype=dtype).exponential_(generator=g).log() for f, n in config["categorical_info"].items()}
    return x, y, t, mask, gumbel


def classifier_forward_train(clf, x, masks):
    """NNClassifier.forward in training mode (nn_classifier.py:8-32) with the three Dropout noises supplied (masks of {0,1})."""
    h, it = x, iter(masks)
    for m in clf.net:
        h = h * (next(it).to(h.dtype) / (1 - m.p)) if isinstance(m, nn.Dropout) else m(h)
    return h


def balanced_class_weights(y, num_classes):
    """[sklearn] compute_class_weight('balanced'): n_samples / (n_classes * bincount(y))   (trainer.py:53)."""
    import numpy as np
    y = np.asarray(y)
    return len(y) / (num_classes * np.bincount(y, minlength=num_classes).astype(np.float64))


def classifier_train_step(clf, optimizer, class_weights, x, y, masks):
    """One iteration of train_classifier's inner loop, trainer.py:78-87 (weighted CrossEntropyLoss :55-57)."""
    clf.train()
    optimizer.zero_grad()
    loss = F.cross_entropy(classifier_forward_train(clf, x, masks), y, weight=class_weights)
    loss.backward()
    optimizer.step()
    return loss.item()

"""Host-side mirror of `conditional_counteRGAN/house_sales_kc_usa/` (the tabular, "prompted" CounteRGAN) on the HIP kernels.

    reference                                              here
    ---------------------------------------------------    -----------------------------------------------------
    config.py                          :13-82               CONFIG (the fields the step reads)
    models/generator.py  FiLM :6-16, ResidualBlock :19-35   FiLM, ResidualBlock (parameter containers)
    models/generator.py  ResidualGenerator :38-92           ResidualGenerator (same ctor args, same state_dict keys)
    models/discriminator.py Discriminator :5-20             Discriminator (spectral-norm keys weight_orig / weight_u / weight_v)
    models/nn_classifier.py NNClassifier :4-32              NNClassifier (frozen / eval use: forward + grad-input)
    trainer.py  cat_norm_maps :205-223                      cat_norm_maps
    trainer.py  train_countergan loop body :241-316         make_optimizers + train_step
    trainer.py  train_countergan :186-378                   train_countergan(generator, config, X_train, y_train, clf_model)

Every network is one autograd node that sequences C-ABI calls (csrc/tabular.hip + the BatchNorm / activation / loss kernels)
and accumulates parameter gradients straight into the FlatModule's flat gradient buffer.  The categorical heads are packed
side by side: `logits` / `samples` are [B, T] with T = sum of category counts (70 for the King-County config), head s
occupying columns seg[s]..seg[s+1]; the reference's dict views are column slices of those.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import ops
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, PcgError, load as _lib_load
from .countergan import CrossEntropyLoss, abs_mean, grad_norm  # noqa: F401  (same loss kernels)
from .nn import FlatModule, affine_fwd, linear_dgrad as _lin_dgrad, linear_fwd as _lin_fwd, linear_wgrad as _lin_wgrad, mean, weighted_sum  # noqa: F401
from .optim import Adam

FEATURES = ["bedrooms", "bathrooms", "sqft_living", "sqft_lot", "floors", "waterfront", "view", "condition", "grade",
            "sqft_above", "sqft_basement", "yr_built", "yr_renovated", "lat", "long", "sqft_living15", "sqft_lot15"]
# config.py:13-82 — category counts of the shipped dataset; `raw_values` come from the data loader (data_utils.py) at run time
CONFIG = {
    "input_dim": 17, "num_classes": 4, "hidden_dim": 32, "lr_G": 1e-3, "lr_D": 1e-3, "batch_size": 128, "epochs": 60,
    "lambda_cls": 2.0, "lambda_reg": 1.0, "lambda_mask": 1.0, "gumbel_tau": 0.5,
    "immutable_idx": [FEATURES.index(f) for f in ("lat", "long", "yr_built", "yr_renovated")],
    "categorical_info": {FEATURES.index(f): {"n": n} for f, n in
                         (("bedrooms", 9), ("bathrooms", 30), ("floors", 6), ("waterfront", 2), ("view", 5), ("condition", 5),
                          ("grade", 13))},
}
CONFIG["continuous_idx"] = [i for i in range(17) if i not in CONFIG["categorical_info"]]


def _ncat(info):
    return int(info["n"]) if isinstance(info, dict) else int(info)


# ---- generator -------------------------------------------------------------------------------------------------------
class FiLM(nn.Module):
    """generator.py:6-16 (container; the arithmetic is pcg_film_fwd/bwd)."""

    def __init__(self, hidden_dim, cond_dim):
        super().__init__()
        self.gamma = nn.Linear(cond_dim, hidden_dim)
        self.beta = nn.Linear(cond_dim, hidden_dim)


class ResidualBlock(nn.Module):
    """generator.py:19-35 (container)."""

    def __init__(self, hidden_dim, cond_dim):
        super().__init__()
        self.fc1 = nn.Linear(hidden_dim, hidden_dim)
        self.bn1 = nn.BatchNorm1d(hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.bn2 = nn.BatchNorm1d(hidden_dim)
        self.film = FiLM(hidden_dim, cond_dim)


class _GFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, target_onehot, mask, noise, tau, hard, *params):
        cont, logits, samples, saved = net._run_forward(x, target_onehot, mask, noise, tau, hard)
        ctx.net, ctx.saved = net, saved
        ctx.set_materialize_grads(False)     # unused outputs (the logits, in the training step) arrive as None, not zeros
        return cont, logits, samples

    @staticmethod
    def backward(ctx, d_cont, d_logits, d_samples):
        ctx.net._run_backward(ctx.saved, d_cont, d_logits, d_samples)
        return (None,) * len(ctx.needs_input_grad)


class ResidualGenerator(FlatModule):
    """generator.py:38-92.  forward() returns what the reference returns — (cont_residual [B, n_cont], {idx: logits},
    {idx: Gumbel-softmax samples}) — the dict entries being column slices of the packed tensors `forward_packed` returns.

    Gumbel noise: `F.gumbel_softmax` draws it from torch's global generator (:90); here it is an explicit input —
    `gumbel=` (a packed [B, T] tensor or a dict idx -> [B, n]) or, when omitted, a draw from `self.rng` (ops.DeviceRNG)."""

    def __init__(self, input_dim, hidden_dim, num_classes, continuous_idx, categorical_info, n_blocks=5, residual_scaling=0.1, tau=0.5):
        super().__init__()
        self.input_dim = input_dim
        self.num_classes = num_classes
        self.continuous_idx = list(continuous_idx)
        self.categorical_info = dict(categorical_info)
        self.cond_dim = input_dim + num_classes
        self.fc_in = nn.Linear(input_dim + self.cond_dim, hidden_dim)
        self.blocks = nn.ModuleList([ResidualBlock(hidden_dim, self.cond_dim) for _ in range(n_blocks)])
        self.fc_cont = nn.Linear(hidden_dim, len(self.continuous_idx))
        self.fc_cat_logits = nn.ModuleDict({str(idx): nn.Linear(hidden_dim, _ncat(info)) for idx, info in self.categorical_info.items()})
        self.residual_scaling = residual_scaling
        self.tau = tau
        self.hidden_dim = hidden_dim
        self.cat_idx = [int(k) for k in self.fc_cat_logits.keys()]
        sizes = [self.fc_cat_logits[str(i)].out_features for i in self.cat_idx]
        self.seg = [0] + [int(v) for v in np.cumsum(sizes)]
        self.total_cat = int(self.seg[-1])
        self.rng = None
        self._idx_dev = None
        self.use_fused = True        # fused segment kernels (csrc/house_fused.hip) when the configuration allows it
        self._fdesc = None

    # -- fused path -------------------------------------------------------------------------------------------------------------
    def _fused_ok(self):
        bn = self.blocks[0].bn1 if len(self.blocks) else None
        return (self.use_fused and self.hidden_dim == 32 and len(self.blocks) == 5 and self.training and bn is not None
                and self.input_dim == 17 and self.num_classes == 4 and self.total_cat <= 96 and len(self.cat_idx) <= 8
                and self.total_cat + len(self.continuous_idx) <= 96 and all(b_ - a_ <= 32 for a_, b_ in zip(self.seg, self.seg[1:]))
                and len(self.continuous_idx) <= 32 and all(b_.bn1.momentum == bn.momentum and b_.bn2.eps == bn.eps for b_ in self.blocks))

    def _fused_desc(self):
        """Element offsets of every parameter in the flat buffer (rebuilt when the module is re-flattened)."""
        from ._lib import HouseGDesc
        key = self._flat.data_ptr()
        if self._fdesc is not None and self._fdesc[0] == key:
            return self._fdesc[1]
        off = {id(p): o for p, o, _ in self._seg}
        d = HouseGDesc()
        d.fc_in_w, d.fc_in_b = off[id(self.fc_in.weight)], off[id(self.fc_in.bias)]
        for k, blk in enumerate(self.blocks):
            d.fc1_w[k], d.fc1_b[k] = off[id(blk.fc1.weight)], off[id(blk.fc1.bias)]
            d.bn1_g[k], d.bn1_b[k] = off[id(blk.bn1.weight)], off[id(blk.bn1.bias)]
            d.fc2_w[k], d.fc2_b[k] = off[id(blk.fc2.weight)], off[id(blk.fc2.bias)]
            d.bn2_g[k], d.bn2_b[k] = off[id(blk.bn2.weight)], off[id(blk.bn2.bias)]
            d.film_gamma_w[k], d.film_gamma_b[k] = off[id(blk.film.gamma.weight)], off[id(blk.film.gamma.bias)]
            d.film_beta_w[k], d.film_beta_b[k] = off[id(blk.film.beta.weight)], off[id(blk.film.beta.bias)]
        d.cont_w, d.cont_b = off[id(self.fc_cont.weight)], off[id(self.fc_cont.bias)]
        for s_, f in enumerate(self.cat_idx):
            head = self.fc_cat_logits[str(f)]
            d.head_w[s_], d.head_b[s_] = off[id(head.weight)], off[id(head.bias)]
        for s_, v in enumerate(self.seg):
            d.seg[s_] = v
        d.nheads, d.ncont, d.D, d.NC = len(self.cat_idx), len(self.continuous_idx), self.input_dim, self.num_classes
        d.hidden, d.nblocks = self.hidden_dim, len(self.blocks)
        self._fdesc = (key, d)
        return d

    def _fused_forward(self, x, target_onehot, mask, noise, tau, hard):
        import ctypes
        from ._lib import HouseGFwdArgs, load
        dev = x.device
        B, T, nc = x.shape[0], self.total_cat, len(self.continuous_idx)
        x, target_onehot, mask, noise = x.contiguous(), target_onehot.contiguous(), mask.contiguous(), noise.contiguous()
        nb = (B + 63) // 64          # PCG_HOUSE_ROWS_PER_BLOCK: one 64-row wave per block
        f32 = dict(dtype=torch.float32, device=dev)
        K = 2 * self.input_dim + self.num_classes
        buf = {"inp": torch.empty((B, K), **f32), "H": torch.empty((6, B, 32), **f32), "Z1": torch.empty((5, B, 32), **f32),
               "Z2": torch.empty((5, B, 32), **f32), "P": torch.empty((10, nb, 2, 32), **f32), "SM": torch.empty((10, 2, 32), **f32),
               "cont": torch.empty((B, nc), **f32), "logits": torch.empty((B, T), **f32), "soft": torch.empty((B, T), **f32),
               "hard": torch.empty((B, T), **f32) if hard else None}
        a = HouseGFwdArgs()
        a.params = self._flat.data_ptr()
        a.x, a.onehot, a.mask, a.noise = x.data_ptr(), target_onehot.data_ptr(), mask.data_ptr(), noise.data_ptr()
        for n in ("inp", "H", "Z1", "Z2", "P", "SM", "cont", "logits", "soft"):
            setattr(a, n, buf[n].data_ptr())
        a.hard = buf["hard"].data_ptr() if hard else None
        for k, blk in enumerate(self.blocks):
            for j, bn in ((2 * k, blk.bn1), (2 * k + 1, blk.bn2)):
                a.running_mean[j], a.running_var[j] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                a.num_batches_tracked[j] = bn.num_batches_tracked.data_ptr()
        bn0 = self.blocks[0].bn1
        a.B, a.eps, a.momentum, a.tau, a.res_scale = B, bn0.eps, bn0.momentum, tau, self.residual_scaling
        ops.check(load().pcg_house_g_fwd(ctypes.byref(self._fused_desc()), ctypes.byref(a), ops._stream()), "pcg_house_g_fwd")
        saved = ("fused", buf, target_onehot, mask, tau, nb)
        return buf["cont"], buf["logits"], (buf["hard"] if hard else buf["soft"]), saved

    def _fused_backward(self, saved, d_cont, d_logits, d_samples):
        import ctypes
        from ._lib import HouseGBwdArgs, load
        _, buf, onehot, mask, tau, nb = saved
        B, T, nc = onehot.shape[0], self.total_cat, len(self.continuous_idx)
        f32 = dict(dtype=torch.float32, device=onehot.device)
        g = {n: torch.empty((5, B, 32), **f32) for n in ("DH", "DZ1", "DZ2", "A1", "DG", "DB")}
        g.update({"DN1": torch.empty((B, 32), **f32), "DZIN": torch.empty((B, 32), **f32), "DL": torch.empty((B, T), **f32),
                  "DC": torch.empty((B, nc), **f32), "Q": torch.empty((9 * nb + (B + 15) // 16, 2, 32), **f32)})
        # BatchNorm gamma / beta gradients are written by the kernels; all of them share the accumulate state of the net
        _, acc = self._grad_view(self.blocks[0].bn1.weight)
        for blk in self.blocks:
            for p_ in (blk.bn1.weight, blk.bn1.bias, blk.bn2.weight, blk.bn2.bias):
                self._grad_view(p_)
        a = HouseGBwdArgs()
        a.params, a.grads = self._flat.data_ptr(), self._gflat.data_ptr()
        a.onehot, a.mask = onehot.data_ptr(), mask.data_ptr()
        for n in ("H", "Z1", "Z2", "SM", "soft"):
            setattr(a, n, buf[n].data_ptr())
        keep = [t.contiguous() if t is not None else None for t in (d_cont, d_logits, d_samples)]
        a.d_cont, a.d_logits, a.d_samples = (t.data_ptr() if t is not None else None for t in keep)
        for n, t in g.items():
            setattr(a, n, t.data_ptr())
        a.B, a.accumulate, a.tau, a.res_scale = B, int(acc), tau, self.residual_scaling
        ops.check(load().pcg_house_g_bwd(ctypes.byref(self._fused_desc()), ctypes.byref(a), ops._stream()), "pcg_house_g_bwd")
        # weight (+ bias) gradients of all 35 Linear layers: ONE launch (deterministic slab reduction per layer)
        inp, K = buf["inp"], buf["inp"].shape[1]
        cond = inp[:, self.input_dim:]
        h_last = buf["H"][5]
        items = []

        def add(lin, x, dy, ldx=None, ldy=None):
            gw, aw = self._grad_view(lin.weight)
            gb, ab = self._grad_view(lin.bias)
            items.append((dy, x, lin.out_features, lin.in_features, gw, gb, ldy if ldy is not None else lin.out_features,
                          ldx if ldx is not None else lin.in_features, aw, ab))
        add(self.fc_in, inp, g["DZIN"])
        for k, blk in enumerate(self.blocks):
            add(blk.fc1, buf["H"][k], g["DZ1"][k])
            add(blk.fc2, g["A1"][k], g["DZ2"][k])
            add(blk.film.gamma, cond, g["DG"][k], ldx=K)
            add(blk.film.beta, cond, g["DB"][k], ldx=K)
        add(self.fc_cont, h_last, g["DC"])
        for s_, f in enumerate(self.cat_idx):
            add(self.fc_cat_logits[str(f)], h_last, g["DL"][:, self.seg[s_]:], ldy=T)
        ops.linear_wgrad_grouped(items, B, onehot.device)

    # -- small device-side index tables (seg offsets, column indices) -------------------------------------------------
    def index_tables(self, device):
        if self._idx_dev is None or self._idx_dev[0].device != device:
            mk = lambda v: torch.tensor(list(v), dtype=torch.int32, device=device)  # noqa: E731
            self._idx_dev = (mk(self.seg), mk(self.cat_idx), mk(self.continuous_idx))
        return self._idx_dev

    def col_src(self):
        """Per feature column, where residual_full takes it from: the index of a continuous output (>= 0) or -(head + 1)."""
        src = [None] * self.input_dim
        for j, c in enumerate(self.continuous_idx):
            src[c] = j
        for s_, c in enumerate(self.cat_idx):
            src[c] = -(s_ + 1)
        if any(v is None for v in src):
            raise PcgError("ResidualGenerator: continuous_idx and categorical_info must cover every feature column")
        return src

    def pack_noise(self, gumbel):
        """dict idx -> [B, n]  ->  packed [B, T] (setup-time helper for tests / supplied draws)."""
        return torch.cat([gumbel[i] for i in self.cat_idx], dim=1).contiguous()

    def _noise(self, gumbel, B, device):
        if gumbel is None:
            if self.rng is None:
                self.rng = ops.DeviceRNG(seed=0)
            return self.rng.gumbel((B, self.total_cat), device)
        if isinstance(gumbel, dict):
            gumbel = self.pack_noise(gumbel)
        if tuple(gumbel.shape) != (B, self.total_cat):
            raise PcgError(f"gumbel noise must be [B, {self.total_cat}], got {tuple(gumbel.shape)}")
        return gumbel

    def forward_packed(self, x, target_onehot, mask=None, temperature=None, hard=False, gumbel=None):
        self._ensure_flat()
        if not x.is_cuda:
            raise PcgError(f"ResidualGenerator: input is on {x.device}; libpcgan_hip has no CPU path")
        B = x.shape[0]
        if mask is None:
            mask = torch.empty_like(x)
            ops.fill(mask, 1.0)                                                               # :70-71
        tau = self.tau if temperature is None else float(temperature)
        noise = self._noise(gumbel, B, x.device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _GFn.apply(self, x, target_onehot, mask, noise, tau, bool(hard), *self.parameters())
        return self._run_forward(x, target_onehot, mask, noise, tau, bool(hard), keep=False)[:3]

    def forward(self, x, target_onehot, mask=None, temperature=None, hard=False, gumbel=None):
        cont, logits, samples = self.forward_packed(x, target_onehot, mask, temperature, hard, gumbel)
        cat_logits = {f: logits[:, self.seg[s]:self.seg[s + 1]] for s, f in enumerate(self.cat_idx)}
        cat_samples = {f: samples[:, self.seg[s]:self.seg[s + 1]] for s, f in enumerate(self.cat_idx)}
        return cont, cat_logits, cat_samples

    # -- one block ----------------------------------------------------------------------------------------------------
    def _bn(self, bn, z):
        C = bn.num_features
        if bn.training:
            mean, invstd = ops.bn_train_stats(z, C, bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked)
            return ops.bn_apply_act(z, C, mean, invstd, bn.weight.data, bn.bias.data, ACT_NONE), mean, invstd
        return ops.bn_apply_act(z, C, bn.running_mean, bn.running_var, bn.weight.data, bn.bias.data, ACT_NONE, var_eps=bn.eps), None, None

    def _run_forward(self, x, target_onehot, mask, noise, tau, hard, keep=True):
        if keep and self._fused_ok():
            return self._fused_forward(x, target_onehot, mask, noise, tau, hard)
        x = x.contiguous()
        seg, cat_idx, cont_idx = self.index_tables(x.device)
        B = x.shape[0]
        cond = ops.concat_cols(target_onehot.contiguous(), mask.contiguous())                # :73
        inp = ops.concat_cols(x, cond)                                                        # :74
        h = _lin_fwd(self.fc_in, inp)
        ops.act_fwd(h, ACT_RELU, 0.0, out=h)                                                  # :75
        blocks = []
        for blk in self.blocks:
            gam = _lin_fwd(blk.film.gamma, cond)                                              # FiLM :13-14 (one module, used twice)
            bet = _lin_fwd(blk.film.beta, cond)
            z1 = _lin_fwd(blk.fc1, h)                                                         # :28
            n1, m1, s1 = self._bn(blk.bn1, z1)
            a1 = ops.film_fwd(gam, n1, bet)                                                   # :29
            ops.act_fwd(a1, ACT_RELU, 0.0, out=a1)                                            # :30
            z2 = _lin_fwd(blk.fc2, a1)                                                        # :31
            n2, m2, s2 = self._bn(blk.bn2, z2)
            f2 = ops.film_fwd(gam, n2, bet)                                                   # :33
            hn = ops.axpby(1.0, h, 1.0, f2, out=f2)                                           # :34
            if keep:
                blocks.append((h, gam, z1, n1, m1, s1, a1, z2, n2, m2, s2))
            h = hn
        cont_raw = _lin_fwd(self.fc_cont, h)
        cont = ops.axpby(self.residual_scaling, cont_raw, out=cont_raw)                       # :81
        logits = torch.empty((B, self.total_cat), dtype=torch.float32, device=x.device)
        for s, f in enumerate(self.cat_idx):
            _lin_fwd(self.fc_cat_logits[str(f)], h, out=logits[:, self.seg[s]:], ldc=self.total_cat)   # :87
        soft, hard_y = ops.gumbel_softmax_fwd(logits, noise, seg, tau, hard=hard)             # :90
        saved = (cond, inp, blocks, h, soft, seg, tau) if keep else None
        return cont, logits, (hard_y if hard else soft), saved

    def _run_backward(self, saved, d_cont, d_logits, d_samples):
        if saved[0] == "fused":
            return self._fused_backward(saved, d_cont, d_logits, d_samples)
        cond, inp, blocks, h_last, soft, seg, tau = saved
        if any(not blk.bn1.training for blk in self.blocks):
            raise PcgError("backward through an eval-mode BatchNorm1d is not implemented")
        B = inp.shape[0]
        T = self.total_cat
        dh = None
        if d_samples is not None:
            dl = ops.gumbel_softmax_bwd(d_samples.contiguous(), soft, seg, tau)               # straight-through for hard=True
            if d_logits is not None:
                ops.axpby(1.0, dl, 1.0, d_logits.contiguous(), out=dl)
        else:
            dl = d_logits.contiguous() if d_logits is not None else None
        if dl is not None:
            dh = torch.empty((B, self.hidden_dim), dtype=torch.float32, device=inp.device)
            for s, f in enumerate(self.cat_idx):
                head = self.fc_cat_logits[str(f)]
                dls = dl[:, self.seg[s]:]
                _lin_wgrad(self, head, h_last, dls, ldy=T)
                _lin_dgrad(head.weight.data, dls, B, ldy=T, out=dh, accumulate=s > 0)
        if d_cont is not None:
            dc = ops.axpby(self.residual_scaling, d_cont.contiguous())
            _lin_wgrad(self, self.fc_cont, h_last, dc)
            dh = _lin_dgrad(self.fc_cont.weight.data, dc, B, out=dh, accumulate=dh is not None)
        if dh is None:
            return
        C = self.hidden_dim
        for blk, (h, gam, z1, n1, m1, s1, a1, z2, n2, m2, s2) in zip(reversed(self.blocks), reversed(blocks)):
            dgam2, dn2 = ops.film_bwd(dh, gam, n2)                                            # d f2 = dh (skip keeps dh too)
            gg, acc = self._grad_view(blk.bn2.weight)
            gb, _ = self._grad_view(blk.bn2.bias)
            dz2 = ops.bn_act_bwd(dn2, z2, None, C, m2, s2, blk.bn2.weight.data, ACT_NONE, 0.0, gg, gb, acc)
            _lin_wgrad(self, blk.fc2, a1, dz2)
            da1 = _lin_dgrad(blk.fc2.weight.data, dz2, B)
            ops.act_bwd(da1, a1, ACT_RELU, 0.0, out=da1)                                      # d f1
            dgam1, dn1 = ops.film_bwd(da1, gam, n1)
            gg, acc = self._grad_view(blk.bn1.weight)
            gb, _ = self._grad_view(blk.bn1.bias)
            dz1 = ops.bn_act_bwd(dn1, z1, None, C, m1, s1, blk.bn1.weight.data, ACT_NONE, 0.0, gg, gb, acc)
            _lin_wgrad(self, blk.fc1, h, dz1)
            # FiLM parameters: gamma sees dgam1 + dgam2, beta sees d f1 + d f2
            dgam = ops.axpby(1.0, dgam1, 1.0, dgam2, out=dgam1)
            dbet = ops.axpby(1.0, da1, 1.0, dh, out=dgam2)
            _lin_wgrad(self, blk.film.gamma, cond, dgam)
            _lin_wgrad(self, blk.film.beta, cond, dbet)
            dh = _lin_dgrad(blk.fc1.weight.data, dz1, B, out=dh, accumulate=True)             # block path + skip path
        ops.act_bwd(dh, blocks[0][0] if blocks else h_last, ACT_RELU, 0.0, out=dh)
        _lin_wgrad(self, self.fc_in, inp, dh)


# ---- residual assembly (trainer.py:266-279) -----------------------------------------------------------------------------
class _AssembleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cont, samples, x, tables, norm_vals):
        seg, cat_idx, cont_idx = tables
        ctx.tables, ctx.norm, ctx.shape = tables, norm_vals, (cont.shape[1], samples.shape[1])
        return ops.assemble_residual_fwd(cont.contiguous(), cont_idx, samples.contiguous(), seg, cat_idx, norm_vals, x.contiguous())

    @staticmethod
    def backward(ctx, dres):
        seg, cat_idx, cont_idx = ctx.tables
        dcont, dsamples = ops.assemble_residual_bwd(dres.contiguous(), ctx.shape[0], cont_idx, seg, ctx.shape[1], cat_idx, ctx.norm)
        return dcont, dsamples, None, None, None


def assemble_residual(generator, cont, samples, x, norm_vals):
    """residual_full [B, D]: continuous columns from `cont`, categorical columns = E_sample[normalised value] - x."""
    return _AssembleFn.apply(cont, samples, x, generator.index_tables(x.device), norm_vals)


def cat_norm_maps(generator, config, device):
    """trainer.py:205-223 — packed [T] tensor of normalised category values, head order of the generator."""
    scaler = config.get("scaler", None)
    vals = []
    for f in generator.cat_idx:
        info = config["categorical_info"][f]
        n = _ncat(info)
        if scaler is not None:
            lo, hi = float(scaler.data_min_[f]), float(scaler.data_max_[f])
            raw = np.asarray(info["raw_values"], dtype=float)
            vals.append((raw - lo) / ((hi - lo) + 1e-12))                                      # :211-216
        else:
            vals.append(np.arange(n, dtype=float) / max(1.0, n - 1))                           # :218-223
    return torch.tensor(np.concatenate(vals), dtype=torch.float32, device=device)


class _MaskMulFn(torch.autograd.Function):
    """masked = res * mask; x_cf = x + masked (trainer.py:281-282)."""

    @staticmethod
    def forward(ctx, res, mask, x):
        ctx.mask = mask
        ctx.set_materialize_grads(False)
        _, masked = ops.scale_mask_fwd(res.contiguous(), mask, 1.0)
        x_cf = ops.axpby(1.0, x, 1.0, masked)
        return masked, x_cf

    @staticmethod
    def backward(ctx, d_masked, d_xcf):
        if d_masked is None:
            dsum = d_xcf.contiguous()
        elif d_xcf is None:
            dsum = d_masked.contiguous()
        else:
            dsum = ops.axpby(1.0, d_masked.contiguous(), 1.0, d_xcf.contiguous())
        return ops.scale_mask_bwd(None, dsum, ctx.mask, 1.0, like=ctx.mask), None, None


_ones_cache = {}


def _one(device):
    """A cached scalar 1.0 on the device: `loss.backward(gradient=_one(dev))` spares autograd its ones_like fill launch."""
    t = _ones_cache.get(device)
    if t is None:
        t = ops.fill(torch.empty((), dtype=torch.float32, device=device), 1.0)
        _ones_cache[device] = t
    return t


# ---- discriminator ------------------------------------------------------------------------------------------------------
class _DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, target_onehot, *params):
        out, saved = net._run_forward(x, target_onehot)
        ctx.net, ctx.saved = net, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        dx = ctx.net._run_backward(ctx.saved, dout, ctx.needs_input_grad[1], any(ctx.needs_input_grad[3:]))
        return (None, dx) + (None,) * (len(ctx.needs_input_grad) - 2)


class Discriminator(FlatModule):
    """discriminator.py:5-20 — a Wasserstein-style critic of four spectral-normalised Linears.  The torch spectral_norm
    containers hold the state (`net.{0,2,4,6}.weight_orig / weight_u / weight_v / bias`); the power iteration, the
    division by sigma and its backward run in pcg_spectral_norm_fwd/bwd.  As in torch, every training-mode forward does
    one power iteration and updates u, v in place."""

    def __init__(self, input_dim, hidden_dim, num_classes):
        super().__init__()
        self.net = nn.Sequential(
            spectral_norm(nn.Linear(input_dim + num_classes, hidden_dim)), nn.LeakyReLU(0.2, inplace=True),
            spectral_norm(nn.Linear(hidden_dim, hidden_dim * 2)), nn.LeakyReLU(0.2, inplace=True),
            spectral_norm(nn.Linear(hidden_dim * 2, hidden_dim * 4)), nn.LeakyReLU(0.2, inplace=True),
            spectral_norm(nn.Linear(hidden_dim * 4, 1)))
        self.input_dim = input_dim

    def _linears(self):
        return [m for m in self.net if isinstance(m, nn.Linear)]

    def forward(self, x, target_onehot):
        self._ensure_flat()
        if not x.is_cuda:
            raise PcgError(f"Discriminator: input is on {x.device}; libpcgan_hip has no CPU path")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _DFn.apply(self, x, target_onehot, *self.parameters())
        return self._run_forward(x, target_onehot, keep=False)[0]

    def _fused_ok(self, x, target_onehot):
        lins = self._linears()
        return (getattr(self, "use_fused", True) and [l.out_features for l in lins] == [32, 64, 128, 1] and lins[0].in_features == 21
                and x.shape[1] + target_onehot.shape[1] == 21)

    def _spectral_norm(self):
        """Power iteration (training mode: updates u, v in place) + W / sigma of all four layers: one launch."""
        lins = self._linears()
        return ops.spectral_norm_fwd_batched([l.weight_orig.data for l in lins], [l.weight_u for l in lins], [l.weight_v for l in lins],
                                             1e-12, self.training)

    def _run_forward(self, x, target_onehot, keep=True, sn=None):
        lins = self._linears()
        if sn is None:
            sn = self._spectral_norm()
        if self._fused_ok(x, target_onehot):
            # the four layers as ONE launch, one thread per row (csrc/house_critic_fused.hip)
            import ctypes
            from ._lib import load
            x, target_onehot = x.contiguous(), target_onehot.contiguous()
            B = x.shape[0]
            f32 = dict(dtype=torch.float32, device=x.device)
            acts = [torch.empty((B, n), **f32) for n in (21, 32, 64, 128)]
            out = torch.empty((B, 1), **f32)
            wb = [t[0] for t in sn]
            ops.check(load().pcg_house_critic_fwd(ops._p(x), ops._p(target_onehot), B, x.shape[1], target_onehot.shape[1], ops._ptr_array(wb),
                                                  ops._ptr_array([l.bias.data for l in lins]), 0.2, ops._p(acts[0]), ops._p(acts[1]),
                                                  ops._p(acts[2]), ops._p(acts[3]), ops._p(out), ops._stream()), "pcg_house_critic_fwd")
            return out, (("fused", acts, sn) if keep else None)
        a = ops.concat_cols(x.contiguous(), target_onehot.contiguous())                       # :19
        layers = []
        for i, lin in enumerate(lins):
            w_bar, sigma, u, v = sn[i]
            z = _lin_fwd(lin, a, weight=w_bar, act=ACT_LRELU if i + 1 < len(lins) else ACT_NONE, slope=0.2)   # LeakyReLU fused
            if keep:
                layers.append((a, z, w_bar, sigma, u, v))
            a = z
        return a, (layers if keep else None)

    def _sn_operands(self):
        """(w_origs, us, vs, eps) of the spectrally normalised Linears: what a power iteration of this critic reads and updates."""
        lins = self._linears()
        return [l.weight_orig.data for l in lins], [l.weight_u for l in lins], [l.weight_v for l in lins], 1e-12

    def _run_pair(self, x_a, onehot_a, cot_a, x_b, onehot_b, cot_b, sn=None, defer_sn_bwd=False):
        """The critic step's two passes as one: D(a) then D(b) — two successive power iterations, as two forward calls make them —
        and the backward of cot_b . D(b) followed by that of cot_a . D(a) into the gradient buffer (the order the chained calls
        of train_step use), in FIVE launches instead of ten (four when the caller made the power iterations, three when it also
        takes the last one as a rider): both power iterations, both forwards, both backwards, all sixteen
        weight / bias reductions, both passes through W / sigma.  Per element the arithmetic of the chained calls (same bits).
        Returns (D(a), D(b)).  `sn`: the result of the two power iterations when the caller has already launched them;
        defer_sn_bwd: leave the last launch (the backward through W / sigma) to the caller, as a rider."""
        import ctypes
        from ._lib import load
        lins = self._linears()
        # sn: the two power iterations already made (they rode with the residual block's launch, ops.house_residual_fwd(sn=...))
        sn_a, sn_b = sn if sn is not None else ops.spectral_norm_fwd_batched_reps(*self._sn_operands(), 2)
        B = x_a.shape[0]
        dev = x_a.device
        f32 = dict(dtype=torch.float32, device=dev)
        ins = [(x_a.contiguous(), onehot_a.contiguous()), (x_b.contiguous(), onehot_b.contiguous())]
        acts = [[torch.empty((B, n), **f32) for n in (21, 32, 64, 128)] for _ in range(2)]
        outs = [torch.empty((B, 1), **f32) for _ in range(2)]
        wb = [t[0] for t in sn_a] + [t[0] for t in sn_b]
        pa = ops._ptr_array
        ops.check(load().pcg_house_critic_fwd_n(2, pa([i[0] for i in ins]), pa([i[1] for i in ins]), B, x_a.shape[1], onehot_a.shape[1], pa(wb),
                                                pa([l.bias.data for l in lins]), 0.2, pa([a[0] for a in acts]), pa([a[1] for a in acts]),
                                                pa([a[2] for a in acts]), pa([a[3] for a in acts]), pa(outs), ops._stream()),
                  "pcg_house_critic_fwd_n")
        d4 = [cot_a.contiguous(), cot_b.contiguous()]
        d3, d2, d1 = ([torch.empty((B, n), **f32) for _ in range(2)] for n in (128, 64, 32))
        null2 = (ctypes.c_void_p * 2)(None, None)
        ops.check(load().pcg_house_critic_bwd_n(2, pa(d4), B, self.input_dim, pa(wb), 0.2, pa([a[1] for a in acts]), pa([a[2] for a in acts]),
                                                pa([a[3] for a in acts]), pa(d3), pa(d2), pa(d1), null2, ops._stream()), "pcg_house_critic_bwd_n")
        # pass b is the first writer of the gradient buffer, pass a adds (bias: through its own buffer, added behind the b pass)
        items, seq_b, seq_a, dws, accs, bias_adds = [], [], [], [], [], []
        for li, lin in enumerate(lins):
            gb, accb = self._grad_view(lin.bias)
            gw, acc = self._grad_view(lin.weight_orig)
            dw_b, dw_a = torch.empty_like(sn_b[li][0]), torch.empty_like(sn_a[li][0])
            gb_a = torch.empty_like(gb)
            O, I = lin.out_features, lin.in_features
            items.append(((d1, d2, d3, d4)[li][1], acts[1][li], O, I, dw_b, gb, O, I, False, accb))
            items.append(((d1, d2, d3, d4)[li][0], acts[0][li], O, I, dw_a, gb_a, O, I, False, False))
            seq_b.append((dw_b, sn_b[li][0], sn_b[li][2], sn_b[li][3], sn_b[li][1]))
            seq_a.append((dw_a, sn_a[li][0], sn_a[li][2], sn_a[li][3], sn_a[li][1]))
            dws.append(gw); accs.append(acc); bias_adds.append((gb, gb_a))
        ops.linear_wgrad_grouped(items, B, dev)
        if defer_sn_bwd:      # the caller launches the backward through W / sigma as a rider: (D(a), D(b), its arguments, what they point to)
            return outs[0], outs[1], ops.sn_bwd_seq_args([seq_b, seq_a], dws, accs, bias_adds), (seq_b, seq_a, dws, bias_adds)
        ops.spectral_norm_bwd_batched_seq([seq_b, seq_a], dws, accs, bias_adds)
        return outs[0], outs[1]

    def _fused_backward(self, saved, dout, need_x, need_p, gtarget=None):
        """gtarget: a flat buffer with the layout of flat_grads that receives this pass's parameter gradients (written, not
        accumulated) instead of the module's gradient buffer — see FlatModule.grad_view_in."""
        from ._lib import load
        _, acts, sn = saved
        lins = self._linears()
        d4 = dout.contiguous()
        B = d4.shape[0]
        f32 = dict(dtype=torch.float32, device=d4.device)
        d3, d2, d1 = (torch.empty((B, n), **f32) for n in (128, 64, 32))
        dx = torch.empty((B, self.input_dim), **f32) if need_x else None
        ops.check(load().pcg_house_critic_bwd(ops._p(d4), B, self.input_dim, ops._ptr_array([t[0] for t in sn]), 0.2, ops._p(acts[1]),
                                              ops._p(acts[2]), ops._p(acts[3]), ops._p(d3), ops._p(d2), ops._p(d1), ops._p(dx),
                                              ops._stream()), "pcg_house_critic_bwd")
        if need_p and lins[0].weight_orig.requires_grad:
            items, sn_items = [], []
            for lin, a, d, (w_bar, sigma, u, v) in zip(lins, acts, (d1, d2, d3, d4), sn):
                dwb = torch.empty_like(w_bar)
                if gtarget is None:
                    gb, accb = self._grad_view(lin.bias)
                    gw, acc = self._grad_view(lin.weight_orig)
                else:
                    gb, accb = self.grad_view_in(gtarget, lin.bias), False
                    gw, acc = self.grad_view_in(gtarget, lin.weight_orig), False
                items.append((d, a, lin.out_features, lin.in_features, dwb, gb, lin.out_features, lin.in_features, False, accb))
                sn_items.append((dwb, w_bar, u, v, sigma, gw, acc))
            ops.linear_wgrad_grouped(items, B, d4.device)                                    # all weight + bias gradients: one launch
            ops.spectral_norm_bwd_batched(sn_items)                                          # through W / sigma: one launch
        return dx

    def _run_backward(self, layers, dout, need_x, need_p, gtarget=None):
        if layers[0] == "fused":
            return self._fused_backward(layers, dout, need_x, need_p, gtarget)
        if gtarget is not None:
            raise PcgError("Discriminator: a separate gradient target needs the fused critic kernels")
        lins = self._linears()
        d = dout.contiguous()
        B = d.shape[0]
        sn_items = []
        for i in range(len(lins) - 1, -1, -1):
            lin = lins[i]
            a, z, w_bar, sigma, u, v = layers[i]
            if i + 1 < len(lins):
                d = ops.act_bwd(d, z, ACT_LRELU, 0.2)
            if need_p and lin.weight_orig.requires_grad:
                dwb = torch.empty_like(w_bar)
                _lin_wgrad(self, lin, a, d, dw_out=dwb, weight_param=lin.weight_orig)
                gw, acc = self._grad_view(lin.weight_orig)
                sn_items.append((dwb, w_bar, u, v, sigma, gw, acc))
            if i > 0 or need_x:
                d = _lin_dgrad(w_bar, d, B)
        if sn_items:
            ops.spectral_norm_bwd_batched(sn_items)                                          # all layers: one launch
        if not need_x:
            return None
        dx, _ = ops.split_cols(d, self.input_dim, d.shape[1] - self.input_dim, need_b=False)
        return dx


# ---- frozen classifier ------------------------------------------------------------------------------------------------
class _CFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x):
        logits, saved = net._run_forward(x)
        ctx.net, ctx.saved = net, saved
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        return None, ctx.net._run_backward(ctx.saved, dlogits)


class _CTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, masks, *params):
        logits, saved = net._train_forward(x, masks)
        ctx.net, ctx.saved = net, saved
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        ctx.net._train_backward(ctx.saved, dlogits)
        return (None,) * len(ctx.needs_input_grad)


class NNClassifier(FlatModule):
    """nn_classifier.py:4-32.  In the GAN step: eval mode, parameters frozen (main.py:27-30) — forward and the gradient with
    respect to the input row; each BatchNorm1d is then a per-column affine map folded into the following Linear once (fp64 at
    pack time): 5 GEMMs and 4 LeakyReLUs.  In training mode (pre-training, trainer.py:18-180 — SURVEY.md section 8f item 3):
    Linear -> LeakyReLU(0.1) -> BatchNorm1d (batch statistics) -> Dropout, with the Dropout masks from `self.rng` (device
    Philox stream) or `self.dropout_masks = [m0 [B,256], m1 [B,256], m2 [B,128]]` for runs that must reproduce given draws."""

    def __init__(self, input_dim, output_dim=4):
        super().__init__()
        self.rng = None
        self.dropout_masks = None
        self.net = nn.Sequential(
            nn.Linear(input_dim, 256), nn.LeakyReLU(0.1), nn.BatchNorm1d(256), nn.Dropout(0.3),
            nn.Linear(256, 256), nn.LeakyReLU(0.1), nn.BatchNorm1d(256), nn.Dropout(0.2),
            nn.Linear(256, 128), nn.LeakyReLU(0.1), nn.BatchNorm1d(128), nn.Dropout(0.1),
            nn.Linear(128, 64), nn.LeakyReLU(0.1), nn.BatchNorm1d(64),
            nn.Linear(64, output_dim))
        self._packed = None

    def _pack(self):
        key = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        if self.training:
            raise PcgError("NNClassifier: only eval mode is implemented (the GAN step uses the frozen classifier)")
        packed, scale, shift = [], None, None
        with torch.no_grad():
            for m in self.net:
                if isinstance(m, nn.Linear):
                    w, b = m.weight.double(), m.bias.double()
                    if scale is not None:            # Linear(bn(a)) = a (W diag(s))^T + (W t + b)
                        b = b + w @ shift
                        w = w * scale[None, :]
                        scale = shift = None
                    packed.append((w.float().contiguous(), b.float().contiguous()))
                elif isinstance(m, nn.BatchNorm1d):
                    s = m.weight.double() / torch.sqrt(m.running_var.double() + m.eps)
                    scale, shift = s, m.bias.double() - m.running_mean.double() * s
        self._packed = (key, packed)
        self._kmajor = None
        return packed

    def _fused_ok(self, x):
        dims = [m.in_features for m in self.net if isinstance(m, nn.Linear)] + [self.net[-1].out_features]
        return getattr(self, "use_fused", True) and x.is_cuda and x.shape[1] == 17 and dims == [17, 256, 256, 128, 64, 4]

    def _pack_kmajor(self):
        """Per layer the folded weight transposed ([K][N]: the forward's B operand is then a coalesced read), layer 0 padded with a
        zero rows to a multiple of the MFMA's reduction depth (17 -> 20); the last layer stays as stored.  Built once per pack."""
        packed = self._pack()
        if getattr(self, "_kmajor", None) is None:
            with torch.no_grad():
                km = [w.t().contiguous() for w, _ in packed[:4]]
                km[0] = torch.cat([km[0], torch.zeros((3, km[0].shape[1]), dtype=km[0].dtype, device=km[0].device)], 0).contiguous()   # K 17 -> 20
                km.append(packed[4][0])
            self._kmajor = km
        return self._kmajor

    def train(self, mode=True):
        # the optimizer kernel updates parameters in place without touching torch's version counters: drop the packed eval
        # image whenever the mode changes so that eval after training sees the new weights
        self._packed = None
        return super().train(mode)

    def forward(self, x):
        if not x.is_cuda:
            raise PcgError(f"NNClassifier: input is on {x.device}; libpcgan_hip has no CPU path")
        if self.training:
            self._ensure_flat()
            masks = self.dropout_masks
            if masks is None:
                if self.rng is None:
                    self.rng = ops.DeviceRNG(seed=0)
                masks = [self.rng.bernoulli((x.shape[0], m_.out_features), x.device, 1.0 - p_) for m_, p_ in self._dropout_sites()]
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                return _CTrainFn.apply(self, x, masks, *self.parameters())
            return self._train_forward(x, masks)[0]
        if torch.is_grad_enabled() and x.requires_grad:
            return _CFn.apply(self, x)
        return self._run_forward(x, keep=False)[0]

    # -- training mode ------------------------------------------------------------------------------------------------------------
    def _stages(self):
        """[(Linear, BatchNorm1d, dropout p or None)] + the final Linear."""
        mods, stages, i = list(self.net), [], 0
        while i < len(mods) - 1:
            lin, bn = mods[i], mods[i + 2]
            p = mods[i + 3].p if i + 3 < len(mods) and isinstance(mods[i + 3], nn.Dropout) else None
            stages.append((lin, bn, p))
            i += 4 if p is not None else 3
        return stages, mods[-1]

    def _dropout_sites(self):
        return [(lin, p) for lin, _, p in self._stages()[0] if p is not None]

    def _train_forward(self, x, masks):
        stages, last = self._stages()
        a = x.contiguous()
        saved, mi = [], 0
        for lin, bn, p in stages:
            z = _lin_fwd(lin, a)
            ops.act_fwd(z, ACT_LRELU, 0.1, out=z)                                            # act output (BatchNorm input)
            C = bn.num_features
            mean, invstd = ops.bn_train_stats(z, C, bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked)
            n = ops.bn_apply_act(z, C, mean, invstd, bn.weight.data, bn.bias.data, ACT_NONE)
            mask = None
            if p is not None:
                mask = masks[mi].contiguous(); mi += 1
                n = ops.dropout_apply(n, mask, p, out=n)
            saved.append((lin, bn, p, a, z, mean, invstd, mask))
            a = n
        logits = _lin_fwd(last, a)
        return logits, (saved, last, a)

    def _train_backward(self, saved_all, dlogits):
        saved, last, a_last = saved_all
        B = a_last.shape[0]
        d = dlogits.contiguous()
        _lin_wgrad(self, last, a_last, d)
        d = _lin_dgrad(last.weight.data, d, B)
        for idx in range(len(saved) - 1, -1, -1):
            lin, bn, p, a, z, mean, invstd, mask = saved[idx]
            if p is not None:
                d = ops.dropout_apply(d, mask, p, out=d)
            gg, acc = self._grad_view(bn.weight)
            gb, _ = self._grad_view(bn.bias)
            d = ops.bn_act_bwd(d, z, None, bn.num_features, mean, invstd, bn.weight.data, ACT_NONE, 0.0, gg, gb, acc)
            ops.act_bwd(d, z, ACT_LRELU, 0.1, out=d)
            _lin_wgrad(self, lin, a, d)
            if idx > 0:
                d = _lin_dgrad(lin.weight.data, d, B)

    def _run_forward(self, x, keep=True, sn_bwd_rider=None, ce=None):
        """sn_bwd_rider: the argument list of a spectral-norm backward (ops.sn_bwd_seq_args) that rides in the fused forward launch
        (the scheduled tabular step: the critic's spectral-norm backward is independent of this classifier).  ce = (target, grad_scale)
        with a rider: the launch also leaves the cross-entropy's gradient and row terms; returns (logits, acts, dlogits, row_loss)."""
        packed = self._pack()
        a = x.contiguous()
        B = a.shape[0]
        if sn_bwd_rider is not None and not self._fused_ok(a):
            raise PcgError("NNClassifier: a rider needs the fused forward launch")
        if self._fused_ok(a):
            # the five layers as ONE launch on the matrix cores (csrc/house_classifier_fused.hip)
            from ._lib import load
            f32 = dict(dtype=torch.float32, device=a.device)
            acts = [torch.empty((B, n), **f32) for n in (256, 256, 128, 64)]
            logits = torch.empty((B, 4), **f32)
            cargs = (ops._p(a), B, ops._ptr_array(self._pack_kmajor()), ops._ptr_array([b for _, b in packed]), ops._p(acts[0]), ops._p(acts[1]),
                     ops._p(acts[2]), ops._p(acts[3]), ops._p(logits))
            if sn_bwd_rider is None:
                ops.check(load().pcg_house_classifier_fwd(*cargs, ops._stream()), "pcg_house_classifier_fwd")
            elif ce is None:
                ops.check(load().pcg_house_classifier_fwd_snbwd(*cargs, *sn_bwd_rider, None, 0.0, None, None, ops._stream()),
                          "pcg_house_classifier_fwd_snbwd")
            else:
                dlog, row_loss = torch.empty((B, 4), **f32), torch.empty((B,), **f32)
                ops.check(load().pcg_house_classifier_fwd_snbwd(*cargs, *sn_bwd_rider, ops._p(ce[0]), float(ce[1]), ops._p(dlog), ops._p(row_loss),
                                                                ops._stream()), "pcg_house_classifier_fwd_snbwd")
                return logits, (acts if keep else None), dlog, row_loss
            return logits, (acts if keep else None)
        acts = []
        for i, (w, b) in enumerate(packed):
            z = affine_fwd(a, w, b, act=ACT_LRELU if i + 1 < len(packed) else ACT_NONE, slope=0.1)
            if i + 1 < len(packed):
                if keep:
                    acts.append(z)
            a = z
        return a, (acts if keep else None)

    def _run_backward(self, acts, dlogits, sn_fwd_rider=None):
        """sn_fwd_rider = (w_origs, us, vs, eps, reps): that training-mode spectral normalisation rides in the fused backward launch;
        the return value is then (dx, its outputs)."""
        packed = self._pack()
        d = dlogits.contiguous()
        B = d.shape[0]
        fused = d.is_cuda and len(acts) == 4 and [t.shape[1] for t in acts] == [256, 256, 128, 64] and getattr(self, "use_fused", True) and d.shape[1] == 4
        if sn_fwd_rider is not None and not fused:
            raise PcgError("NNClassifier: a rider needs the fused backward launch")
        if fused:
            from ._lib import load
            dx = torch.empty((B, 17), dtype=torch.float32, device=d.device)
            cargs = (ops._p(d), B, ops._ptr_array([w for w, _ in packed]), ops._p(acts[0]), ops._p(acts[1]), ops._p(acts[2]), ops._p(acts[3]), ops._p(dx))
            if sn_fwd_rider is None:
                ops.check(load().pcg_house_classifier_bwd(*cargs, ops._stream()), "pcg_house_classifier_bwd")
                return dx
            outs, sn_args = ops.sn_fwd_reps_args(*sn_fwd_rider)
            ops.check(load().pcg_house_classifier_bwd_snfwd(*cargs, *sn_args, ops._stream()), "pcg_house_classifier_bwd_snfwd")
            return dx, outs
        for i in range(len(packed) - 1, -1, -1):
            if i + 1 < len(packed):
                d = ops.act_bwd(d, acts[i], ACT_LRELU, 0.1, out=d)
            d = _lin_dgrad(packed[i][0], d, B)
        return d


# ---- trainer ------------------------------------------------------------------------------------------------------------
def make_optimizers(generator, discriminator, config=CONFIG):
    """trainer.py:229-230."""
    return Adam(generator.parameters(), lr=config["lr_G"]), Adam(discriminator.parameters(), lr=config["lr_D"])


def train_step(generator, discriminator, classifier, opt_g, opt_d, x, y, target_y, mask, norm_vals, config=CONFIG, gumbel=None,
               ce=None, skip_dead_d_wgrad=True, branch=None, onehots=None, diag=None):
    """One iteration of train_countergan's loop body (trainer.py:241-316) for a batch already on the GPU.  The draws —
    `target_y` (:248-249), `mask` (:253-255) and the Gumbel noise inside G (gumbel=None: generator.rng) — are inputs.
    Returns device tensors; the reference's `.item()` calls are the caller's.

    branch: a second HIP stream.  The step is a chain of ~100 tiny kernels (64 blocks each on a 256-CU chip), i.e. bound by the
    length of its dependency chain; the frozen classifier's pass (trainer.py:301-302: forward, cross-entropy and — in backward —
    the gradient with respect to x_cf; ~25 % of the chain) depends only on x_cf, so it is forked onto `branch` as soon as x_cf
    exists and runs beside the whole critic update; it joins where g_loss is formed.  autograd runs a node's backward on the
    stream its forward ran on, so the classifier's backward overlaps the critic's in the G step too.  Same kernels, same inputs:
    results are bit-identical to the single-stream order; captured in a HIP graph the two streams become parallel branches.

    diag: {"logits_orig": the frozen classifier's logits of the original rows ([B, nc], or [N, nc] of the whole training set with
    "src_rows" [B]), "acc": float64[8] epoch accumulators or None, "eps": 1e-3} — the trainer's per-iteration diagnostics
    (trainer.py:318-343: pred_gain, sparsity, reg_loss_l2, class_flip_rate) as one fused device reduction, returned as out["diag"]
    ([4] device tensor) and summed into `acc` together with D_loss / G_loss (read once per epoch, no .item() per iteration)."""
    if branch is not None:
        return _train_step_branch(generator, discriminator, classifier, opt_g, opt_d, x, y, target_y, mask, norm_vals, config, gumbel,
                                  branch, skip_dead_d_wgrad, onehots, diag)
    nc = config["num_classes"]
    ce = ce if ce is not None else CrossEntropyLoss()
    target_onehot = ops.onehot(target_y, nc)                                                  # :250
    cont, _, samples = generator.forward_packed(x, target_onehot, mask, temperature=config["gumbel_tau"], hard=False,
                                                gumbel=gumbel)                                # :259-261
    residual_full = assemble_residual(generator, cont, samples, x, norm_vals)                 # :266-279
    masked_residual, x_cf = _MaskMulFn.apply(residual_full, mask, x)                          # :281-282
    mask_penalty_pre = abs_mean(residual_full, mask, one_minus=True)                          # :287
    # ---- D step
    d_real = discriminator(x, ops.onehot(y, nc))                                              # :290
    d_fake = discriminator(x_cf.detach(), target_onehot)                                      # :291
    d_loss = weighted_sum([mean(d_fake), mean(d_real)], [1.0, -1.0])                          # :292
    opt_d.zero_grad()
    d_loss.backward(gradient=_one(x.device))
    opt_d.step()                                                                              # :293-295
    # ---- G step
    if skip_dead_d_wgrad:            # the reference computes D's weight gradients here and never uses them (next zero_grad)
        for p in discriminator.parameters():
            p.requires_grad_(False)
    try:
        d_fake_for_g = discriminator(x_cf, target_onehot)                                     # :298
        m_fake = mean(d_fake_for_g)
        clf_preds = classifier(x_cf)                                                          # :301
        g_cls = ce(clf_preds, target_y)                                                       # :302
        am = abs_mean(masked_residual)                                                        # :305  mean_b ||.||_1 = D * mean|.|
        d_feat = float(x.shape[1])
        g_loss = weighted_sum([m_fake, g_cls, am, mask_penalty_pre],
                              [-1.0, config["lambda_cls"], config["lambda_reg"] * d_feat, config["lambda_mask"]])   # :299, :307-312
        with torch.no_grad():                                                                 # logged values
            g_adv = weighted_sum([m_fake], [-1.0])
            g_reg = weighted_sum([am], [d_feat])
        opt_g.zero_grad()
        g_loss.backward(gradient=_one(x.device))                                              # :314-315
    finally:
        if skip_dead_d_wgrad:
            for p in discriminator.parameters():
                p.requires_grad_(True)
    opt_g.step()                                                                              # :316
    out = {"D_loss": d_loss, "G_loss": g_loss, "g_adv": g_adv, "g_cls": g_cls, "reg": g_reg, "mask_pen": mask_penalty_pre,
           "D_real": d_real, "D_fake_forG": d_fake_for_g, "x_cf": x_cf, "masked_residual": masked_residual}
    if diag is not None:                                                                      # :318-343
        with torch.no_grad():
            out["diag"] = ops.house_diag(clf_preds.detach().contiguous(), diag["logits_orig"], target_y, masked_residual.detach().contiguous(),
                                         eps=diag.get("eps", 1e-3), src_rows=diag.get("src_rows"), acc=diag.get("acc"))
            _acc_losses(diag.get("acc"), d_loss, g_loss)
    return out


def _acc_losses(acc, d_loss, g_loss):
    """acc[0] += D_loss, acc[1] += G_loss, acc[6] += 1 (the schedule of train_step(branch=...) does this inside its scalars launch;
    the reference-order step with three tiny device adds — setup / test path, not the benched one)."""
    if acc is not None:
        acc[0] += d_loss.detach().double()
        acc[1] += g_loss.detach().double()
        acc[6] += 1.0


_cot_cache = {}
_alt_cache = {}


def _alt_grads(net):
    """A second, zero-initialised gradient buffer with the layout of net.flat_grads (kept per net: passes that target it WRITE
    every parameter's gradient, the padding between parameters stays zero)."""
    g = net.flat_grads
    key = (id(net), g.data_ptr())
    a = _alt_cache.get(key)
    if a is None:
        a = ops.fill(torch.empty_like(g), 0.0)
        _alt_cache[key] = a
    return a



def _mean_cotangents(B, device):
    """(+1/B, -1/B) as [B, 1] tensors: d(mean(out))/d(out) scaled by the Wasserstein losses' weights (+1 and -1, trainer.py:292, :299).
    They are constants of the step, so the critic's backward can start the moment its forward is done — no loss kernels on the
    chain.  Built once with the very kernels autograd would run (weighted_sum's backward, then mean's backward): same bits."""
    key = (int(B), device)
    c = _cot_cache.get(key)
    if c is None:
        like = torch.empty((B, 1), dtype=torch.float32, device=device)
        gpos, gneg = ops.weighted_sum_bwd((1.0, -1.0), _one(device).view(1), (True, True))
        c = (ops.mean_bwd(gpos.contiguous(), 1.0, like), ops.mean_bwd(gneg.contiguous(), 1.0, like))
        _cot_cache[key] = c
    return c


def _train_step_branch(generator, discriminator, classifier, opt_g, opt_d, x, y, target_y, mask, norm_vals, config, gumbel, branch,
                       skip_dead_d_wgrad, onehots=None, diag=None):
    """train_step scheduled for the length of its dependency chain (the step is a chain of small kernels, 36 launches on one stream): what the
    reference's loop body computes, bit for bit (tests/test_hip_house.py: graph vs eager vs the reference-order autograd step), with

      * no autograd graph: every backward is called directly, in the order autograd would run it;
      * residual assembly, mask, x_cf and both L1 penalties (:266-287, :305) in one launch, and the whole way back from
        dLoss/dx_cf (critic: -1/B through D; classifier: from the branch) to the generator's outputs in one (:314-315);
      * the frozen classifier's whole term (:301-302) — forward, cross-entropy value AND gradient in one launch (grad_scale =
        lambda_cls, exactly the product autograd forms), grad-input sweep — on the `branch` stream beside the critic update,
        the logged scalars (D_loss, G_loss, g_adv, g_reg) there too;
      * the critic step's two passes as one (Discriminator._run_pair): the cotangents of the Wasserstein means are the constants
        +-1/B (_mean_cotangents), the fake pass is the first writer of the gradient buffer, the real pass adds;
      * no gradient zero-fills: every parameter of both nets receives a gradient, the first writer overwrites (0 + g == g);
      * riders (one launch whose blocks split between two independent kernel bodies): the critic step's two power iterations in the
        residual block's forward launch, the logged scalars in its backward launch; on ONE stream also the critic's backward through
        W / sigma in the classifier's forward launch (whose tail is the cross-entropy) and the power iteration of the generator step's
        critic call in its backward launch.
    Captured in a HIP graph the two streams are parallel branches; branch="inline": everything on one stream (the default)."""
    nc, dev, B = config["num_classes"], x.device, x.shape[0]
    if classifier.training or any(p.requires_grad for p in classifier.parameters()):
        raise PcgError("train_step(branch=...): the classifier must be frozen and in eval mode (main.py:27-30)")
    branch, branch2 = branch if isinstance(branch, (tuple, list)) else (branch, None)
    main = torch.cuda.current_stream()
    if isinstance(branch, str):        # "inline": this schedule's kernels, all on the current stream (no fork: every wait is a no-op)
        branch = main
    discriminator._ensure_flat(); generator._ensure_flat()
    cot_pos, cot_neg = _mean_cotangents(B, dev)
    d_feat = float(x.shape[1])
    if not x.is_cuda:
        raise PcgError(f"train_step: input is on {x.device}; libpcgan_hip has no CPU path")
    seg, cat_idx, cont_idx = generator.index_tables(dev)
    x, mask = x.contiguous(), mask.contiguous()
    # onehots = (one_hot(target_y), one_hot(y)) already made (draw_batch_randoms writes them with the draws): two launches less
    target_onehot = onehots[0] if onehots is not None else ops.onehot(target_y, nc)            # :250
    with torch.no_grad():            # no autograd graph anywhere in this schedule: every backward below is called directly
        cont, _, samples, g_saved = generator._run_forward(x, target_onehot, mask, generator._noise(gumbel, B, dev),
                                                           float(config["gumbel_tau"]), False)            # :259-261
        # residual assembly, mask, x_cf and both L1 penalties (:266-287, :305): one launch — and the critic step's two power
        # iterations ride in it (they only read the critic's weights: one launch less on the chain)
        onehot_y = onehots[1] if onehots is not None else ops.onehot(y, nc)
        # (training mode: every critic call of the step makes a power iteration — the batched / riding forms assume it)
        pair = (branch2 is None and discriminator.training and discriminator._fused_ok(x, onehot_y) and
                all(p.requires_grad for p in discriminator.parameters()))
        rf = ops.house_residual_fwd(cont.contiguous(), samples.contiguous(), seg, norm_vals, x, mask, generator.col_src(),
                                    sn=discriminator._sn_operands() + (2,) if pair else None)
        residual_full, masked_residual, x_cf, mask_penalty_pre, am = rf[:5]
        sn_pair = rf[5] if pair else None
    # the zero-fills of the two gradient buffers are not launched: every parameter of both nets receives a gradient in this step,
    # so the first writer overwrites (0 + g == g)
    discriminator.drop_grads(); generator.drop_grads()
    # One stream and the fused critic pair: the classifier's two launches carry the critic's spectral-norm work that is on the chain
    # at the same time (riders) — its forward the backward through W / sigma of the critic step, its backward the power iteration
    # of the generator step's critic call.  Otherwise the classifier's term runs on the branch beside the critic update.
    xc = x_cf.detach().contiguous()
    riders = pair and branch is main and classifier._fused_ok(xc)
    if not riders:
        # ---- fork
        branch.wait_stream(main)
        with torch.cuda.stream(branch):
            with torch.no_grad():
                logits_c, acts_c = classifier._run_forward(xc, keep=True)                                         # :301
                g_cls, dlog = ops.cross_entropy_fwd_bwd(logits_c.contiguous(), target_y, need_loss=True, need_grad=True,
                                                        grad_scale=float(config["lambda_cls"]))                  # :302
                dx_cls = classifier._run_backward(acts_c, dlog)
                g_cls = g_cls.view(())
        for t in (x_cf, target_y):
            t.record_stream(branch)
    # ---- D step (:290-295)
    xd = x_cf.detach()
    with torch.no_grad():
        if riders:
            d_real, d_fake, snb_args, snb_keep = discriminator._run_pair(x, onehot_y, cot_neg, xd, target_onehot, cot_pos, sn=sn_pair,
                                                                         defer_sn_bwd=True)                   # :290-294, three launches
            # :301 + the fourth; with the logged scalars riding below, the cross-entropy (:302) is the tail of this launch too
            ce_tail = B <= 16 * 1024
            cf = classifier._run_forward(xc, keep=True, sn_bwd_rider=snb_args,
                                         ce=(target_y, float(config["lambda_cls"])) if ce_tail else None)
            logits_c, acts_c = cf[:2]
            del snb_keep
        elif pair:
            d_real, d_fake = discriminator._run_pair(x, onehot_y, cot_neg, xd, target_onehot, cot_pos, sn=sn_pair)   # :290-294, four launches
        elif branch2 is None or not discriminator._fused_ok(x, onehot_y):
            d_real, sv_r = discriminator._run_forward(x, onehot_y, keep=True)                 # :290
            d_fake, sv_f = discriminator._run_forward(xd, target_onehot, keep=True)           # :291
            discriminator._run_backward(sv_f, cot_pos, False, True)                           # d(mean(d_fake) - mean(d_real))
            discriminator._run_backward(sv_r, cot_neg, False, True)
        else:
            # (Opt-in, `GraphedTrainStep(overlap="critic")`: measured r02 without gain — 0.857 vs 0.856 ms — because the classifier
            # branch is then the longest chain, and every cross-queue edge of the graph costs ~10 us; with the zero-fills moved onto
            # this third stream as well the step went to 1.10 ms.)
            # The real pass and the fake pass only share the spectral-norm state: the reference's two forward calls each run a
            # power iteration, in this order (so the passes use different sigma and cannot be batched).  After the first one the
            # WHOLE real pass — critic forward, backward, weight gradients, the backward through W/sigma — runs on a third
            # stream into a second gradient buffer, beside the second power iteration and the fake pass; one add joins them
            # ((0 + a) + b either way).
            sn_r = discriminator._spectral_norm()                                             # power iteration of the D(real) call
            alt = _alt_grads(discriminator)
            branch2.wait_stream(main)                                                         # (behind the zero-fills it carries)
            with torch.cuda.stream(branch2):
                d_real, sv_r = discriminator._run_forward(x, onehot_y, keep=True, sn=sn_r)    # :290
                discriminator._run_backward(sv_r, cot_neg, False, True, gtarget=alt)
            for t in (x, onehot_y, cot_neg) + tuple(tt for grp in sn_r for tt in grp):
                t.record_stream(branch2)
            d_fake, sv_f = discriminator._run_forward(xd, target_onehot, keep=True)           # :291 (its own power iteration)
            discriminator._run_backward(sv_f, cot_pos, False, True)
            main.wait_stream(branch2)
            d_real.record_stream(main)
            ops.axpby(1.0, discriminator.flat_grads, 1.0, alt, out=discriminator.flat_grads)
    opt_d.step()                                                                              # :295
    # ---- G step (:298-316)
    with torch.no_grad():
        sn_g = None
        if riders:
            if ce_tail:
                dlog, g_cls = cf[2], cf[3]        # g_cls: still the row terms; the launch that logs the scalars forms their mean
            else:
                g_cls, dlog = ops.cross_entropy_fwd_bwd(logits_c.contiguous(), target_y, need_loss=True, need_grad=True,
                                                        grad_scale=float(config["lambda_cls"]))              # :302
                g_cls = g_cls.view(())
            dx_cls, sn_out = classifier._run_backward(acts_c, dlog, sn_fwd_rider=discriminator._sn_operands() + (1,))
            sn_g = sn_out[0]                                                                  # the power iteration of the :298 call
        d_fake_for_g, sv_g = discriminator._run_forward(xd, target_onehot, keep=True, sn=sn_g)   # :298
        ride = B <= 16 * 1024  # the five logged scalars ride in the launch of the residual block's backward (same trees, same bits)
        if not ride:
            fwd_done = torch.cuda.Event()
            fwd_done.record(main)
        dx_d = discriminator._run_backward(sv_g, cot_neg, True, not skip_dead_d_wgrad)        # d(-mean(D(x_cf)))/d(x_cf)
    if not ride:               # on the branch, as soon as the last critic forward is out
        branch.wait_event(fwd_done)
        with torch.cuda.stream(branch), torch.no_grad():
            m_fake = ops.mean_fwd(d_fake_for_g.contiguous()).view(())
            d_loss = ops.weighted_sum_fwd([ops.mean_fwd(d_fake.contiguous()), ops.mean_fwd(d_real.contiguous())], [1.0, -1.0]).view(())
            g_loss = ops.weighted_sum_fwd([m_fake, g_cls, am, mask_penalty_pre],
                                          [-1.0, config["lambda_cls"], config["lambda_reg"] * d_feat, config["lambda_mask"]]).view(())
            g_adv = ops.weighted_sum_fwd([m_fake], [-1.0]).view(())
            g_reg = ops.weighted_sum_fwd([am], [d_feat]).view(())
        for t in (d_real, d_fake, d_fake_for_g):
            t.record_stream(branch)
    main.wait_stream(branch)                                                                  # join: dx_cls (and g_cls)
    dx_cls.record_stream(main); g_cls.record_stream(main)
    with torch.no_grad():
        # :314-315 from dLoss/dx_cf = dx_d + dx_cls and the two penalty weights down to the generator's outputs: one launch
        rb = ops.house_residual_bwd(residual_full, masked_residual, mask, dx_d.contiguous(), dx_cls.contiguous(),
                                    config["lambda_mask"], config["lambda_reg"] * d_feat, len(generator.continuous_idx), cont_idx,
                                    seg, generator.total_cat, cat_idx, norm_vals,
                                    losses=(d_real, d_fake, d_fake_for_g, g_cls, am, mask_penalty_pre, float(config["lambda_cls"]),
                                            float(config["lambda_reg"] * d_feat), float(config["lambda_mask"]), d_feat) if ride else None,
                                    diag=(logits_c.contiguous(), diag["logits_orig"], diag.get("src_rows"), target_y, diag.get("eps", 1e-3),
                                          diag.get("acc")) if (ride and diag is not None) else None)
        d_cont, d_samples = rb[:2]
        if ride:
            d_loss, g_loss, g_adv, g_reg = rb[2][0], rb[2][1], rb[2][2], rb[2][3]             # :292, :307-312
            if g_cls.numel() > 1:
                g_cls = rb[2][5]
        generator._run_backward(g_saved, d_cont, None, d_samples)
    opt_g.step()                                                                              # :316
    main.wait_stream(branch)                                                                  # final join
    for t in (d_loss, g_loss, g_adv, g_reg, g_cls):
        t.record_stream(main)
    out = {"D_loss": d_loss, "G_loss": g_loss, "g_adv": g_adv, "g_cls": g_cls, "reg": g_reg, "mask_pen": mask_penalty_pre,
           "D_real": d_real, "D_fake_forG": d_fake_for_g, "x_cf": x_cf, "masked_residual": masked_residual}
    if diag is not None:                                                                      # :318-343
        if ride:
            out["diag"] = rb[3]                      # rode in the residual block's backward launch, with the accumulators
        else:
            with torch.no_grad():
                out["diag"] = ops.house_diag(logits_c.contiguous(), diag["logits_orig"], target_y, masked_residual, eps=diag.get("eps", 1e-3),
                                             src_rows=diag.get("src_rows"), acc=diag.get("acc"))
                _acc_losses(diag.get("acc"), d_loss, g_loss)
    return out


# ---- evaluation ------------------------------------------------------------------------------
def build_counterfactuals(G, x, target_onehot, config, gumbel=None, norm_vals=None):
    """eval_utils.py:25-181 for this generator's output signature: all features modifiable except the immutable ones, hard
    Gumbel-softmax samples (:76-77), residual assembly (:127-171), clamp to [0, 1] (:180).  Returns (masked_residual, x_cf)."""
    device = x.device
    mask = torch.empty_like(x)
    ops.fill(mask, 1.0)
    imm = list(config.get("immutable_idx", []))
    if imm:
        mask[:, imm] = 0.0                                                                  # :48-50 (setup-sized strided fill)
    norm_vals = norm_vals if norm_vals is not None else cat_norm_maps(G, config, device)    # :56-67
    cont, _, samples = G.forward_packed(x, target_onehot, mask, temperature=config.get("gumbel_tau", None), hard=True, gumbel=gumbel)
    residual_full = ops.assemble_residual_fwd(cont.contiguous(), G.index_tables(device)[2], samples.contiguous(),
                                              G.index_tables(device)[0], G.index_tables(device)[1], norm_vals, x.contiguous())
    _, masked = ops.scale_mask_fwd(residual_full, mask, 1.0)                                # :174-177
    x_cf = ops.clamp_add_fwd(x.contiguous(), masked, 0.0, 1.0)                              # :180
    return masked, x_cf


def compute_metrics_per_target(generator, classifier, X, y, config, gumbel_per_call=None, rng=None, max_vis=500):
    """eval_utils.py:185-289 — per target class: class-flip rate, prediction gain, mean |masked residual| over the samples whose
    class differs from the target, averaged over batches.  X, y: numpy arrays (already MinMax-scaled).  `gumbel_per_call`: an
    iterator of packed noise tensors, one per generator call, for runs that must reproduce given draws; otherwise `rng`."""
    import numpy as np
    device = next(generator.parameters()).device
    bs_cfg = int(config.get("batch_size", 128))
    X_t, y_t = torch.as_tensor(X, dtype=torch.float32), torch.as_tensor(y, dtype=torch.long)
    num_classes = int(np.unique(np.asarray(y)).size)
    generator.eval(); classifier.eval()
    if gumbel_per_call is None and generator.rng is None:
        generator.rng = rng if rng is not None else ops.DeviceRNG(seed=0)
    norm_vals = cat_norm_maps(generator, config, device)
    noise_it = iter(gumbel_per_call) if gumbel_per_call is not None else None
    results, originals, cfs = [], [], []
    with torch.no_grad():
        for target in range(num_classes):
            flips, gains, actions = [], [], []
            for i in range(0, X_t.shape[0], bs_cfg):
                xb, yb = X_t[i:i + bs_cfg], y_t[i:i + bs_cfg]
                sel = yb != target                                                          # :224-226 (host-side selection)
                if int(sel.sum()) == 0:
                    continue
                x = xb[sel].to(device).contiguous()
                bs = x.shape[0]
                target_vec = torch.full((bs,), target, dtype=torch.long, device=device)
                target_onehot = ops.onehot(target_vec, num_classes)
                noise = next(noise_it) if noise_it is not None else None
                masked, _ = build_counterfactuals(generator, x, target_onehot, config, gumbel=noise, norm_vals=norm_vals)
                x_cf = ops.axpby(1.0, x, 1.0, masked)                                       # :243 (unclamped, as the reference)
                m = ops.cf_metrics(classifier(x_cf).contiguous(), target_vec, logits_ref=classifier(x).contiguous()).cpu()
                flips.append(float(m[0])); gains.append(float(m[1])); actions.append(abs_mean(masked).item())
                if sum(o.shape[0] for o in originals) < max_vis:
                    originals.append(x.cpu()); cfs.append(x_cf.cpu())
            results.append({"target_class": target, "class_flip": float(np.mean(flips)) if flips else float("nan"),
                            "prediction_gain": float(np.mean(gains)) if gains else float("nan"),
                            "avg_actionability": float(np.mean(actions)) if actions else float("nan")})
    originals = torch.cat(originals, 0).numpy() if originals else np.empty((0, X_t.shape[1]))
    cfs = torch.cat(cfs, 0).numpy() if cfs else np.empty((0, X_t.shape[1]))
    return results, originals, cfs


class GraphedTrainStep:
    """The whole training step — G forward, critic step, G step, both Adam updates: ~400 kernels as an op chain, 36 as scheduled by
    train_step(branch=...) — captured once in a HIP graph and replayed with one host call: this path is latency bound (SURVEY.md
    section 8a row a15), and the graph removes the per-kernel host cost.  overlap="inline" (default): the scheduled step on one
    stream; True: the frozen classifier's term on a parallel graph branch (a graph with branches is launched node by node by the
    host, a chain is not: 0.28 vs 0.03 ms of host time per replay; wall 0.434 vs 0.422 ms at batch 4096 once the classifier term
    itself was down to 47 us); "critic": a third stream for the critic's real pass (no gain); False: the reference-order autograd
    step.  All bit-identical.  Inputs live in static device buffers (`x, y, target_y, mask, noise`): write the next
    batch into them (`load(...)`, or draw straight into them) and call `replay()`; outputs are the static tensors in `out`.
    Capture needs warm-up executions of real steps; the parameters, buffers and optimizer state are snapshotted before and
    restored after, so constructing this object does not advance training."""

    def __init__(self, generator, discriminator, classifier, opt_g, opt_d, norm_vals, batch, config=CONFIG, warmup=3, overlap="inline",
                 rng=None, dataset=None, diag=None, with_critic_wgrad_variant=False):
        """with_critic_wgrad_variant: also capture the step WITH the critic weight gradients of the generator step
        (skip_dead_d_wgrad=False: what the reference's autograd computes, :314) as a second graph over the same static buffers —
        `replay(full=True)`; the trainer runs it as the last iteration of an epoch, whose D.grad the epoch summary prints (:362).
        dataset = (X [N, D] float32, Y [N] int64) resident on the device (needs rng): the draws launch also TAKES the batch — rows
        perm[cursor .. cursor + batch) of the epoch's permutation (`new_epoch(perm)` uploads it and rewinds the cursor) — so a replay
        needs no host-side copy (trainer.py:198's DataLoader(shuffle=True, drop_last=True) without collation / PCIe / copy launches).
        diag = {"logits_orig": [N, nc] (dataset mode: gathered through the batch's source rows) or [batch, nc], "acc": float64[8] or
        None, "eps": 1e-3}: the trainer's per-iteration diagnostics + epoch accumulators ride in the step (train_step(diag=...)).
        rng (an ops.DeviceRNG): the per-iteration draws (draw_batch_randoms: target class, feature mask, Gumbel noise, one-hot rows) are
        the first launch of the captured step, their Philox offsets read from a device counter that the launch advances itself — write x
        and y into the static buffers (`load_batch`) and replay(); the numbers are those draw_batch_randoms(rng, ...) would have drawn
        before each eager step, and rng.offset is kept in step on the host."""
        dev = norm_vals.device
        # train_step's parallel branch: the classifier term and the logged sums; overlap="critic" adds a third stream for the
        # critic's real pass (bit-identical, no gain measured: see _train_step_branch)
        if overlap == "critic":
            self.branch = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        elif overlap == "inline":
            self.branch = "inline"
        else:
            self.branch = torch.cuda.Stream(device=dev) if overlap else None
        D_in, T = config["input_dim"], generator.total_cat
        self.x = torch.zeros((batch, D_in), dtype=torch.float32, device=dev)
        self.y = torch.zeros((batch,), dtype=torch.int64, device=dev)
        self.target_y = torch.ones((batch,), dtype=torch.int64, device=dev)
        self.mask = torch.ones((batch, D_in), dtype=torch.float32, device=dev)
        self.noise = torch.zeros((batch, T), dtype=torch.float32, device=dev)
        nets = (generator, discriminator)
        for n in nets:
            n._ensure_flat()
        saved = [(n.flat_params.clone(), [b.clone() for b in n.buffers()]) for n in nets]
        osnap = [o.snapshot() for o in (opt_g, opt_d)]

        nc = config["num_classes"]
        # one-hot rows of target_y and y: static inputs of the scheduled step (draw_batch_randoms(onehots=self.onehots) fills them with
        # the draws; load() computes them)
        self.onehots = (torch.zeros((batch, nc), dtype=torch.float32, device=dev), torch.zeros((batch, nc), dtype=torch.float32, device=dev))
        ops.onehot(self.target_y, nc, out=self.onehots[0]); ops.onehot(self.y, nc, out=self.onehots[1])

        self._rng = rng
        self._span = ops.DeviceRNG.house_draws_span(batch, D_in, T)
        self._dataset = dataset
        if dataset is not None and rng is None:
            raise PcgError("GraphedTrainStep(dataset=...) needs rng: the batch is taken by the draws launch")
        self._ctr = rng.device_counter(dev, cursor=dataset is not None) if rng is not None else None
        self.src = self.perm = None
        if dataset is not None:
            X, Y = dataset
            if X.dtype != torch.float32 or Y.dtype != torch.int64 or X.shape[1] != D_in or X.shape[0] != Y.shape[0] or X.shape[0] < batch:
                raise PcgError(f"GraphedTrainStep(dataset=...): need X [N >= {batch}, {D_in}] float32 and Y [N] int64 on the device")
            self.src = torch.zeros((batch,), dtype=torch.int64, device=dev)
            self.perm = torch.arange((X.shape[0] // batch) * batch, dtype=torch.int64, device=dev)      # identity until new_epoch()
            self._imm = torch.tensor(list(config.get("immutable_idx", [])), dtype=torch.int32, device=dev)
        ctr0 = self._ctr.clone() if rng is not None else None
        sdiag = None
        if diag is not None:
            sdiag = dict(diag)
            if dataset is not None and diag["logits_orig"].shape[0] != batch:
                sdiag["src_rows"] = self.src
        self._diag = sdiag
        acc0 = sdiag["acc"].clone() if sdiag is not None and sdiag.get("acc") is not None else None

        def step(skip_dead=True):
            if dataset is not None:
                rng.house_batch_draws(dataset[0], dataset[1], self.perm, nc, T, self._imm if self._imm.numel() else None,
                                      (self.x, self.y, self.target_y, self.mask, self.noise), self.onehots, self._ctr, src_out=self.src)
            elif rng is not None:
                draw_batch_randoms(rng, generator, self.y, config, dev, out=(self.target_y, self.mask, self.noise),
                                   onehots=self.onehots if self.branch is not None else None, counter=self._ctr)
            return train_step(generator, discriminator, classifier, opt_g, opt_d, self.x, self.y, self.target_y, self.mask, norm_vals,
                              config, gumbel=self.noise, branch=self.branch, onehots=self.onehots if self.branch is not None else None,
                              diag=sdiag, skip_dead_d_wgrad=skip_dead)
        self._nc = nc
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                step()
            if with_critic_wgrad_variant:
                step(False)
        torch.cuda.current_stream().wait_stream(side)
        import gc
        gc.collect()                 # no finalizer may run while the stream captures (see nn._HipGraphCapture.begin)
        gc_on = gc.isenabled()
        gc.disable()
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = step()
            self.graph_full = self.out_full = None
            if with_critic_wgrad_variant:
                self.graph_full = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_full):
                    self.out_full = step(False)
        finally:
            if gc_on:
                gc.enable()
        for n, (fp, bufs) in zip(nets, saved):
            n.flat_params.copy_(fp)
            for b, b0 in zip(n.buffers(), bufs):
                b.copy_(b0)
        for o, sn in zip((opt_g, opt_d), osnap):
            o.restore(sn)
        if rng is not None:
            self._ctr.copy_(ctr0)        # the warm-up steps drew from the counter (and walked the cursor): back to the stream's position
        if acc0 is not None:
            sdiag["acc"].copy_(acc0)     # ... and added to the epoch accumulators

    def new_epoch(self, perm):
        """dataset mode: upload this epoch's row order (int64, at least steps * batch entries; only whole batches are used —
        drop_last) and rewind the cursor.  Two small copies per EPOCH; the replays in between touch no host memory."""
        if self._dataset is None:
            raise PcgError("new_epoch: this GraphedTrainStep was not built with dataset=...")
        n = self.perm.numel()
        if perm.numel() < n:
            raise PcgError(f"new_epoch: permutation of {perm.numel()} entries, {n} needed")
        self.perm.copy_(perm[:n].to(torch.int64), non_blocking=False)
        self._ctr[2:3].zero_()

    @property
    def steps_per_epoch(self):
        return self.perm.numel() // self.x.shape[0] if self.perm is not None else None

    def load_batch(self, x, y):
        """With rng: the data half of a batch (the draws are made by the replay)."""
        self.x.copy_(x); self.y.copy_(y)

    def load(self, x, y, target_y, mask, noise):
        self.x.copy_(x); self.y.copy_(y); self.target_y.copy_(target_y); self.mask.copy_(mask); self.noise.copy_(noise)
        ops.onehot(self.target_y, self._nc, out=self.onehots[0]); ops.onehot(self.y, self._nc, out=self.onehots[1])

    def replay(self, full=False):
        """full=True: the variant that also computes the generator step's critic weight gradients (with_critic_wgrad_variant)."""
        if full:
            if self.graph_full is None:
                raise PcgError("replay(full=True): build the GraphedTrainStep with with_critic_wgrad_variant=True")
            self.graph_full.replay()
        else:
            self.graph.replay()
        if self._rng is not None:
            self._rng.offset += self._span       # the host-side mirror of the device counter
        return self.out_full if full else self.out


class WeightedCrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(weight=class_weights) (trainer.py:55-57)."""

    def __init__(self, weight):
        super().__init__()
        self.register_buffer("weight", weight)

    def forward(self, input, target):
        return _WCEFn.apply(input, target, self.weight)


class _WCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight):
        zc = logits.contiguous()
        ctx.save_for_backward(zc, target, weight)
        loss, _ = ops.cross_entropy_weighted_fwd_bwd(zc, target, weight, need_grad=False)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        z, target, weight = ctx.saved_tensors
        _, dz = ops.cross_entropy_weighted_fwd_bwd(z, target, weight, need_loss=False, grad_out=g.contiguous().view(1))
        return dz, None, None


def train_classifier(X_train_all, X_test, y_train_all, y_test, scaler, config, device=None, verbose=True):
    """trainer.py:18-180 without the plotting tail: stratified validation split, class-weighted CrossEntropyLoss, AdamW,
    ReduceLROnPlateau(factor 0.5, patience 4), early stopping on the validation loss; returns the model with its best state."""
    import copy
    import numpy as np
    from sklearn.model_selection import train_test_split
    from sklearn.utils.class_weight import compute_class_weight
    from torch.utils.data import DataLoader, TensorDataset
    from .optim import AdamW
    device = torch.device(device if device is not None else config.get("cuda", "cuda:0"))
    seed = config.get("seed", 42)
    torch.manual_seed(seed); np.random.seed(seed)                                             # :20-22
    X_train, X_val, y_train, y_val = train_test_split(X_train_all, y_train_all, test_size=config.get("val_frac", 0.10),
                                                      random_state=seed, stratify=y_train_all)   # :29-31
    num_classes = int(np.unique(y_train_all).size)
    bs = config.get("clf_batch_size", config.get("batch_size", 128))
    mk = lambda X, y: TensorDataset(torch.tensor(X, dtype=torch.float32), torch.tensor(y, dtype=torch.long))  # noqa: E731
    train_loader = DataLoader(mk(X_train, y_train), batch_size=bs, shuffle=True, drop_last=False)
    val_loader = DataLoader(mk(X_val, y_val), batch_size=bs, shuffle=False)
    model = NNClassifier(config["input_dim"], output_dim=num_classes).to(device)              # :50
    cw = compute_class_weight("balanced", classes=np.arange(num_classes), y=y_train)          # :53
    criterion = WeightedCrossEntropyLoss(torch.tensor(cw, dtype=torch.float32, device=device))
    optimizer = AdamW(model.parameters(), lr=config.get("clf_lr", 1e-3), weight_decay=config.get("clf_wd", 1e-4))   # :58
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.5, patience=4)           # :59
    best_val, best_state, wait = float("inf"), None, 0
    patience, epochs = config.get("clf_early_stopping", 15), config.get("clf_epochs", 100)
    history = []
    for epoch in range(1, epochs + 1):
        model.train()
        run_loss, correct, total = 0.0, 0.0, 0
        for xb, yb in train_loader:
            xb, yb = xb.to(device), yb.to(device)
            optimizer.zero_grad()
            logits = model(xb)
            loss = criterion(logits, yb)
            loss.backward()
            optimizer.step()
            n = xb.size(0)
            run_loss += loss.item() * n
            correct += ops.cf_metrics(logits.detach().contiguous(), yb, other=yb)[0].item() * n
            total += n
        model.eval()
        v_loss, v_correct, v_total = 0.0, 0.0, 0
        with torch.no_grad():
            for xb, yb in val_loader:
                xb, yb = xb.to(device), yb.to(device)
                logits = model(xb).contiguous()
                n = xb.size(0)
                v_loss += ops.cross_entropy_weighted_fwd_bwd(logits, yb, criterion.weight, need_grad=False)[0].item() * n
                v_correct += ops.cf_metrics(logits, yb, other=yb)[0].item() * n
                v_total += n
        val_loss = v_loss / v_total
        scheduler.step(val_loss)                                                              # :128
        history.append((run_loss / total, correct / total, val_loss, v_correct / v_total))
        if val_loss < best_val - 1e-6:                                                        # :129-139
            best_val, wait = val_loss, 0
            best_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        else:
            wait += 1
        if verbose:
            print(f"[Epoch {epoch}/{epochs}] train_loss={history[-1][0]:.4f}, train_acc={history[-1][1]:.4f} | "
                  f"val_loss={val_loss:.4f}, val_acc={history[-1][3]:.4f} | wait={wait}")
        if wait >= patience:
            break
    if best_state is not None:
        model.load_state_dict(best_state)                                                     # :150-151
    model.history = history
    return model


def draw_batch_randoms(rng, generator, y, config, device, out=None, onehots=None, counter=None):
    """The per-iteration draws of trainer.py:248-255 + generator.py:90 on the device: (target_y != y, feature mask, Gumbel noise).
    out = (target_y, mask, noise): draw straight into these buffers (the static inputs of a GraphedTrainStep).  counter: the device
    counter form of the launch (ops.DeviceRNG.house_draws)."""
    B = y.shape[0]
    imm = getattr(generator, "_imm_dev", None)
    if imm is None or imm.device != device:
        imm = torch.tensor(list(config.get("immutable_idx", [])), dtype=torch.int32, device=device)
        generator._imm_dev = imm
    if out is None:
        out = (torch.empty((B,), dtype=torch.int64, device=device), torch.empty((B, config["input_dim"]), dtype=torch.float32, device=device),
               torch.empty((B, generator.total_cat), dtype=torch.float32, device=device))
    # one launch; the values of randint(exclude=y), feature_mask, gumbel called in this order
    return rng.house_draws(y.contiguous(), config["num_classes"], config["input_dim"], generator.total_cat, imm if imm.numel() else None, out,
                           onehots=onehots, counter=counter)


def grad_norm_sum(net, out=None):
    """trainer.py:182-183 — this trainer's `grad_norm`: the SUM over parameters of ||p.grad||_2 (not the root of the summed squares
    of mnist/trainer.py:41-42).  One launch over the flat gradient buffer (pcg_norm_sum); `out` (a [1] device tensor): leave the
    value there and return it unread (the trainer reads a whole history at once), else one host read."""
    net._ensure_flat()
    seg = getattr(net, "_norm_seg", None)
    if seg is None or seg[0] != net.flat_grads.data_ptr():
        table = torch.tensor([[off, n] for p, off, n in net._seg if p.requires_grad], dtype=torch.int64, device=net.flat_grads.device)
        seg = net._norm_seg = (net.flat_grads.data_ptr(), table)
    if any(p.grad is None for p, _, _ in net._seg if p.requires_grad):
        raise PcgError("grad_norm_sum: a parameter has no gradient yet (call after a training step)")
    res = ops.norm_sum(net.flat_grads, seg[1], out=out)
    return res if out is not None else float(res.item())


def epoch_permutation(n):
    """The row order DataLoader(TensorDataset, shuffle=True) walks in one epoch, with its exact consumption of torch's global CPU
    generator (trainer.py:198): the loader iterator draws its base seed, the RandomSampler a seed for a private generator, and
    that generator the permutation (tests/test_host_logic.py pins this to torch.utils.data.DataLoader itself)."""
    torch.empty((), dtype=torch.int64).random_()                                   # _BaseDataLoaderIter: base seed (unused here)
    seed = int(torch.empty((), dtype=torch.int64).random_().item())                # RandomSampler.__iter__
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


def train_countergan(generator, config, X_train, y_train, clf_model, *, rng=None, draws=None, graph=None, log_every=100, verbose=True,
                     save=True, discriminator=None):
    """house_sales_kc_usa/trainer.py:186-378 `train_countergan(generator, config, X_train, y_train, clf_model)` on the HIP kernels —
    same signature, same body:

      :187-189  device = config['cuda'] (default "cuda"); torch.manual_seed / np.random.seed(config.get('seed', 42))
      :191-198  num_classes from y_train, batch size, DataLoader(shuffle=True, drop_last=True) — here the training set is uploaded
                ONCE and stays in HBM; the epoch's row order is DataLoader's (epoch_permutation), batches are taken on the device
      :200-223  cat_norm_maps from config['scaler'] (or the n-1 fallback)
      :226-231  the Discriminator is built HERE, after the seeding (same torch draws as the reference's), Adam x2
      :233-234  classifier to the device, eval()
      :241-316  the iteration = train_step (draws: target class, feature mask, Gumbel noise from `rng`, an ops.DeviceRNG seeded with
                the config seed by default; `draws(epoch, batch_idx, y) -> (target_y, mask, gumbel)` supplies them instead — parity runs)
      :318-343  pred_gain, sparsity, L2 reg, class-flip rate: one fused device reduction riding in the step (pcg_house_diag); the
                classifier's logits of the ORIGINAL rows are evaluated once for the whole training set (the classifier is frozen)
      :345-355  the log line every 100 batches; D_loss / G_loss and the four diagnostics are summed on the device and read ONCE per
                epoch (the reference: eight .item() calls per iteration)
      :357-366  epoch summary with G_grad / D_grad (grad_norm_sum).  D's .grad there = the D step's gradients + the G step's critic
                weight gradients (no zero_grad in between, :293 -> :314): the last iteration of an epoch therefore runs with
                skip_dead_d_wgrad=False, all others skip that dead work
      :369-378  torch.save(generator.state_dict(), config['generator_path']) (the loss-curve PNG is plotting: left out)

    graph (default: when the batch size divides into full batches and no `draws` hook is given): every full-size iteration is ONE
    HIP-graph replay (GraphedTrainStep(dataset=..., diag=...)) — the benched rate through the reference-shaped API.
    Returns {"d_losses", "g_losses", "pred_gain", "sparsity", "l2_reg", "class_flip_rate", "G_grad", "D_grad", "discriminator"}
    (per-epoch lists; the reference returns None)."""
    device = torch.device(config.get("cuda", "cuda"))
    if device.type != "cuda":
        raise PcgError(f"train_countergan: config['cuda'] = {device}; libpcgan_hip has no CPU path")
    seed = config.get("seed", 42)
    torch.manual_seed(seed)                                                                    # :188
    np.random.seed(seed)                                                                       # :189
    y_np = np.asarray(y_train)
    num_classes = int(np.unique(y_np).size)                                                    # :191
    bs = int(config.get("batch_size", 128))                                                    # :192
    X_t = torch.tensor(np.asarray(X_train), dtype=torch.float32)                               # :195
    y_t = torch.tensor(y_np, dtype=torch.long)                                                 # :196
    N = X_t.shape[0]
    steps = N // bs                                                                            # drop_last (:198)
    if steps < 1:
        raise PcgError(f"train_countergan: {N} rows do not fill one batch of {bs} (drop_last=True leaves no iteration)")
    X_dev, y_dev = X_t.to(device), y_t.to(device)                                              # the whole training set: resident
    cfg = dict(config, num_classes=num_classes)
    norm_vals = cat_norm_maps(generator, cfg, device)                                          # :200-223
    G = generator.to(device)
    D = discriminator if discriminator is not None else Discriminator(config["input_dim"], config["hidden_dim"], num_classes)   # :227
    D = D.to(device)
    opt_g, opt_d = make_optimizers(G, D, cfg)                                                  # :230-231
    clf_model = clf_model.to(device)                                                           # :233
    clf_model.eval()                                                                           # :234
    frozen = not any(p.requires_grad for p in clf_model.parameters())
    rng = rng if rng is not None else ops.DeviceRNG(seed=seed)
    # the classifier on the original rows (:320): frozen, so ONE evaluation for the whole training set serves every iteration
    with torch.no_grad():
        logits_orig = torch.cat([clf_model(X_dev[i:i + 65536]).detach() for i in range(0, N, 65536)]).contiguous()
    acc = torch.zeros(8, dtype=torch.float64, device=device)
    diag = {"logits_orig": logits_orig, "acc": acc, "eps": 1e-3}
    if graph is None:
        graph = draws is None and frozen
    gs = None
    if graph:
        if draws is not None:
            raise PcgError("train_countergan(graph=True): supplied draws cannot be replayed from a captured step; use graph=False")
        gs = GraphedTrainStep(G, D, clf_model, opt_g, opt_d, norm_vals, bs, cfg, rng=rng, dataset=(X_dev, y_dev), diag=diag,
                              with_critic_wgrad_variant=True)
    epochs = int(config["epochs"])
    # per-epoch results stay on the device until someone needs them: verbose prints (and reads) every epoch like the reference,
    # otherwise ONE read after the last epoch — the host never waits for the GPU inside the loop
    acc_hist = torch.zeros((epochs, 8), dtype=torch.float64, device=device)
    gn_hist = torch.zeros((epochs, 2), dtype=torch.float32, device=device)
    names = ("d_losses", "g_losses", "pred_gain", "sparsity", "l2_reg", "class_flip_rate")
    hist = {k: [] for k in names + ("G_grad", "D_grad")}

    def read_epochs(lo, hi):
        a, gn = acc_hist[lo:hi].cpu().numpy(), gn_hist[lo:hi].cpu().numpy()
        for e in range(hi - lo):
            n_it = max(float(a[e, 6]), 1.0)
            for k, v in zip(names, a[e, :6] / n_it):
                hist[k].append(float(v))
            hist["G_grad"].append(float(gn[e, 0])); hist["D_grad"].append(float(gn[e, 1]))

    sched = "inline" if (frozen and not clf_model.training) else None
    for epoch in range(epochs):                                                                # :237
        perm = epoch_permutation(N)[:steps * bs]                                               # :241 (the loader's order)
        perm_dev = perm.to(device)
        ops.fill(acc.view(torch.float32), 0.0)
        if gs is not None:
            gs.new_epoch(perm_dev)
        for batch_idx in range(steps):
            last = batch_idx == steps - 1          # keeps the G step's critic weight gradients: D_grad of the epoch summary (:362)
            if gs is not None:
                out = gs.replay(full=last)
            else:
                idx = perm_dev[batch_idx * bs:(batch_idx + 1) * bs]
                x, y = X_dev.index_select(0, idx), y_dev.index_select(0, idx)                  # :242-243
                if draws is not None:
                    target_y, mask, noise = draws(epoch, batch_idx, y)                          # :248-255, generator.py:90
                    target_y, mask = target_y.to(device), mask.to(device)
                    noise = G.pack_noise(noise) if isinstance(noise, dict) else noise.to(device)
                else:
                    target_y, mask, noise = draw_batch_randoms(rng, G, y, cfg, device)
                out = train_step(G, D, clf_model, opt_g, opt_d, x, y, target_y, mask, norm_vals, cfg, gumbel=noise, branch=sched,
                                 diag=dict(diag, src_rows=idx), skip_dead_d_wgrad=not last)
            if verbose and batch_idx % log_every == 0:                                         # :345-348
                print(f"[Epoch {epoch + 1}/{epochs}] batch {batch_idx} :: "
                      f"D(real)={torch.sigmoid(out['D_real']).mean().item():.3f}, D(fake)={torch.sigmoid(out['D_fake_forG']).mean().item():.3f}, "
                      f"g_adv={out['g_adv'].item():.4f}, g_cls={out['g_cls'].item():.4f}, reg={out['reg'].item():.6f}, "
                      f"mask_pen={out['mask_pen'].item():.5f}")
        acc_hist[epoch].copy_(acc)                                                             # :349-355 (device to device, no sync)
        grad_norm_sum(G, out=gn_hist[epoch, 0:1]); grad_norm_sum(D, out=gn_hist[epoch, 1:2])   # :362
        if verbose:
            read_epochs(epoch, epoch + 1)
            print(f"[{epoch + 1}/{epochs}] D: {hist['d_losses'][-1]:.4f}, G: {hist['g_losses'][-1]:.4f}, "
                  f"pred_gain={hist['pred_gain'][-1]:.4f}, sparsity={hist['sparsity'][-1]:.4f}, l2_reg={hist['l2_reg'][-1]:.4f}, "
                  f"class_flip_rate={hist['class_flip_rate'][-1]:.4f} | G_grad: {hist['G_grad'][-1]:.4f}, D_grad: {hist['D_grad'][-1]:.4f}")
    if not verbose:
        read_epochs(0, epochs)
    if save and config.get("generator_path"):
        import os
        os.makedirs(os.path.dirname(os.path.abspath(config["generator_path"])) or ".", exist_ok=True)
        torch.save({k: v.detach().cpu().contiguous() for k, v in generator.state_dict().items()}, config["generator_path"])   # :377
        if verbose:
            print(f"Saved generator model to {config['generator_path']}")
    hist["discriminator"] = D
    return hist


def train_countergan_loop(generator, discriminator, classifier, loader, config, device, rng=None, log_every=50):
    """The bare loop over a user-supplied iterable of (x, y) batches (r01 form; the reference-shaped entry point is train_countergan):
    per-epoch means of D and G loss."""
    opt_g, opt_d = make_optimizers(generator, discriminator, config)
    norm_vals = cat_norm_maps(generator, config, device)
    rng = rng if rng is not None else ops.DeviceRNG(seed=0)
    classifier.eval()
    history = []
    for epoch in range(config["epochs"]):
        pending = []
        for batch_idx, (x, y) in enumerate(loader):
            x, y = x.to(device), y.to(device)
            target_y, mask, noise = draw_batch_randoms(rng, generator, y, config, device)
            out = train_step(generator, discriminator, classifier, opt_g, opt_d, x, y, target_y, mask, norm_vals, config, gumbel=noise)
            pending.append((out["D_loss"].detach(), out["G_loss"].detach()))   # no grad_fn kept alive over the epoch
            if batch_idx % log_every == 0:
                print(f"[Epoch {epoch + 1}/{config['epochs']}] batch {batch_idx}: D_loss={out['D_loss'].item():.4f}, "
                      f"G_loss={out['G_loss'].item():.4f}, g_adv={out['g_adv'].item():.4f}, g_cls={out['g_cls'].item():.4f}, "
                      f"reg={out['reg'].item():.4f}, mask_pen={out['mask_pen'].item():.6f}")
        n = max(len(pending), 1)
        history.append((sum(d.item() for d, _ in pending) / n, sum(g.item() for _, g in pending) / n))
    return history


def build(device, config=CONFIG, seed=0):
    """(G, D, clf) on `device`, constructed in the reference's order (main.py: classifier, generator; trainer.py:227: D)."""
    torch.manual_seed(seed)
    clf = NNClassifier(config["input_dim"], config["num_classes"])
    G = ResidualGenerator(config["input_dim"], config["hidden_dim"], config["num_classes"], config["continuous_idx"],
                          config["categorical_info"], tau=config["gumbel_tau"])
    D = Discriminator(config["input_dim"], config["hidden_dim"], config["num_classes"])
    clf.eval()
    for p in clf.parameters():
        p.requires_grad = False
    return G.to(device), D.to(device), clf.to(device)

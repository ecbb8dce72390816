"""Host-side mirror of `conditional_counteRGAN/mnist/` on the HIP kernels.

    reference                                              here
    ---------------------------------------------------    -----------------------------------------------------
    config.py            Config            :3-28            Config (the fields the step reads)
    models/generator.py  ResidualGenerator :25-86           ResidualGenerator (same ctor args, same state_dict keys)
    models/discriminator.py Discriminator  :5-38            Discriminator
    models/classifier.py CNNClassifier     :4-28            CNNClassifier (frozen / eval use: forward + grad-input)
    trainer.py           build_mask        :45-72           build_mask (same draws from torch's RNG)
    trainer.py           train_countergan  :76-123          make_optimizers + train_step (+ train_countergan loop)

The torch layer objects are kept as parameter containers (identical `state_dict()` keys: `embed.weight`,
`conv_in.*`, `resblocks.{0..5}.{conv1,bn1,conv2,bn2}.*`, `conv_mid.*`, `conv_out.*`; `cond_embed.weight`, `main.{0,2,4,6}.weight`,
`adv_head.*`; `conv.{0,2,4}.*`, `fc.{1,4}.*`), forward/backward are one autograd node per network sequencing C-ABI
calls on NHWC activations.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, PcgError
from .nn import weighted_sum, FlatModule
from .optim import Adam


class Config:
    """config.py:3-28."""
    batch_size = 128
    num_epochs_gan = 20
    d_lr = 1e-5
    g_lr = 5e-5
    cls_lr = 1e-3
    num_epochs_clf = 10
    lambda_adv = 1.0
    lambda_cls = 1.0
    lambda_reg = 2.5
    lambda_mask = 2.0
    patch_size = 7
    num_modifiable_patches = 10
    img_shape = (1, 28, 28)
    num_classes = 10


# ---- conv helpers on NHWC activations ------------------------------------------------------------------------------
def _geom(conv, B, H, W):
    k, s, p = conv.kernel_size, conv.stride, conv.padding
    return ops.conv_geom(B, H, W, conv.in_channels, conv.out_channels, k[0], k[1], s[0], p[0])


def _conv_fwd(conv, a, act=ACT_NONE, slope=0.0):
    B, H, W, _ = a.shape
    g = _geom(conv, B, H, W)
    z = ops.conv2d_fwd(g, a, ops.ohwi(conv.weight.data), conv.bias.data if conv.bias is not None else None, act=act, slope=slope)
    return g, z


def _conv_wgrad(net, conv, g, a, dz, need_p=True, bias_done=False):
    """Accumulate weight/bias gradients into net's flat buffer (bias_done: the bias gradient already came out of the BatchNorm
    backward's apply pass — ops.bn_act_bwd / bn_bwd_partial with dcol=)."""
    if need_p and conv.weight.requires_grad:
        gw, acc = net._grad_view(conv.weight)
        ops.conv2d_wgrad(g, a, dz, ops.ohwi(gw), acc)
        if conv.bias is not None and not bias_done:
            gb, accb = net._grad_view(conv.bias)
            ops.colsum(dz.numel() // conv.out_channels, conv.out_channels, dz, gb, accb)


def _conv_bwd(net, conv, g, a, dz, need_p, need_x):
    """Accumulate weight/bias gradients into net's flat buffer; return grad wrt the conv input (or None)."""
    _conv_wgrad(net, conv, g, a, dz, need_p)
    return ops.conv2d_dgrad(g, dz, ops.ohwi(conv.weight.data)) if need_x else None


def _dgrad_act(g, dz, w, a_below, act, slope):
    """conv_dgrad(dz, w) * act'(a_below): one launch when the layer is an MFMA layer, grad-input + pcg_act_bwd otherwise."""
    res = ops.conv_bwd_data_fused(g, dz, w, False, act, slope, a_below=a_below) if FUSE_BACKWARD_EPILOGUE else None
    if res is not None:
        return res[0]
    d = ops.conv2d_dgrad(g, dz, w)
    return ops.act_bwd(d, a_below.view(d.shape), act, slope, out=d)


S1_DGRAD_AS_FWD = os.environ.get("PCG_S1_DGRAD_AS_FWD", "1") != "0"   # A/B switch: stride-1 grad-inputs on the forward kernel (adjoint weight)
FUSE_SKIP_BNSUM = os.environ.get("PCG_SKIP_BNSUM", "1") != "0"   # A/B switch: bn2's backward column sums out of the previous block's skip-add grad-input epilogue
DEFER_SLAB_REDUCTIONS = os.environ.get("PCG_SLAB_DEFER", "0") != "0"   # opt-in A/B switch: one slab-reduction launch per backward sweep (ops.slab_reductions_deferred; 26.68 -> 26.64 ms)
GRAD_INPUT_LABEL_CHANNEL_ONLY = True   # A/B switch: conv_in's grad-input for the label-map channel only (see _run_backward)
FUSE_BIAS_COLSUM = True         # A/B switch: conv-bias gradients in front of a BatchNorm out of the BatchNorm backward's apply pass
FUSE_BACKWARD_EPILOGUE = True   # A/B switch for the tests: activation derivative / BatchNorm-backward sums / skip-connection add in
                                # the grad-input kernel's epilogue (ops.conv_bwd_data_fused, conv2d_dgrad_add) vs separate passes


def _as_rows(t, B):
    """[B,1,H,W] / [B,H,W] tensor as a contiguous fp32 [B, H*W] view (C == 1: NCHW and NHWC are the same bytes)."""
    v = t.reshape(B, -1)
    if not v.is_contiguous():
        v = v.contiguous()
    if v.dtype != torch.float32:
        raise PcgError(f"expected float32, got {v.dtype}")
    return v


# ---- small autograd nodes for the trainer's elementwise glue (trainer.py:97,99,119) ---------------------------------
class _ClampAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, lo, hi):
        xc, rc = x.contiguous(), r.contiguous()
        ctx.save_for_backward(xc, rc)
        ctx.lo, ctx.hi = lo, hi
        return ops.clamp_add_fwd(xc, rc, lo, hi)

    @staticmethod
    def backward(ctx, dy):
        x, r = ctx.saved_tensors
        return None, ops.clamp_add_bwd(dy.contiguous(), x, r, ctx.lo, ctx.hi), None, None


def clamp_add(x, r, lo=-1.0, hi=1.0):
    """torch.clamp(x + r, lo, hi) with gradient to r only (x is data)."""
    return _ClampAdd.apply(x, r, lo, hi)


class _AbsMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, m, one_minus):
        ac = a.contiguous()
        mc = m.contiguous() if m is not None else None
        ctx.save_for_backward(ac, mc)
        ctx.one_minus = one_minus
        return ops.abs_mean_fwd(ac, mc, one_minus).view(())

    @staticmethod
    def backward(ctx, g):
        a, m = ctx.saved_tensors
        return ops.abs_mean_bwd(a, m, ctx.one_minus, g.contiguous().view(1)), None, None


def abs_mean(a, m=None, one_minus=False):
    """mean(|a * w|) with w = m, (1 - m) or 1."""
    return _AbsMean.apply(a, m, one_minus)


class _BCELogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, target_const):
        zc = z.contiguous()
        ctx.save_for_backward(zc)
        ctx.t = target_const
        loss, _ = ops.bce_logits_fwd_bwd(zc, target_const, need_grad=False)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        _, dz = ops.bce_logits_fwd_bwd(z, ctx.t, need_loss=False, grad_out=g.contiguous().view(1))
        return dz, None


class BCEWithLogitsLoss(nn.Module):
    """nn.BCEWithLogitsLoss() against an all-ones / all-zeros target (trainer.py:106-107,117 use ones_like / zeros_like)."""

    def forward(self, input, target):
        t = float(target) if not torch.is_tensor(target) else None
        if t is None:
            raise PcgError("BCEWithLogitsLoss here takes the constant target 0.0 or 1.0 (ones_like/zeros_like in the reference)")
        return _BCELogitsFn.apply(input, t)


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        zc = logits.contiguous()
        ctx.save_for_backward(zc, target)
        loss, _ = ops.cross_entropy_fwd_bwd(zc, target, need_grad=False)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        z, target = ctx.saved_tensors
        _, dz = ops.cross_entropy_fwd_bwd(z, target, need_loss=False, grad_out=g.contiguous().view(1))
        return dz, None


class CrossEntropyLoss(nn.Module):
    def forward(self, input, target):
        return _CEFn.apply(input, target)


# ---- generator -----------------------------------------------------------------------------------------------------
class _ResBlock(nn.Module):
    """generator.py:5-22 (parameter container; executed by ResidualGenerator)."""

    def __init__(self, channels, activation):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(channels)
        self.act = activation
        self.conv2 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm2d(channels)


class _GFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, target, mask, *params):
        raw, masked, saved = net._run_forward(x, target, mask)
        ctx.net, ctx.saved = net, saved
        return raw, masked

    @staticmethod
    def backward(ctx, d_raw, d_masked):
        with ops.slab_reductions_deferred(DEFER_SLAB_REDUCTIONS):
            ctx.net._run_backward(ctx.saved, d_raw, d_masked, any(ctx.needs_input_grad[4:]))
        return (None,) * len(ctx.needs_input_grad)


class ResidualGenerator(FlatModule):
    """generator.py:25-86 — returns (raw_residual, masked_residual)."""

    def __init__(self, img_shape=(1, 28, 28), num_classes=10, base_ch=64, n_resblocks=6, residual_scaling=0.1):
        super().__init__()
        C, H, W = img_shape
        if C != 1:
            raise PcgError("ResidualGenerator: single-channel images only (the reference's MNIST configuration)")
        self.embed = nn.Embedding(num_classes, H * W)
        self.conv_in = nn.Conv2d(C + 2, base_ch, kernel_size=3, padding=1)
        self.act = nn.LeakyReLU(0.2, inplace=True)
        self.resblocks = nn.Sequential(*[_ResBlock(base_ch, self.act) for _ in range(n_resblocks)])
        self.conv_mid = nn.Conv2d(base_ch, base_ch, kernel_size=3, padding=1)
        self.conv_out = nn.Conv2d(base_ch, 1, kernel_size=3, padding=1)
        self.residual_scaling = residual_scaling
        self._hw = (H, W)
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
        if mask is None:
            raise PcgError("ResidualGenerator: a mask is required (the reference only warns and then fails in torch.cat)")
        self._ensure_flat()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raw, masked = _GFn.apply(self, x, target, mask, *self.parameters())
        else:
            raw, masked, _ = self._run_forward(x, target, mask, keep=False)
        return raw, masked

    def _conv_bn(self, conv, bn, a, act, slope, residual=None, alpha=1.0):
        """conv -> BatchNorm (training: statistics fused into the conv epilogue) -> act, y = residual + alpha*act(bn(z))."""
        B, H, W, _ = a.shape
        g = _geom(conv, B, H, W)
        C = bn.num_features
        w, bias = ops.ohwi(conv.weight.data), (conv.bias.data if conv.bias is not None else None)
        if bn.training:
            z, mean, invstd = ops.conv_bn_train(g, a, w, bias, False, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                                bn.num_batches_tracked)
            y = ops.bn_apply_act(z, C, mean, invstd, bn.weight.data, bn.bias.data, act, slope, residual=residual, alpha=alpha)
            return g, z, y, mean, invstd
        z = ops.conv2d_fwd(g, a, w, bias)
        y = ops.bn_apply_act(z, C, bn.running_mean, bn.running_var, bn.weight.data, bn.bias.data, act, slope, var_eps=bn.eps,
                             residual=residual, alpha=alpha)
        return g, z, y, None, None

    def _run_forward(self, x, target, mask, keep=True):
        B = x.shape[0]
        H, W = self._hw
        xr, mr = _as_rows(x, B), _as_rows(mask, B)
        slope = float(self.act.negative_slope)
        inp = ops.embed_concat_fwd(xr, target, self.embed.weight.data, mr).view(B, H, W, 3)
        g_in, h = _conv_fwd(self.conv_in, inp, ACT_LRELU, slope)
        blocks = []
        for blk in self.resblocks:
            g1, z1, a1, m1, s1 = self._conv_bn(blk.conv1, blk.bn1, h, ACT_LRELU, slope)
            g2, z2, hn, m2, s2 = self._conv_bn(blk.conv2, blk.bn2, a1, ACT_NONE, 0.0, residual=h, alpha=0.1)   # x + 0.1*out (:20)
            if keep:
                blocks.append((g1, h, z1, a1, m1, s1, g2, z2, m2, s2))
            h = hn
        g_mid, hm = _conv_fwd(self.conv_mid, h, ACT_LRELU, slope)
        g_out, c = _conv_fwd(self.conv_out, hm)
        raw, masked = ops.scale_mask_fwd(c, mr, self.residual_scaling)
        saved = (inp, g_in, blocks, h, g_mid, hm, g_out, mr, target) if keep else None
        shp = (B, 1, H, W)
        return raw.view(shp), masked.view(shp), saved

    def _run_backward(self, saved, d_raw, d_masked, need_p):
        inp, g_in, blocks, h_last, g_mid, hm, g_out, mr, target = saved
        if not need_p:
            return
        if any(blk.bn1.training is False for blk in self.resblocks):
            raise PcgError("backward through an eval-mode BatchNorm2d is not implemented")
        slope = float(self.act.negative_slope)
        B = inp.shape[0]
        dr = d_raw.contiguous() if d_raw is not None else None
        dm = d_masked.contiguous() if d_masked is not None else None
        dc = ops.scale_mask_bwd(dr, dm, mr, self.residual_scaling, like=mr).view(g_out.B, g_out.OH, g_out.OW, 1)
        _conv_wgrad(self, self.conv_out, g_out, hm, dc, True)
        d = _dgrad_act(g_out, dc, ops.ohwi(self.conv_out.weight.data), hm, ACT_LRELU, slope)     # the mask rides in the thin expand kernel
        order = list(zip(reversed(self.resblocks), reversed(blocks)))
        # adjoint weights of every stride-1 conv whose grad-input runs on the forward kernel: ONE launch for all of them (r04; was one
        # 6 us launch in front of each of the 12 grad-inputs)
        adjw = {}
        if FUSE_BACKWARD_EPILOGUE and S1_DGRAD_AS_FWD:
            convs = [c for blk, sv in order for c, g_ in ((blk.conv2, sv[6]), (blk.conv1, sv[0])) if g_.stride == 1 and ops.xform_ok(g_, "x")]
            if FUSE_SKIP_BNSUM and order and g_mid.stride == 1 and ops.xform_ok(g_mid, "x"):
                convs.append(self.conv_mid)
            shapes = {tuple(c.weight.shape) for c in convs}
            if convs and len(shapes) == 1 and len(convs) <= 16:
                for c, wa in zip(convs, ops.conv_weight_adjoint_many([ops.ohwi(c.weight.data) for c in convs])):
                    adjw[id(c)] = wa

        def adjoint_of(conv):
            wa = adjw.get(id(conv))
            return wa if wa is not None else ops.conv_weight_adjoint(ops.ohwi(conv.weight.data))
        pending = None      # (partial, nparts): bn2's backward sums of the gradient `dh`, left by the previous block's skip-add epilogue
        # conv_mid's backward: weight / bias gradients, then its grad-input `dh` = the gradient of the last block's output — with the column
        # sums that block's bn2 backward needs, out of the same epilogue (r04: the head of the chain takes the no-addend form)
        _conv_wgrad(self, self.conv_mid, g_mid, h_last, d, True)
        head = order[0][1] if order else None
        adjm = FUSE_BACKWARD_EPILOGUE and S1_DGRAD_AS_FWD and FUSE_SKIP_BNSUM and head is not None and g_mid.stride == 1 and ops.xform_ok(g_mid, "x")
        if adjm:
            dh, part, nparts = ops.conv2d_dgrad_add(ops.adjoint_geom(g_mid), d, adjoint_of(self.conv_mid), None,
                                                    bnsum=(head[7], head[8], head[9], 0.1), transposed=True)
            pending = (part, nparts)
        else:
            dh = ops.conv2d_dgrad(g_mid, d, ops.ohwi(self.conv_mid.weight.data))
        h0_masked = False
        for bi, (blk, (g1, h, z1, a1, m1, s1, g2, z2, m2, s2)) in enumerate(order):
            C = blk.bn2.num_features
            dg2, acc = self._grad_view(blk.bn2.weight)
            db2, _ = self._grad_view(blk.bn2.bias)
            # the conv biases in front of the BatchNorms: their gradient is the column sum of dz, taken in the apply pass
            cb2, accb2 = self._grad_view(blk.conv2.bias) if (blk.conv2.bias is not None and FUSE_BIAS_COLSUM) else (None, False)
            if pending is not None:     # sums of 0.1*dh and 0.1*dh*xhat2 came with dh: no reduction pass over (dh, z2)
                dz2 = ops.bn_bwd_partial(dh, z2, C, m2, s2, blk.bn2.weight.data, pending[0], pending[1], dg2, db2, acc, dcol=cb2,
                                         accumulate_col=accb2, dm_scale=0.1)
            else:
                dz2 = ops.bn_act_bwd(dh, z2, None, C, m2, s2, blk.bn2.weight.data, ACT_NONE, 0.0, dg2, db2, acc, dy_scale=0.1,
                                     dcol=cb2, accumulate_col=accb2)
            pending = None
            _conv_wgrad(self, blk.conv2, g2, a1, dz2, bias_done=cb2 is not None)
            dg1, acc = self._grad_view(blk.bn1.weight)
            db1, _ = self._grad_view(blk.bn1.bias)
            # stride-1 layers: the grad-input runs on the FORWARD kernel with the adjoint weight (both operands K-major)
            adj = FUSE_BACKWARD_EPILOGUE and S1_DGRAD_AS_FWD and g2.stride == 1 and ops.xform_ok(g2, "x")
            if adj:
                res = ops.conv_bwd_data_fused(ops.adjoint_geom(g2), dz2, adjoint_of(blk.conv2), True,
                                              ACT_LRELU, slope, z_below=z1, bn=(m1, s1, blk.bn1.weight.data, blk.bn1.bias.data))
            else:
                res = (ops.conv_bwd_data_fused(g2, dz2, ops.ohwi(blk.conv2.weight.data), False, ACT_LRELU, slope, z_below=z1,
                                               bn=(m1, s1, blk.bn1.weight.data, blk.bn1.bias.data)) if FUSE_BACKWARD_EPILOGUE else None)
            cb1, accb1 = self._grad_view(blk.conv1.bias) if (blk.conv1.bias is not None and FUSE_BIAS_COLSUM) else (None, False)
            if res is not None:      # LeakyReLU mask + BatchNorm-backward column sums came out of conv2's grad-input epilogue
                dz1 = ops.bn_bwd_partial(res[0], z1, C, m1, s1, blk.bn1.weight.data, res[1], res[2], dg1, db1, acc, out=res[0],
                                         dcol=cb1, accumulate_col=accb1)
            else:
                da1 = ops.conv2d_dgrad(g2, dz2, ops.ohwi(blk.conv2.weight.data))
                dz1 = ops.bn_act_bwd(da1, z1, None, C, m1, s1, blk.bn1.weight.data, ACT_LRELU, slope, dg1, db1, acc, beta=blk.bn1.bias.data,
                                     dcol=cb1, accumulate_col=accb1)
            _conv_wgrad(self, blk.conv1, g1, h, dz1, bias_done=cb1 is not None)
            if FUSE_BACKWARD_EPILOGUE:   # skip path + block path: the add happens in conv1's grad-input epilogue, in place
                nxt = order[bi + 1][1] if (bi + 1 < len(order) and FUSE_SKIP_BNSUM) else None
                adj1 = S1_DGRAD_AS_FWD and g1.stride == 1 and ops.xform_ok(g1, "x")
                ga, wa = ((ops.adjoint_geom(g1), adjoint_of(blk.conv1)) if adj1
                          else (g1, ops.ohwi(blk.conv1.weight.data)))
                if nxt is not None:      # ... together with the BatchNorm-backward sums the NEXT block's bn2 needs from this sum
                    dh, part, nparts = ops.conv2d_dgrad_add(ga, dz1, wa, dh, out=dh, bnsum=(nxt[7], nxt[8], nxt[9], 0.1), transposed=adj1)
                    pending = (part, nparts)
                elif bi + 1 == len(order):   # the chain's last sum is the gradient w.r.t. h0 = LeakyReLU(conv_in(inp)): its mask in the same epilogue
                    dh = ops.conv2d_dgrad_add_mask(ga, dz1, wa, dh, h, ACT_LRELU, slope, out=dh, transposed=adj1)
                    h0_masked = True
                else:
                    dh = ops.conv2d_dgrad_add(ga, dz1, wa, dh, out=dh, transposed=adj1)
            else:
                dconv = ops.conv2d_dgrad(g1, dz1, ops.ohwi(blk.conv1.weight.data))
                dh = ops.axpby(1.0, dh, 1.0, dconv, out=dconv)
        if not h0_masked:
            ops.act_bwd(dh, blocks[0][1] if blocks else h_last, ACT_LRELU, slope, out=dh)   # h0 = LeakyReLU(conv_in(inp))
        _conv_wgrad(self, self.conv_in, g_in, inp, dh, True)
        ge, acc = self._grad_view(self.embed.weight)
        if GRAD_INPUT_LABEL_CHANNEL_ONLY and g_in.Cin == 3:
            # of conv_in's three input channels (image, label map, mask: generator.py:73-74) only the label map has a consumer — the
            # embedding table.  Its grad-input is the one-channel convolution with that channel's weights: 9 tap products per pixel
            # instead of 27 (r04 census: 118 + 46 us -> 55 + 26)
            g1 = ops.conv_geom(g_in.B, g_in.IH, g_in.IW, 1, g_in.Cout, g_in.KH, g_in.KW, g_in.stride, g_in.pad)
            d1 = ops.conv2d_dgrad(g1, dh, ops.gather_channel(ops.ohwi(self.conv_in.weight.data), 1))
            ops.embed_table_grad(d1, target, 1, 0, self.embed.num_embeddings, ge, accumulate=acc)
        else:
            dinp = ops.conv2d_dgrad(g_in, dh, ops.ohwi(self.conv_in.weight.data))
            ops.embed_concat_bwd(dinp, target, 3, self.embed.num_embeddings, dtable=ge, accumulate=acc)


# ---- discriminator ----------------------------------------------------------------------------------------------------
class _DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, cond_idx, *params):
        out, saved = net._run_forward(x, cond_idx)
        ctx.net, ctx.saved = net, saved
        return out

    @staticmethod
    def backward(ctx, dlogits):
        with ops.slab_reductions_deferred(DEFER_SLAB_REDUCTIONS):
            dx = ctx.net._run_backward(ctx.saved, dlogits, ctx.needs_input_grad[1], any(ctx.needs_input_grad[3:]))
        return (None, dx) + (None,) * (len(ctx.needs_input_grad) - 2)


class Discriminator(FlatModule):
    """discriminator.py:5-38 — logits [B, 1]."""

    def __init__(self, img_shape=(1, 28, 28), num_classes=Config.num_classes):
        super().__init__()
        C, H, W = img_shape
        self.cond_embed = nn.Embedding(num_classes, H * W)
        self.img_channel = 2
        self.d_hidden = 64
        d = self.d_hidden
        self.main = nn.Sequential(
            nn.Conv2d(self.img_channel, d, 3, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d, d * 2, 3, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d * 2, d * 4, 3, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(d * 4, d * 4, 3, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
            nn.AdaptiveAvgPool2d(1),
        )
        self.flatten = nn.Flatten()
        self.adv_head = nn.Linear(d * 4, 1)
        self._hw = (H, W)

    def _convs(self):
        return [m for m in self.main if isinstance(m, nn.Conv2d)]

    def forward(self, x, cond_idx):
        self._ensure_flat()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _DFn.apply(self, x, cond_idx, *self.parameters())
        return self._run_forward(x, cond_idx, keep=False)[0]

    def _run_forward(self, x, cond_idx, keep=True):
        B = x.shape[0]
        H, W = self._hw
        a = ops.embed_concat_fwd(_as_rows(x, B), cond_idx, self.cond_embed.weight.data, None).view(B, H, W, 2)
        layers = []
        for conv in self._convs():
            g, z = _conv_fwd(conv, a, ACT_LRELU, 0.2)
            if keep:
                layers.append((g, a, z))
            a = z
        Bq, Hq, Wq, Cq = a.shape
        pooled = ops.avgpool_fwd(a, B, Hq * Wq, Cq)
        gl = ops.conv_geom(B, 1, 1, Cq, 1, 1, 1, 1, 0)
        logits = ops.conv2d_fwd(gl, pooled, self.adv_head.weight.data, self.adv_head.bias.data).view(B, 1)
        saved = (layers, pooled, gl, (Hq * Wq, Cq), cond_idx) if keep else None
        return logits, saved

    def _run_backward(self, saved, dlogits, need_x, need_p):
        layers, pooled, gl, (HWq, Cq), cond_idx = saved
        B = pooled.shape[0]
        dl = dlogits.contiguous()
        head = self.adv_head
        if need_p and head.weight.requires_grad:
            gw, acc = self._grad_view(head.weight)
            ops.conv2d_wgrad(gl, pooled, dl, gw, acc)
            gb, accb = self._grad_view(head.bias)
            ops.colsum(B, 1, dl, gb, accb)
        dpool = ops.conv2d_dgrad(gl, dl, head.weight.data)
        d = ops.avgpool_bwd(dpool, B, HWq, Cq).view(layers[-1][2].shape)
        convs = self._convs()
        masked = False          # d already carries the LeakyReLU derivative of layer i (applied in layer i+1's grad-input epilogue)
        for i in range(len(convs) - 1, -1, -1):
            g, a, y = layers[i]
            if not masked:
                ops.act_bwd(d, y, ACT_LRELU, 0.2, out=d)
            last = i == 0
            need_in = (not last) or need_x or (need_p and self.cond_embed.weight.requires_grad)
            _conv_wgrad(self, convs[i], g, a, d, need_p)
            masked = False
            if not need_in:
                d = None
                continue
            w = ops.ohwi(convs[i].weight.data)
            if last and not need_x and GRAD_INPUT_LABEL_CHANNEL_ONLY and g.Cin == 2:
                # D(real) / D(fake.detach()): of the entry conv's two input channels (image, label map: discriminator.py:35-36) only
                # the label map has a consumer, the embedding table — 9 tap products per pixel instead of 18
                g1 = ops.conv_geom(g.B, g.IH, g.IW, 1, g.Cout, g.KH, g.KW, g.stride, g.pad)
                d1 = ops.conv2d_dgrad(g1, d, ops.gather_channel(w, 1))
                ge, acc = self._grad_view(self.cond_embed.weight)
                ops.embed_table_grad(d1, cond_idx, 1, 0, self.cond_embed.num_embeddings, ge, accumulate=acc)
                return None
            res = ops.conv_bwd_data_fused(g, d, w, False, ACT_LRELU, 0.2, a_below=a) if (not last and FUSE_BACKWARD_EPILOGUE) else None
            if res is not None:
                d, masked = res[0], True
            else:
                d = ops.conv2d_dgrad(g, d, w)
        if d is None:
            return None
        ge, acc = (self._grad_view(self.cond_embed.weight) if need_p and self.cond_embed.weight.requires_grad else (None, False))
        dx = ops.embed_concat_bwd(d, cond_idx, 2, self.cond_embed.num_embeddings, dtable=ge, accumulate=acc, need_dx=need_x)
        H, W = self._hw
        return dx.view(B, 1, H, W) if need_x else None


# ---- frozen classifier -----------------------------------------------------------------------------------------------
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


class CNNClassifier(FlatModule):
    """classifier.py:4-28.  In the GAN step it is frozen and in eval mode (main.py:30-33): forward and the gradient with
    respect to the input image, on packed weights.  In training mode (pre-training, trainer.py:8-39 — SURVEY.md section 8f
    item 3) it is a regular trainable net: Dropout2d(0.25) / Dropout(0.5) masks come from `self.rng` (a device Philox
    stream) or, for runs that must reproduce given draws, from `self.dropout_masks = [mask2d [B,128], mask [B,256]]`."""

    def __init__(self, num_classes=10):
        super().__init__()
        self.rng = None
        self.dropout_masks = None
        self.conv = nn.Sequential(
            nn.Conv2d(1, 32, 3, 1, 1), nn.ReLU(),
            nn.Conv2d(32, 64, 3, 2, 1), nn.ReLU(),
            nn.Conv2d(64, 128, 3, 2, 1), nn.ReLU(),
            nn.Dropout2d(0.25),
        )
        self.fc = nn.Sequential(nn.Flatten(), nn.Linear(128 * 7 * 7, 256), nn.ReLU(), nn.Dropout(0.5), nn.Linear(256, num_classes))
        self.num_classes = num_classes
        self._packed = None

    def _pack(self):
        """NHWC / 4-aligned images of the frozen weights (rebuilt if the parameters change): conv weights OHWI; fc1 columns
        re-ordered from (c,h,w) to (h,w,c) so it consumes the NHWC activation; fc2 padded from 10 to 12 outputs (16-byte
        rows).  One-time setup copies, not part of the step."""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        if self.training:
            raise PcgError("CNNClassifier: only eval mode is implemented (the GAN step uses the frozen classifier)")
        convs = [m for m in self.conv if isinstance(m, nn.Conv2d)]
        fc1, fc2 = self.fc[1], self.fc[4]
        with torch.no_grad():
            cw = [(c, c.weight.permute(0, 2, 3, 1).contiguous(), c.bias.contiguous()) for c in convs]
            w1 = fc1.weight.view(fc1.out_features, 128, 7, 7).permute(0, 2, 3, 1).contiguous().view(fc1.out_features, -1)
            kp = (self.num_classes + 3) // 4 * 4
            w2 = torch.zeros(kp, fc2.in_features, device=fc2.weight.device)
            w2[: self.num_classes] = fc2.weight
            b2 = torch.zeros(kp, device=fc2.weight.device)
            b2[: self.num_classes] = fc2.bias
        packed = (cw, w1, fc1.bias.contiguous(), w2, b2, kp)
        self._packed = (key, packed)
        return packed

    def train(self, mode=True):
        # the optimizer kernel updates parameters in place without touching torch's version counters: drop the packed eval
        # image whenever the mode changes so that eval after training sees the new weights
        self._packed = None
        return super().train(mode)

    def forward(self, x):
        if not x.is_cuda:
            raise PcgError(f"CNNClassifier: input is on {x.device}; libpcgan_hip has no CPU path")
        if self.training:
            self._ensure_flat()
            B = x.shape[0]
            masks = self.dropout_masks
            if masks is None:
                if self.rng is None:
                    self.rng = ops.DeviceRNG(seed=0)
                masks = [self.rng.bernoulli((B, 128), x.device, 0.75), self.rng.bernoulli((B, 256), x.device, 0.5)]
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                return _CTrainFn.apply(self, x, masks, *self.parameters())
            return self._train_forward(x, masks)[0]
        if torch.is_grad_enabled() and x.requires_grad:
            return _CFn.apply(self, x)
        return self._run_forward(x, keep=False)[0]

    # -- training mode (trainer.py:15-20) ------------------------------------------------------------------------------------
    def _train_forward(self, x, masks):
        from .nn import linear_fwd
        B = x.shape[0]
        a = _as_rows(x, B).view(B, 28, 28, 1)
        layers = []
        for conv in (m for m in self.conv if isinstance(m, nn.Conv2d)):
            g, y = _conv_fwd(conv, a, ACT_RELU)
            layers.append((conv, g, a, y))
            a = y
        m2d, m1 = masks
        HW, C = a.shape[1] * a.shape[2], a.shape[3]
        ad = ops.dropout_apply(a, m2d.contiguous(), 0.25, inner=HW, C=C)                       # Dropout2d(0.25) :13
        flat = ops.nhwc_to_nchw_flat(ad, B, HW, C).view(B, HW * C)                            # nn.Flatten of the NCHW tensor :16
        fc1, fc2 = self.fc[1], self.fc[4]
        h = linear_fwd(fc1, flat)
        ops.act_fwd(h, ACT_RELU, 0.0, out=h)
        hd = ops.dropout_apply(h, m1.contiguous(), 0.5)                                       # Dropout(0.5) :19
        logits = linear_fwd(fc2, hd)
        return logits, (layers, m2d, m1, flat, h, hd, (HW, C))

    def _train_backward(self, saved, dlogits):
        from .nn import linear_dgrad, linear_wgrad
        layers, m2d, m1, flat, h, hd, (HW, C) = saved
        B = flat.shape[0]
        fc1, fc2 = self.fc[1], self.fc[4]
        dl = dlogits.contiguous()
        linear_wgrad(self, fc2, hd, dl)
        d = linear_dgrad(fc2.weight.data, dl, B)
        d = ops.dropout_apply(d, m1.contiguous(), 0.5, out=d)
        ops.act_bwd(d, h, ACT_RELU, 0.0, out=d)
        linear_wgrad(self, fc1, flat, d)
        d = linear_dgrad(fc1.weight.data, d, B)
        d = ops.nhwc_to_nchw_flat(d, B, HW, C, inverse=True).view(layers[-1][3].shape)
        d = ops.dropout_apply(d, m2d.contiguous(), 0.25, inner=HW, C=C, out=d)
        for i in range(len(layers) - 1, -1, -1):
            conv, g, a, y = layers[i]
            ops.act_bwd(d, y, ACT_RELU, 0.0, out=d)
            d = _conv_bwd(self, conv, g, a, d, True, i > 0)

    def _run_forward(self, x, keep=True):
        cw, w1, b1, w2, b2, kp = self._pack()
        B = x.shape[0]
        a = _as_rows(x, B).view(B, 28, 28, 1)
        layers = []
        for conv, w, b in cw:
            Bq, H, W, _ = a.shape
            g = _geom(conv, B, H, W)
            z = ops.conv2d_fwd(g, a, w, b, act=ACT_RELU)
            if keep:
                layers.append((g, w, z))
            a = z
        feat = a.numel() // B
        g1 = ops.conv_geom(B, 1, 1, feat, w1.shape[0], 1, 1, 1, 0)
        h = ops.conv2d_fwd(g1, a.view(B, 1, 1, feat), w1, b1, act=ACT_RELU)
        g2 = ops.conv_geom(B, 1, 1, w1.shape[0], kp, 1, 1, 1, 0)
        logits = ops.conv2d_fwd(g2, h, w2, b2).view(B, kp)[:, : self.num_classes]
        saved = (layers, g1, w1, h, g2, w2, kp) if keep else None
        return logits, saved

    def _run_backward(self, saved, dlogits):
        layers, g1, w1, h, g2, w2, kp = saved
        B = dlogits.shape[0]
        dl = torch.empty((B, kp), dtype=torch.float32, device=dlogits.device)
        ops.fill(dl, 0.0)
        dl[:, : self.num_classes].copy_(dlogits)      # pad 10 -> 12 columns (tiny strided copy)
        d = _dgrad_act(g2, dl, w2, h, ACT_RELU, 0.0)
        d = _dgrad_act(g1, d, w1, layers[-1][2], ACT_RELU, 0.0)
        for idx in range(len(layers) - 1, -1, -1):
            g, w, y = layers[idx]
            d = d.view(y.shape)
            d = _dgrad_act(g, d, w, layers[idx - 1][2], ACT_RELU, 0.0) if idx > 0 else ops.conv2d_dgrad(g, d, w)
        return d.view(B, 1, 28, 28)


# ---- trainer ------------------------------------------------------------------------------------------------------------
def build_mask(x, patch_size, device, num_modifiable_patches=None):
    """trainer.py:45-72 — same draws from torch's RNG as the reference (host loop over the batch; moving this to the
    device is SURVEY.md §8f item 1)."""
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


def build_mask_device(rng, x, patch_size, num_modifiable_patches):
    """build_mask on the GPU (one kernel instead of a host loop of B randperm calls): same distribution — every sample
    gets exactly `num_modifiable_patches` distinct patches, uniformly — from the engine's Philox stream `rng`
    (ops.DeviceRNG) rather than torch's generator."""
    bs, c, h, w = x.shape
    if c != 1:
        raise PcgError("build_mask_device: single-channel images only")
    total = (h // patch_size) * (w // patch_size)
    if num_modifiable_patches is None or num_modifiable_patches >= total:
        raise PcgError("build_mask_device: the Bernoulli-per-patch branch (trainer.py:58-60) is not implemented")
    return rng.patch_mask(bs, h, w, patch_size, num_modifiable_patches, x.device)


def evaluate_counterfactuals(generator, classifier, x, y_true, y_target, device=None):
    """eval_utils.py:46-79 — eval-mode generator with an all-ones mask, clamp to [-1, 1], classifier; returns
    ({class_flip_rate, prediction_gain, actionability}, (x_vis, x_cf_vis) in [0, 1])."""
    classifier.eval(); generator.eval()
    device = device if device is not None else next(generator.parameters()).device
    x, y_true, y_target = x.to(device), y_true.to(device), y_target.to(device)
    with torch.no_grad():
        ones = torch.empty_like(x)
        ops.fill(ones, 1.0)
        residual = generator(x, y_target, ones)[1]                                          # :55
        x_cf = clamp_add(x, residual, -1.0, 1.0)                                            # :57
        logits = classifier(x_cf)                                                           # :58
        m = ops.cf_metrics(logits.contiguous(), y_target, other=y_true).cpu()               # :61-64
        diff = ops.axpby(1.0, x_cf, -1.0, x.contiguous())
        actionability = abs_mean(diff).item()                                               # :66
        x_vis = ops.axpby(0.5, x.contiguous(), out=None).add_(0.5).cpu()                    # :69-70 ([-1,1] -> [0,1]; host-side view)
        x_cf_vis = ops.axpby(0.5, x_cf).add_(0.5).cpu()
    return {"class_flip_rate": float(m[0]), "prediction_gain": float(m[1]), "actionability": actionability}, (x_vis, x_cf_vis)


def generate_counterfactuals(generator, classifier, x, y, y_target, mask, device=None):
    """eval_utils.py:489-497."""
    generator.eval(); classifier.eval()
    with torch.no_grad():
        raw_residual, masked_residual = generator(x, y_target, mask)
        x_cf = clamp_add(x, masked_residual, -1.0, 1.0)
    return raw_residual, masked_residual, x_cf


def train_classifier(classifier, train_loader, valid_loader, cfg, device, save=True):
    """trainer.py:8-39 — Adam(cls_lr), CrossEntropyLoss, per-epoch validation accuracy, best state saved to
    cfg.classifier_path."""
    optimizer = Adam(classifier.parameters(), lr=cfg.cls_lr)
    criterion = CrossEntropyLoss()
    best_acc, history = 0.0, []
    for epoch in range(cfg.num_epochs_clf):
        classifier.train()
        for x, y in train_loader:
            x, y = x.to(device), y.to(device)
            optimizer.zero_grad()
            loss = criterion(classifier(x), y)
            loss.backward()
            optimizer.step()
        classifier.eval()
        correct, total = 0.0, 0
        with torch.no_grad():
            for x, y in valid_loader:
                x, y = x.to(device), y.to(device)
                acc_b = ops.cf_metrics(classifier(x).contiguous(), y, other=y)[0].item()     # mean(argmax == y)
                correct += acc_b * y.size(0)
                total += y.size(0)
        acc = correct / max(total, 1)
        history.append(acc)
        print(f"[Classifier] Epoch {epoch + 1}/{cfg.num_epochs_clf} | Val Acc: {acc:.4f}")
        if acc > best_acc:
            best_acc = acc
            if save and getattr(cfg, "classifier_path", None):
                torch.save({k: v.detach().cpu().contiguous() for k, v in classifier.state_dict().items()}, cfg.classifier_path)
    return history


def grad_norm(net):
    """trainer.py:41-42 — sqrt(sum ||p.grad||^2) over a net's parameters (per-epoch diagnostic, :142-143): one launch over
    the flat gradient buffer (its alignment padding is zero), one host read."""
    out = torch.empty(1, dtype=torch.float32, device=net.flat_grads.device)
    ops.sumsq(net.flat_grads, out)
    return float(out.item()) ** 0.5


def make_optimizers(generator, discriminator, cfg=Config):
    """trainer.py:77-80."""
    opt_g = Adam(generator.parameters(), lr=cfg.g_lr)
    opt_d = Adam(discriminator.parameters(), lr=cfg.d_lr)
    return opt_g, opt_d, BCEWithLogitsLoss(), CrossEntropyLoss()


def train_step(generator, discriminator, classifier, opt_g, opt_d, bce, ce, x, y, target_y, mask, cfg=Config, dp=None,
               skip_dead_d_wgrad=True):
    """One iteration of train_countergan's loop body (trainer.py:89-123) for a batch already on the GPU; `target_y`
    (:94) and `mask` (:95) are passed in.  Returns device tensors; the reference's `.item()` calls are the caller's.

    dp (parallel.GradSync): D's bucket is averaged in stream order (Adam(D) needs it at once); G's bucket + Adam(G) run on the
    side stream and overlap with the NEXT iteration's D(real) forward — which is therefore issued before the generator forward
    when data-parallel (it reads only x, y and D's weights; D has no BatchNorm, so the hoist changes no number)."""
    d_real_logits = None
    if dp is not None:
        d_real_logits = discriminator(x, y)                                            # :103, hoisted above the wait
        dp.wait(generator)                                                             # previous iteration's G all-reduce + Adam(G)
    raw_residual, masked_residual = generator(x, target_y, mask)                       # :96
    x_cf = clamp_add(x, masked_residual, -1.0, 1.0)                                    # :97
    mask_penalty_pre = abs_mean(raw_residual, mask, one_minus=True)                    # :99
    # Discriminator update
    opt_d.zero_grad()                                                                  # :102
    if d_real_logits is None:
        d_real_logits = discriminator(x, y)                                            # :103
    d_fake_logits = discriminator(x_cf.detach(), target_y)                             # :104
    d_loss = weighted_sum([bce(d_real_logits, 1.0), bce(d_fake_logits, 0.0)], [1.0, 1.0])   # :106-107 (one launch each way, no ATen add)
    d_loss.backward()                                                                  # :111
    if dp is not None:
        dp.sync_now(discriminator)
    opt_d.step()                                                                       # :112
    # Generator update
    opt_g.zero_grad()                                                                  # :115
    if skip_dead_d_wgrad:
        for p in discriminator.parameters():
            p.requires_grad_(False)
    try:
        g_fake_logits = discriminator(x_cf, target_y)                                  # :116
        g_adv = bce(g_fake_logits, 1.0)                                                # :117
        g_cls = ce(classifier(x_cf), target_y)                                         # :118
        reg_l1 = abs_mean(masked_residual)                                             # :119
        g_loss = weighted_sum([g_adv, g_cls, reg_l1, mask_penalty_pre],
                              [cfg.lambda_adv, cfg.lambda_cls, cfg.lambda_reg, cfg.lambda_mask])   # :121 (was seven ATen mul / add launches)
        g_loss.backward()                                                              # :122
    finally:
        if skip_dead_d_wgrad:
            for p in discriminator.parameters():
                p.requires_grad_(True)
    if dp is not None:
        dp.sync_then(generator, opt_g.step)                                            # overlapped with the next D(real) forward
    else:
        opt_g.step()                                                                   # :123
    return {"d_loss": d_loss, "g_loss": g_loss, "g_adv": g_adv, "g_cls": g_cls, "reg_l1": reg_l1, "mask_pen": mask_penalty_pre,
            "d_real_logits": d_real_logits, "d_fake_logits": d_fake_logits, "x_cf": x_cf}


def _lookahead(it):
    """(item, is_last) over any iterable — the loader is `any iterable of (x, y)`, it need not have a length."""
    it = iter(it)
    try:
        prev = next(it)
    except StopIteration:
        return
    for cur in it:
        yield prev, False
        prev = cur
    yield prev, True


def train_countergan(generator, discriminator, classifier, train_loader, cfg, device, log_every=100, device_rng=None, draws=None,
                     save=True, verbose=True):
    """conditional_counteRGAN/mnist/trainer.py:76-163 `train_countergan(generator, discriminator, classifier, train_loader, cfg,
    device)`: the same loop, the same per-epoch means (:139-141), the per-epoch `G_grad` / `D_grad` line (:142-147 — the one place
    grad_norm :41-42 is used), `residual_mean` in the batch log line (:137) and the generator checkpoint (:162); the loss-curve PNG
    (:149-160) is plotting and is left out.

    D's .grad at the epoch summary = the D step's gradients + the generator step's critic weight gradients (no zeroing between :111
    and :122): the LAST iteration of an epoch runs with skip_dead_d_wgrad=False, every other one skips that dead work.
    Host reads: the reference calls .item() five times per iteration; here the per-iteration scalars stay on the device and are
    read once per epoch (and at the log lines), summed in the same order.
    device_rng: an ops.DeviceRNG — draw targets and masks on the GPU instead of torch's RNG;
    draws(epoch, batch_idx, x) -> (target_y, mask): supplied draws (parity runs).  Returns the per-epoch history:
    {"g_losses", "d_losses", "g_cls_losses", "G_grad", "D_grad"}."""
    opt_g, opt_d, bce, ce = make_optimizers(generator, discriminator, cfg)                     # :77-80
    hist = {k: [] for k in ("g_losses", "d_losses", "g_cls_losses", "G_grad", "D_grad")}
    for epoch in range(cfg.num_epochs_gan):                                                    # :84
        pending = []
        for batch_idx, ((x, y), last) in enumerate(_lookahead(train_loader)):                  # :89
            x, y = x.to(device), y.to(device)                                                  # :90
            bs = x.size(0)
            if draws is not None:
                target_y, mask = (t.to(device) for t in draws(epoch, batch_idx, x))
            elif device_rng is not None:
                target_y = device_rng.randint(0, cfg.num_classes, bs, x.device)                # :94
                mask = build_mask_device(device_rng, x, cfg.patch_size, cfg.num_modifiable_patches)   # :95
            else:
                target_y = torch.randint(0, cfg.num_classes, (bs,), device=device)             # :94
                mask = build_mask(x, cfg.patch_size, device, cfg.num_modifiable_patches)       # :95
            out = train_step(generator, discriminator, classifier, opt_g, opt_d, bce, ce, x, y, target_y, mask, cfg,
                             skip_dead_d_wgrad=not last)                                        # :96-123
            # :130-132, read at the epoch's end.  Detached: a scalar with its grad_fn would keep every custom node's saved
            # activations (ctx.saved is a plain attribute, not released by backward) alive until the epoch ends — 0.67 GB per
            # iteration at the reference batch
            pending.append((out["g_loss"].detach(), out["d_loss"].detach(), out["g_cls"].detach()))
            if verbose and batch_idx % log_every == 0:                                         # :134-137
                reg = out["reg_l1"].item()
                print(f"[Epoch {epoch + 1}/{cfg.num_epochs_gan}] batch {batch_idx} :: "
                      f"D(real)={torch.sigmoid(out['d_real_logits']).mean().item():.3f}, "
                      f"D(fake)={torch.sigmoid(out['d_fake_logits']).mean().item():.3f}, g_adv={out['g_adv'].item():.4f}, "
                      f"g_cls={out['g_cls'].item():.4f}, reg={reg:.6f}, residual_mean={reg:.4f}")   # masked_residual.abs().mean() IS reg_l1 (:119)
        n = len(pending)
        if n == 0:
            raise PcgError("train_countergan: the loader yielded no batch")
        g_epoch = d_epoch = cls_epoch = 0.0                                                    # same order of additions as :130-132
        for g_l, d_l, c_l in pending:
            g_epoch += g_l.item(); d_epoch += d_l.item(); cls_epoch += c_l.item()
        hist["g_losses"].append(g_epoch / n); hist["d_losses"].append(d_epoch / n); hist["g_cls_losses"].append(cls_epoch / n)   # :139-141
        hist["G_grad"].append(grad_norm(generator)); hist["D_grad"].append(grad_norm(discriminator))                           # :142-143
        if verbose:
            print(f"[GAN] Epoch {epoch + 1}/{cfg.num_epochs_gan} | G: {hist['g_losses'][-1]:.4f}, D: {hist['d_losses'][-1]:.4f}, "
                  f"G_cls: {hist['g_cls_losses'][-1]:.4f}, G_grad: {hist['G_grad'][-1]:.4f}, D_grad: {hist['D_grad'][-1]:.4f}")   # :145-147
    if save and getattr(cfg, "generator_path", None):
        import os
        os.makedirs(os.path.dirname(os.path.abspath(cfg.generator_path)) or ".", exist_ok=True)
        torch.save({k: v.detach().cpu().contiguous() for k, v in generator.state_dict().items()}, cfg.generator_path)         # :162
        if verbose:
            print(f"Generator saved to {cfg.generator_path}")
    return hist

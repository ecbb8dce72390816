"""Host-side mirror of `conditional_gan/mnist/mnist_wgan_conditional.py` (conditional WGAN-GP) on the HIP kernels.

    reference                                      here
    -------------------------------------------    ------------------------------------------------------------------
    Hyperparameter :21-31                          Hyperparameter
    Generator :51-78                               Generator (same state_dict keys: latent_embedding.0.*, condition_embedding.0.*,
                                                   tcnn.{0,1,3,4,6,7,9}.*)
    Critic :80-108                                 Critic (condition_embedding.0.*, cnn_net.{0,1,3,4,6,7}.*, Critic_net.{0,2}.*)
    optimizers :118-119                            make_optimizers (AdamW lr 1e-4, betas (0, 0.9))
    critic update :133-155                         critic_step       (gradient penalty :146-150 = gradient_penalty / interpolate)
    generator update :157-168                      generator_step
    loop :129-168                                  train (draws on the device; graphed=True replays GraphedSteps)
    -                                              GraphedSteps: both updates captured once as HIP graphs, one host call each

The gradient penalty needs the gradient of the critic output with respect to its input image as a DIFFERENTIABLE quantity
(`autograd.grad(..., create_graph=True)`, :149) and then `critic_loss.backward()` (:154) differentiates through it.  The
critic is one autograd node whose backward is itself an autograd node (`_CriticGradFn`); that node's backward is the
hand-derived backward-of-backward: per layer, the transposed-convolution/convolution pair swaps roles, InstanceNorm
contributes the three cotangents of pcg_instnorm_bwd_bwd, LeakyReLU masks are piecewise constant, and the cotangents that
land on forward activations (only InstanceNorm's backward depends on them) are swept down through the forward graph once.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_LRELU, PcgError
from .nn import FlatModule, GraphedStep, SequentialConvNet, _compile, linear_dgrad, linear_fwd, linear_wgrad, mean, weighted_sum
from .optim import AdamW


@dataclass
class Hyperparameter:
    """:21-31 (data_path / num_epochs belong to the input pipeline, not the step)."""
    num_classes: int = 10
    batchsize: int = 128
    num_epochs: int = 20
    latent_size: int = 32
    n_critic: int = 5
    critic_size: int = 1024
    generator_size: int = 1024
    critic_hidden_size: int = 1024
    gp_lambda: float = 10.0


# ---- generator ----------------------------------------------------------------------------------------------------------
class _GFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, latent, condition, *params):
        y, saved = net._gen_forward(latent, condition, keep=True)
        ctx.net, ctx.saved = net, saved
        return y

    @staticmethod
    def backward(ctx, dy):
        ctx.net._gen_backward(ctx.saved, dy, any(ctx.needs_input_grad[3:]))
        return (None,) * len(ctx.needs_input_grad)


class Generator(SequentialConvNet):
    """:51-78 — two embeddings (Linear) concatenated into a [B, generator_size, 1, 1] seed, then ConvT/BN/ReLU x3, ConvT, Tanh."""

    def __init__(self, hp=None):
        FlatModule.__init__(self)
        hp = hp if hp is not None else Hyperparameter()
        gs = hp.generator_size
        self.hp = hp
        self.latent_embedding = nn.Sequential(nn.Linear(hp.latent_size, gs // 2))
        self.condition_embedding = nn.Sequential(nn.Linear(hp.num_classes, gs // 2))
        self.tcnn = nn.Sequential(
            nn.ConvTranspose2d(gs, gs, 4, 1, 0), nn.BatchNorm2d(gs), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(gs, gs // 2, 3, 2, 1), nn.BatchNorm2d(gs // 2), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(gs // 2, gs // 4, 4, 2, 1), nn.BatchNorm2d(gs // 4), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(gs // 4, 1, 4, 2, 1), nn.Tanh())
        self._blocks = None

    @property
    def main(self):
        return self.tcnn

    def forward(self, latent, condition):
        self._ensure_flat()
        if self._blocks is None:
            self._blocks = _compile(self.tcnn)
        if not latent.is_cuda:
            raise PcgError(f"Generator: input is on {latent.device}; libpcgan_hip has no CPU path")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _GFn.apply(self, latent, condition, *self.parameters())
        return self._gen_forward(latent, condition, keep=False)[0]

    def _gen_forward(self, latent, condition, keep):
        latent, condition = latent.contiguous(), condition.contiguous()
        B = latent.shape[0]
        vec_latent = linear_fwd(self.latent_embedding[0], latent)                       # :73
        vec_class = linear_fwd(self.condition_embedding[0], condition)                  # :74
        seed = ops.concat_cols(vec_latent, vec_class).view(B, 1, 1, self.hp.generator_size)   # :75-76
        y, saved = SequentialConvNet._run_forward(self, seed, keep=keep)                # :77
        return y.permute(0, 3, 1, 2), ((latent, condition, saved) if keep else None)    # one channel: NHWC memory == NCHW memory

    def _gen_backward(self, saved, dy, need_p):
        if not need_p:
            return
        latent, condition, seq = saved
        B = latent.shape[0]
        half = self.hp.generator_size // 2
        d = SequentialConvNet._run_backward(self, seq, dy.permute(0, 2, 3, 1), True, True).view(B, 2 * half)
        d_latent, d_class = ops.split_cols(d, half, half)
        linear_wgrad(self, self.latent_embedding[0], latent, d_latent)
        linear_wgrad(self, self.condition_embedding[0], condition, d_class)


# ---- critic -------------------------------------------------------------------------------------------------------------
class _CriticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, image, condition, *params):
        out, saved = net._run_forward(image, condition, keep=True)
        ctx.net, ctx.saved = net, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        n_in = len(ctx.needs_input_grad)
        if torch.is_grad_enabled():
            # create_graph=True (:149): the caller wants d(out)/d(image) as a differentiable tensor.  Parameter gradients are
            # not produced on this path — autograd.grad(..., only_inputs=True) would discard them anyway.
            if not ctx.needs_input_grad[1]:
                raise PcgError("Critic: create_graph=True is implemented for the gradient with respect to the image (WGAN-GP)")
            dx = _CriticGradFn.apply(net, ctx.saved, dout, *net.parameters())
            return (None, dx) + (None,) * (n_in - 2)
        dx, _ = net._run_backward(ctx.saved, dout, ctx.needs_input_grad[1], any(ctx.needs_input_grad[3:]), keep=False)
        return (None, dx) + (None,) * (n_in - 2)


class _CriticGradFn(torch.autograd.Function):
    """image-gradient of the critic as a function of (dout, parameters): forward = the critic's backward sweep (keeping
    its intermediates), backward = the backward-of-backward."""

    @staticmethod
    def forward(ctx, net, saved, dout, *params):
        dx, first = net._run_backward(saved, dout, True, False, keep=True)
        ctx.net, ctx.saved, ctx.first = net, saved, first
        return dx

    @staticmethod
    def backward(ctx, r):
        if torch.is_grad_enabled():
            raise PcgError("Critic: third-order differentiation is not implemented")
        ctx.net._run_double_backward(ctx.saved, ctx.first, r)
        return (None,) * (3 + len(list(ctx.net.parameters())))


class Critic(FlatModule):
    """:80-108 — three Conv(k3,s2)/InstanceNorm/LeakyReLU stages on the image, a Linear embedding of the condition, and a
    two-layer head on their concatenation.  Output [B, 1] (no sigmoid: Wasserstein critic)."""

    def __init__(self, hp=None):
        super().__init__()
        hp = hp if hp is not None else Hyperparameter()
        cs = hp.critic_size
        self.hp = hp
        self.condition_embedding = nn.Sequential(nn.Linear(hp.num_classes, cs * 4))
        self.cnn_net = nn.Sequential(
            nn.Conv2d(1, cs // 4, 3, 2), nn.InstanceNorm2d(cs // 4, affine=True), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(cs // 4, cs // 2, 3, 2), nn.InstanceNorm2d(cs // 2, affine=True), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(cs // 2, cs, 3, 2), nn.InstanceNorm2d(cs, affine=True), nn.LeakyReLU(0.2, inplace=True),
            nn.Flatten())
        self.Critic_net = nn.Sequential(nn.Linear(cs * 8, hp.critic_hidden_size), nn.LeakyReLU(0.2, inplace=True),
                                        nn.Linear(hp.critic_hidden_size, 1))

    def _stages(self):
        m = self.cnn_net
        return [(m[0], m[1]), (m[3], m[4]), (m[6], m[7])]

    def forward(self, image, condition):
        self._ensure_flat()
        if not image.is_cuda:
            raise PcgError(f"Critic: input is on {image.device}; libpcgan_hip has no CPU path")
        if torch.is_grad_enabled() and (image.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _CriticFn.apply(self, image, condition, *self.parameters())
        return self._run_forward(image, condition, keep=False)[0]

    # -- forward ---------------------------------------------------------------------------------------------------------------
    def _run_forward(self, image, condition, keep=True):
        B, _, H, W = image.shape
        a = image.contiguous().view(B, H, W, 1)                                          # one channel: NCHW memory == NHWC memory
        condition = condition.contiguous()
        stages = []
        for conv, inorm in self._stages():
            g = ops.conv_geom(B, a.shape[1], a.shape[2], conv.in_channels, conv.out_channels, 3, 3, 2, 0)
            z = ops.conv2d_fwd(g, a, ops.ohwi(conv.weight.data), conv.bias.data)
            y, mean_, invstd = ops.instnorm_fwd(z, B, g.OH * g.OW, g.Cout, inorm.weight.data, inorm.bias.data, inorm.eps, ACT_LRELU, 0.2)
            if keep:
                stages.append((g, a, z, mean_, invstd, y))
            a = y
        HW, C = a.shape[1] * a.shape[2], a.shape[3]
        feats = ops.nhwc_to_nchw_flat(a, B, HW, C).view(B, HW * C)                       # nn.Flatten of the NCHW tensor (:96)
        vec_condition = linear_fwd(self.condition_embedding[0], condition)              # :105
        u = ops.concat_cols(feats, vec_condition)                                        # :107
        h = linear_fwd(self.Critic_net[0], u)
        ops.act_fwd(h, ACT_LRELU, 0.2, out=h)
        out = linear_fwd(self.Critic_net[2], h)                                          # :108
        saved = (stages, condition, u, h, (HW, C)) if keep else None
        return out, saved

    # -- first-order backward (also the forward of _CriticGradFn when keep=True) ------------------------------------------------------
    def _run_backward(self, saved, dout, need_x, need_p, keep=False):
        stages, condition, u, h, (HW, C) = saved
        B = u.shape[0]
        dout = dout.contiguous()
        l1, l2 = self.Critic_net[0], self.Critic_net[2]
        if need_p:
            linear_wgrad(self, l2, h, dout)
        dh = linear_dgrad(l2.weight.data, dout, B)
        dp = ops.act_bwd(dh, h, ACT_LRELU, 0.2, out=dh)
        if need_p:
            linear_wgrad(self, l1, u, dp)
        du = linear_dgrad(l1.weight.data, dp, B)
        nf = HW * C
        d_feats, d_cond = ops.split_cols(du, nf, u.shape[1] - nf, need_b=need_p)
        if need_p:
            linear_wgrad(self, self.condition_embedding[0], condition, d_cond)
        d = ops.nhwc_to_nchw_flat(d_feats, B, HW, C, inverse=True).view(stages[-1][5].shape)
        first = []
        stage_mods = self._stages()
        for i in range(len(stages) - 1, -1, -1):
            g, a, z, mean_, invstd, y = stages[i]
            conv, inorm = stage_mods[i]
            # LeakyReLU' -> InstanceNorm' in one launch, with the per-sample partials of dgamma, dbeta and the conv bias gradient
            dz, dn, dgp, dbp, dsp = ops.instnorm_bwd_fused(d, z, B, g.OH * g.OW, g.Cout, mean_, invstd, inorm.weight.data, act_y=y, slope=0.2,
                                                           keep_dn=keep, need_params=need_p, need_dxsum=need_p)
            if need_p:
                self._stage_vector_grads(inorm, conv, dgp, dbp, dsp, B)
                self._conv_param_grads(conv, g, a, dz, bias=False)
            if keep:
                first.append((dn, dz))
            if i == 0 and not need_x:
                d = None
                break
            d = ops.conv2d_dgrad(g, dz, ops.ohwi(conv.weight.data))
        dx = d.view(B, 1, d.shape[1], d.shape[2]) if d is not None else None
        return dx, ((first[::-1], dp, dout) if keep else None)

    def _stage_vector_grads(self, inorm, conv, dgp, dbp, dsp, B):
        """dgamma, dbeta (InstanceNorm affine) and the conv bias gradient from their [B, C] per-sample partials: one launch."""
        items = []
        for prm, part in ((inorm.weight, dgp), (inorm.bias, dbp), (conv.bias, dsp)):
            if part is not None:
                gv, acc = self._grad_view(prm)
                items.append((part, gv, acc))
        if items:
            ops.rowsum3(items, B, inorm.num_features)

    def _conv_param_grads(self, conv, g, a, dz, x_side=None, bias=True):
        gw, acc = self._grad_view(conv.weight)
        ops.conv2d_wgrad(g, a if x_side is None else x_side, dz, ops.ohwi(gw), acc)
        if x_side is None and bias:
            gb, accb = self._grad_view(conv.bias)
            ops.colsum(dz.numel() // g.Cout, g.Cout, dz, gb, accb)

    # -- backward of the backward ---------------------------------------------------------------------------------------------------
    def _run_double_backward(self, saved, first, r):
        """r: cotangent on dx = d(out)/d(image).  Accumulates d<r, dx>/d(parameters) into the flat gradient buffer."""
        stages, condition, u, h, (HW, C) = saved
        firsts, dp, dout = first
        B = u.shape[0]
        stage_mods = self._stages()
        rl = r.contiguous().view(stages[0][1].shape)                                   # cotangent on the image-side tensor of stage 1
        ezs = []
        # up: dx_{l-1} = dgrad_l(dz_l; W_l) is bilinear in (dz_l, W_l); dz_l = IN_bwd(dn_l; z_l, gamma_l); dn_l = da_l * lrelu'(.)
        for i, (g, a, z, mean_, invstd, y) in enumerate(stages):
            conv, inorm = stage_mods[i]
            dn, dz = firsts[i]
            ddz = ops.conv2d_fwd(g, rl, ops.ohwi(conv.weight.data), None)              # adjoint of dgrad in dz
            self._conv_param_grads(conv, g, a, dz, x_side=rl)                          # d/dW of <r, dgrad(dz; W)>
            rl, ez, dgp = ops.instnorm_bwd_bwd(ddz, dn, z, B, g.OH * g.OW, g.Cout, mean_, invstd, inorm.weight.data, act_y=y, slope=0.2)
            self._stage_vector_grads(inorm, conv, dgp, None, None, B)                  # (rl: the cotangent on da_l, LeakyReLU' applied)
            ezs.append(ez)
        # head: da_3 = unflatten(du[:, :nf]); du = dp W1; dp = dh * lrelu'(p); dh = dout W2
        l1, l2 = self.Critic_net[0], self.Critic_net[2]
        nf = HW * C
        ddf = ops.nhwc_to_nchw_flat(rl, B, HW, C).view(B, nf)
        zeros = torch.empty((B, u.shape[1] - nf), dtype=torch.float32, device=u.device)
        ops.fill(zeros, 0.0)
        ddu = ops.concat_cols(ddf, zeros)
        ddp = linear_fwd(l1, ddu, use_bias=False)                                        # cotangent on dp
        linear_wgrad(self, l1, ddu, dp, use_bias=False)                                  # dW1 += dp^T ddu
        ddh = ops.act_bwd(ddp, h, ACT_LRELU, 0.2, out=ddp)
        linear_wgrad(self, l2, ddh, dout, use_bias=False)                                # dW2 += dout^T ddh
        # down: the cotangents ez_l sit on forward activations z_l — ordinary backward through the forward graph
        e = ezs[-1]
        bias_done = False                          # the last stage's e comes from instnorm_bwd_bwd: its bias gradient is a colsum of its own
        for i in range(len(stages) - 1, -1, -1):
            g, a, z, mean_, invstd, y = stages[i]
            conv, inorm = stage_mods[i]
            self._conv_param_grads(conv, g, a, e, bias=not bias_done)
            if i == 0:
                break
            gp_, ap, zp, meanp, invstdp, yp = stages[i - 1]
            convp, inormp = stage_mods[i - 1]
            da = ops.conv2d_dgrad(g, e, ops.ohwi(conv.weight.data))
            # e_{l-1} = IN'(lrelu'(da)) + ez_{l-1}, and the per-sample partial of its bias gradient, in one launch
            e, _, dgp, dbp, esp = ops.instnorm_bwd_fused(da, zp, B, gp_.OH * gp_.OW, gp_.Cout, meanp, invstdp, inormp.weight.data, act_y=yp, slope=0.2,
                                                         addend=ezs[i - 1], need_params=True, need_dxsum=True)
            self._stage_vector_grads(inormp, convp, dgp, dbp, esp, B)
            bias_done = True


# ---- gradient penalty -------------------------------------------------------------------------------------------------------
class _GradientPenaltyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gradients, lam):
        B = gradients.shape[0]
        g = gradients.contiguous()
        pen, norms = ops.gradient_penalty_fwd(g, B, lam)
        ctx.g, ctx.norms, ctx.lam = g, norms, lam
        return pen.view(())

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.g
        return ops.gradient_penalty_bwd(g, ctx.norms, grad_out.contiguous(), g.shape[0], ctx.lam).view(g.shape), None


def gradient_penalty(gradients, gp_lambda):
    """gp_lambda * ((gradients.view(B, -1).norm(dim=1) - 1) ** 2).mean()   (:150) as one fused op."""
    return _GradientPenaltyFn.apply(gradients, float(gp_lambda))


# ---- trainer ----------------------------------------------------------------------------------------------------------------
def make_optimizers(critic, generator):
    """:118-119."""
    return (AdamW(critic.parameters(), lr=1e-4, betas=(0.0, 0.9)), AdamW(generator.parameters(), lr=1e-4, betas=(0.0, 0.9)))


def _ones_like_out(critic, B, device):
    t = getattr(critic, "_grad_tensor", None)
    if t is None or t.shape[0] != B or t.device != device:
        t = torch.empty((B, 1), dtype=torch.float32, device=device)
        ops.fill(t, 1.0)
        critic._grad_tensor = t                                                          # grad_tensor (:126)
    return t


def _sub_saved(saved, lo, hi):
    """The saved forward state of rows lo..hi of a batched critic pass (all tensors are batch-major: contiguous slices)."""
    stages, condition, u, h, hwc = saved
    B = u.shape[0]
    sub = []
    for g, a, z, mean_, invstd, y in stages:
        g2 = ops.conv_geom(hi - lo, g.IH, g.IW, g.Cin, g.Cout, g.KH, g.KW, g.stride, g.pad)
        C = g.Cout
        sub.append((g2, a[lo:hi], z[lo:hi], mean_.view(B, C)[lo:hi].reshape(-1), invstd.view(B, C)[lo:hi].reshape(-1), y[lo:hi]))
    return sub, condition[lo:hi], u[lo:hi], h[lo:hi], hwc


def critic_step_batched(critic, generator, critic_optimizer, hp, real_images, real_class_labels, noise, alpha, dp=None):
    """The critic update (:133-155) with the three critic passes (real, fake, interpolates) run as ONE batch of 3B rows: the
    critic has no BatchNorm (InstanceNorm statistics are per sample), so the passes are independent per row and use the same
    weights; the small-M layers (conv3: M = 4 rows per sample) fill the GPU three times better.  Then: first-order backward for the
    real + fake rows (cotangents -1/B and +1/B), image-gradient sweep + gradient penalty + backward-of-backward for the
    interpolates rows.  No autograd graph is built; gradients accumulate into the critic's flat buffer as in critic_step."""
    B = real_images.shape[0]
    dev = real_images.device
    critic._ensure_flat()
    critic.drop_grads()                                                                  # :136 — every critic parameter receives a gradient below: the first writer overwrites (no fill)
    with torch.no_grad():
        fake_image = generator(noise, real_class_labels)                                 # :142
        real_c, fake_c = real_images.contiguous(), fake_image.contiguous()
        x3 = ops.interpolate_stack(alpha.contiguous(), real_c, fake_c)                   # :147 and the batch-major staging of the three passes, one launch
        c3 = torch.cat([real_class_labels] * 3, 0).contiguous()
        out3, saved = critic._run_forward(x3, c3, keep=True)                             # :138, :143, :148 in one pass
        loss_real, loss_fake = ops.mean_fwd(out3[:B].contiguous()), ops.mean_fwd(out3[B:2 * B].contiguous())   # :139, :144
        cot = getattr(critic, "_rf_cot", None)
        if cot is None or cot.shape[0] != 2 * B or cot.device != dev:
            cot = torch.empty((2 * B, 1), dtype=torch.float32, device=dev)
            ops.fill(cot[:B], -1.0 / B); ops.fill(cot[B:], 1.0 / B)                       # d(-mean(real) + mean(fake)) / d(out)
            critic._rf_cot = cot
        critic._run_backward(_sub_saved(saved, 0, 2 * B), cot, False, True, keep=False)
        sv_i = _sub_saved(saved, 2 * B, 3 * B)
        gradients, first = critic._run_backward(sv_i, _ones_like_out(critic, B, dev), True, False, keep=True)   # :149
        gp, norms = ops.gradient_penalty_fwd(gradients.contiguous(), B, hp.gp_lambda)    # :150
        r = ops.gradient_penalty_bwd(gradients.contiguous(), norms, None, B, hp.gp_lambda)
        critic._run_double_backward(sv_i, first, r.view(gradients.shape))                # :154 (the penalty's part)
        critic_loss = ops.weighted_sum_fwd([loss_real, loss_fake, gp], [-1.0, 1.0, 1.0])  # :152
    if dp is not None:
        dp.sync_now(critic)
    critic_optimizer.step()                                                              # :155
    return {"critic_loss": critic_loss.view(()), "loss_real": loss_real.view(()), "loss_fake": loss_fake.view(()),
            "gradient_penalty": gp.view(()), "gradients": gradients, "fake_image": fake_image}


def critic_step(critic, generator, critic_optimizer, hp, real_images, real_class_labels, noise, alpha, dp=None, batched=True):
    """:133-155 for a batch already on the GPU; the draws (`noise` :141, `alpha` :146) are inputs.  Returns device tensors.
    batched=True: critic_step_batched (same result, one critic pass of 3B rows); False: statement by statement through autograd,
    exactly as the reference's loop body reads."""
    if batched:
        return critic_step_batched(critic, generator, critic_optimizer, hp, real_images, real_class_labels, noise, alpha, dp)
    B = real_images.shape[0]
    critic_optimizer.zero_grad()                                                         # :136
    critic_loss_real = mean(critic(real_images, real_class_labels))                      # :138-139
    with torch.no_grad():
        fake_image = generator(noise, real_class_labels)                                 # :142
    critic_loss_fake = mean(critic(fake_image, real_class_labels))                       # :143-144
    interpolates = ops.interpolate(alpha.contiguous(), real_images.contiguous(), fake_image.contiguous()).requires_grad_(True)   # :147
    d_interpolates = critic(interpolates, real_class_labels)                             # :148
    gradients = torch.autograd.grad(d_interpolates, interpolates, _ones_like_out(critic, B, real_images.device),
                                    create_graph=True, only_inputs=True)[0]              # :149
    gp = gradient_penalty(gradients, hp.gp_lambda)                                       # :150
    critic_loss = weighted_sum([critic_loss_real, critic_loss_fake, gp], [-1.0, 1.0, 1.0])   # :152
    critic_loss.backward()                                                               # :154
    if dp is not None:
        dp.sync_now(critic)
    critic_optimizer.step()                                                              # :155
    return {"critic_loss": critic_loss, "loss_real": critic_loss_real, "loss_fake": critic_loss_fake, "gradient_penalty": gp,
            "gradients": gradients, "fake_image": fake_image}


def generator_step(critic, generator, generator_optimizer, fake_class_labels, noise, dp=None, skip_dead_critic_wgrad=True):
    """:157-168 (labels :161 and noise :162 supplied).  skip_dead_critic_wgrad: the reference also accumulates the critic's
    weight gradients here and zeroes them at the next :136 without using them."""
    generator.drop_grads()                                                               # :159 — every generator parameter receives a gradient in this backward
    if skip_dead_critic_wgrad:
        for p in critic.parameters():
            p.requires_grad_(False)
    try:
        fake_image = generator(noise, fake_class_labels)                                 # :163
        generator_loss = weighted_sum([mean(critic(fake_image, fake_class_labels))], [-1.0])   # :164-165 (no ATen neg launch)
        generator_loss.backward()                                                        # :167
    finally:
        if skip_dead_critic_wgrad:
            for p in critic.parameters():
                p.requires_grad_(True)
    if dp is not None:
        dp.sync_now(generator)
    generator_optimizer.step()                                                           # :168
    return {"generator_loss": generator_loss}


class GraphedSteps:
    """The two updates of the loop body (:133-155, :157-168) captured once each as HIP graphs (nn.GraphedStep) and replayed with
    one host call per update: at 256 images per GPU an update is ~70 / ~115 kernel launches of 5-200 us, and launched one by one
    the host leaves the GPU idle for a quarter of the iteration.  The draws stay outside the graphs (a captured Philox draw would
    replay the same numbers): the caller passes noise / alpha / labels exactly as to critic_step / generator_step and they are
    copied into the static inputs the graphs read.  Results are bit-identical to the eager calls (same kernels, same order).

        gs = GraphedSteps(critic, generator, c_opt, g_opt, hp, B, device)
        out = gs.critic_step(images, onehot, noise, alpha);  out = gs.generator_step(fake_onehot, noise)

    Data-parallel: pass the parallel.GradSync as `dp`; each graph is cut at its gradient exchange (nn.GraphedStep)."""

    def __init__(self, critic, generator, critic_optimizer, generator_optimizer, hp, B, device, dp=None, warmup=2):
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=device)
        self.B = int(B)
        self._c_in = {"images": z(B, 1, 28, 28), "labels": z(B, hp.num_classes), "noise": z(B, hp.latent_size), "alpha": z(B, 1)}
        self._g_in = {"labels": z(B, hp.num_classes), "noise": z(B, hp.latent_size)}
        ops.fill(self._c_in["alpha"], 0.5)
        for d in (self._c_in, self._g_in):                       # one-hot rows: the warm-up steps run on well-formed inputs
            d["labels"][:, 0] = 1.0
        ci, gi = self._c_in, self._g_in
        mods, opts = [critic, generator], [critic_optimizer, generator_optimizer]
        cfn = lambda d=None: critic_step_batched(critic, generator, critic_optimizer, hp, ci["images"], ci["labels"], ci["noise"], ci["alpha"], dp=d)
        gfn = lambda d=None: generator_step(critic, generator, generator_optimizer, gi["labels"], gi["noise"], dp=d)
        self._critic = GraphedStep(cfn, ci, mods, opts, warmup=warmup, dp=dp)
        self._generator = GraphedStep(gfn, gi, mods, opts, warmup=warmup, dp=dp)

    def _check(self, t):
        if t.shape[0] != self.B:
            raise PcgError(f"GraphedSteps was captured for batches of {self.B} rows, got {t.shape[0]} (run the ragged last batch eagerly)")

    def critic_step(self, real_images, real_class_labels, noise, alpha):
        self._check(real_images)
        self._critic.load(images=real_images, labels=real_class_labels, noise=noise, alpha=alpha)
        return self._critic.replay()

    def generator_step(self, fake_class_labels, noise):
        self._check(fake_class_labels)
        self._generator.load(labels=fake_class_labels, noise=noise)
        return self._generator.replay()


def train(critic, generator, dataloader, hp, device, rng=None, epochs=None, graphed=False):
    """:129-189 without the plotting tail: `dataloader` yields (images [B,1,28,28], class indices [B]); one-hot rows come from
    an identity table (:123,133); noise / alpha / fake labels are drawn on the device.  graphed=True: full batches of hp.batchsize
    rows replay GraphedSteps (same numbers); a ragged last batch runs eagerly."""
    critic_optimizer, generator_optimizer = make_optimizers(critic, generator)
    gs = GraphedSteps(critic, generator, critic_optimizer, generator_optimizer, hp, hp.batchsize, device) if graphed else None
    rng = rng if rng is not None else ops.DeviceRNG(seed=1)
    history = []
    generator_loss = None
    for epoch in range(hp.num_epochs if epochs is None else epochs):
        d_sum = g_sum = 0.0
        n = 0
        for batch_idx, (images, labels) in enumerate(dataloader):
            images, labels = images.to(device), labels.to(device)
            B = images.shape[0]
            onehot = ops.onehot(labels, hp.num_classes)                                  # all_labels[data[1]] (:133)
            replay = gs is not None and B == gs.B
            noise, alpha = rng.randn((B, hp.latent_size), device), rng.rand((B, 1), device)
            out = (gs.critic_step(images, onehot, noise, alpha) if replay else
                   critic_step(critic, generator, critic_optimizer, hp, images, onehot, noise, alpha))
            if batch_idx % hp.n_critic == 0:                                             # :157
                fake_labels = ops.onehot(rng.randint(0, hp.num_classes, B, device), hp.num_classes)   # :161
                noise = rng.randn((B, hp.latent_size), device)
                generator_loss = (gs.generator_step(fake_labels, noise) if replay else
                                  generator_step(critic, generator, generator_optimizer, fake_labels, noise))["generator_loss"]
            d_sum += out["critic_loss"].item(); g_sum += generator_loss.item()           # :177-178
            n += 1
        history.append((d_sum / max(n, 1), g_sum / max(n, 1)))
    return history


def build(device, hp=None, seed=1):
    """(critic, generator) on `device`, constructed in the reference's order (:116) after torch.manual_seed(seed) (:13)."""
    hp = hp if hp is not None else Hyperparameter()
    torch.manual_seed(seed)
    critic, generator = Critic(hp), Generator(hp)
    return critic.to(device), generator.to(device)

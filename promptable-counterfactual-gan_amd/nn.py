"""nn.Module drop-ins backed by libpcgan_hip.so.

`SequentialConvNet` takes the very `nn.Sequential` the reference scripts build (Conv2d / ConvTranspose2d /
BatchNorm2d / ReLU / LeakyReLU / Tanh / Sigmoid — mnist_dcgan.py:75-90,99-113) and runs it on the HIP kernels:
the torch layer objects stay as parameter containers, so `state_dict()` keys, `.apply(weights_init)`,
`parameters()`, `.train()/.eval()` and `load_state_dict()` behave exactly as in the reference, while forward and
backward are one autograd node that sequences C-ABI calls on NHWC activations.

Parameters live in ONE flat fp32 buffer per net (conv weights physically OHWI = channels_last), gradients in a
second one: the fused Adam step is a single launch over the flat buffer and the data-parallel all-reduce is a
single RCCL call on the flat gradient bucket.  Backward accumulates straight into the flat gradient buffer
(`p.grad` are views of it), which is autograd's `.grad` accumulation contract without extra add kernels.
"""
import contextlib
import os

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, PcgError


def _pad4(n):
    return (n + 3) // 4 * 4


class FlatModule(nn.Module):
    """Keeps every parameter of the module tree as a view of one flat buffer (and .grad as views of another)."""

    def __init__(self):
        super().__init__()
        self._flat = None
        self._gflat = None
        self._seg = None  # list of (param, offset, numel)

    # -- layout ------------------------------------------------------------------------------------
    def _flatten(self, device=None):
        params = list(self.parameters())
        if not params:
            raise PcgError("FlatModule has no parameters")
        device = device or params[0].device
        if device.type != "cuda":
            raise PcgError(f"parameters are on {device}; libpcgan_hip has no CPU path — move the module to the GPU")
        total, seg = 0, []
        for p in params:
            seg.append((p, total, p.numel()))
            total += _pad4(p.numel())
        flat = torch.empty(total, dtype=torch.float32, device=device)
        gflat = torch.empty(total, dtype=torch.float32, device=device)
        ops.fill(flat, 0.0)
        ops.fill(gflat, 0.0)
        with torch.no_grad():
            for p, off, n in seg:
                pv, gv = self._views(flat, gflat, p, off, n)
                pv.copy_(p.data)          # one-time relayout (ATen copy: setup, not the step)
                if p.grad is not None:
                    gv.copy_(p.grad)
                p.data = pv
                p.grad = gv
        self._flat, self._gflat, self._seg = flat, gflat, seg

    @staticmethod
    def _views(flat, gflat, p, off, n):
        if p.dim() == 4:  # conv weight: physical [d0, KH, KW, d1], logical [d0, d1, KH, KW] (channels_last)
            o, i, kh, kw = p.shape
            pv = flat[off:off + n].view(o, kh, kw, i).permute(0, 3, 1, 2)
            gv = gflat[off:off + n].view(o, kh, kw, i).permute(0, 3, 1, 2)
        else:
            pv = flat[off:off + n].view(p.shape)
            gv = gflat[off:off + n].view(p.shape)
        return pv, gv

    def _ensure_flat(self):
        if self._flat is None:
            self._flatten()
            return
        p, off, _ = self._seg[0]
        if p.data_ptr() != self._flat.data_ptr() + 4 * off or not p.is_cuda:
            self._flatten()  # someone replaced .data (e.g. .to(device)): rebuild the flat image from current values

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._flat = None  # .to()/.cuda()/.float() re-create storages; re-flatten lazily
        return r

    # -- grads ---------------------------------------------------------------------------------------
    def _grad_view(self, p):
        """(view, accumulate): the flat-buffer gradient view of parameter p, and whether it already holds a
        gradient to add to (autograd semantics: p.grad is None means 'no gradient yet')."""
        for q, off, n in self._seg:
            if q is p:
                gv = self._views(self._flat, self._gflat, p, off, n)[1]
                acc = p.grad is not None
                if not acc or p.grad.data_ptr() != gv.data_ptr():
                    if acc:  # a foreign .grad tensor: adopt its values
                        with torch.no_grad():
                            gv.copy_(p.grad)
                    p.grad = gv
                return gv, acc
        raise PcgError("parameter does not belong to this FlatModule")

    def grad_view_in(self, buf, p):
        """The view of parameter p's gradient inside `buf`, a flat buffer with the layout of flat_grads (a second gradient
        accumulator: two backward passes that run on parallel streams write one buffer each and are added once)."""
        for q, off, n in self._seg:
            if q is p:
                return self._views(self._flat, buf, p, off, n)[1]
        raise PcgError("parameter does not belong to this FlatModule")

    def drop_grads(self):
        """Forget the gradients without touching memory: the next backward OVERWRITES (p.grad is None = 'no gradient yet', see
        _grad_view) instead of adding to a zero-filled buffer — 0 + g == g, one fill launch less.  Only for steps in which every
        parameter receives a gradient (the padding between parameters stays as allocated: zero)."""
        self._ensure_flat()
        for p, _, _ in self._seg:
            p.grad = None

    def zero_grad(self, set_to_none=False):
        """Zero the flat gradient buffer in one launch and keep the .grad views (set_to_none is ignored: the
        views are the gradient storage; values after backward are identical either way)."""
        self._ensure_flat()
        ops.fill(self._gflat, 0.0)
        for p, off, n in self._seg:
            if p.grad is None or p.grad.data_ptr() != self._gflat.data_ptr() + 4 * off:
                p.grad = self._views(self._flat, self._gflat, p, off, n)[1]

    @property
    def flat_params(self):
        self._ensure_flat()
        return self._flat

    @property
    def flat_grads(self):
        self._ensure_flat()
        return self._gflat


# ---------------------------------------------------------------------------------------------------
class _LinearAsConv:
    """nn.Linear seen as a 1x1 convolution on a [B, 1, 1, F] activation: its [out, in] weight is already OHWI."""

    def __init__(self, lin):
        self.lin = lin
        self.in_channels, self.out_channels = lin.in_features, lin.out_features
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = (1, 1), (1, 1), (0, 0), (1, 1), 1

    @property
    def weight(self):
        return self.lin.weight

    @property
    def bias(self):
        return self.lin.bias


def _w_ohwi(t):
    """Physical OHWI view of a conv weight (4-D, channels_last) or a Linear weight (2-D [out, in] = [out][1][1][in])."""
    return t if t.dim() == 2 else ops.ohwi(t)


class _Block:
    """conv (or transposed conv, or Linear) -> [BatchNorm2d] -> [activation]"""

    def __init__(self, conv):
        if isinstance(conv, nn.Linear):
            conv = _LinearAsConv(conv)
        self.conv = conv
        self.transposed = isinstance(conv, nn.ConvTranspose2d)
        self.bn = None
        self.act = ACT_NONE
        self.slope = 0.0
        k, s, p = conv.kernel_size, conv.stride, conv.padding
        if s[0] != s[1] or p[0] != p[1] or conv.dilation != (1, 1) or conv.groups != 1:
            raise PcgError(f"unsupported convolution configuration: {conv}")
        if self.transposed and conv.output_padding != (0, 0):
            raise PcgError(f"output_padding is not supported: {conv}")
        self.kh, self.kw, self.stride, self.pad = k[0], k[1], s[0], p[0]
        self.flat = False     # set by geom(): transposed conv on a 1x1 input run as one plain GEMM

    def geom(self, B, H, W):
        """Geometry of the (adjoint) convolution and the output spatial size, for an input [B, H, W, C]."""
        c = self.conv
        if not self.transposed:
            g = ops.conv_geom(B, H, W, c.in_channels, c.out_channels, self.kh, self.kw, self.stride, self.pad)
            return g, g.OH, g.OW
        OH = (H - 1) * self.stride - 2 * self.pad + self.kh
        OW = (W - 1) * self.stride - 2 * self.pad + self.kw
        if H == 1 and W == 1 and self.pad == 0 and (self.kh * self.kw * c.out_channels) % 4 == 0 and c.in_channels % 4 == 0 \
                and c.in_channels > 3:
            # a 1x1 input: the layer is ONE plain GEMM  out[b][(kh,kw,co)] = sum_ci a[b][ci] W[ci][(kh,kw,co)], and the
            # ConvTranspose weight [ci][kh][kw][co] is exactly the OHWI weight of a 1x1 convolution with KH*KW*co "input"
            # channels.  The same three kernels serve it (N = KH*KW*co instead of a 16-tap gather with one live tap per row).
            g = ops.conv_geom(B, 1, 1, self.kh * self.kw * c.out_channels, c.in_channels, 1, 1, 1, 0)
            self.flat = True
            return g, OH, OW
        self.flat = False
        # adjoint conv: x side = this layer's output (OH x OW x out_channels), y side = its input
        g = ops.conv_geom(B, OH, OW, c.out_channels, c.in_channels, self.kh, self.kw, self.stride, self.pad)
        if g.OH != H or g.OW != W:
            raise PcgError(f"transposed-conv geometry mismatch for input {H}x{W}: {c}")
        return g, OH, OW


def _compile(seq):
    blocks = []
    for m in seq:
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
            blocks.append(_Block(m))
            continue
        if not blocks:
            raise PcgError(f"{type(m).__name__} before the first convolution is not supported")
        b = blocks[-1]
        if isinstance(m, nn.BatchNorm2d):
            if b.bn is not None or b.act != ACT_NONE:
                raise PcgError("expected conv -> [BatchNorm2d] -> [activation]")
            if not m.affine or not m.track_running_stats or m.momentum is None:
                raise PcgError(f"unsupported BatchNorm2d configuration: {m}")
            b.bn = m
        elif isinstance(m, nn.LeakyReLU):
            b.act, b.slope = ACT_LRELU, float(m.negative_slope)
        elif isinstance(m, nn.ReLU):
            b.act = ACT_RELU
        elif isinstance(m, nn.Tanh):
            b.act = ACT_TANH
        elif isinstance(m, nn.Sigmoid):
            b.act = ACT_SIGMOID
        else:
            raise PcgError(f"layer {type(m).__name__} has no libpcgan_hip implementation")
    if not blocks:
        raise PcgError("empty network")
    return blocks


class _WgradDefer:
    """Experiment (r04, `SequentialConvNet.wgrad_overlap = "bn"`): a layer's weight gradient is launched on a side stream AFTER the
    layer's grad-input has been queued, so that it runs beside the next layer's BatchNorm-backward finalize + apply pass (HBM-bound,
    no LDS, few waves) instead of before it; the next grad-input waits for it, so two MFMA grids never share the chip (that form —
    `wgrad_stream` — measured slower in r03)."""

    def __init__(self, side, main):
        self.side, self.main, self.job, self.inflight = side, main, None, False

    def submit(self, fn, tensors):
        self.job = (fn, tensors)

    def start_pending(self):
        if self.job is None:
            return
        fn, tensors = self.job
        self.job = None
        self.side.wait_stream(self.main)          # everything queued so far (the layer's grad-input included) comes first
        for t in tensors:
            t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            fn()
        self.inflight = True

    def join(self):
        if self.inflight:
            self.main.wait_stream(self.side)
            self.inflight = False

    def finish(self):
        if self.job is not None:                  # the last layer's: nothing left to run beside it
            fn, _ = self.job
            self.job = None
            self.join()
            fn()
        self.join()


class _SeqFn(torch.autograd.Function):
    """One autograd node for the whole stack.  Parameters are passed only so that autograd knows the output
    depends on them; their gradients are accumulated into the flat buffer inside backward (returned as None)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        y, saved = net._run_forward(x)
        ctx.net, ctx.saved = net, saved
        return y

    @staticmethod
    def backward(ctx, dy):
        need_x = ctx.needs_input_grad[1]
        need_p = any(ctx.needs_input_grad[2:])
        dx = ctx.net._run_backward(ctx.saved, dy, need_x, need_p)
        return (None, dx) + (None,) * (len(ctx.needs_input_grad) - 2)


class _SeqGroupFn(torch.autograd.Function):
    """The stack on G independent batches side by side (SequentialConvNet.forward_groups): one node, parameter gradients only."""

    @staticmethod
    def forward(ctx, net, ngroups, *rest):
        xs = rest[:ngroups]
        y, saved = net._run_forward_groups(xs)
        ctx.net, ctx.saved, ctx.ngroups = net, saved, ngroups
        return y

    @staticmethod
    def backward(ctx, dy):
        if any(ctx.needs_input_grad[2:2 + ctx.ngroups]):
            raise PcgError("forward_groups: gradients w.r.t. the inputs are not implemented (detach them: the D step of mnist_dcgan.py:159)")
        if any(ctx.needs_input_grad[2 + ctx.ngroups:]):
            ctx.net._run_backward_groups(ctx.saved, dy)
        return (None,) * len(ctx.needs_input_grad)


class SequentialConvNet(FlatModule):
    """Runs `self.main` (an nn.Sequential of torch conv / BN / activation layers) on the HIP kernels.

    Input and output follow the reference's NCHW *shape* convention; internally everything is NHWC.  Inputs with
    one channel (MNIST images, z as [B, z, 1, 1]) are NHWC already, so no relayout happens at the boundary.
    """

    fuse_backward_epilogue = True   # A/B switch (tests compare both forms): activation derivative / BatchNorm-backward sums of the
                                    # layer below taken in the grad-input kernel's epilogue instead of in separate passes
    fold_bn_apply = False           # A/B switch: BatchNorm(train) + ReLU / LeakyReLU applied inside the next convolution's gathers
    fold_bn_apply_thin = os.environ.get("PCG_FOLD_THIN", "1") != "0"   # ... inside a thin one-channel ConvTranspose2d's loads (G5 behind G4: no apply pass, no activated copy; bit-identical)
                                    # (forward and weight gradient) instead of as a pass that writes the activated tensor.
                                    # Bit-identical results; OFF by default because it is SLOWER on gfx950 (profiles/README.md r02:
                                    # DCGAN step 10.97 -> 11.26 ms): the ~16 VALU per gathered float4 in the producer waves are not
                                    # hidden behind the consumers' MFMAs (+10..20 % on every conv kernel that carries a transform),
                                    # while the pass it removes streams at 6 TB/s (0.21 ms per step)

    def __init__(self, main):
        super().__init__()
        self.main = main
        self._blocks = None

    # -- execution -------------------------------------------------------------------------------------
    def forward(self, input):
        self._ensure_flat()
        if self._blocks is None:
            self._blocks = _compile(self.main)
        flat_in = input.dim() == 2            # MLP: [B, F] is a [B, 1, 1, F] activation
        if flat_in:
            input = input.view(input.shape[0], input.shape[1], 1, 1)
        if input.dim() != 4:
            raise PcgError(f"expected a [B, C, H, W] or [B, F] input, got shape {tuple(input.shape)}")
        x = input.permute(0, 2, 3, 1)
        if not x.is_contiguous():
            x = x.contiguous()
        if x.dtype != torch.float32:
            raise PcgError(f"expected float32 input, got {x.dtype}")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            y = _SeqFn.apply(self, x, *self.parameters())
        else:
            y, _ = self._run_forward(x, keep=False)
        y = y.permute(0, 3, 1, 2)
        return y.reshape(y.shape[0], y.shape[1]) if flat_in else y

    def _can_consume_xform(self, nxt, B, H, W):
        """Should block `nxt` read its input [B, H, W, C] through an input transform?  MFMA consumers: opt-in (fold_bn_apply: measured
        slower, their gathers are the bottleneck); a thin one-channel ConvTranspose2d: yes (fold_bn_apply_thin: its kernels are HBM-bound)."""
        if nxt.transposed:
            OH = (H - 1) * nxt.stride - 2 * nxt.pad + nxt.kh
            OW = (W - 1) * nxt.stride - 2 * nxt.pad + nxt.kw
            if H == 1 and W == 1 and nxt.pad == 0:
                return False                    # 1x1 input: plain-GEMM form
            g = ops.conv_geom(B, OH, OW, nxt.conv.out_channels, nxt.conv.in_channels, nxt.kh, nxt.kw, nxt.stride, nxt.pad)
        else:
            g = ops.conv_geom(B, H, W, nxt.conv.in_channels, nxt.conv.out_channels, nxt.kh, nxt.kw, nxt.stride, nxt.pad)
        if ops.xform_ok(g, "dy" if nxt.transposed else "x"):
            return self.fold_bn_apply
        return self.fold_bn_apply_thin and nxt.transposed and g.Cin == 1 and ops.xform_thin_ok(g)

    def _next_reads_bn_input(self, idx, B, H, W, groups):
        """Block idx + 1 is a full-window one-channel convolution that can read block idx's PRE-BatchNorm output (ops.BnInput): D's last
        layer behind D4's BatchNorm + LeakyReLU.  B: samples in the launch (all groups)."""
        if not self.fold_bn_apply_thin or idx + 1 >= len(self._blocks):
            return False
        b, nxt = self._blocks[idx], self._blocks[idx + 1]
        if b.bn is None or not b.bn.training or b.act not in (ACT_NONE, ACT_RELU, ACT_LRELU) or nxt.transposed or nxt.bn is not None:
            return False
        if nxt.conv.out_channels != 1 or nxt.kh != H or nxt.kw != W or nxt.pad != 0:
            return False
        return ops.bnin_full_ok(ops.conv_geom(B, H, W, nxt.conv.in_channels, 1, nxt.kh, nxt.kw, nxt.stride, nxt.pad), groups)

    def _run_forward(self, x, keep=True):
        B, H, W, C = x.shape
        saved = []
        a = x
        xf = None     # input transform pending on `a`: a is then the producer's PRE-BatchNorm output (see fold_bn_apply)
        nblk = len(self._blocks)
        for idx, b in enumerate(self._blocks):
            c = b.conv
            if C != c.in_channels:
                raise PcgError(f"channel mismatch: activation has {C}, layer expects {c.in_channels}")
            g, OH, OW = b.geom(B, H, W)
            w = _w_ohwi(c.weight.data)
            bias = c.bias.data if c.bias is not None else None
            C = c.out_channels
            mean = invstd = None
            xf_in, xf = xf, None
            # BatchNorm(train) + ReLU / LeakyReLU of this block folded into the NEXT block's gathers: this block then never
            # writes its activated output
            fold = (idx + 1 < nblk and b.bn is not None and b.bn.training
                    and b.act in (ACT_NONE, ACT_RELU, ACT_LRELU) and self._can_consume_xform(self._blocks[idx + 1], B, OH, OW))
            if b.transposed and b.flat:
                # plain-GEMM form: bias is shared by the KH*KW positions of a channel, statistics are per channel over B*KH*KW rows
                z = ops.conv2d_dgrad(g, a, w, None, xf=xf_in).view(B, OH, OW, C)
                if bias is not None:
                    ops.add_bias_rows(z, C, bias)
                if b.bn is not None:
                    bn = b.bn
                    if bn.training and fold:
                        mean, invstd, coef = ops.bn_train_stats(z, C, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                                                bn.num_batches_tracked, gamma=bn.weight.data, beta=bn.bias.data)
                        y, xf = None, ops.InputXform(coef, b.act, b.slope)
                    elif bn.training:
                        mean, invstd = ops.bn_train_stats(z, C, bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked)
                        y = ops.bn_apply_act(z, C, mean, invstd, bn.weight.data, bn.bias.data, b.act, b.slope)
                    else:
                        y = ops.bn_apply_act(z, C, bn.running_mean, bn.running_var, bn.weight.data, bn.bias.data, b.act, b.slope,
                                             var_eps=bn.eps, out=z)
                        z = None
                elif b.act != ACT_NONE:
                    y = ops.act_fwd(z, b.act, b.slope, out=z)
                    z = None
                else:
                    y, z = z, None
            elif b.bn is not None and b.bn.training:
                bn = b.bn
                if fold:
                    z, mean, invstd, coef = ops.conv_bn_train(g, a, w, bias, b.transposed, bn.eps, bn.momentum, bn.running_mean,
                                                              bn.running_var, bn.num_batches_tracked, xf=xf_in,
                                                              gamma=bn.weight.data, beta=bn.bias.data)
                    y, xf = None, ops.InputXform(coef, b.act, b.slope)
                else:
                    z, mean, invstd = ops.conv_bn_train(g, a, w, bias, b.transposed, bn.eps, bn.momentum, bn.running_mean,
                                                        bn.running_var, bn.num_batches_tracked, xf=xf_in)
                    if self._next_reads_bn_input(idx, B, OH, OW, 1):
                        y, xf = None, ops.BnInput(mean, invstd, bn.weight.data, bn.bias.data, b.act, b.slope)     # no apply pass, no activated copy
                    else:
                        y = ops.bn_apply_act(z, C, mean, invstd, bn.weight.data, bn.bias.data, b.act, b.slope)
            elif b.bn is not None:
                bn = b.bn
                z = ops.conv2d_dgrad(g, a, w, bias, xf=xf_in) if b.transposed else ops.conv2d_fwd(g, a, w, bias, xf=xf_in)
                y = ops.bn_apply_act(z, C, bn.running_mean, bn.running_var, bn.weight.data, bn.bias.data, b.act, b.slope,
                                     var_eps=bn.eps, out=z)
                z = None
            elif b.act != ACT_NONE:   # activation fused into the conv's output write
                y = (ops.conv2d_dgrad(g, a, w, bias, act=b.act, slope=b.slope, xf=xf_in) if b.transposed
                     else ops.conv2d_fwd(g, a, w, bias, act=b.act, slope=b.slope, xf=xf_in))
                z = None
            else:
                y = ops.conv2d_dgrad(g, a, w, bias, xf=xf_in) if b.transposed else ops.conv2d_fwd(g, a, w, bias, xf=xf_in)
                z = None
            if keep:
                saved.append((g, a, z, mean, invstd, y, b.bn is not None and not b.bn.training, xf_in))
            a, H, W = (y if xf is None else z), OH, OW
        return a, saved

    # -- G independent batches side by side (r04) ------------------------------------------------------------------------------
    def supports_groups(self, shape, groups=2):
        """Can `forward_groups` run `groups` inputs of NCHW shape `shape`?  (Conv2d / Linear stacks whose first block has no
        BatchNorm; training-mode BatchNorm layers behind MFMA convolutions with whole 128-row tiles per group; see
        include/pcgan_hip.h "grouped batches".)  Cached per (shape, groups, training)."""
        self._ensure_flat()
        if self._blocks is None:
            self._blocks = _compile(self.main)
        key = (tuple(shape), int(groups), self.training)
        cache = self.__dict__.setdefault("_group_ok", {})
        if key in cache:
            return cache[key]
        ok = self._group_check(tuple(shape), int(groups))
        cache[key] = ok
        return ok

    def _group_check(self, shape, G):
        if len(shape) != 4 or G < 2 or G > 8:
            return False
        B, C, H, W = shape
        for idx, b in enumerate(self._blocks):
            c = b.conv
            if b.transposed or C != c.in_channels:
                return False
            g = ops.conv_geom(B if idx == 0 else G * B, H, W, c.in_channels, c.out_channels, b.kh, b.kw, b.stride, b.pad)
            if idx == 0 and b.bn is not None:
                return False
            if b.bn is not None:
                if not b.bn.training or b.act not in (ACT_NONE, ACT_RELU, ACT_LRELU) or not ops.group_fwd_ok(g, G):
                    return False
            C, H, W = c.out_channels, g.OH, g.OW
        return True

    def forward_groups(self, inputs):
        """The stack applied to G independent batches in ONE pass: `inputs` = G tensors of the same [B, C, H, W] shape (the real
        batch and fake.detach() of mnist_dcgan.py:151,159); returns the [G*B, ...] output, group after group.  Per layer one
        convolution launch over G*B images; BatchNorm statistics, running-statistics updates and the backward's means stay per
        group (see include/pcgan_hip.h).  Inputs get no gradient."""
        inputs = list(inputs)
        if not self.supports_groups(inputs[0].shape, len(inputs)):
            raise PcgError(f"forward_groups: this stack / shape {tuple(inputs[0].shape)} x {len(inputs)} is not eligible (supports_groups)")
        xs = []
        for t in inputs:
            if t.shape != inputs[0].shape or t.dtype != torch.float32:
                raise PcgError("forward_groups: the inputs must be float32 tensors of one shape")
            x = t.permute(0, 2, 3, 1)
            xs.append(x if x.is_contiguous() else x.contiguous())
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            y = _SeqGroupFn.apply(self, len(xs), *xs, *self.parameters())
        else:
            y, _ = self._run_forward_groups(xs, keep=False)
        return y.permute(0, 3, 1, 2)

    def _run_forward_groups(self, xs, keep=True):
        G = len(xs)
        B, H, W, C = xs[0].shape
        saved = []
        a = None
        xf = None     # ops.BnInput pending on `a` (then the producer's PRE-BatchNorm output): see _next_reads_bn_input
        for idx, b in enumerate(self._blocks):
            c = b.conv
            w = _w_ohwi(c.weight.data)
            bias = c.bias.data if c.bias is not None else None
            mean = invstd = z = None
            xf_in, xf = xf, None
            if idx == 0:
                # the groups' inputs are separate tensors: the first layer runs once per group into its slice of ONE output
                g = ops.conv_geom(B, H, W, c.in_channels, c.out_channels, b.kh, b.kw, b.stride, b.pad)
                y = torch.empty((G * B, g.OH, g.OW, c.out_channels), dtype=torch.float32, device=xs[0].device)
                for k, x in enumerate(xs):
                    ops.conv2d_fwd(g, x, w, bias, out=y[k * B:(k + 1) * B], act=b.act, slope=b.slope)
                a_in = tuple(xs)
            else:
                g = ops.conv_geom(G * B, H, W, c.in_channels, c.out_channels, b.kh, b.kw, b.stride, b.pad)
                a_in = a
                if b.bn is not None and b.bn.training:
                    bn = b.bn
                    z, mean, invstd = ops.conv_bn_train_g(g, a, w, bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                                          bn.num_batches_tracked, G)
                    if self._next_reads_bn_input(idx, G * B, g.OH, g.OW, G):
                        y, xf = None, ops.BnInput(mean, invstd, bn.weight.data, bn.bias.data, b.act, b.slope, G)
                    else:
                        y = ops.bn_apply_act_g(z, c.out_channels, mean, invstd, bn.weight.data, bn.bias.data, b.act, b.slope, G)
                elif b.bn is not None:
                    bn = b.bn
                    z = ops.conv2d_fwd(g, a, w, bias)
                    y = ops.bn_apply_act(z, c.out_channels, bn.running_mean, bn.running_var, bn.weight.data, bn.bias.data, b.act, b.slope,
                                         var_eps=bn.eps, out=z)
                    z = None
                else:
                    y = ops.conv2d_fwd(g, a, w, bias, act=b.act, slope=b.slope, xf=xf_in)
            if xf_in is not None and (idx == 0 or b.bn is not None):
                raise PcgError("grouped forward: a BatchNorm input reached a layer that cannot read it")     # (_next_reads_bn_input rules this out)
            if keep:
                saved.append((g, a_in, z, mean, invstd, y, b.bn is not None and not b.bn.training, xf_in))
            a, H, W = (y if xf is None else z), g.OH, g.OW
        return a, (saved, G, B)

    def _run_backward_groups(self, saved_all, dy):
        saved, G, B = saved_all
        if not dy.is_contiguous():
            dy = dy.contiguous()
        d, own = dy, False
        fused = None           # ("bn", partial, nparts, nphases) | ("mask",) | None — see _run_backward_impl
        defer = self._defer()
        try:
            with ops.slab_reductions_deferred(self.defer_slab_reductions and defer is None):
                self._run_backward_groups_impl(saved, G, B, d, own, fused, defer)
        finally:
            if defer is not None:
                defer.finish()

    def _run_backward_groups_impl(self, saved, G, B, d, own, fused, defer):
        for idx in range(len(self._blocks) - 1, -1, -1):
            b = self._blocks[idx]
            g, a, z, mean, invstd, y, bn_eval, xf_in = saved[idx]
            c = b.conv
            C = c.out_channels
            if defer is not None:
                defer.start_pending()
            if b.bn is not None:
                if bn_eval:
                    raise PcgError("backward through an eval-mode BatchNorm2d is not implemented")
                bn = b.bn
                dg = db = None
                acc = False
                if bn.weight.requires_grad:
                    dg, acc = self._grad_view(bn.weight)
                    db, _ = self._grad_view(bn.bias)
                if fused is not None and fused[0] == "done":
                    dz = d               # the full-window layer above pushed its gradient through this BatchNorm already
                elif fused is not None:
                    dz = ops.bn_bwd_partial_g(d, z, C, mean, invstd, bn.weight.data, fused[1], fused[2], fused[3], dg, db, acc, G, out=d)
                else:
                    dz = ops.bn_act_bwd_g(d, z, C, mean, invstd, bn.weight.data, bn.bias.data, b.act, b.slope, dg, db, acc, G,
                                          out=d if own else None)
            elif fused is not None or b.act == ACT_NONE:
                dz = d
            else:
                dz = ops.act_bwd(d, y, b.act, b.slope, out=d if own else None)
            own = True
            if c.weight.requires_grad:
                gw, acc = self._grad_view(c.weight)
                gw = _w_ohwi(gw)

                def wgrad_job(idx=idx, g=g, a=a, dz=dz, gw=gw, acc=acc, c=c, C=C, xf_in=xf_in):
                    if idx == 0:
                        for k, x in enumerate(a):       # per group: the inputs are separate tensors (.grad accumulation of :153,161)
                            ops.conv2d_wgrad(g, x, dz[k * B:(k + 1) * B], gw, acc or k > 0)
                    else:
                        ops.conv2d_wgrad(g, a, dz, gw, acc, xf_x=xf_in)      # ONE sum over the pixels of all groups
                    if c.bias is not None and c.bias.requires_grad:
                        gb, accb = self._grad_view(c.bias)
                        ops.colsum(dz.numel() // C, C, dz, gb, accb)
                if defer is not None:
                    defer.join()
                if defer is not None and idx > 0 and g.Cin > 3 and g.Cout > 3:
                    defer.submit(wgrad_job, (dz, a))
                else:
                    wgrad_job()
            elif defer is not None:
                defer.join()
            if idx == 0:
                return None
            lo = self._blocks[idx - 1]
            _, _, zl, ml, il, yl, lo_eval, _ = saved[idx - 1]
            w = _w_ohwi(c.weight.data)
            fused = None
            mfma = g.Cin > 3 and g.Cout > 3 and g.Cin % 4 == 0 and g.Cout % 4 == 0 and g.stride <= 2
            if self.fuse_backward_epilogue and mfma and lo.act in (ACT_NONE, ACT_RELU, ACT_LRELU):
                if lo.bn is not None and not lo_eval and zl is not None and ops.group_dgrad_ok(g, G):
                    d, partial, nparts, nphases = ops.conv_bwd_data_fused_g(g, dz, w, lo.act, lo.slope, zl,
                                                                            (ml, il, lo.bn.weight.data, lo.bn.bias.data), G)
                    fused = ("bn", partial, nparts, nphases)
                    continue
                if lo.bn is None and lo.act != ACT_NONE:
                    res = ops.conv_bwd_data_fused(g, dz, w, False, lo.act, lo.slope, a_below=yl)
                    if res is not None:
                        d, fused = res[0], ("mask",)
                        continue
            if (self.fuse_backward_epilogue and lo.bn is not None and lo.act in (ACT_NONE, ACT_RELU, ACT_LRELU) and not lo_eval and zl is not None
                    and lo.bn.weight.requires_grad and self.fuse_full_window_bn and ops.full_dgrad_bn_bwd_ok(g, G)):
                dgl, accl = self._grad_view(lo.bn.weight)
                dbl, _ = self._grad_view(lo.bn.bias)
                d = ops.full_dgrad_bn_bwd(g, dz, w, zl, ml, il, lo.bn.weight.data, lo.bn.bias.data, lo.act, lo.slope, dgl, dbl, accl, groups=G)
                fused = ("done",)
                continue
            d = ops.conv2d_dgrad(g, dz, w)

    # opt-in A/B: one slab-reduction launch per backward sweep instead of one per weight gradient (ops.slab_reductions_deferred): bit-identical;
    # measured r04: DCGAN 10.174 / 10.178 ms, CounteRGAN 26.68 -> 26.64 — the one launch (16.5 us for four entries) reads slabs that have left
    # the caches by then, the per-layer launches (5-7 us each) read them right behind the kernel that wrote them
    defer_slab_reductions = False

    fuse_full_window_bn = True   # A/B: a full-window one-channel conv's grad-input through the BatchNorm backward below it, unwritten (pcg_conv2d_dgrad_bnbwd_full)

    wgrad_stream = None      # opt-in A/B: a second HIP stream for the weight gradients (they and the grad-input of a layer both need only dz)

    wgrad_overlap = None     # experiment: "bn" = weight gradients beside the next layer's BatchNorm-backward passes (_WgradDefer)

    def _defer(self):
        if self.wgrad_overlap != "bn":
            return None
        side = self.__dict__.get("_overlap_stream")
        if side is None:
            side = self.__dict__["_overlap_stream"] = torch.cuda.Stream()
        return _WgradDefer(side, torch.cuda.current_stream())

    def _run_backward(self, saved, dy, need_x, need_p):
        if not dy.is_contiguous():
            dy = dy.contiguous()
        side = self.wgrad_stream
        main = torch.cuda.current_stream() if side is not None else None
        defer = self._defer() if side is None else None
        try:
            with ops.slab_reductions_deferred(self.defer_slab_reductions and side is None and defer is None):
                return self._run_backward_impl(saved, dy, need_x, need_p, side, main, defer)
        finally:
            if side is not None:
                main.wait_stream(side)
            if defer is not None:
                defer.finish()

    def _run_backward_impl(self, saved, dy, need_x, need_p, side, main, defer=None):
        d = dy
        own = False  # never write into autograd's incoming grad tensor; deeper gradients are ours to overwrite
        nblk = len(self._blocks)
        # `fused`: how the gradient `d` arriving at block idx was produced by the grad-input kernel of block idx+1:
        #   None                  plain gradient w.r.t. the block's output
        #   ("mask",)             already multiplied by the block's ReLU / LeakyReLU derivative
        #   ("bn", partial, n)    the same, plus BatchNorm-backward's column sums in `partial` (n partial rows)
        fused = None
        for idx in range(nblk - 1, -1, -1):
            b = self._blocks[idx]
            g, a, z, mean, invstd, y, bn_eval, xf_in = saved[idx]
            c = b.conv
            C = c.out_channels
            if defer is not None:
                defer.start_pending()     # the layer above's weight gradient: beside this layer's BatchNorm backward
            if b.bn is not None:
                if bn_eval:
                    raise PcgError("backward through an eval-mode BatchNorm2d is not implemented")
                bn = b.bn
                dg = db = None
                acc = False
                if need_p and bn.weight.requires_grad:
                    dg, acc = self._grad_view(bn.weight)
                    db, acc2 = self._grad_view(bn.bias)
                    if acc != acc2:
                        raise PcgError("inconsistent .grad state on BatchNorm weight/bias")
                if fused is not None and fused[0] == "done":
                    dz = d               # the thin layer above pushed its gradient through this BatchNorm already (thin_fwd_bn_bwd)
                elif fused is not None:
                    dz = ops.bn_bwd_partial(d, z, C, mean, invstd, bn.weight.data, fused[1], fused[2], dg, db, acc, out=d)
                else:
                    # ReLU / LeakyReLU: the mask is recomputed from z (no read of y)
                    dz = ops.bn_act_bwd(d, z, None if b.act in (ACT_RELU, ACT_LRELU) else y, C, mean, invstd, bn.weight.data, b.act, b.slope, dg,
                                        db, acc, beta=bn.bias.data)
            elif fused is not None or b.act == ACT_NONE:
                dz = d
            else:
                dz = ops.act_bwd(d, y, b.act, b.slope, out=d if own else None)
            own = True
            if need_p and c.weight.requires_grad:
                gw, acc = self._grad_view(c.weight)
                gw = _w_ohwi(gw)
                if side is not None:
                    side.wait_stream(main)                       # dz is complete
                    dz.record_stream(side); a.record_stream(side)

                def wgrad_job(b=b, g=g, a=a, dz=dz, gw=gw, acc=acc, xf_in=xf_in, c=c, C=C):
                    if not b.transposed:      # `a` may be the layer below's pre-BatchNorm output read through its transform
                        ops.conv2d_wgrad(g, a, dz, gw, acc, xf_x=xf_in)
                    else:
                        ops.conv2d_wgrad(g, dz, a, gw, acc, xf_dy=xf_in)
                    if c.bias is not None and c.bias.requires_grad:
                        gb, accb = self._grad_view(c.bias)
                        ops.colsum(dz.numel() // C, C, dz, gb, accb)
                mfma = g.Cin > 3 and g.Cout > 3
                if defer is not None and mfma and idx > 0:
                    defer.join()                                 # (an earlier deferred gradient: one MFMA grid at a time)
                    defer.submit(wgrad_job, (dz, a))             # launched at the top of the next iteration, behind this layer's grad-input
                else:
                    if defer is not None:
                        defer.join()
                    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                        wgrad_job()
            elif defer is not None:
                defer.join()
            last = idx == 0
            if last and not need_x:
                return None
            # grad-input of this block = gradient w.r.t. the output of the block below; apply that block's activation derivative
            # (and take its BatchNorm-backward sums) in the epilogue when both layers allow it
            fused = None
            if not last and self.fuse_backward_epilogue:
                lo = self._blocks[idx - 1]
                _, _, zl, ml, il, yl, lo_eval, _ = saved[idx - 1]
                w = _w_ohwi(c.weight.data)
                res = None
                if lo.act in (ACT_NONE, ACT_RELU, ACT_LRELU) and not (b.transposed and b.flat):
                    if lo.bn is not None and not lo_eval and zl is not None:
                        res = ops.conv_bwd_data_fused(g, dz, w, b.transposed, lo.act, lo.slope, z_below=zl,
                                                      bn=(ml, il, lo.bn.weight.data, lo.bn.bias.data))
                        if res is not None:
                            d, fused = res[0], ("bn", res[1], res[2])
                    elif lo.bn is None and lo.act != ACT_NONE:
                        res = ops.conv_bwd_data_fused(g, dz, w, b.transposed, lo.act, lo.slope, a_below=yl)
                        if res is not None:
                            d, fused = res[0], ("mask",)
                if res is not None:
                    continue
            if not b.transposed:
                lo = self._blocks[idx - 1] if not last else None
                if (lo is not None and self.fuse_backward_epilogue and lo.bn is not None and lo.act in (ACT_NONE, ACT_RELU, ACT_LRELU)
                        and not saved[idx - 1][6] and saved[idx - 1][2] is not None
                        and self.fuse_full_window_bn and ops.full_dgrad_bn_bwd_ok(g)):
                    # a full-window one-channel convolution above a BatchNorm layer (D5 above D4): its grad-input — one multiply per
                    # element — goes through that BatchNorm's backward without being written (with or without parameter gradients:
                    # the G step's pass through D wants none)
                    _, _, zl, ml, il, _, _, _ = saved[idx - 1]
                    dgl = dbl = None
                    accl = False
                    if need_p and lo.bn.weight.requires_grad:
                        dgl, accl = self._grad_view(lo.bn.weight)
                        dbl, _ = self._grad_view(lo.bn.bias)
                    d = ops.full_dgrad_bn_bwd(g, dz, _w_ohwi(c.weight.data), zl, ml, il, lo.bn.weight.data, lo.bn.bias.data, lo.act, lo.slope,
                                              dgl, dbl, accl)
                    fused = ("done",)
                    continue
                d = ops.conv2d_dgrad(g, dz, _w_ohwi(c.weight.data))
            else:
                lo = self._blocks[idx - 1] if not last else None
                if (lo is not None and self.fuse_backward_epilogue and lo.bn is not None and lo.act in (ACT_NONE, ACT_RELU, ACT_LRELU)
                        and not saved[idx - 1][6] and saved[idx - 1][2] is not None and lo.bn.weight.requires_grad and need_p
                        and ops.thin_fwd_bn_bwd_ok(g)):
                    # a one-channel ConvTranspose2d above a BatchNorm layer (G5 above G4): its grad-input goes through that
                    # BatchNorm's backward without being written
                    _, _, zl, ml, il, _, _, _ = saved[idx - 1]
                    dgl, accl = self._grad_view(lo.bn.weight)
                    dbl, _ = self._grad_view(lo.bn.bias)
                    d = ops.thin_fwd_bn_bwd(g, dz, _w_ohwi(c.weight.data), zl, ml, il, lo.bn.weight.data, lo.bn.bias.data, lo.act, lo.slope,
                                            dgl, dbl, accl)
                    fused = ("done",)
                    continue
                d = ops.conv2d_fwd(g, dz, _w_ohwi(c.weight.data))
        return d


class HipSequential(SequentialConvNet):
    """Drop-in for a bare `nn.Sequential(...)` (state_dict keys "0.weight", "2.bias", ... with no prefix), e.g. the MLPs
    of simple_gan/moons/make_moons_gan.py:33-46."""

    def __init__(self, *layers):
        FlatModule.__init__(self)
        for i, layer in enumerate(layers):
            self.add_module(str(i), layer)
        self._blocks = None

    @property
    def main(self):
        return list(self.children())

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i):
        return list(self.children())[i]


# ---- nn.Linear on rows [B, in] --------------------------------------------------------------------------------------------
def _lin_mfma(B, I, O):
    """Big layers go through the MFMA implicit-GEMM kernels as 1x1 convolutions (they need in_features % 4 == 0); narrow,
    ragged (widths 1, 10, 17, 21, 38 ...) or tiny ones through the bounds-checked small GEMM."""
    return I % 4 == 0 and I >= 64 and O >= 32 and B * I * O >= (1 << 24)


def affine_fwd(x, w, bias=None, act=ACT_NONE, slope=0.0):
    """y = act(x w^T (+ bias)) for a raw [out, in] weight (frozen / folded layers); act: none / ReLU / LeakyReLU, fused."""
    B = x.shape[0]
    O, I = w.shape
    if _lin_mfma(B, I, O):
        return ops.conv2d_fwd(ops.conv_geom(B, 1, 1, I, O, 1, 1, 1, 0), x, w, bias, act=act, slope=slope).view(B, O)
    return ops.gemm(x, w, B, O, I, transB=True, bias=bias, act=act, slope=slope)


def linear_fwd(lin, x, weight=None, out=None, ldc=None, use_bias=True, act=ACT_NONE, slope=0.0):
    """y = act(x W^T (+ b)).  `weight` overrides lin.weight (spectral-norm layers pass W / sigma); act: none / ReLU / LeakyReLU."""
    w = lin.weight.data if weight is None else weight
    B = x.shape[0]
    O, I = w.shape
    bias = lin.bias.data if (use_bias and lin.bias is not None) else None
    if out is None and _lin_mfma(B, I, O):
        g = ops.conv_geom(B, 1, 1, I, O, 1, 1, 1, 0)
        return ops.conv2d_fwd(g, x, w, bias, act=act, slope=slope).view(B, O)
    return ops.gemm(x, w, B, O, I, transB=True, bias=bias, out=out, ldc=ldc, act=act, slope=slope)


def linear_dgrad(w, dy, B, ldy=None, out=None, accumulate=False):
    """dx = dy W (optionally accumulated into `out`)."""
    O, I = w.shape
    if ldy is None and out is None and _lin_mfma(B, I, O):
        g = ops.conv_geom(B, 1, 1, I, O, 1, 1, 1, 0)
        return ops.conv2d_dgrad(g, dy, w).view(B, I)
    return ops.gemm(dy, w, B, I, O, lda=ldy if ldy is not None else O, out=out, accumulate=accumulate)


def linear_wgrad(net, lin, x, dy, ldy=None, dw_out=None, weight_param=None, use_bias=True, ldx=None):
    """dW += dy^T x, db += column sums of dy, accumulated into net's flat gradient buffer.  `dw_out`: write dW there
    instead (spectral-norm layers post-process it); `weight_param`: the parameter that owns the weight gradient when it is
    not `lin.weight` (spectral norm: weight_orig)."""
    B = x.shape[0]
    I = lin.in_features
    O = lin.out_features
    wp = weight_param if weight_param is not None else getattr(lin, "weight", None)
    has_b = use_bias and lin.bias is not None and lin.bias.requires_grad
    gb, accb = net._grad_view(lin.bias) if has_b else (None, False)
    if dw_out is not None:
        gw, acc = dw_out, False
    else:
        gw, acc = net._grad_view(wp)
    if ldy is None and ldx is None and _lin_mfma(B, I, O):
        g = ops.conv_geom(B, 1, 1, I, O, 1, 1, 1, 0)
        ops.conv2d_wgrad(g, x, dy, gw, acc)
        if gb is not None:
            ops.colsum(B, O, dy, gb, accb)
        return
    ops.linear_wgrad(dy, x, B, O, I, gw, gb, ldy=ldy, ldx=ldx, accumulate_w=acc, accumulate_b=accb)


class _MeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.like = x
        return ops.mean_fwd(x.contiguous()).view(())

    @staticmethod
    def backward(ctx, g):
        return ops.mean_bwd(g.contiguous(), 1.0, ctx.like)


def mean(x):
    """tensor.mean() of a critic output (Wasserstein losses)."""
    return _MeanFn.apply(x)


class _WeightedSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.weights, ctx.shapes = weights, [t.shape for t in terms]
        return ops.weighted_sum_fwd([t.contiguous() for t in terms], weights).view(())

    @staticmethod
    def backward(ctx, g):
        gs = ops.weighted_sum_bwd(ctx.weights, g.contiguous(), ctx.needs_input_grad[1:])
        return (None,) + tuple(t.view(s) if t is not None else None for t, s in zip(gs, ctx.shapes))


def weighted_sum(terms, weights):
    """sum_i weights[i] * terms[i] for one-element loss tensors — the loss compositions of the training loops
    (d_loss = -D_real.mean() + D_fake.mean(); G_loss = G_adv + lambda_cls G_cls + ...) as one launch forward and one backward."""
    return _WeightedSumFn.apply(tuple(float(w) for w in weights), *terms)


class _CutDP:
    """Stands in for a `parallel.GradSync` while a step is being captured: every exchange point (wait / sync_now / sync_then) ends
    the graph segment being captured, runs the real operation eagerly and starts the next segment.  RCCL collectives and the
    cross-stream event waits therefore stay ordinary stream operations between graph launches."""

    def __init__(self, dp, owner):
        self._dp, self._owner = dp, owner

    def wait(self, net):
        self._owner._cut(lambda: self._dp.wait(net))

    def sync_now(self, net):
        self._owner._cut(lambda: self._dp.sync_now(net))

    def sync_then(self, net, fn):
        self._owner._cut(lambda: self._dp.sync_then(net, fn))

    def wait_all(self):
        self._owner._cut(self._dp.wait_all)


class _HipGraphCapture:
    """Capture backend of GraphedStep: HIP graphs (torch.cuda.CUDAGraph), every segment in ONE memory pool."""

    def __init__(self, device, tick=True):
        self._pool = None
        self._tick = torch.zeros(1, device=device) if tick else None     # tick=False: one segment holding the whole step, never empty

    def warmup(self, run, n, dp):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(n):
                run()
            if dp is not None:
                dp.wait_all()
        torch.cuda.current_stream().wait_stream(side)

    def begin(self):
        # Python's cyclic collector must not run while a stream captures: freeing a graph, an event or a cached block from a finalizer
        # is a HIP call the capture forbids, and an error raised inside a finalizer aborts the process.  Collect now, hold it off.
        import gc
        gc.collect()
        self._gc_was_on = gc.isenabled()
        gc.disable()
        try:
            self._g = torch.cuda.CUDAGraph()
            # thread_local: calls made by other threads (the RCCL watchdog polling events) must not invalidate the capture
            self._ctx = torch.cuda.graph(self._g, pool=self._pool, capture_error_mode="thread_local")
            self._ctx.__enter__()
        except BaseException:
            self._gc_restore()         # the capture never began: end() will not run, the collector must not stay off
            raise
        if self._tick is not None:
            self._tick.add_(1.0)   # no segment is ever empty (an empty capture cannot be instantiated)

    def _gc_restore(self):
        if getattr(self, "_gc_was_on", False):
            import gc
            gc.enable()
        self._gc_was_on = False

    def end(self):
        try:
            self._ctx.__exit__(None, None, None)
        finally:
            self._gc_restore()
        if self._pool is None:
            self._pool = self._g.pool()
        return self._g             # .replay()

    abort = end


class GraphedStep:
    """A whole training step captured once in a HIP graph and replayed with one host call — the host then cannot starve the GPU,
    whether the step is short (small per-GPU batches, the tabular nets) or the host cores are slow or shared.

        gs = GraphedStep(lambda: train_step(G, D, ..., x, y, ...), inputs={"x": x, "y": y, ...}, modules=[G, D], optimizers=[opt_g, opt_d])
        gs.load(x=next_x, y=next_y, ...); out = gs.replay()

    `inputs` are the static device tensors the step closes over (load() copies new values into them); `out` is whatever the step
    returned (static tensors too).  Capture needs warm-up executions of real steps: parameters, buffers and optimizer state of the
    given modules / optimizers are snapshotted before and restored after, so building the object does not advance training.

    Data-parallel steps: pass the `parallel.GradSync` as `dp`; `step_fn` then takes one argument (the object to hand to the
    training step as its `dp`).  The step is captured as a chain of graph segments cut at the exchange points; replay() launches
    segment, exchange, segment, ... in the captured order (all segments share one memory pool, so the order is fixed).

    `capture`: the capture backend (begin / end -> object with replay() / abort / warmup); default HIP graphs.  The CPU rehearsal of
    the segment program (tests/test_parallel_gloo.py, world size 2 over gloo) passes a recording backend instead."""

    def __init__(self, step_fn, inputs, modules, optimizers, warmup=3, dp=None, capture=None):
        self.inputs = dict(inputs)
        # the captured launches carry raw pointers into the modules' flat buffers, the optimizers' state and whatever the step
        # closes over: keep all of it alive as long as the graph can be replayed (a caller that drops its own references — a
        # builder function returning only the GraphedStep — must not leave the graph writing into freed memory)
        self._keepalive = (step_fn, list(modules), list(optimizers), dp)
        if dp is not None and getattr(dp, "sync_bn", False):
            raise PcgError("exact-BatchNorm mode (GradSync(sync_bn=True)) issues RCCL collectives inside the BatchNorm calls, which "
                           "cannot be captured in a HIP graph: run the step eagerly")
        for m in modules:
            m._ensure_flat()
        saved = [(m.flat_params.clone(), [b.clone() for b in m.buffers()]) for m in modules]
        osnap = [o.snapshot() for o in optimizers]
        run = step_fn if dp is None else (lambda: step_fn(dp))
        if capture is None:
            # the segments of a data-parallel program may hold no launch (the paired step starts with wait(G)): they carry a one-element add;
            # the single graph of a one-rank step does not need it (4.6 us per replay)
            capture = _HipGraphCapture(self.inputs[next(iter(self.inputs))].device if self.inputs else torch.device("cuda"), tick=dp is not None)
        self._capture = capture
        capture.warmup(run, max(1, warmup), dp)
        self.program = []          # [(graph, eager operation after it or None)]
        capture.begin()
        try:
            self.out = step_fn() if dp is None else step_fn(_CutDP(dp, self))
        except BaseException:
            capture.abort()
            raise
        self._end(None)
        if dp is not None:
            dp.wait_all()
        self.graph = self.program[0][0]
        for m, (fp, bufs) in zip(modules, saved):
            m.flat_params.copy_(fp)
            for b, b0 in zip(m.buffers(), bufs):
                b.copy_(b0)
        for o, sn in zip(optimizers, osnap):
            o.restore(sn)

    def _end(self, op):
        self.program.append((self._capture.end(), op))

    def _cut(self, op):
        self._end(op)
        op()                       # capture executes nothing; the exchange itself runs (state is restored after capture)
        self._capture.begin()

    def load(self, **tensors):
        for k, v in tensors.items():
            self.inputs[k].copy_(v)

    def replay(self):
        for g, op in self.program:
            g.replay()
            if op is not None:
                op()
        return self.out


class BCELoss(nn.Module):
    """nn.BCELoss() (reduction='mean') on the HIP kernel (mnist_dcgan.py:125)."""

    def forward(self, input, target):
        return _BCEFn.apply(input, target)


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, target):
        p = p.contiguous()
        target = target.contiguous()
        loss, _ = ops.bce_fwd_bwd(p, target, 0.0, need_grad=False)
        ctx.save_for_backward(p, target)
        return loss.view(())

    @staticmethod
    def backward(ctx, grad_out):
        p, target = ctx.saved_tensors
        _, dp = ops.bce_fwd_bwd(p, target, 0.0, need_loss=False, grad_out=grad_out.contiguous().view(1))
        return dp, None


class _BCEPairFn(torch.autograd.Function):
    """(BCE(p[:n], t0), BCE(p[n:], t1), their sum) in one launch forward and one backward (ops.bce_pair)."""

    @staticmethod
    def forward(ctx, p, t0, t1):
        p = p.contiguous()
        n = p.numel() // 2
        loss, _ = ops.bce_pair(p, n, t0, t1, need_loss=True)
        ctx.save_for_backward(p)
        ctx.t = (float(t0), float(t1), n)
        ctx.set_materialize_grads(False)
        return loss[0], loss[1], loss[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        (p,) = ctx.saved_tensors
        t0, t1, n = ctx.t
        cot = tuple(None if g is None else g.contiguous().view(1) for g in (g0, g1, g2))
        _, dp = ops.bce_pair(p, n, t0, t1, need_loss=False, need_grad=True, cotangents=cot)
        return dp, None, None


def bce_pair(p, target0, target1):
    """nn.BCELoss()(p[:n], full(target0)), nn.BCELoss()(p[n:], full(target1)) and their sum (errD_real, errD_fake, errD of
    mnist_dcgan.py:152,160,163) for the output of a two-group discriminator pass."""
    if p.numel() % 2:
        raise PcgError("bce_pair: expected an even number of outputs (two groups)")
    return _BCEPairFn.apply(p, float(target0), float(target1))


_ones = {}


def backward(loss):
    """loss.backward() with the seed gradient taken from a cached one-element tensor: autograd's own `ones_like(loss)` is an ATen
    fill launch per call (three per DCGAN step)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _ones.get(key)
    if one is None:
        one = ops.fill(torch.empty(loss.shape, dtype=loss.dtype, device=loss.device), 1.0) if loss.numel() else torch.ones_like(loss)
        _ones[key] = one
    torch.autograd.backward(loss, grad_tensors=one)

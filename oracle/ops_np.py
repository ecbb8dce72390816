"""ORACLE — test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline); never the product path.

NumPy restatement (float64 accumulation unless said otherwise) of the arithmetic behind the PyTorch operators the
reference's hot path calls, written in the layouts of the C ABI (include/pcgan_hip.h): activations NHWC
[B,H,W,C], conv weights OHWI [Cout,KH,KW,Cin].  The reference pins PyTorch 2.4.1 (Dockerfile:1); the operator
semantics restated here ([torch] in SURVEY.md) are stable across 2.4 - 2.10 and are re-checked against
torch.nn.functional on the CPU by tests/test_oracle_ops.py.

Call sites restated: nn.Conv2d / nn.ConvTranspose2d (mnist_dcgan.py:76-88,100-111), nn.BatchNorm2d in training
mode (:77-87,103-110), nn.ReLU / nn.LeakyReLU(0.2) / nn.Tanh / nn.Sigmoid, nn.BCELoss (:125),
nn.BCEWithLogitsLoss (conditional_counteRGAN/mnist/trainer.py:79), optim.Adam (:126-127).
"""
import numpy as np

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4


def out_size(i, k, s, p):
    return (i + 2 * p - k) // s + 1


# ---- convolution ------------------------------------------------------------------------------------
def conv2d_fwd(x, w, bias, stride, pad):
    """y[b,oh,ow,co] = bias[co] + sum_{kh,kw,ci} x[b,oh*s-p+kh,ow*s-p+kw,ci] * w[co,kh,kw,ci]."""
    B, IH, IW, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH, OW = out_size(IH, KH, stride, pad), out_size(IW, KW, stride, pad)
    xp = np.zeros((B, IH + 2 * pad, IW + 2 * pad, Cin), np.float64)
    xp[:, pad:pad + IH, pad:pad + IW] = x
    y = np.zeros((B, OH, OW, Cout), np.float64)
    for kh in range(KH):
        for kw in range(KW):
            patch = xp[:, kh:kh + stride * OH:stride, kw:kw + stride * OW:stride, :]     # [B,OH,OW,Cin]
            y += patch @ w[:, kh, kw, :].astype(np.float64).T
    if bias is not None:
        y += bias
    return y


def conv2d_dgrad(dy, w, in_hw, stride, pad, bias_x=None):
    """dx[b,ih,iw,ci] = sum over (oh,kh): ih = oh*s-p+kh of dy[b,oh,ow,co] * w[co,kh,kw,ci]  (= ConvTranspose2d fwd)."""
    B, OH, OW, Cout = dy.shape
    _, KH, KW, Cin = w.shape
    IH, IW = in_hw
    dxp = np.zeros((B, IH + 2 * pad + stride, IW + 2 * pad + stride, Cin), np.float64)
    for kh in range(KH):
        for kw in range(KW):
            dxp[:, kh:kh + stride * OH:stride, kw:kw + stride * OW:stride, :] += dy.astype(np.float64) @ w[:, kh, kw, :].astype(np.float64)
    dx = dxp[:, pad:pad + IH, pad:pad + IW].copy()
    if bias_x is not None:
        dx += bias_x
    return dx


def conv2d_wgrad(x, dy, ksize, stride, pad):
    """dw[co,kh,kw,ci] = sum_{b,oh,ow} dy[b,oh,ow,co] * x[b,oh*s-p+kh,ow*s-p+kw,ci]."""
    B, IH, IW, Cin = x.shape
    _, OH, OW, Cout = dy.shape
    KH, KW = ksize
    xp = np.zeros((B, IH + 2 * pad + stride, IW + 2 * pad + stride, Cin), np.float64)
    xp[:, pad:pad + IH, pad:pad + IW] = x
    dw = np.zeros((Cout, KH, KW, Cin), np.float64)
    d2 = dy.reshape(-1, Cout).astype(np.float64)
    for kh in range(KH):
        for kw in range(KW):
            patch = xp[:, kh:kh + stride * OH:stride, kw:kw + stride * OW:stride, :].reshape(-1, Cin)
            dw[:, kh, kw, :] = d2.T @ patch
    return dw


# ---- activations ---------------------------------------------------------------------------------------
def act_fwd(v, act, slope=0.0):
    v = np.asarray(v, np.float64)
    if act == ACT_RELU:
        return np.where(v > 0, v, 0.0)
    if act == ACT_LRELU:
        return np.where(v > 0, v, v * slope)
    if act == ACT_TANH:
        return np.tanh(v)
    if act == ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-v))
    return v


def act_grad_from_out(y, act, slope=0.0):
    """d act / d input, written through the activation's OUTPUT ([torch]: ReLU/LeakyReLU use the negative-side
    slope at 0; with slope > 0 the sign of the output equals the sign of the input)."""
    y = np.asarray(y, np.float64)
    if act == ACT_RELU:
        return (y > 0).astype(np.float64)
    if act == ACT_LRELU:
        return np.where(y > 0, 1.0, slope)
    if act == ACT_TANH:
        return 1.0 - y * y
    if act == ACT_SIGMOID:
        return y * (1.0 - y)
    return np.ones_like(y)


# ---- BatchNorm (training) -------------------------------------------------------------------------------
def bn_train_stats(x, eps, momentum=0.1, running_mean=None, running_var=None):
    """[torch] batch_norm(training=True): biased variance for normalisation, unbiased for the running estimate."""
    C = x.shape[-1]
    x2 = x.reshape(-1, C).astype(np.float64)
    n = x2.shape[0]
    mean = x2.mean(0)
    var = x2.var(0)
    invstd = 1.0 / np.sqrt(var + eps)
    new_rm = new_rv = None
    if running_mean is not None:
        new_rm = (1 - momentum) * running_mean + momentum * mean
        new_rv = (1 - momentum) * running_var + momentum * var * (n / max(n - 1, 1))
    return mean, invstd, new_rm, new_rv


def bn_apply_act(x, mean, invstd, gamma, beta, act, slope=0.0):
    v = (x.astype(np.float64) - mean) * invstd * gamma + beta
    return act_fwd(v, act, slope)


def bn_act_bwd(dy, x, y, mean, invstd, gamma, act, slope=0.0):
    """Returns (dx, dgamma, dbeta) for y = act(bn(x)) in training mode."""
    C = x.shape[-1]
    dz = dy.astype(np.float64) * act_grad_from_out(y, act, slope)
    xh = (x.astype(np.float64) - mean) * invstd
    d2, h2 = dz.reshape(-1, C), xh.reshape(-1, C)
    dbeta, dgamma = d2.sum(0), (d2 * h2).sum(0)
    n = d2.shape[0]
    dx = gamma * invstd * (dz - dbeta / n - xh * dgamma / n)
    return dx, dgamma, dbeta


# ---- losses ------------------------------------------------------------------------------------------------
def bce(p, t):
    """[torch] binary_cross_entropy, reduction='mean': log terms clamped at -100; grad (p-t)/max(p(1-p),1e-12)/n."""
    p = np.asarray(p, np.float64)
    t = np.broadcast_to(np.asarray(t, np.float64), p.shape)
    with np.errstate(divide="ignore"):
        lp, lq = np.maximum(np.log(p), -100.0), np.maximum(np.log1p(-p), -100.0)
    loss = np.mean(-(t * lp + (1 - t) * lq))
    grad = (p - t) / np.maximum(p * (1 - p), 1e-12) / p.size
    return loss, grad


def bce_with_logits(z, t):
    z = np.asarray(z, np.float64)
    t = np.broadcast_to(np.asarray(t, np.float64), z.shape)
    mx = np.maximum(-z, 0.0)
    loss = np.mean((1 - t) * z + mx + np.log(np.exp(-mx) + np.exp(-z - mx)))
    grad = (1.0 / (1.0 + np.exp(-z)) - t) / z.size
    return loss, grad


# ---- Adam ------------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, beta1, beta2, eps=1e-8, weight_decay=0.0, decoupled=False):
    """[torch] optim.Adam (amsgrad=False) / AdamW single-tensor update; `step` is 1-based."""
    p, g, m, v = (np.asarray(a, np.float64).copy() for a in (p, g, m, v))
    if weight_decay:
        if decoupled:
            p *= 1 - lr * weight_decay
        else:
            g = g + weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p = p - (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)
    return p, m, v

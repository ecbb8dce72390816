"""CPU: the NumPy restatement (oracle/ops_np.py) against the PyTorch operators the reference calls."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops_np as O

torch.set_num_threads(4)


def _nhwc(t):  # NCHW torch -> NHWC numpy
    return t.detach().permute(0, 2, 3, 1).contiguous().numpy()


def _ohwi(w):
    return w.detach().permute(0, 2, 3, 1).contiguous().numpy()


CASES = [  # B, Cin, Cout, H, W, k, s, p
    (2, 8, 12, 8, 8, 4, 2, 1),
    (3, 4, 8, 7, 7, 3, 1, 1),
    (2, 4, 4, 7, 9, 3, 2, 1),
    (2, 16, 1, 4, 4, 4, 1, 0),
    (2, 1, 8, 10, 10, 4, 2, 1),
    (4, 20, 12, 1, 1, 1, 1, 0),
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,s,p", CASES)
def test_conv_fwd_dgrad_wgrad(B, Cin, Cout, H, W, k, s, p):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    np.testing.assert_allclose(O.conv2d_fwd(_nhwc(x), _ohwi(w), b.numpy(), s, p), _nhwc(y), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.conv2d_dgrad(_nhwc(dy), _ohwi(w), (H, W), s, p), _nhwc(x.grad), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.conv2d_wgrad(_nhwc(x), _nhwc(dy), (k, k), s, p), _ohwi(w.grad), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("B,Cin,Cout,H,k,s,p", [(2, 8, 4, 4, 4, 2, 1), (3, 12, 8, 1, 4, 1, 0), (2, 4, 1, 8, 4, 2, 1)])
def test_conv_transpose_is_dgrad_of_adjoint(B, Cin, Cout, H, k, s, p):
    """ConvTranspose2d(Cin->Cout) forward == conv2d_dgrad of the adjoint conv (Cout->Cin) with the same weight memory."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cin, Cout, k, k, generator=g, dtype=torch.float64, requires_grad=True)  # ConvT weight [Cin,Cout,k,k]
    y = F.conv_transpose2d(x, w, None, stride=s, padding=p)
    OH = y.shape[2]
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    w_ohwi = _ohwi(w)  # [Cin_T, k, k, Cout_T] = OHWI of the adjoint conv (Cout_adj = Cin_T)
    np.testing.assert_allclose(O.conv2d_dgrad(_nhwc(x), w_ohwi, (OH, OH), s, p), _nhwc(y), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.conv2d_fwd(_nhwc(dy), w_ohwi, None, s, p), _nhwc(x.grad), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.conv2d_wgrad(_nhwc(dy), _nhwc(x), (k, k), s, p), _ohwi(w.grad), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("act,slope", [(O.ACT_RELU, 0.0), (O.ACT_LRELU, 0.2), (O.ACT_NONE, 0.0)])
def test_batchnorm_train_fwd_bwd(act, slope):
    g = torch.Generator().manual_seed(2)
    B, C, H = 6, 8, 5
    x = (torch.randn(B, C, H, H, generator=g, dtype=torch.float64) * 2 + 0.5).requires_grad_(True)
    gamma = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    beta = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    z = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    y = {O.ACT_RELU: F.relu, O.ACT_LRELU: lambda t: F.leaky_relu(t, 0.2), O.ACT_NONE: lambda t: t}[act](z)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    mean, invstd, nrm, nrv = O.bn_train_stats(_nhwc(x), 1e-5, 0.1, np.zeros(C), np.ones(C))
    np.testing.assert_allclose(nrm, rm.numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(nrv, rv.numpy(), rtol=1e-12, atol=1e-12)
    yn = O.bn_apply_act(_nhwc(x), mean, invstd, gamma.detach().numpy(), beta.detach().numpy(), act, slope)
    np.testing.assert_allclose(yn, _nhwc(y), rtol=1e-11, atol=1e-11)
    dx, dg, db = O.bn_act_bwd(_nhwc(dy), _nhwc(x), yn, mean, invstd, gamma.detach().numpy(), act, slope)
    np.testing.assert_allclose(dx, _nhwc(x.grad), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(dg, gamma.grad.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(db, beta.grad.numpy(), rtol=1e-10, atol=1e-10)


def test_bce_and_logits():
    g = torch.Generator().manual_seed(3)
    p = torch.rand(37, generator=g, dtype=torch.float64).clamp(1e-6, 1 - 1e-6).requires_grad_(True)
    for t in (0.0, 1.0):
        p.grad = None
        loss = F.binary_cross_entropy(p, torch.full_like(p, t))
        loss.backward()
        l, gr = O.bce(p.detach().numpy(), t)
        np.testing.assert_allclose(l, loss.item(), rtol=1e-12)
        np.testing.assert_allclose(gr, p.grad.numpy(), rtol=1e-10)
    # saturated probabilities: the -100 clamp ([torch])
    ps = torch.tensor([0.0, 1.0, 0.5], dtype=torch.float64)
    np.testing.assert_allclose(O.bce(ps.numpy(), 1.0)[0], F.binary_cross_entropy(ps, torch.ones_like(ps)).item(), rtol=1e-12)
    z = (torch.randn(41, generator=g, dtype=torch.float64) * 4).requires_grad_(True)
    for t in (0.0, 1.0):
        z.grad = None
        loss = F.binary_cross_entropy_with_logits(z, torch.full_like(z, t))
        loss.backward()
        l, gr = O.bce_with_logits(z.detach().numpy(), t)
        np.testing.assert_allclose(l, loss.item(), rtol=1e-12)
        np.testing.assert_allclose(gr, z.grad.numpy(), rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize("betas,wd,decoupled", [((0.5, 0.999), 0.0, False), ((0.9, 0.999), 0.0, False), ((0.0, 0.9), 0.01, True)])
def test_adam(betas, wd, decoupled):
    g = torch.Generator().manual_seed(4)
    p = torch.randn(50, generator=g, dtype=torch.float64, requires_grad=True)
    cls = torch.optim.AdamW if decoupled else torch.optim.Adam
    opt = cls([p], lr=2e-4, betas=betas, weight_decay=wd)
    pn, m, v = p.detach().numpy().copy(), np.zeros(50), np.zeros(50)
    for step in range(1, 4):
        gr = torch.randn(50, generator=g, dtype=torch.float64)
        p.grad = gr.clone()
        opt.step()
        pn, m, v = O.adam_step(pn, gr.numpy(), m, v, step, 2e-4, betas[0], betas[1], 1e-8, wd, decoupled)
        np.testing.assert_allclose(pn, p.detach().numpy(), rtol=1e-12, atol=1e-14)

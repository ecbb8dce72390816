"""Bisect D(real) backward: tensors entering / leaving each BatchNorm backward, HIP vs float64 vs CPU fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgan_amd
from pcgan_amd import dcgan as D, ops
from pcgan_amd.nn import SequentialConvNet
from oracle import dcgan_ref as R
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DEV = "cuda:0"
torch.set_num_threads(16)
refG, refD = R.build(None, seed=1)
r64D = R.Discriminator().double(); r64D.load_state_dict({k: v.double() for k, v in refD.state_dict().items()})
real, noise = R.synthetic_batch(B, seed=0)
crit = torch.nn.BCELoss()
cap = {}
def hook(tag, name):
    def h(mod, gin, gout):
        cap[(tag, name, "gout")] = gout[0].detach().permute(0, 2, 3, 1).contiguous().double().numpy()
        if gin[0] is not None:
            cap[(tag, name, "gin")] = gin[0].detach().permute(0, 2, 3, 1).contiguous().double().numpy()
    return h
def fhook(tag, name):
    def h(mod, inp, out):
        cap[(tag, name, "pre")] = inp[0].detach().permute(0, 2, 3, 1).contiguous().double().numpy()
    return h
for tag, net in (("c32", refD), ("c64", r64D)):
    for i, m in enumerate(net.main):
        if isinstance(m, torch.nn.LeakyReLU):
            m.register_forward_hook(fhook(tag, i))
    for i, m in enumerate(net.main):
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.LeakyReLU)):
            m.register_full_backward_hook(hook(tag, i))
    # inplace LeakyReLU + full backward hooks do not mix: rebuild non-inplace
for net in (refD, r64D):
    for m in net.main:
        if isinstance(m, torch.nn.LeakyReLU): m.inplace = False
crit(refD(real), torch.ones(B)).backward()
crit(r64D(real.double()), torch.ones(B).double()).backward()
def rl2(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
SequentialConvNet.fuse_backward_epilogue = False
rec = []
orig = ops.bn_act_bwd
def spy(dy, x, y, C, mean, invstd, gamma, act, slope, dgamma, dbeta, accumulate, out=None, dy_scale=1.0, beta=None):
    d_in = dy.clone()
    r = orig(dy, x, y, C, mean, invstd, gamma, act, slope, dgamma, dbeta, accumulate, out=out, dy_scale=dy_scale, beta=beta)
    sc = gamma * invstd
    pre = torch.addcmul(beta - mean * sc, x.view(-1, C), sc)
    rec.append((C, d_in.cpu().double().numpy(), r.clone().cpu().double().numpy(), mean.cpu().double().numpy(), invstd.cpu().double().numpy(),
                (pre > 0).cpu().numpy(), x.cpu().double().numpy()))
    return r
import pcgan_amd.nn as NN
NN.ops.bn_act_bwd = spy
netD = D.Discriminator(); netD.load_state_dict(refD.state_dict()); netD.to(DEV)
c2 = D.make_optimizers(D.Generator().to(DEV), netD)[0]
netD.zero_grad()
c2(netD(real.to(DEV)), torch.ones(B, device=DEV)).backward()
# BN modules at indices 9 (C=512), 6 (256), 3 (128); LeakyReLU after them at 10, 7, 4.  grad wrt BN+act output = gout of the LeakyReLU
for (C, d_in, d_out, mean, invstd, pos, zhip), bn_i in zip(rec, (9, 6, 3)):
    tpre, cpre = cap[("c64", bn_i + 1, "pre")].reshape(-1, C), cap[("c32", bn_i + 1, "pre")].reshape(-1, C)
    flips_h, flips_c = int((pos != (tpre > 0)).sum()), int(((cpre > 0) != (tpre > 0)).sum())
    near = int((np.abs(tpre) < 3e-6 * tpre.std()).sum())
    print(f"BN{bn_i}: {pos.size} elements; activation-mask flips vs float64: HIP {flips_h}, CPU fp32 {flips_c}; elements with |pre| < 3e-6 std: {near}")
    t_in, t_out = cap[("c64", bn_i + 1, "gout")], cap[("c64", bn_i, "gin")]
    c_in, c_out = cap[("c32", bn_i + 1, "gout")], cap[("c32", bn_i, "gin")]
    cm = lambda a: a.reshape(-1, C).mean(0)
    print(f"BN{bn_i} C={C}: d_in  HIP {rl2(d_in, t_in):.2e} CPU {rl2(c_in, t_in):.2e} | colmean(d_in) HIP {rl2(cm(d_in), cm(t_in)):.2e} CPU {rl2(cm(c_in), cm(t_in)):.2e}"
          f" | |colmean|/rms {np.abs(cm(t_in)).mean() / t_in.std():.2e}")
    print(f"          dz    HIP {rl2(d_out, t_out):.2e} CPU {rl2(c_out, t_out):.2e} | colmean(dz) HIP {np.abs(cm(d_out)).max():.2e} CPU {np.abs(cm(c_out)).max():.2e} truth {np.abs(cm(t_out)).max():.2e}  rms(dz) {t_out.std():.2e}")

"""Three-way gradient distances of one full-width DCGAN step (lr = 0): HIP fp32 vs float64 truth, PyTorch-CPU fp32 vs float64 truth.
Usage: python scripts/grad_noise_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgan_amd
from pcgan_amd import dcgan as D
from oracle import dcgan_ref as R
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DEV = "cuda:0"
torch.set_num_threads(16)
cfg = {"lr": 0.0}
refG, refD = R.build(None, seed=1)
r64G, r64D = R.build(None, seed=1); r64G.double(); r64D.double()
r64G.load_state_dict({k: v.double() for k, v in refG.state_dict().items()}); r64D.load_state_dict({k: v.double() for k, v in refD.state_dict().items()})
netG, netD = D.Generator(), D.Discriminator()
netG.load_state_dict(refG.state_dict()); netD.load_state_dict(refD.state_dict()); netG.to(DEV); netD.to(DEV)
real, noise = R.synthetic_batch(B, seed=0)
R.dcgan_step(refG, refD, *R.make_optimizers(refG, refD, cfg), real, noise)
R.dcgan_step(r64G, r64D, *R.make_optimizers(r64G, r64D, cfg), real.double(), noise.double())
D.train_step(netG, netD, *D.make_optimizers(netG, netD, cfg), real.to(DEV), noise.to(DEV), skip_dead_d_wgrad=False)
def rl2(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
print(f"batch {B}: rel-L2 to float64 truth      HIP fp32    CPU fp32    (HIP vs CPU fp32)")
for tag, n3 in (("G", (netG, refG, r64G)), ("D", (netD, refD, r64D))):
    for (n, p), (_, q), (_, t) in zip(*(m.named_parameters() for m in n3)):
        g, c, t64 = p.grad.cpu().double().numpy(), q.grad.double().numpy(), t.grad.numpy()
        print(f"  {tag}.{n:16s} {rl2(g, t64):10.2e}  {rl2(c, t64):10.2e}   {rl2(g, c):10.2e}")

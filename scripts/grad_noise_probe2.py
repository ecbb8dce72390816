"""D(real) backward only: BN beta/gamma grads, fused vs unfused epilogue, against float64 truth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgan_amd
from pcgan_amd import dcgan as D
from pcgan_amd.nn import SequentialConvNet
from oracle import dcgan_ref as R
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DEV = "cuda:0"
torch.set_num_threads(16)
refG, refD = R.build(None, seed=1)
r64D = R.Discriminator().double(); r64D.load_state_dict({k: v.double() for k, v in refD.state_dict().items()})
real, noise = R.synthetic_batch(B, seed=0)
crit = torch.nn.BCELoss()
crit(refD(real), torch.ones(B)).backward()
crit(r64D(real.double()), torch.ones(B).double()).backward()
def rl2(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
for fuse in (True, False):
    SequentialConvNet.fuse_backward_epilogue = fuse
    netD = D.Discriminator(); netD.load_state_dict(refD.state_dict()); netD.to(DEV)
    c2 = D.make_optimizers(D.Generator().to(DEV), netD)[0]
    netD.zero_grad()
    c2(netD(real.to(DEV)), torch.ones(B, device=DEV)).backward()
    print(f"fuse={fuse}  batch {B}: rel-L2 to float64 truth: HIP, CPU fp32")
    for (n, p), (_, q), (_, t) in zip(netD.named_parameters(), refD.named_parameters(), r64D.named_parameters()):
        g, c, t64 = p.grad.cpu().double().numpy(), q.grad.double().numpy(), t.grad.numpy()
        print(f"  D.{n:16s} {rl2(g, t64):10.2e}  {rl2(c, t64):10.2e}   |truth| {np.linalg.norm(t64):.3e}")

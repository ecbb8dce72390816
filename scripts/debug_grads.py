"""Per-parameter gradient error of the HIP DCGAN step vs the CPU oracle (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgan_amd
from pcgan_amd import dcgan as D
from oracle import dcgan_ref as R

torch.set_num_threads(16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = None
refG, refD = R.build(cfg, seed=1)
netG, netD = D.Generator(cfg), D.Discriminator(cfg)
netG.load_state_dict(refG.state_dict()); netD.load_state_dict(refD.state_dict())
netG.to("cuda:0"); netD.to("cuda:0")
rcrit, roptD, roptG = R.make_optimizers(refG, refD)
crit, optD, optG = D.make_optimizers(netG, netD)
real, noise = R.synthetic_batch(B, seed=10)

def rel(a, b):
    a = a.detach().cpu().double().numpy(); b = b.detach().double().numpy()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)

# forward-only comparisons
with torch.no_grad():
    pass
# D(real) backward only
netD.zero_grad(); refD.zero_grad()
o = netD(real.cuda()); l = crit(o, torch.ones(B, device="cuda:0")); l.backward()
ro = refD(real); rl = rcrit(ro, torch.ones(B)); rl.backward()
print("D(real) out rel", rel(o, ro), "loss", l.item(), rl.item())
for (n, p), (_, q) in zip(netD.named_parameters(), refD.named_parameters()):
    print(f"  D(real) grad {n:16s} rel-L2 {rel(p.grad, q.grad):.2e}")
# G path: fake = G(z); D(fake) -> backward into G
netG.zero_grad(); refG.zero_grad()
fake = netG(noise.cuda()); rfake = refG(noise)
print("G out rel", rel(fake, rfake))
o = netD(fake); l = crit(o, torch.ones(B, device="cuda:0")); l.backward()
ro = refD(rfake); rl = rcrit(ro, torch.ones(B)); rl.backward()
print("D(fake) out rel", rel(o, ro))
for (n, p), (_, q) in zip(netG.named_parameters(), refG.named_parameters()):
    print(f"  G grad {n:16s} rel-L2 {rel(p.grad, q.grad):.2e}  |ref| {q.grad.norm().item():.3e}")
# input gradient of D wrt image
x = real.clone().cuda().requires_grad_(True); rx = real.clone().requires_grad_(True)
netD.zero_grad(); refD.zero_grad()
crit(netD(x), torch.ones(B, device="cuda:0")).backward(); rcrit(refD(rx), torch.ones(B)).backward()
print("dD/dx rel", rel(x.grad, rx.grad))

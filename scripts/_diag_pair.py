import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgan_amd
from pcgan_amd import dcgan as D
from pcgan_amd.nn import GraphedStep
pcgan_amd.load()
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pair = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
torch.manual_seed(1)
netG, netD = D.build(None, device="cpu")
netG.to(dev); netD.to(dev)
crit, optD, optG = D.make_optimizers(netG, netD)
g = torch.Generator().manual_seed(1234)
real = (torch.rand(B, 1, 64, 64, generator=g) * 2 - 1).to(dev)
noise = torch.randn(B, 100, 1, 1, generator=g).to(dev)
print("building graph", flush=True)
gs = GraphedStep(lambda: D.train_step(netG, netD, crit, optD, optG, real, noise, pair=pair), {"real": real, "noise": noise}, [netG, netD], [optD, optG])
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(3):
    out = gs.replay()
    torch.cuda.synchronize()
    print("replay", i, {k: float(out[k].item()) for k in ("errD_real", "errD_fake", "errG")}, flush=True)

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dcgan_ref as R
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    netG, netD = R.build(None, seed=1)
    crit, optD, optG = R.make_optimizers(netG, netD)
    real, noise = R.synthetic_batch(128, seed=0)
    R.dcgan_step(netG, netD, crit, optD, optG, real, noise)
    t0 = time.perf_counter(); R.dcgan_step(netG, netD, crit, optD, optG, real, noise); dt = time.perf_counter() - t0
    print(f"threads {th}: {128/dt:.1f} img/s ({dt:.2f} s/step at bs128)", flush=True)

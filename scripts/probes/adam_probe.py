"""Adam update kernel: host-stepped vs device-stepped (capturable) at the DCGAN sizes, us per launch (HIP events over 200 launches)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, pcgan_amd
from pcgan_amd import ops
dev = torch.device("cuda:0")
for n in (3_576_704, 2_765_568, 100_000):
    p, g, m, v = (torch.randn(n, device=dev) for _ in range(4)); v.abs_()
    step = torch.zeros(1, dtype=torch.int64, device=dev); hyper = torch.zeros(128, device=dev)
    def a(): ops.adam_step(p, g, m, v, 2e-4, 0.5, 0.999, 1e-8, 0.0, False, 7)
    def b(): ops.adam_step_capturable(p, g, m, v, 2e-4, 0.5, 0.999, 1e-8, 0.0, False, step, hyper)
    for name, fn in (("host-stepped", a), ("device-stepped", b)):
        for _ in range(10): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"n={n} {name}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us/launch  ({28 * n / (e0.elapsed_time(e1) / 200 * 1e-3) / 1e12:.2f} TB/s)")

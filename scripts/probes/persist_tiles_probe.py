import sys, os
sys.path.insert(0, os.getcwd())
import torch, pcgan_amd
from pcgan_amd import ops
sys.path.insert(0, 'scripts')
import conv_microbench as M
pcgan_amd.load()
dev = torch.device('cuda:0')
B = 512
def bench(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name in ("D2", "R64", "G4", "D3", "WC2"):
    Cin, Cout, H, k, s, p = M.LAYERS[name]
    g = ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    dy = torch.randn(B, g.OH, g.OW, Cout, device=dev)
    y = torch.empty(B, g.OH, g.OW, Cout, device=dev); dx = torch.empty_like(x)
    flops = 2.0 * B * g.OH * g.OW * Cout * k * k * Cin
    for op, fn in (("fwd", lambda: ops.conv2d_fwd(g, x, w, None, out=y)), ("dgrad", lambda: ops.conv2d_dgrad(g, dy, w, None, out=dx))):
        res = {}
        for rnd in range(3):
            for cfg in ((0, -1), (1, -1), (1, 2), (1, 4), (1, 8)):
                ops.tune("persistent", cfg[0]); ops.tune("persist_tiles", cfg[1])
                res.setdefault(cfg, []).append(bench(fn))
        ops.tune("persistent", -1); ops.tune("persist_tiles", -1)
        print(name, op, " | ".join(f"p={c[0]} T={c[1]}: {sorted(v)[1]*1e3:6.1f} us {flops/sorted(v)[1]/1e9:6.1f} TF" for c, v in res.items()), flush=True)

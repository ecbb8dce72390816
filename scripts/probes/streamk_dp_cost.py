import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcgan_amd
from pcgan_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from streamk_scan import bench
pcgan_amd.load()
dev = torch.device("cuda:0")
for name, op, B, Cin, Cout, H, k, s, p in [("D3 fwd", "fwd", 512, 128, 256, 16, 4, 2, 1), ("D4 dgrad", "dgrad", 512, 256, 512, 8, 4, 2, 1), ("D3 dgrad", "dgrad", 512, 128, 256, 16, 4, 2, 1),
                                           ("critic conv2 dgrad B512", "dgrad", 512, 256, 512, 13, 3, 2, 0), ("critic conv2 dgrad B256", "dgrad", 256, 256, 512, 13, 3, 2, 0)]:
    g = ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)
    w = torch.randn(Cout, k, k, Cin, device=dev) * 0.02
    if op == "fwd":
        x = torch.randn(B, H, H, Cin, device=dev); y = torch.empty(B, g.OH, g.OW, Cout, device=dev)
        fn = lambda: ops.conv2d_fwd(g, x, w, None, out=y)
    else:
        dy = torch.randn(B, g.OH, g.OW, Cout, device=dev); dx = torch.empty(B, H, H, Cin, device=dev)
        fn = lambda: ops.conv2d_dgrad(g, dy, w, out=dx)
    r = {}
    for rep in range(2):
        for mode in (0, 3):
            ops.tune("stream_k", mode)
            r.setdefault(mode, []).append(bench(fn))
    print(name, "base", [round(v, 1) for v in r[0]], "sk-kernel all-DP", [round(v, 1) for v in r[3]], flush=True)

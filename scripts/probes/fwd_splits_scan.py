#!/usr/bin/env python3
"""Scan the forward split-K count (pcg_tune_set("fwd_splits", s)) over the small-M / awkward-tile-count forward launches of the
WGAN-GP updates (width 1024): time per launch incl. the slab reduction, against the built-in plan.  Diagnostic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import pcgan_amd  # noqa: E402
from pcgan_amd import ops  # noqa: E402

# (name, B, Cin, Cout, H, k, s, p) of the forward-kernel launches
SHAPES = [("critic conv2 B768", 768, 256, 512, 13, 3, 2, 0), ("critic conv3 B768", 768, 512, 1024, 6, 3, 2, 0),
          ("critic Linear B768", 768, 8192, 1024, 1, 1, 1, 0), ("critic conv2 B256", 256, 256, 512, 13, 3, 2, 0),
          ("critic conv3 B256", 256, 512, 1024, 6, 3, 2, 0), ("critic Linear B256", 256, 8192, 1024, 1, 1, 1, 0),
          ("G ConvT3 grad-input B256 (fwd 256->512 14->7 k4)", 256, 256, 512, 14, 4, 2, 1),
          ("G ConvT2 grad-input B256 (fwd 512->1024 7->4 k3)", 256, 512, 1024, 7, 3, 2, 1),
          ("G ConvT1 grad-input B256 (fwd 16384->1024)", 256, 16384, 1024, 1, 1, 1, 0)]


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    pcgan_amd.load()
    dev = torch.device("cuda:0")
    for name, B, Cin, Cout, H, k, s, p in SHAPES:
        g = ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)
        x = torch.randn(B, H, H, Cin, device=dev)
        w = torch.randn(Cout, k, k, Cin, device=dev) * 0.02
        y = torch.empty(B, g.OH, g.OW, Cout, device=dev)
        flops = 2.0 * B * g.OH * g.OW * Cout * k * k * Cin
        M = B * g.OH * g.OW
        tiles = -(-M // 128) * -(-Cout // (128 if Cout > 64 else 64))
        kt = k * k * -(-Cin // 32)
        fn = lambda: ops.conv2d_fwd(g, x, w, None, out=y)
        ops.tune("fwd_splits", -1)
        base = bench(fn)
        res = []
        for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            if sp > kt // 4:
                continue
            ops.tune("fwd_splits", sp)
            res.append((bench(fn), sp))
        ops.tune("fwd_splits", -1)
        best = min(res)
        print(f"{name:52s} tiles {tiles:4d} ktiles {kt:4d} | built-in {base:6.1f} us ({flops / base / 1e6:5.1f} TF) | best s={best[1]:2d}: {best[0]:6.1f} us "
              f"({flops / best[0] / 1e6:5.1f} TF) | " + " ".join(f"{sp}:{t:.0f}" for t, sp in res), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Stream-K A/B over the forward / grad-input launches of the WGAN-GP updates (width 1024) and the DCGAN / counteRGAN layers whose
tile counts leave a remainder: time per launch with pcg_tune_set("stream_k", 0 / 1 / 2) and the sk_blocks choices.  Diagnostic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import pcgan_amd  # noqa: E402
from pcgan_amd import ops  # noqa: E402

# (name, op, B, Cin, Cout, H, k, s, p)
ONLY = os.environ.get('SK_ONLY')
SHAPES = [("critic conv2 B768", "fwd", 768, 256, 512, 13, 3, 2, 0), ("critic conv3 B768", "fwd", 768, 512, 1024, 6, 3, 2, 0),
          ("critic conv2 B256", "fwd", 256, 256, 512, 13, 3, 2, 0),
          ("G ConvT3 grad-input B256 (fwd k4)", "fwd", 256, 256, 512, 14, 4, 2, 1), ("G ConvT2 grad-input B256 (fwd k3)", "fwd", 256, 512, 1024, 7, 3, 2, 1),
          ("critic conv3 dgrad B256", "dgrad", 256, 512, 1024, 6, 3, 2, 0), ("critic conv3 dgrad B512", "dgrad", 512, 512, 1024, 6, 3, 2, 0),
          ("critic conv2 dgrad B256", "dgrad", 256, 256, 512, 13, 3, 2, 0), ("critic conv2 dgrad B512", "dgrad", 512, 256, 512, 13, 3, 2, 0),
          ("critic Linear dgrad B256", "dgrad", 256, 8192, 1024, 1, 1, 1, 0), ("critic Linear dgrad B512", "dgrad", 512, 8192, 1024, 1, 1, 1, 0),
          ("G ConvT1 fwd B256 (dgrad 1x1)", "dgrad", 256, 16384, 1024, 1, 1, 1, 0),
          ("G ConvT2 fwd B256 (dgrad k3 7->4)", "dgrad", 256, 512, 1024, 7, 3, 2, 1), ("G ConvT3 fwd B256 (dgrad k4 14->7)", "dgrad", 256, 256, 512, 14, 4, 2, 1),
          ("critic conv3 wgrad B512", "wgrad", 512, 512, 1024, 6, 3, 2, 0), ("critic conv3 wgrad B256", "wgrad", 256, 512, 1024, 6, 3, 2, 0),
          ("G ConvT2 wgrad B256", "wgrad", 256, 512, 1024, 7, 3, 2, 1),
          ("DCGAN D3 fwd B512", "fwd", 512, 128, 256, 16, 4, 2, 1), ("DCGAN D4 fwd B512", "fwd", 512, 256, 512, 8, 4, 2, 1),
          ("DCGAN D4 dgrad B512", "dgrad", 512, 256, 512, 8, 4, 2, 1)]


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
    for name, op, B, Cin, Cout, H, k, s, p in SHAPES:
        if ONLY and ONLY not in op:
            continue
        g = ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)
        w = torch.randn(Cout, k, k, Cin, device=dev) * 0.02
        flops = 2.0 * B * g.OH * g.OW * Cout * k * k * Cin
        if op == "fwd":
            x = torch.randn(B, H, H, Cin, device=dev)
            y = torch.empty(B, g.OH, g.OW, Cout, device=dev)
            fn = lambda: ops.conv2d_fwd(g, x, w, None, out=y)
        elif op == "wgrad":
            x = torch.randn(B, H, H, Cin, device=dev)
            dy = torch.randn(B, g.OH, g.OW, Cout, device=dev)
            dw = torch.zeros_like(w)
            fn = lambda: ops.conv2d_wgrad(g, x, dy, dw, True)
        else:
            dy = torch.randn(B, g.OH, g.OW, Cout, device=dev)
            dx = torch.empty(B, H, H, Cin, device=dev)
            fn = lambda: ops.conv2d_dgrad(g, dy, w, out=dx)
        res = {}
        for mode, blocks in ((0, -1), (1, -1), (2, 512), (2, 256), (2, 128)):
            ops.tune("stream_k", mode); ops.tune("sk_blocks", blocks)
            res[(mode, blocks)] = bench(fn)
        ops.tune("stream_k", -1); ops.tune("sk_blocks", -1)
        print(f"{name:38s} {op:5s} | off {res[(0, -1)]:6.1f} us ({flops / res[(0, -1)] / 1e6:5.1f} TF) | auto {res[(1, -1)]:6.1f} ({flops / res[(1, -1)] / 1e6:5.1f} TF) | "
              f"forced 512: {res[(2, 512)]:6.1f}  256: {res[(2, 256)]:6.1f}  128: {res[(2, 128)]:6.1f}", flush=True)


if __name__ == "__main__":
    main()

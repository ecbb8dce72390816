#!/usr/bin/env python3
"""What would ONE grid holding a layer's grad-input and weight-gradient launches gain?  Both need only dz.  Upper bound from two
streams (eager launches, event-timed): serial on one stream vs the two kernels free to overlap their tails.  Diagnostic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import pcgan_amd  # noqa: E402
from pcgan_amd import ops  # noqa: E402


def main():
    pcgan_amd.load()
    dev = torch.device("cuda:0")
    for name, (B, Cin, Cout, H, k, s, p) in {"D2": (512, 64, 128, 32, 4, 2, 1), "D3": (512, 128, 256, 16, 4, 2, 1), "D4": (512, 256, 512, 8, 4, 2, 1)}.items():
        g = ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)
        x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, g.OH, g.OW, Cout, device=dev)
        w = torch.randn(Cout, k, k, Cin, device=dev) * 0.02
        dx = torch.empty_like(x); dw = torch.zeros_like(w)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        p_hi, p_lo = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)   # r04: does a priority order the two grids (tail overlap only)?

        def serial():
            ops.conv2d_dgrad(g, dy, w, out=dx)
            ops.conv2d_wgrad(g, x, dy, dw, False)

        def overlapped():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                ops.conv2d_dgrad(g, dy, w, out=dx)
            with torch.cuda.stream(s2):
                ops.conv2d_wgrad(g, x, dy, dw, False)
            cur.wait_stream(s1); cur.wait_stream(s2)

        def prioritized():
            cur = torch.cuda.current_stream()
            p_hi.wait_stream(cur); p_lo.wait_stream(cur)
            with torch.cuda.stream(p_hi):
                ops.conv2d_dgrad(g, dy, w, out=dx)
            with torch.cuda.stream(p_lo):
                ops.conv2d_wgrad(g, x, dy, dw, False)
            cur.wait_stream(p_hi); cur.wait_stream(p_lo)

        res = {}
        for label, fn in (("serial", serial), ("two streams", overlapped), ("priority", prioritized), ("serial", serial), ("two streams", overlapped),
                          ("priority", prioritized)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(label, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"{name}: grad-input + weight gradient  serial {min(res['serial']):.1f} us | on two streams {min(res['two streams']):.1f} us | "
              f"high- / low-priority streams {min(res['priority']):.1f} us", flush=True)


if __name__ == "__main__":
    main()

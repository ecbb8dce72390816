#!/usr/bin/env python3
"""Visibility stress of the stream-K partial-tile exchange: the same forced stream-K launches over and over (the scratch slots are
reused by every launch, so a stale line in any XCD's L2 shows up as a changed result), interleaved with launches that dirty the
L2s.  Prints the number of runs whose output differs from the first.  Diagnostic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import pcgan_amd  # noqa: E402
from pcgan_amd import ops  # noqa: E402


def main():
    pcgan_amd.load()
    dev = torch.device("cuda:0")
    bad_total = 0
    for (B, Cin, Cout, H, k, s, p) in [(1024, 1024, 4608, 1, 1, 1, 0), (256, 256, 512, 13, 3, 2, 0), (768, 512, 1024, 6, 3, 2, 0)]:
        g = ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)
        ops.tune("stream_k", 0)
        x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.02
        ref0 = ops.conv2d_fwd(g, x, w, None).clone()
        ops.tune("stream_k", 2)
        first = None
        bad = 0
        worst = 0.0
        for it in range(200):
            if it % 3 == 0:   # other data through the caches and through the same scratch slots
                x2 = torch.randn(B, H, H, Cin, device=dev)
                ops.conv2d_fwd(g, x2, w, None)
            y = ops.conv2d_fwd(g, x, w, None)
            if first is None:
                first = y.clone()
            elif not torch.equal(y, first):
                bad += 1
            worst = max(worst, float((y - ref0).abs().max()))
        print(f"shape {(B, Cin, Cout, H, k, s, p)}: {bad} of 199 runs differ from the first; max |stream-K - plain| {worst:.3e}", flush=True)
        bad_total += bad
    ops.tune("stream_k", -1)
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()

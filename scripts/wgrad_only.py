#!/usr/bin/env python3
"""One conv-family op in a loop, for PMC passes (rocprofv3 --pmc ... -- python3 scripts/wgrad_only.py LAYER OP BATCH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgan_amd
from pcgan_amd import ops
sys.argv = [sys.argv[0]] + sys.argv[1:]
from scripts.conv_microbench import LAYERS
name, op, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
Cin, Cout, H, k, s, p = LAYERS[name]
dev = torch.device("cuda:0")
g = ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
dy = torch.randn(B, g.OH, g.OW, Cout, device=dev)
y = torch.empty(B, g.OH, g.OW, Cout, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
fn = {"fwd": lambda: ops.conv2d_fwd(g, x, w, None, out=y), "dgrad": lambda: ops.conv2d_dgrad(g, dy, w, None, out=dx),
      "wgrad": lambda: ops.conv2d_wgrad(g, x, dy, dw, False)}[op]
for _ in range(5):
    fn()
torch.cuda.synchronize()

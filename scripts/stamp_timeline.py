#!/usr/bin/env python3
"""Per-block timeline of conv_fwd_kernel from a -DPCG_STAMPS build (diagnostic).  PCG_LIB must point at such a build."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgan_amd
from pcgan_amd import ops
lib = pcgan_amd.load()
layer = sys.argv[1] if len(sys.argv) > 1 else "D4"
L = {"D2": (64, 128, 32), "D3": (128, 256, 16), "D4": (256, 512, 8)}[layer]
B = 512
g = ops.conv_geom(B, L[2], L[2], L[0], L[1], 4, 4, 2, 1)
dev = torch.device("cuda:0")
x = torch.randn(B, L[2], L[2], L[0], device=dev); w = torch.randn(L[1], 4, 4, L[0], device=dev) * 0.05
y = torch.empty(B, g.OH, g.OW, L[1], device=dev)
for _ in range(5):
    ops.conv2d_fwd(g, x, w, None, out=y)
torch.cuda.synchronize()
nblk = (B * g.OH * g.OW // 128) * (L[1] // 128)
buf = (ctypes.c_ulonglong * (nblk * 5))()
lib.pcg_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.pcg_debug_read_stamps(buf, nblk * 5)
st = np.array(buf[:], dtype=np.float64).reshape(nblk, 5)[:, :4] / 100.0   # us (100 MHz realtime)
t0 = st[:, 0].min()
st -= t0
print(f"{layer}: {nblk} blocks; kernel span {st[:, 3].max():.1f} us")
print("block start   : min %.1f  median %.1f  max %.1f" % (st[:, 0].min(), np.median(st[:, 0]), st[:, 0].max()))
print("loader init   : median %.2f us  max %.2f" % (np.median(st[:, 1] - st[:, 0]), (st[:, 1] - st[:, 0]).max()))
print("main loop     : median %.2f us  min %.2f max %.2f" % (np.median(st[:, 2] - st[:, 1]), (st[:, 2] - st[:, 1]).min(), (st[:, 2] - st[:, 1]).max()))
print("epilogue      : median %.2f us  max %.2f" % (np.median(st[:, 3] - st[:, 2]), (st[:, 3] - st[:, 2]).max()))
print("block end     : min %.1f  median %.1f  max %.1f" % (st[:, 3].min(), np.median(st[:, 3]), st[:, 3].max()))
order = np.argsort(st[:, 0])
late = st[order[-min(256, nblk):], 0]
print("start of the last 256 blocks: min %.1f median %.1f" % (late.min(), np.median(late)))
ktiles = 16 * L[0] // 32
print("ideal loop at 2.4 GHz: %.2f us (%d k-tiles x 4096 cycles)" % (ktiles * 4096 / 2400.0, ktiles))

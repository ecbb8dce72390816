"""Input pipeline pieces of the reference's training scripts.

    reference                                                        here
    -------------------------------------------------------------    -------------------------------------------------
    dset.MNIST(root, train=True) raw idx files                        read_idx_images / read_idx_labels (host, numpy)
    transforms.Resize(64) + ToTensor + Normalize((0.5,), (0.5,))      ResizeNormalize (device, bit-exact Pillow bilinear)
      dconv_gan/mnist/mnist_dcgan.py:42-46
"""
import gzip
import math
import struct

import numpy as np
import torch

from . import ops, _lib
from ._lib import PcgError

_PRECISION_BITS = 32 - 8 - 2


def read_idx_images(path):
    """MNIST `*-images-idx3-ubyte[.gz]`: big-endian magic 2051, count, rows, cols, then uint8 pixels -> [N, rows, cols] uint8."""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        magic, n, h, w = struct.unpack(">IIII", f.read(16))
        if magic != 2051:
            raise PcgError(f"{path}: not an idx3 image file (magic {magic})")
        return np.frombuffer(f.read(n * h * w), dtype=np.uint8).reshape(n, h, w)


def read_idx_labels(path):
    """MNIST `*-labels-idx1-ubyte[.gz]`: magic 2049, count, then uint8 labels -> [N] int64."""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        magic, n = struct.unpack(">II", f.read(8))
        if magic != 2049:
            raise PcgError(f"{path}: not an idx1 label file (magic {magic})")
        return np.frombuffer(f.read(n), dtype=np.uint8).astype(np.int64)


def _pillow_bilinear_coeffs(in_size, out_size):
    """[Pillow] Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter:
    per output coordinate the source window and its integer weights (22 fractional bits)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)])
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, (-0.5 + kk * (1 << _PRECISION_BITS)).astype(np.int64), (0.5 + kk * (1 << _PRECISION_BITS)).astype(np.int64))
    return bounds, ik.astype(np.int32), ksize


class ResizeNormalize:
    """transforms.Compose([Resize(size), ToTensor(), Normalize((mean,), (std,))]) for uint8 [N, H, W] batches, on the device.
    Square-image Resize(size) as the reference uses it (28x28 -> 64x64)."""

    def __init__(self, in_hw, out_hw, mean=0.5, std=0.5, device="cuda:0"):
        self.in_hw, self.out_hw, self.mean, self.std = tuple(in_hw), tuple(out_hw), float(mean), float(std)
        dev = torch.device(device)
        yb, yk, self.yks = _pillow_bilinear_coeffs(in_hw[0], out_hw[0])
        xb, xk, self.xks = _pillow_bilinear_coeffs(in_hw[1], out_hw[1])
        self.tables = tuple(torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (xb, xk, yb, yk))

    def __call__(self, images_u8):
        if images_u8.dtype != torch.uint8 or not images_u8.is_cuda or not images_u8.is_contiguous():
            raise PcgError("ResizeNormalize: expected a contiguous uint8 tensor [N, H, W] on the GPU")
        N = images_u8.shape[0]
        if tuple(images_u8.shape[-2:]) != self.in_hw:
            raise PcgError(f"ResizeNormalize: built for {self.in_hw} images, got {tuple(images_u8.shape[-2:])}")
        out = torch.empty((N, 1) + self.out_hw, dtype=torch.float32, device=images_u8.device)
        xb, xk, yb, yk = self.tables
        ops.check(_lib.load().pcg_resize8_normalize(ops._p(images_u8), N, self.in_hw[0], self.in_hw[1], self.out_hw[0], self.out_hw[1],
                                                    ops._p(xb), ops._p(xk), self.xks, ops._p(yb), ops._p(yk), self.yks, self.mean, self.std,
                                                    ops._p(out), ops._stream()), "pcg_resize8_normalize")
        return out

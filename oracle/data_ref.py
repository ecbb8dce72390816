"""ORACLE — test infrastructure only.  CPU restatement of the DCGAN input transform (dconv_gan/mnist/mnist_dcgan.py:42-46):
Resize(64) as Pillow computes it for 8-bit images (Resample.c: precompute_coeffs, normalize_coeffs_8bpc, horizontal then
vertical pass with rounding to uint8 after each), ToTensor (float32 / 255) and Normalize((0.5,), (0.5,)).
Pinned to Pillow itself: tests/golden/mnist_resize.npz."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_coeffs(in_size, out_size):
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                                  # bilinear filter support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) / filterscale)) for x in range(xmax)]
        ww = sum(w)
        w = [v / ww for v in w] if ww != 0.0 else w
        out.append((xmin, [int(0.5 + v * (1 << PRECISION_BITS)) if v >= 0 else int(-0.5 + v * (1 << PRECISION_BITS)) for v in w]))
    return out, ksize


def _pass(img, coeffs, axis):
    src = np.moveaxis(img.astype(np.int64), axis, -1)
    out = np.empty(src.shape[:-1] + (len(coeffs),), np.int64)
    for i, (x0, k) in enumerate(coeffs):
        ss = np.full(src.shape[:-1], 1 << (PRECISION_BITS - 1), np.int64)
        for j, kv in enumerate(k):
            ss = ss + src[..., x0 + j] * kv
        out[..., i] = np.clip(ss >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, -1, axis).astype(np.uint8)


def resize_u8(images, out_hw):
    """[N, H, W] uint8 -> [N, OH, OW] uint8, horizontal pass then vertical pass (ImagingResample order)."""
    xc, _ = bilinear_coeffs(images.shape[2], out_hw[1])
    yc, _ = bilinear_coeffs(images.shape[1], out_hw[0])
    return _pass(_pass(images, xc, 2), yc, 1)


def to_tensor_normalize(u8, mean=0.5, std=0.5):
    t = u8.astype(np.float32) / np.float32(255)
    return ((t - np.float32(mean)) / np.float32(std)).astype(np.float32)

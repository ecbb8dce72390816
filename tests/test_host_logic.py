"""CPU: host-side logic of the shim and of the launch planning (no kernels run)."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fastdiv_is_exact(tmp_path):
    """pcg::FastDiv (magic-number division used for every pixel decomposition) against '/' and '%' for n < 2^31."""
    src = tmp_path / "fd.cpp"
    src.write_text(r'''
#include "pcg_common.h"
#include <cstdio>
#include <cstdlib>
int main() {
  unsigned ds[] = {1, 2, 3, 4, 5, 7, 8, 13, 14, 16, 28, 32, 49, 64, 100, 196, 255, 256, 784, 1000, 4096, 65535, 65536, 1000003, 2147483647u};
  unsigned long long bad = 0, seed = 88172645463325252ull;
  for (unsigned d : ds) {
    pcg::FastDiv f(d);
    for (int i = 0; i < 2000000; ++i) {
      seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
      unsigned n = (unsigned)(seed >> 33);            // < 2^31
      if (i < 70000) n = (unsigned)i;                  // dense small range
      if (i >= 70000 && i < 70100) n = 2147483647u - (unsigned)(i - 70000);
      unsigned q, r; f.divmod(n, q, r);
      if (q != n / d || r != n % d) ++bad;
    }
  }
  printf("bad=%llu\n", bad);
  return bad != 0;
}
''')
    exe = tmp_path / "fd"
    inc = os.path.join(ROOT, "promptable-counterfactual-gan_amd", "csrc")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", inc, "-o", str(exe), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "bad=0" in r.stdout, r.stdout + r.stderr


def test_sequential_compiles_to_blocks():
    from pcgan_amd import dcgan, nn as pnn
    from pcgan_amd._lib import ACT_LRELU, ACT_RELU, ACT_SIGMOID, ACT_TANH
    g = dcgan.Generator({"g_hidden": 8, "z_dim": 16})
    blocks = pnn._compile(g.main)
    assert [b.transposed for b in blocks] == [True] * 5
    assert [b.bn is not None for b in blocks] == [True, True, True, True, False]
    assert [b.act for b in blocks] == [ACT_RELU] * 4 + [ACT_TANH]
    d = dcgan.Discriminator({"d_hidden": 8})
    blocks = pnn._compile(d.main)
    assert [b.bn is not None for b in blocks] == [False, True, True, True, False]
    assert [b.act for b in blocks] == [ACT_LRELU] * 4 + [ACT_SIGMOID]
    assert abs(blocks[0].slope - 0.2) < 1e-12
    # geometry of a transposed block = the adjoint convolution: x side is the layer's output
    geo, oh, ow = pnn._compile(g.main)[1].geom(4, 4, 4)
    assert (oh, ow) == (8, 8) and (geo.IH, geo.OH, geo.Cin, geo.Cout) == (8, 4, 32, 64)
    with pytest.raises(Exception, match="no libpcgan_hip implementation"):
        pnn._compile(nn.Sequential(nn.Conv2d(4, 4, 3), nn.Softplus()))


def test_state_dict_surface_matches_oracle_classes():
    from oracle import dcgan_ref as R
    from pcgan_amd import dcgan
    cfg = {"g_hidden": 8, "d_hidden": 8, "z_dim": 16}
    for ours, ref in ((dcgan.Generator(cfg), R.Generator(cfg)), (dcgan.Discriminator(cfg), R.Discriminator(cfg))):
        a, b = ours.state_dict(), ref.state_dict()
        assert list(a) == list(b)
        assert [tuple(v.shape) for v in a.values()] == [tuple(v.shape) for v in b.values()]
        ours.load_state_dict(b)  # round trip on the CPU (before any flattening)


def test_weights_init_matches_reference_rng_stream():
    from oracle import dcgan_ref as R
    from pcgan_amd import dcgan
    cfg = {"g_hidden": 8, "d_hidden": 8, "z_dim": 16}
    torch.manual_seed(1)
    a = dcgan.Generator(cfg); a.apply(dcgan.weights_init)
    torch.manual_seed(1)
    b = R.Generator(cfg); b.apply(R.weights_init)
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k


def test_adam_dense_span_detection():
    from pcgan_amd.optim import Adam
    flat = torch.zeros(64)
    w = flat[:32].view(2, 2, 2, 4).permute(0, 3, 1, 2)  # channels_last conv weight
    assert not w.is_contiguous()
    assert Adam._dense_span(w) == (flat.data_ptr(), 32)
    assert Adam._dense_span(flat[::2]) is None
    assert Adam._dense_span(flat[32:40]) == (flat.data_ptr() + 128, 8)


def test_epoch_permutation_is_dataloaders_order():
    """house.epoch_permutation must walk the rows exactly as DataLoader(TensorDataset, shuffle=True, drop_last=True) does after the
    same torch.manual_seed (house_sales_kc_usa/trainer.py:188,198), epoch after epoch: the trainer keeps the data set on the device
    and takes its batches there, so the row order is the one thing it has to reproduce of the loader."""
    from torch.utils.data import DataLoader, TensorDataset
    from pcgan_amd import house as H
    n, bs = 1000, 128
    torch.manual_seed(42)
    nn.Linear(4, 4)                                            # whatever consumes the generator before the loop (the critic's init)
    dl = DataLoader(TensorDataset(torch.arange(n)), batch_size=bs, shuffle=True, drop_last=True)
    want = [torch.cat([b[0] for b in dl]) for _ in range(3)]
    torch.manual_seed(42)
    nn.Linear(4, 4)
    got = [H.epoch_permutation(n)[:(n // bs) * bs] for _ in range(3)]
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    assert all(len(set(p.tolist())) == (n // bs) * bs for p in got)


def test_conv_launch_planning_at_the_bench_shapes():
    """The launch form each bench-shape convolution takes (csrc/conv_igemm.hip: plan_fwd, fwd_use_t64, plan_sk_shape, plan_skn_shape,
    dgrad_as_gemm, plan_wgrad) through pcg_conv_plan_describe — host logic, no device.  Pins the decisions DESIGN.md 3.1.1 tabulates:
    DCGAN's tile counts stay data-parallel (D4's 256 big tiles become 512 tiles of 64x128), the WGAN-GP layers whose tile counts do
    not fill the chip's 512 workgroup slots take stream-K when the stream has scratch and the r02 forms when it has none."""
    import ctypes
    from pcgan_amd import _lib
    lib = _lib.load()

    def plan(B, Cin, Cout, H, k, s, p, op, scratch):
        OH = (H + 2 * p - k) // s + 1
        g = _lib.ConvGeom(B, H, H, Cin, OH, OH, Cout, k, k, s, p)
        buf = ctypes.create_string_buffer(512)
        assert lib.pcg_conv_plan_describe(ctypes.byref(g), op, scratch, buf, 512) == 0
        return buf.value.decode()

    FWD, DGRAD, WGRAD = 0, 1, 2
    d2, d3, d4 = (512, 64, 128, 32, 4, 2, 1), (512, 128, 256, 16, 4, 2, 1), (512, 256, 512, 8, 4, 2, 1)
    for scratch in (0, 1):                                           # DCGAN: the same forms with or without scratch
        assert plan(*d2, FWD, scratch) == "128x128 tiles: 1024"
        assert plan(*d3, FWD, scratch) == "128x128 tiles: 512"
        assert plan(*d4, FWD, scratch) == "64x128 tiles: 512"
        assert plan(*d2, DGRAD, scratch) == "4 phases: 128x64 tiles: 1024 per phase"
        assert plan(*d3, DGRAD, scratch) == "4 phases: 128x128 tiles: 256 per phase"
        assert plan(*d3, WGRAD, scratch).startswith("128x128 tiles: 32 x 16 K-slices")
        assert plan(512, 1, 64, 64, 4, 2, 1, FWD, scratch).startswith("thin")
    # WGAN-GP, width 1024
    c2_768, c3_256 = (768, 256, 512, 13, 3, 2, 0), (256, 512, 1024, 6, 3, 2, 0)
    assert plan(*c2_768, FWD, 1) == "stream-K: 512 whole tiles + 352 tiles x 72 k-tiles over 512 ranges"
    assert plan(*c2_768, FWD, 0) == "128x128 tiles: 864"
    assert plan(256, 256, 512, 13, 3, 2, 0, FWD, 0).startswith("128x128 tiles: 288 x 3 K-slices")     # r02's answer to 288 tiles
    assert plan(256, 256, 512, 13, 3, 2, 0, FWD, 1).startswith("stream-K: 0 whole tiles + 288 tiles")
    assert plan(*c3_256, FWD, 1).startswith("128x128 tiles: 64 x 8 K-slices")                           # deep K split stays slabs
    assert plan(*c3_256, DGRAD, 1) == "GEMM + col2im: stream-K: 0 whole tiles + 288 tiles x 32 k-tiles over 512 ranges"
    assert plan(*c3_256, DGRAD, 0) == "GEMM + col2im: 128x128 tiles: 288"
    assert plan(*c3_256, WGRAD, 1).startswith("stream-K: 0 whole tiles + 288 tiles x 32 k-tiles over 512 ranges")
    t2 = (256, 512, 1024, 7, 3, 2, 1)                               # generator ConvT 1024 -> 512, 4x4 -> 7x7 (adjoint geometry)
    assert plan(*t2, DGRAD, 1).startswith("4 phases: stream-K over unequal phases: 392 tiles, 25600 k-tile iterations over 512 ranges")
    assert plan(*t2, DGRAD, 0) == "GEMM + col2im: 128x128 tiles: 1152"
    assert plan(256, 256, 512, 13, 3, 2, 0, DGRAD, 1).startswith("GEMM + col2im")                      # 13 -> 6: the phase form has MORE MACs
    assert plan(256, 8192, 1024, 1, 1, 1, 0, WGRAD, 1) == "128x128 tiles: 512, dw written by the epilogue"
    # counteRGAN 3x3 64 -> 64
    assert plan(1024, 64, 64, 28, 3, 1, 1, FWD, 1) == "128x64 tiles: 6272"
    assert plan(1024, 64, 64, 28, 3, 1, 1, WGRAD, 1).startswith("64x192 tiles: 3 x")

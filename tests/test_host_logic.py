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

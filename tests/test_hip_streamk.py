"""Hybrid stream-K launches of the forward / grad-input conv kernels (csrc/conv_igemm.hip: conv_fwd_sk_kernel, conv_dgrad_sk_kernel)
against the one-tile-per-workgroup launches of the same library and against fp64 samples: same terms, another order of the K sum.
The layers are the WGAN-GP shapes the form was built for (mnist_wgan_conditional.py:61-70,87-95 at width 1024) plus ragged ones."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


@pytest.fixture()
def pcg():
    import pcgan_amd
    pcgan_amd.load()
    yield pcgan_amd
    pcgan_amd.ops.tune("stream_k", -1)
    pcgan_amd.ops.tune("sk_blocks", -1)


# (B, Cin, Cout, H, k, s, p)
FWD = [(1024, 1024, 4608, 1, 1, 1, 0),      # the 288-tile GEMM of the critic's conv3 grad-input (as a 1x1 forward)
       (256, 256, 512, 13, 3, 2, 0),        # critic conv2: 288 tiles x 72 k-tiles
       (96, 96, 640, 9, 3, 1, 1),           # ragged M and K (Cin % 32 != 0), 61 x 5 tiles
       (512, 128, 256, 16, 4, 2, 1)]        # DCGAN D3 (512 tiles: stays data-parallel unless forced; forced it has no remainder)
DGRAD = [(256, 8192, 1024, 1, 1, 1, 0),     # Linear 8192 -> 1024 grad-input: 128 tiles x 32 k-tiles
         (256, 256, 512, 14, 4, 2, 1),      # G ConvT3: four uniform sub-pixel phases, 784 tiles
         (64, 96, 160, 10, 4, 2, 1)]        # ragged phases' tiles


def _geom(ops, B, Cin, Cout, H, k, s, p):
    return ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)


def _rows(B):
    """Images spread over the whole batch (r04; the first rounds checked [:4] only, i.e. the first M-tiles): the first, the last, and
    a few in between — with 1x1 geometry an image is one GEMM row, so these rows land in tiles across all of M."""
    return sorted({0, 1, B // 5, B // 3 + 1, B // 2, (2 * B) // 3 + 1, (4 * B) // 5, B - 2, B - 1} & set(range(B)))


@pytest.mark.parametrize("shape", FWD)
@pytest.mark.parametrize("blocks", [-1, 256])
def test_forward_streamk_matches_plain(pcg, shape, blocks):
    ops = pcg.ops
    B, Cin, Cout, H, k, s, p = shape
    g = _geom(ops, *shape)
    gen = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn((B, H, H, Cin), generator=gen).to(DEV)
    w = (torch.randn((Cout, k, k, Cin), generator=gen) * 0.05).to(DEV)
    b = torch.randn(Cout, generator=gen).to(DEV)
    ops.tune("stream_k", 0)
    ref = ops.conv2d_fwd(g, x, w, b, act=pcg.ops.ACT_LRELU, slope=0.2).clone()
    ops.tune("stream_k", 2); ops.tune("sk_blocks", blocks)
    y1 = ops.conv2d_fwd(g, x, w, b, act=pcg.ops.ACT_LRELU, slope=0.2).clone()
    y2 = ops.conv2d_fwd(g, x, w, b, act=pcg.ops.ACT_LRELU, slope=0.2).clone()
    assert torch.equal(y1, y2)                                   # the sum's order does not depend on who arrives last
    K = k * k * Cin
    tol = 16 * 2.0 ** -24 * K * float(x.abs().mean() * w.abs().mean()) * 4 + 1e-6     # same bound family as test_hip_benchshape
    assert float((y1 - ref).abs().max()) <= tol
    parts, arrivals = ops._sk_streams[(0, torch.cuda.current_stream().cuda_stream)]
    assert int(arrivals.view(torch.int32).abs().sum()) == 0      # counters are back to zero
    # fp64 samples
    rows = _rows(B)
    xs = x[rows].double().permute(0, 3, 1, 2).cpu(); ws = w.double().permute(0, 3, 1, 2).cpu()
    want = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xs, ws, b.double().cpu(), stride=s, padding=p), 0.2).permute(0, 2, 3, 1)
    assert float((y1[rows].double().cpu() - want).abs().max()) <= tol


@pytest.mark.parametrize("shape", DGRAD)
def test_grad_input_streamk_matches_plain(pcg, shape):
    ops = pcg.ops
    B, Cin, Cout, H, k, s, p = shape
    g = _geom(ops, *shape)
    gen = torch.Generator(device="cpu").manual_seed(11)
    dy = torch.randn((B, g.OH, g.OW, Cout), generator=gen).to(DEV)
    w = (torch.randn((Cout, k, k, Cin), generator=gen) * 0.05).to(DEV)
    ops.tune("stream_k", 0)
    ref = ops.conv2d_dgrad(g, dy, w).clone()
    ops.tune("stream_k", 2)
    d1 = ops.conv2d_dgrad(g, dy, w).clone()
    d2 = ops.conv2d_dgrad(g, dy, w).clone()
    assert torch.equal(d1, d2)
    K = k * k * Cout
    tol = 16 * 2.0 ** -24 * K * float(dy.abs().mean() * w.abs().mean()) * 4 + 1e-6
    assert float((d1 - ref).abs().max()) <= tol
    rows = _rows(B)
    want = torch.nn.functional.conv_transpose2d(dy[rows].double().permute(0, 3, 1, 2).cpu(), w.double().permute(0, 3, 1, 2).cpu(), stride=s, padding=p,
                                                output_padding=H - ((g.OH - 1) * s - 2 * p + k)).permute(0, 2, 3, 1)
    assert float((d1[rows].double().cpu() - want).abs().max()) <= tol


@pytest.mark.parametrize("shape", [(512, 512, 1024, 6, 3, 2, 0),     # critic conv3 at B = 512: 288 tiles x 64 k-tiles
                                   (256, 512, 1024, 7, 3, 2, 1),     # generator ConvT2's gradient: 288 tiles x 128 k-tiles
                                   (40, 96, 160, 9, 3, 1, 1)])       # ragged
@pytest.mark.parametrize("accumulate", [False, True])
def test_grad_weight_streamk_matches_slabs(pcg, shape, accumulate):
    ops = pcg.ops
    B, Cin, Cout, H, k, s, p = shape
    g = _geom(ops, *shape)
    gen = torch.Generator(device="cpu").manual_seed(13)
    x = torch.randn((B, H, H, Cin), generator=gen).to(DEV)
    dy = (torch.randn((B, g.OH, g.OW, Cout), generator=gen) * 0.1).to(DEV)
    base = torch.randn((Cout, k, k, Cin), generator=gen).to(DEV)
    outs = []
    for mode in (0, 2, 2):
        ops.tune("stream_k", mode)
        dw = base.clone()
        ops.conv2d_wgrad(g, x, dy, dw, accumulate)
        outs.append(dw)
    assert torch.equal(outs[1], outs[2])
    K = B * g.OH * g.OW
    tol = 16 * 2.0 ** -24 * K * float(x.abs().mean() * dy.abs().mean()) * 4 + 1e-6
    assert float((outs[1] - outs[0]).abs().max()) <= tol
    # fp64 sample: four output channels
    xs = x.double().permute(0, 3, 1, 2).cpu(); dys = dy.double().permute(0, 3, 1, 2).cpu()
    w0 = torch.zeros((Cout, Cin, k, k), dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xs, w0, None, stride=s, padding=p).backward(dys)
    ch = _rows(Cout)                                                # output channels = GEMM rows of the weight gradient, across all tiles
    want = w0.grad.permute(0, 2, 3, 1)[ch] + (base[ch].double().cpu() if accumulate else 0.0)
    assert float((outs[1][ch].double().cpu() - want).abs().max()) <= tol


@pytest.mark.parametrize("shape", [(256, 256, 512, 13, 3, 2, 0),     # critic conv2: phases of 49 / 42 / 42 / 36 pixels and 4 / 2 / 2 / 1 taps
                                   (256, 512, 1024, 6, 3, 2, 0),     # critic conv3
                                   (256, 512, 1024, 7, 3, 2, 1),     # generator ConvT2 (4 -> 7)
                                   (24, 96, 160, 9, 3, 2, 1)])       # ragged tiles
@pytest.mark.parametrize("blocks", [-1, 256])
def test_grad_input_with_unequal_phases_streamk(pcg, shape, blocks):
    """conv_dgrad_skn_kernel (every tile stream-K, any number of segments per workgroup) against the plain phase kernel and fp64."""
    ops = pcg.ops
    B, Cin, Cout, H, k, s, p = shape
    g = _geom(ops, *shape)
    gen = torch.Generator(device="cpu").manual_seed(17)
    dy = torch.randn((B, g.OH, g.OW, Cout), generator=gen).to(DEV)
    w = (torch.randn((Cout, k, k, Cin), generator=gen) * 0.05).to(DEV)
    bias = torch.randn(Cin, generator=gen).to(DEV)
    ops.tune("dgrad_gemm", 0)                                    # the phase form, not GEMM + col2im
    try:
        ops.tune("stream_k", 0)
        ref = ops.conv2d_dgrad(g, dy, w, bias, act=pcg.ops.ACT_LRELU, slope=0.2).clone()
        ops.tune("stream_k", 1); ops.tune("sk_blocks", blocks)
        d1 = ops.conv2d_dgrad(g, dy, w, bias, act=pcg.ops.ACT_LRELU, slope=0.2).clone()
        d2 = ops.conv2d_dgrad(g, dy, w, bias, act=pcg.ops.ACT_LRELU, slope=0.2).clone()
    finally:
        ops.tune("dgrad_gemm", -1)
    assert torch.equal(d1, d2)
    K = k * k * Cout
    tol = 16 * 2.0 ** -24 * K * float(dy.abs().mean() * w.abs().mean()) * 4 + 1e-6
    assert float((d1 - ref).abs().max()) <= tol
    parts, arrivals = ops._sk_streams[(0, torch.cuda.current_stream().cuda_stream)]
    assert int(arrivals.view(torch.int32).abs().sum()) == 0
    rows = _rows(B)
    want = torch.nn.functional.conv_transpose2d(dy[rows].double().permute(0, 3, 1, 2).cpu(), w.double().permute(0, 3, 1, 2).cpu(), bias.double().cpu(),
                                                stride=s, padding=p, output_padding=H - ((g.OH - 1) * s - 2 * p + k))
    want = torch.nn.functional.leaky_relu(want, 0.2).permute(0, 2, 3, 1)
    assert float((d1[rows].double().cpu() - want).abs().max()) <= tol


def test_convtranspose_with_fused_statistics_takes_the_unequal_phase_form(pcg):
    """The generator's ConvTranspose2d(1024 -> 512, k3 s2 p1, 4x4 -> 7x7) + BatchNorm statistics at the bench batch
    (mnist_wgan_conditional.py:64-66): grad-input kernel with the statistics in its epilogue, phases of 16 / 12 / 12 / 9 pixels and
    4 / 2 / 2 / 1 taps -> conv_dgrad_skn_kernel.  Output, batch statistics and running statistics against the plain launch."""
    ops = pcg.ops
    B, Cin, Cout, H, k, s, p = 256, 512, 1024, 7, 3, 2, 1          # adjoint geometry: the ConvT's output is this conv's input
    g = _geom(ops, B, Cin, Cout, H, k, s, p)
    gen = torch.Generator(device="cpu").manual_seed(23)
    a = torch.randn((B, g.OH, g.OW, Cout), generator=gen).to(DEV)
    w = (torch.randn((Cout, k, k, Cin), generator=gen) * 0.03).to(DEV)
    res = []
    for mode in (0, 1, 1):
        ops.tune("stream_k", mode)
        rm, rv, nbt = torch.zeros(Cin, device=DEV), torch.ones(Cin, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        z, mean, invstd = ops.conv_bn_train(g, a, w, None, True, 1e-5, 0.1, rm, rv, nbt)
        res.append((z.clone(), mean.clone(), invstd.clone(), rm.clone(), rv.clone()))
    for x, y in zip(res[1], res[2]):
        assert torch.equal(x, y)                                    # run-to-run identical
    K = k * k * Cout
    tol = 16 * 2.0 ** -24 * K * float(a.abs().mean() * w.abs().mean()) * 4 + 1e-6
    assert float((res[1][0] - res[0][0]).abs().max()) <= tol
    for i in (1, 2, 3, 4):
        np.testing.assert_allclose(res[1][i].cpu().numpy(), res[0][i].cpu().numpy(), rtol=2e-4, atol=2e-5)
    zd = res[1][0].double()
    np.testing.assert_allclose(res[1][1].cpu().numpy(), zd.mean((0, 1, 2)).cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(1024, 1024, 4608, 1, 1, 1, 0), (256, 256, 512, 13, 3, 2, 0), (768, 512, 1024, 6, 3, 2, 0)])
def test_partial_tile_exchange_is_stable_under_cache_pressure(pcg, shape):
    """The visibility stress of scripts/probes/streamk_stress.py as a test (20 rounds, not 199): the same forced stream-K launch over
    and over — every launch reuses the same scratch slots, so a stale line in any XCD's L2 shows as a changed result — interleaved
    with launches on other data that dirty the caches and the slots.  (The r03 inline-asm store bug differed in 199 of 199 runs.)"""
    ops = pcg.ops
    B, Cin, Cout, H, k, s, p = shape
    g = _geom(ops, *shape)
    x = torch.randn(B, H, H, Cin, device=DEV)
    w = torch.randn(Cout, k, k, Cin, device=DEV) * 0.02
    ops.tune("stream_k", 0)
    ref = ops.conv2d_fwd(g, x, w, None).clone()
    ops.tune("stream_k", 2)
    first = None
    for it in range(20):
        if it % 3 == 0:
            ops.conv2d_fwd(g, torch.randn(B, H, H, Cin, device=DEV), w, None)
        y = ops.conv2d_fwd(g, x, w, None)
        if first is None:
            first = y.clone()
        else:
            assert torch.equal(y, first), f"run {it} differs from the first"
    tol = 16 * 2.0 ** -24 * k * k * Cin * float(x.abs().mean() * w.abs().mean()) * 4 + 1e-6
    assert float((first - ref).abs().max()) <= tol

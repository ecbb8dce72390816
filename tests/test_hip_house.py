"""GPU parity, tabular CounteRGAN (conditional_counteRGAN/house_sales_kc_usa): the drop-ins in pcgan_amd.house against
  (a) one batch of the reference's own train_countergan (tests/golden/house_ref_b64.npz, the draws it made recorded),
  (b) the reference modules with the checkpoints the reference ships (tests/golden/house_trained_eval.npz), and
  (c) the oracle restatement (oracle/house_ref.py) evaluated live in float64 (truth) and float32 (noise floor).
One-hot / concatenation / hard samples must be bit-exact; floating point within the tolerances stated in each assert."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import house_ref as HR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pcg():
    import pcgan_amd
    from pcgan_amd import house  # noqa: F401
    return pcgan_amd


@pytest.fixture(scope="module")
def hgold(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "house_ref_b64.npz")))


def _dev(t):
    return t.to(DEV).contiguous()


def _close(a, b, rtol, atol, msg=""):
    np.testing.assert_allclose(a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy(), rtol=rtol, atol=atol, err_msg=msg)


# ---------------------------------------------------------------------------------------------------------------------
def test_gemm_all_layouts(pcg):
    """pcg_gemm against float64 matmul: ragged sizes (the layer widths of this model are 38, 21, 17, 10, 9, 30, ...),
    transposes, strided column slices, bias, accumulate.  fp32 FMA accumulation over K <= 300: rtol 2e-5."""
    ops = pcg.ops
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in [(1, 1, 1), (64, 32, 38), (130, 9, 32), (7, 70, 21), (256, 256, 17), (33, 1, 128), (1, 13, 300)]:
        for tA in (False, True):
            for tB in (False, True):
                A = torch.randn((K, M) if tA else (M, K), generator=g)
                Bm = torch.randn((N, K) if tB else (K, N), generator=g)
                bias = torch.randn(N, generator=g)
                ref = (A.double().T if tA else A.double()) @ (Bm.double().T if tB else Bm.double()) + bias.double()
                out = ops.gemm(_dev(A), _dev(Bm), M, N, K, transA=tA, transB=tB, bias=_dev(bias))
                _close(out, ref, 2e-5, 2e-5, f"{M}x{N}x{K} tA={tA} tB={tB}")
    # long K, 1..4 output columns (the critic heads: Linear(1024 -> 1)): the one-wave-per-row kernel; K % 4 != 0 and an unaligned
    # row stride take its scalar paths; accumulate + fused LeakyReLU on top
    for (M, N, K, lda) in [(768, 1, 1024, 1024), (5, 4, 1027, 1027), (130, 3, 300, 304), (9, 2, 256, 260)]:
        Aw, Bm, bias, C0 = torch.randn(M, lda, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
        ref = Aw[:, :K].double() @ Bm.double().T + bias.double()
        _close(ops.gemm(_dev(Aw), _dev(Bm), M, N, K, transB=True, lda=lda, bias=_dev(bias)), ref, 2e-5, 1e-4, f"rowdot {M}x{N}x{K}")
        Cd = _dev(C0)
        ops.gemm(_dev(Aw), _dev(Bm), M, N, K, transB=True, lda=lda, bias=_dev(bias), out=Cd, accumulate=True, act=pcg.ops.ACT_LRELU, slope=0.2)
        _close(Cd, torch.nn.functional.leaky_relu(ref + C0.double(), 0.2), 2e-5, 1e-4, f"rowdot acc+act {M}x{N}x{K}")
    # column slices: A = columns 3..3+K of a wider buffer, C = columns 5..5+N of a wider buffer, accumulate on top
    M, N, K = 50, 9, 12
    Aw, Bm, Cw = torch.randn(M, 40, generator=g), torch.randn(N, K, generator=g), torch.randn(M, 30, generator=g)
    Cd = _dev(Cw)
    ops.gemm(_dev(Aw)[:, 3:], _dev(Bm), M, N, K, transB=True, lda=40, out=Cd[:, 5:], ldc=30, accumulate=True)
    ref = Cw.double().clone()
    ref[:, 5:5 + N] += Aw[:, 3:3 + K].double() @ Bm.double().T
    _close(Cd, ref, 2e-5, 2e-5)


def test_linear_wgrad_slab_split(pcg):
    """pcg_linear_wgrad: weight and bias gradient in one launch; batches from one slab to many; strided dy; accumulate;
    repeated calls (the ticket buffer must come back zero) and bit-identical results run to run (fixed slab order)."""
    ops = pcg.ops
    g = torch.Generator().manual_seed(11)
    for (B, O, I) in [(1, 1, 1), (64, 32, 38), (128, 9, 32), (1000, 32, 21), (4096, 256, 256), (5000, 64, 128), (20000, 1, 128), (777, 13, 32)]:
        dy, x = torch.randn(B, O + 5, generator=g), torch.randn(B, I, generator=g)
        dW0, db0 = torch.randn(O, I, generator=g), torch.randn(O, generator=g)
        refW = dy[:, 2:2 + O].double().T @ x.double()
        refb = dy[:, 2:2 + O].double().sum(0)
        dyd, xd = _dev(dy), _dev(x)
        tolW, tolb = 3e-6 * float(refW.abs().max()) + 1e-6, 3e-6 * float(refb.abs().max()) + 1e-6   # fp32 sums over B rows
        outs = []
        for rep in range(2):
            dW, db = torch.empty(O, I, device=DEV), torch.empty(O, device=DEV)
            ops.linear_wgrad(dyd[:, 2:], xd, B, O, I, dW, db, ldy=O + 5)
            _close(dW, refW, 2e-5, tolW * 10, f"dW {B}x{O}x{I}"); _close(db, refb, 2e-5, tolb * 10, f"db {B}x{O}x{I}")
            outs.append((dW.clone(), db.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        dW, db = _dev(dW0), _dev(db0)
        ops.linear_wgrad(dyd[:, 2:], xd, B, O, I, dW, db, ldy=O + 5, accumulate_w=True, accumulate_b=True)
        _close(dW, refW + dW0.double(), 2e-5, tolW * 10); _close(db, refb + db0.double(), 2e-5, tolb * 10)
        dW = torch.empty(O, I, device=DEV)
        ops.linear_wgrad(dyd[:, 2:], xd, B, O, I, dW, None, ldy=O + 5)                     # no bias
        assert torch.equal(dW, outs[0][0])


@pytest.mark.parametrize("B", [64, 1000, 4096])
def test_grouped_weight_gradient_on_the_matrix_cores(pcg, B):
    """pcg_linear_wgrad_grouped: the generator's 29 layers (strided operands, shared inputs, ragged widths 2..38) and a 128x64 layer
    (eight 32x32 tiles) in ONE launch against float64; batches on either side of the one-block / eight-partials-per-tile switch and a
    ragged one; accumulate on top of existing values; two runs are bit-identical (fixed reduction order, no float atomics)."""
    ops = pcg.ops
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(B)
    rnd = lambda *sh: _dev(torch.randn(*sh, generator=g) * 0.01)
    T, K = 70, 38
    inp = _dev(torch.rand(B, K, generator=g)); cond = inp[:, 17:]
    Hs = _dev(torch.rand(6, B, 32, generator=g)); A1 = _dev(torch.rand(5, B, 32, generator=g))
    DZIN, DZ1, DZ2, DG, DB, DC, DL = rnd(B, 32), rnd(5, B, 32), rnd(5, B, 32), rnd(5, B, 32), rnd(5, B, 32), rnd(B, 10), rnd(B, T)
    big_dy, big_x = rnd(B, 128), _dev(torch.rand(B, 64, generator=g))
    seg = [0, 9, 39, 45, 47, 52, 57, 70]
    spec = [(DZIN, inp, 32, 38, None, None)]
    for k in range(5):
        spec += [(DZ1[k], Hs[k], 32, 32, None, None), (DZ2[k], A1[k], 32, 32, None, None), (DG[k], cond, 32, 21, None, K), (DB[k], cond, 32, 21, None, K)]
    spec.append((DC, Hs[5], 10, 32, None, None))
    spec += [(DL[:, seg[s_]:], Hs[5], seg[s_ + 1] - seg[s_], 32, T, None) for s_ in range(7)]
    spec.append((big_dy, big_x, 128, 64, None, None))

    def run():
        items = []
        for dy, x, O, I, ldy, ldx in spec:
            dW = torch.full((O, I), 0.5, device=dev); db = torch.full((O,), -0.25, device=dev)
            items.append((dy, x, O, I, dW, db, ldy or O, ldx or I, True, True))
        ops.linear_wgrad_grouped(items, B, dev)
        return items
    a, b = run(), run()
    for (dy, x, O, I, ldy, ldx), ia, ib in zip(spec, a, b):
        rw = dy[:, :O].double().T @ x[:, :I].double() + 0.5
        rb = dy[:, :O].double().sum(0) - 0.25
        _close(ia[4], rw, 0, 2e-6 * float((rw - 0.5).abs().max()) + 1e-7, f"dW {O}x{I}")
        _close(ia[5], rb, 0, 2e-6 * float((rb + 0.25).abs().max()) + 1e-7, f"db {O}")
        assert torch.equal(ia[4], ib[4]) and torch.equal(ia[5], ib[5])


def test_tabular_elementwise_ops(pcg):
    ops = pcg.ops
    g = torch.Generator().manual_seed(2)
    B = 77
    idx = torch.randint(0, 4, (B,), generator=g)
    assert torch.equal(ops.onehot(_dev(idx), 4).cpu(), F.one_hot(idx, 4).float())                      # bit-exact
    a, b = torch.randn(B, 17, generator=g), torch.randn(B, 4, generator=g)
    cat = ops.concat_cols(_dev(a), _dev(b))
    assert torch.equal(cat.cpu(), torch.cat([a, b], 1))                                                # bit-exact
    da, db = ops.split_cols(cat, 17, 4)
    assert torch.equal(da.cpu(), a) and torch.equal(db.cpu(), b)
    da, db = ops.split_cols(cat, 17, 4, need_b=False)
    assert db is None and torch.equal(da.cpu(), a)
    # FiLM
    gm, h, bt, dy = (torch.randn(B, 32, generator=g) for _ in range(4))
    _close(ops.film_fwd(_dev(gm), _dev(h), _dev(bt)), gm.double() * h.double() + bt.double(), 1e-6, 1e-6)
    dg, dh = ops.film_bwd(_dev(dy), _dev(gm), _dev(h))
    _close(dg, dy * h, 1e-6, 1e-7); _close(dh, dy * gm, 1e-6, 1e-7)
    # mean
    x = torch.randn(B, 1, generator=g)
    _close(ops.mean_fwd(_dev(x)), x.double().mean().view(1), 1e-6, 1e-7)
    gout = torch.tensor(-1.5)
    _close(ops.mean_bwd(_dev(gout), 1.0, _dev(x)), torch.full_like(x, -1.5 / B), 1e-6, 0)


def test_gumbel_softmax_heads_and_residual_assembly(pcg):
    ops = pcg.ops
    g = torch.Generator().manual_seed(3)
    B, sizes, D = 53, [9, 30, 6, 2, 5, 5, 13], 17
    cat_idx, cont_idx = [0, 1, 4, 5, 6, 7, 8], [2, 3, 9, 10, 11, 12, 13, 14, 15, 16]
    seg = [0] + list(np.cumsum(sizes)); T = int(seg[-1])
    mk = lambda v: torch.tensor([int(i) for i in v], dtype=torch.int32, device=DEV)  # noqa: E731
    segd, catd, contd = mk(seg), mk(cat_idx), mk(cont_idx)
    logits = torch.randn(B, T, generator=g, dtype=torch.float64, requires_grad=True)
    noise = -torch.empty(B, T, dtype=torch.float64).exponential_(generator=g).log()
    cont = torch.randn(B, len(cont_idx), generator=g, dtype=torch.float64, requires_grad=True)
    x = torch.rand(B, D, generator=g, dtype=torch.float64)
    norm = torch.cat([torch.arange(n, dtype=torch.float64) / max(1, n - 1) for n in sizes])
    tau = 0.5
    # float64 reference with autograd, written as the reference trainer writes it (trainer.py:266-279)
    soft = torch.cat([((logits[:, seg[s]:seg[s + 1]] + noise[:, seg[s]:seg[s + 1]]) / tau).softmax(-1) for s in range(len(sizes))], 1)
    res = torch.zeros(B, D, dtype=torch.float64)
    for i, f in enumerate(cont_idx):
        res[:, f] = cont[:, i]
    for s, f in enumerate(cat_idx):
        res[:, f] = soft[:, seg[s]:seg[s + 1]].matmul(norm[seg[s]:seg[s + 1]]) - x[:, f]
    dres = torch.randn(B, D, generator=g, dtype=torch.float64)
    res.backward(dres)
    y, yh = ops.gumbel_softmax_fwd(_dev(logits.detach().float()), _dev(noise.float()), segd, tau, hard=True)
    _close(y, soft, 2e-5, 1e-7)                                                    # expf / fp32 softmax
    for s in range(len(sizes)):                                                    # hard: one-hot of the soft argmax — exact
        blk = yh[:, seg[s]:seg[s + 1]].cpu()
        assert torch.equal(blk.sum(1), torch.ones(B)) and torch.equal(blk.argmax(1), soft[:, seg[s]:seg[s + 1]].argmax(1))
    r = ops.assemble_residual_fwd(_dev(cont.detach().float()), contd, y, segd, catd, _dev(norm.float()), _dev(x.float()))
    _close(r, res, 2e-5, 2e-6)
    dcont, dsamp = ops.assemble_residual_bwd(_dev(dres.float()), len(cont_idx), contd, segd, T, catd, _dev(norm.float()))
    _close(dcont, cont.grad, 1e-6, 1e-7)
    dl = ops.gumbel_softmax_bwd(dsamp, y, segd, tau)
    _close(dl, logits.grad, 1e-4, 1e-6)                                             # y*(dy - <dy,y>): cancellation at fp32 eps * |dy|


def test_spectral_norm_matches_torch(pcg):
    """pcg_spectral_norm_fwd/bwd against torch.nn.utils.spectral_norm on the CPU: the in-place power iteration (u, v after
    each forward), W / sigma, and the gradient that reaches weight_orig."""
    ops = pcg.ops
    torch.manual_seed(4)
    for (O, I) in [(32, 21), (64, 32), (128, 64), (1, 128)]:
        lin = torch.nn.utils.spectral_norm(torch.nn.Linear(I, O))
        w_orig, u, v = _dev(lin.weight_orig.detach()), _dev(lin.weight_u.clone()), _dev(lin.weight_v.clone())
        x = torch.randn(40, I)
        for it in range(2):                                                        # two training-mode forwards: u, v keep moving
            lin.zero_grad()
            out = lin(x)
            out.square().sum().backward()
            w_bar, sigma, uu, vu = ops.spectral_norm_fwd(w_orig, u, v, 1e-12, True)
            _close(u, lin.weight_u, 1e-5, 1e-6, f"u {O}x{I} it{it}"); _close(v, lin.weight_v, 1e-5, 1e-6, f"v {O}x{I} it{it}")
            assert torch.equal(uu, u) and torch.equal(vu, v)
            _close(w_bar, lin.weight.detach(), 1e-5, 1e-6, f"w_bar {O}x{I}")
            dwb = (2 * out.detach()).T @ x                                          # d(sum out^2)/d w_bar
            gw = torch.zeros(O, I, device=DEV)
            ops.spectral_norm_bwd(_dev(dwb), w_bar, uu, vu, sigma, gw, False)
            _close(gw, lin.weight_orig.grad, 2e-4, 1e-5 * float(lin.weight_orig.grad.abs().max()), f"dW {O}x{I}")
        lin.eval()                                                                  # eval: no power iteration
        ub = u.clone()
        w_bar, _, _, _ = ops.spectral_norm_fwd(w_orig, u, v, 1e-12, False)
        lin(x)
        assert torch.equal(ub, u)
        _close(w_bar, lin.weight.detach(), 1e-5, 1e-6)


# ---------------------------------------------------------------------------------------------------------------------
def _load_golden_nets(pcg, hgold):
    H = pcg.house
    torch.manual_seed(0)
    C = H.NNClassifier(17, 4)                                   # same construction order / seed as make_golden.py
    G = H.ResidualGenerator(17, 32, 4, H.CONFIG["continuous_idx"], H.CONFIG["categorical_info"], tau=0.5)
    D = H.Discriminator(17, 32, 4)
    assert [f"init.G.{k}" for k in G.state_dict()] == [k for k in hgold if k.startswith("init.G.")]
    assert [f"init.D.{k}" for k in D.state_dict()] == [k for k in hgold if k.startswith("init.D.")]   # weight_orig / _u / _v
    G.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in hgold.items() if k.startswith("init.G.")})
    D.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in hgold.items() if k.startswith("init.D.")})
    C.eval()
    for p in C.parameters():
        p.requires_grad = False
    return G.to(DEV), D.to(DEV), C.to(DEV)


def _oracle_nets(hgold, dtype):
    G, D, clf = HR.build(seed=0)
    G.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in hgold.items() if k.startswith("init.G.")})
    D.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in hgold.items() if k.startswith("init.D.")})
    return G.to(dtype), D.to(dtype), clf.to(dtype)


def test_golden_reference_train_loop_batch(pcg, hgold):
    """One batch of the reference's train_countergan: logged losses, G gradients, G parameters after Adam; the critic's
    state after its Adam step (weight_orig, bias, u, v) against the float32 oracle run on the same draws."""
    H = pcg.house
    G, D, C = _load_golden_nets(pcg, hgold)
    x, y, t, m = (_dev(torch.from_numpy(hgold[f"in.{n}"])) for n in ("x", "y", "target_y", "mask"))
    gumbel = {int(k[7:]): torch.from_numpy(v) for k, v in hgold.items() if k.startswith("gumbel.")}
    noise = G.pack_noise({f: _dev(v) for f, v in gumbel.items()})
    opt_g, opt_d = H.make_optimizers(G, D)
    norm = H.cat_norm_maps(G, H.CONFIG, torch.device(DEV))
    out = H.train_step(G, D, C, opt_g, opt_d, x, y, t, m, norm, gumbel=noise, skip_dead_d_wgrad=False)
    log = str(hgold["log"])

    def logged(pat):
        return float(re.search(pat, log).group(1))
    # printed with 3-6 decimals: half a unit of the last printed digit + fp32 noise
    assert abs(torch.sigmoid(out["D_real"]).mean().item() - logged(r"D\(real\)=([0-9.]+)")) <= 6e-4
    assert abs(torch.sigmoid(out["D_fake_forG"]).mean().item() - logged(r"D\(fake\)=([0-9.]+)")) <= 6e-4
    assert abs(out["g_adv"].item() - logged(r"g_adv=(-?[0-9.]+)")) <= 7e-5 and abs(out["g_cls"].item() - logged(r"g_cls=([0-9.]+)")) <= 7e-5
    assert abs(out["reg"].item() - logged(r"reg=([0-9.]+)")) <= 2e-6 and abs(out["mask_pen"].item() - logged(r"mask_pen=([0-9.]+)")) <= 7e-6
    assert abs(out["D_loss"].item() - logged(r"\] D: (-?[0-9.]+)")) <= 7e-5 and abs(out["G_loss"].item() - logged(r", G: (-?[0-9.]+)")) <= 7e-5
    lr = H.CONFIG["lr_G"]
    # The G gradients pass through the critic AFTER its Adam step, and Adam's first step is ~lr * sign(g): a critic weight whose
    # gradient is at the fp32 noise level lands 2*lr apart under a different (equally valid) summation order, which moves the G
    # gradients by ~1e-4 of their scale.  The floor is measured, not assumed: the oracle on this batch in float32 and in float64.
    noise = {}
    for dt in (torch.float32, torch.float64):
        oG, oD, oC = _oracle_nets(hgold, dt)
        o_g, o_d = HR.make_optimizers(oG, oD)
        HR.house_step(oG, oD, oC, o_g, o_d, x.cpu().to(dt), y.cpu(), t.cpu(), m.cpu().to(dt), {f: v.to(dt) for f, v in gumbel.items()},
                      {f: v.to(dt) for f, v in HR.cat_norm_maps().items()})
        noise[dt] = {n: p.grad.double().clone() for n, p in oG.named_parameters()}
    for n, p in G.named_parameters():
        ref = hgold[f"grad.G.{n}"]
        floor = 3 * float((noise[torch.float32][n] - noise[torch.float64][n]).abs().max())
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=5e-4, atol=max(2e-6 + 2e-5 * np.abs(ref).max(), floor), err_msg=f"grad {n}")
    for k, v in G.state_dict().items():
        gk = f"grad.G.{k}"
        if gk in hgold and np.abs(hgold[gk]).max() < 1e-6:
            # a Linear bias in front of BatchNorm1d has an exactly-zero gradient; what is stored is fp32 noise whose sign
            # Adam turns into a +-lr step: only bound the move
            assert np.abs(v.cpu().numpy() - hgold[f"final.G.{k}"]).max() <= 2.2 * lr, k
            continue
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(hgold[f"final.G.{k}"])
            continue
        d = np.abs(v.cpu().numpy().astype(np.float64) - hgold[f"final.G.{k}"])
        bad = d > 3e-5 + 1e-4 * np.abs(hgold[f"final.G.{k}"])
        # Adam's first step is ~lr * sign(g): an entry whose gradient is at the fp32 noise level may land elsewhere within +-lr
        assert bad.sum() <= max(2, 0.01 * d.size) and d.max() <= 2.2 * lr, (f"final {k}", int(bad.sum()), float(d.max()))
    # critic: float32 oracle on the same draws (the golden file holds only D's initial state)
    oG, oD, oC = _oracle_nets(hgold, torch.float32)
    o_g, o_d = HR.make_optimizers(oG, oD)
    HR.house_step(oG, oD, oC, o_g, o_d, x.cpu(), y.cpu(), t.cpu(), m.cpu(), gumbel, HR.cat_norm_maps())
    for k, v in D.state_dict().items():
        ref = oD.state_dict()[k]
        if k.endswith("weight_u") or k.endswith("weight_v"):
            _close(v, ref, 1e-4, 1e-5, k)                     # three power iterations per step (D(real), D(fake), D(fake) for G)
        else:
            # Adam's first step moves every weight by ~lr*sign(g): entries whose gradient is at noise level may flip (the Wasserstein
            # critic's gradient is a difference of two batch means: a few entries per layer cancel to fp32 noise, and which way they
            # fall depends on the summation order — here MFMA accumulation vs the oracle's sequential sums)
            d = (v.cpu() - ref).abs()
            assert float(d.max()) <= 2.2 * H.CONFIG["lr_D"] and int((d > 2e-5).sum()) <= max(2, 0.03 * d.numel()), (k, float(d.max()), int((d > 2e-5).sum()))


@pytest.mark.parametrize("batch", [256, 4096])
def test_step_vs_oracle_float64(pcg, hgold, batch):
    """Synthetic batch: every loss, every G gradient and every D gradient of one step against the
    float64 oracle; tolerance = max(1e-4 relative to the tensor's max, 3x the float32 oracle's own distance from float64).
    batch 4096 = BASELINE config 5 (r04: the whole step at the bench size against the oracle; ~0.1 s of CPU work)."""
    H = pcg.house
    G, D, C = _load_golden_nets(pcg, hgold)
    x, y, t, m, gumbel = HR.synthetic_batch(batch, seed=9, dtype=torch.float64)
    res = {}
    for dt in (torch.float64, torch.float32):
        oG, oD, oC = _oracle_nets(hgold, dt)
        o_g, o_d = HR.make_optimizers(oG, oD)
        losses = HR.house_step(oG, oD, oC, o_g, o_d, x.to(dt), y, t, m.to(dt), {f: v.to(dt) for f, v in gumbel.items()},
                               {f: v.to(dt) for f, v in HR.cat_norm_maps().items()})
        res[dt] = (losses, {n: p.grad.double().clone() for n, p in oG.named_parameters()},
                   {n: p.grad.double().clone() for n, p in oD.named_parameters()},
                   {k: v.double().clone() for k, v in oG.state_dict().items()})
    opt_g, opt_d = H.make_optimizers(G, D)
    norm = H.cat_norm_maps(G, H.CONFIG, torch.device(DEV))
    noise = G.pack_noise({f: _dev(v.float()) for f, v in gumbel.items()})
    out = H.train_step(G, D, C, opt_g, opt_d, _dev(x.float()), _dev(y), _dev(t), _dev(m.float()), norm, gumbel=noise,
                       skip_dead_d_wgrad=False)
    l64, g64, d64, s64 = res[torch.float64]
    l32, g32, d32, _ = res[torch.float32]
    for k in ("D_loss", "G_loss", "g_adv", "g_cls", "reg", "mask_pen"):
        tol = max(1e-5 + 1e-5 * abs(l64[k]), 3 * abs(l32[k] - l64[k]))
        assert abs(out[k].item() - l64[k]) <= tol, (k, out[k].item(), l64[k], tol)

    def check(mine, truth, noise32, label, net_scale):
        # net_scale: the largest gradient entry of the whole net — a Linear bias in front of BatchNorm1d has a true gradient
        # of exactly 0, and what any fp32 implementation returns there is summation noise relative to the net's scale
        scale = float(truth.abs().max())
        tol = max(1e-4 * scale, 3 * float((noise32 - truth).abs().max()), 1e-6 * net_scale)
        err = float((mine.detach().cpu().double() - truth).abs().max())
        assert err <= tol, (label, err, tol, scale)
    g_scale, d_scale = max(float(v.abs().max()) for v in g64.values()), max(float(v.abs().max()) for v in d64.values())
    for n, p in G.named_parameters():
        check(p.grad, g64[n], g32[n], f"G grad {n}", g_scale)
    for n, p in D.named_parameters():
        check(p.grad, d64[n], d32[n], f"D grad {n}", d_scale)                  # D-step gradient + the G step's dead contribution, as in torch
    # running statistics of the BatchNorm1d layers (updated once per step)
    for k, v in G.state_dict().items():
        if "running" in k:
            _close(v, s64[k], 1e-4, 1e-6, k)
        if k.endswith("num_batches_tracked"):
            assert int(v) == 1


def test_skip_dead_d_wgrad_is_equivalent(pcg, hgold):
    """skip_dead_d_wgrad=True only drops work whose result the reference discards: both nets end in the same state."""
    H = pcg.house
    x, y, t, m, gumbel = HR.synthetic_batch(64, seed=3)
    states = []
    for skip in (False, True):
        G, D, C = _load_golden_nets(pcg, hgold)
        opt_g, opt_d = H.make_optimizers(G, D)
        norm = H.cat_norm_maps(G, H.CONFIG, torch.device(DEV))
        noise = G.pack_noise({f: _dev(v) for f, v in gumbel.items()})
        for _ in range(2):
            H.train_step(G, D, C, opt_g, opt_d, _dev(x), _dev(y), _dev(t), _dev(m), norm, gumbel=noise, skip_dead_d_wgrad=skip)
        states.append({**{f"G.{k}": v.clone() for k, v in G.state_dict().items()}, **{f"D.{k}": v.clone() for k, v in D.state_dict().items()}})
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k


@pytest.mark.parametrize("rows,hard", [(300, False), (7, False), (1000, True), (4096, False)])
def test_fused_generator_kernels_match_the_op_chain(pcg, hgold, rows, hard):
    """csrc/house_fused.hip (12 + 11 launches on the matrix cores: 64-row segment blocks, 16-row head blocks) against the per-op
    path on the same inputs: the three outputs, every parameter gradient, the BatchNorm buffers.  Same arithmetic up to summation
    order: 2e-5 of scale.  300 / 1000 rows end in ragged 64- and 16-row blocks, 7 rows is one partial block of either kind;
    hard: the straight-through one-hot samples (first maximum) must be the same one-hot rows."""
    H = pcg.house
    x, y, t, m, gumbel = HR.synthetic_batch(rows, seed=5)
    res = []
    for fused in (True, False):
        G, _, _ = _load_golden_nets(pcg, hgold)
        G.use_fused = fused
        noise = G.pack_noise({f: _dev(v) for f, v in gumbel.items()})
        cont, logits, samples = G.forward_packed(_dev(x), pcg.ops.onehot(_dev(t), 4), _dev(m), temperature=0.5, hard=hard, gumbel=noise)
        g = torch.Generator().manual_seed(1)
        dc, dl, ds = (torch.randn(v.shape, generator=g) for v in (cont, logits, samples))
        torch.autograd.backward([cont, logits, samples], [_dev(dc), _dev(dl), _dev(ds)])
        res.append((cont.detach(), logits.detach(), samples.detach(), {n: p.grad.clone() for n, p in G.named_parameters()},
                    {n: b.clone() for n, b in G.named_buffers()}))
    if hard:     # one-hot rows: equal wherever the two largest probabilities of a head are not within rounding of each other
        same = (res[0][2] == res[1][2]).all(dim=1).float().mean().item()
        assert same >= 0.995, same
        assert bool(((res[0][2] == 0) | (res[0][2] == 1)).all())
        res = [(r[0], r[1], r[1]) + r[3:] for r in res]            # (the samples were compared above)
    for a, b in zip(res[0][:3], res[1][:3]):
        _close(a, b, 2e-5, 2e-5 * float(b.abs().max()))
    scale = max(float(v.abs().max()) for v in res[1][3].values())
    for n in res[0][3]:
        _close(res[0][3][n], res[1][3][n], 1e-4, 2e-5 * float(res[1][3][n].abs().max()) + 2e-6 * scale, f"grad {n}")
    for n in res[0][4]:
        _close(res[0][4][n], res[1][4][n], 1e-5, 1e-6, f"buffer {n}")


@pytest.mark.parametrize("rows", [4096, 300])
def test_fused_residual_block_equals_the_op_chain_bitwise(pcg, hgold, rows):
    """pcg_house_residual_fwd / _bwd (one launch each) against the seven + nine single-op launches they replace, through autograd:
    residual_full, masked residual, x_cf, both penalties, and the gradients reaching the generator's outputs — bit for bit, at a
    size on either side of abs_mean's one-block / 256-block switch.  pcg_house_draws against the three separate draws."""
    H, ops = pcg.house, pcg.ops
    G, _, _ = _load_golden_nets(pcg, hgold)
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(rows)
    B, D, T, nc = rows, 17, G.total_cat, len(G.continuous_idx)
    x, mask = _dev(torch.rand(B, D, generator=g)), _dev((torch.rand(B, D, generator=g) > 0.4).float())
    cont = _dev(torch.randn(B, nc, generator=g) * 0.1).requires_grad_(True)
    samples = _dev(torch.softmax(torch.randn(B, T, generator=g), 1)).requires_grad_(True)
    gx_a, gx_b = _dev(torch.randn(B, D, generator=g) * 1e-3), _dev(torch.randn(B, D, generator=g) * 1e-3)
    norm = H.cat_norm_maps(G, H.CONFIG, dev)
    seg, cat_idx, cont_idx = G.index_tables(dev)
    lam_mask, w_reg = 1.0, 1.0 * D
    # the op chain, as train_step (reference order) composes it
    res = H.assemble_residual(G, cont, samples, x, norm)
    masked, x_cf = H._MaskMulFn.apply(res, mask, x)
    pen, am = H.abs_mean(res, mask, one_minus=True), H.abs_mean(masked)
    rest = H.weighted_sum([am, pen], [w_reg, lam_mask])
    torch.autograd.backward([rest, x_cf], [H._one(dev), ops.axpby(1.0, gx_a, 1.0, gx_b)])
    # fused
    with torch.no_grad():
        r2, m2, xc2, pen2, am2 = ops.house_residual_fwd(cont.detach(), samples.detach(), seg, norm, x, mask, G.col_src())
        dc2, ds2 = ops.house_residual_bwd(r2, m2, mask, gx_a, gx_b, lam_mask, w_reg, nc, cont_idx, seg, T, cat_idx, norm)
    for a, b, what in ((res, r2, "residual_full"), (masked, m2, "masked"), (x_cf, xc2, "x_cf"), (pen, pen2, "mask penalty"), (am, am2, "am"),
                       (cont.grad, dc2, "d_cont"), (samples.grad, ds2, "d_samples")):
        assert torch.equal(a.detach().reshape(-1), b.reshape(-1)), what
    y = _dev(torch.randint(0, 4, (B,), generator=g))
    imm = torch.tensor(H.CONFIG["immutable_idx"], dtype=torch.int32, device=dev)
    r1, r2 = ops.DeviceRNG(seed=11), ops.DeviceRNG(seed=11)
    t1 = r1.randint(0, 4, B, dev, exclude=y); k1 = r1.feature_mask(B, D, dev, imm); n1 = r1.gumbel((B, T), dev)
    oh = (torch.empty((B, 4), device=dev), torch.empty((B, 4), device=dev))
    t2, k2, n2 = H.draw_batch_randoms(r2, G, y, H.CONFIG, dev, onehots=oh)
    assert torch.equal(t1, t2) and torch.equal(k1, k2) and torch.equal(n1, n2) and r1.offset == r2.offset
    assert torch.equal(oh[0], ops.onehot(t1, 4)) and torch.equal(oh[1], ops.onehot(y, 4))


def test_fused_critic_kernels_match_the_op_chain(pcg, hgold):
    """csrc/house_critic_fused.hip (one forward + one backward launch, one thread per row; weight gradients through the grouped
    reduction, a 128x64 layer as four tiles) against the per-op path: output, input gradient, every parameter gradient and the
    power-iteration buffers."""
    H = pcg.house
    x, y, t, m, _ = HR.synthetic_batch(300, seed=6)
    res = []
    for fused in (True, False):
        _, D, _ = _load_golden_nets(pcg, hgold)
        D.use_fused = fused
        xi = _dev(x).requires_grad_(True)
        out = D(xi, pcg.ops.onehot(_dev(t), 4))
        g = torch.Generator().manual_seed(2)
        out.backward(_dev(torch.randn(out.shape, generator=g)))
        res.append((out.detach(), xi.grad.clone(), {n: p.grad.clone() for n, p in D.named_parameters()}, {n: b.clone() for n, b in D.named_buffers()}))
    _close(res[0][0], res[1][0], 2e-5, 2e-6)
    _close(res[0][1], res[1][1], 1e-4, 2e-5 * float(res[1][1].abs().max()))
    scale = max(float(v.abs().max()) for v in res[1][2].values())
    for n in res[0][2]:
        _close(res[0][2][n], res[1][2][n], 1e-4, 2e-5 * float(res[1][2][n].abs().max()) + 2e-6 * scale, f"grad {n}")
    for n in res[0][3]:
        _close(res[0][3][n], res[1][3][n], 1e-6, 1e-7, f"buffer {n}")


@pytest.mark.parametrize("rows", [300, 4096, 5])
def test_fused_classifier_kernels_match_the_op_chain(pcg, hgold, rows):
    """csrc/house_classifier_fused.hip (the frozen classifier's five layers as one MFMA launch forward, one backward) against the
    per-layer GEMM path and against float64 torch on the folded weights: logits and the input gradient of a cross-entropy on
    them.  Same arithmetic up to summation order: 2e-5 of scale; ragged last block, a batch smaller than a block."""
    H, ops = pcg.house, pcg.ops
    _, _, C = _load_golden_nets(pcg, hgold)
    g = torch.Generator().manual_seed(rows)
    with torch.no_grad():       # trained-looking BatchNorm statistics, so that the folding matters
        for m in C.net:
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(_dev(torch.randn(m.num_features, generator=g) * 0.1)); m.running_var.copy_(_dev(torch.rand(m.num_features, generator=g) + 0.5))
    x = _dev(torch.rand(rows, 17, generator=g)); t = _dev(torch.randint(0, 4, (rows,), generator=g))
    res = []
    for fused in (True, False):
        C.use_fused = fused
        with torch.no_grad():
            logits, acts = C._run_forward(x, keep=True)
            loss, dlog = ops.cross_entropy_fwd_bwd(logits.contiguous(), t, need_loss=True, need_grad=True, grad_scale=2.0)
            dx = C._run_backward(acts, dlog)
        res.append((logits.clone(), dx.clone(), [a.clone() for a in acts]))
    C.use_fused = True
    # float64 truth from the folded weights
    a = x.double().cpu()
    packed = [(w.double().cpu(), b.double().cpu()) for w, b in C._pack()]
    a.requires_grad_(True)
    h = a
    for i, (w, b) in enumerate(packed):
        h = h @ w.T + b
        if i + 1 < len(packed):
            h = F.leaky_relu(h, 0.1)
    (2.0 * F.cross_entropy(h, t.cpu())).backward()
    for k, (lg, dx, acts) in enumerate(res):
        _close(lg, h.detach(), 2e-5, 2e-5 * float(h.detach().abs().max()), f"logits fused={k == 0}")
        _close(dx, a.grad, 1e-4, 2e-5 * float(a.grad.abs().max()), f"dx fused={k == 0}")
    for u, v in zip(res[0][2], res[1][2]):
        _close(u, v, 2e-5, 2e-5 * float(v.abs().max()), "saved activation")


@pytest.mark.parametrize("overlap,batch", [(True, 128), ("critic", 128), (False, 128), ("inline", 128), (True, 4096), ("inline", 4096), ("inline", 1000), ("inline", 20000)])
def test_graphed_step_equals_eager(pcg, hgold, overlap, batch):
    """GraphedTrainStep (one HIP-graph replay per step) leaves the nets exactly where the eager step does, and constructing
    it (warm-up + capture) does not advance the training state.  overlap=True: the schedule with the classifier term on a parallel
    branch and the critic passes run directly with constant cotangents (house._train_step_branch); "critic": additionally the
    critic's real pass on a third stream into a second gradient buffer; False: the reference-order single-stream step; "inline"
    (the default): the schedule on one stream, with the rider launches (spectral-norm work inside the residual block's and the
    classifier's launches, the logged scalars inside the residual block's backward, the cross-entropy as the tail of the classifier's
    forward).  All are bit-identical to the eager autograd step — parameters, buffers, D_loss, G_loss and g_cls."""
    H = pcg.house
    # 4096: the multi-block forms of every reduction (the bench batch); 20000: past the one-block forms of the logged scalars / cross-entropy tail
    batches = [HR.synthetic_batch(batch, seed=s) for s in (1, 2, 3)]
    states = []
    for graphed in (False, True):
        G, D, C = _load_golden_nets(pcg, hgold)
        opt_g, opt_d = H.make_optimizers(G, D)
        norm = H.cat_norm_maps(G, H.CONFIG, torch.device(DEV))
        gs = H.GraphedTrainStep(G, D, C, opt_g, opt_d, norm, batch, overlap=overlap) if graphed else None
        losses = []
        for (x, y, t, m, gumbel) in batches:
            noise = G.pack_noise({f: _dev(v) for f, v in gumbel.items()})
            if graphed:
                gs.load(_dev(x), _dev(y), _dev(t), _dev(m), noise)
                out = gs.replay()
            else:
                out = H.train_step(G, D, C, opt_g, opt_d, _dev(x), _dev(y), _dev(t), _dev(m), norm, gumbel=noise)
            losses.append((out["D_loss"].item(), out["G_loss"].item(), out["g_cls"].item(), out["reg"].item()))
        states.append(({**{f"G.{k}": v.clone() for k, v in G.state_dict().items()}, **{f"D.{k}": v.clone() for k, v in D.state_dict().items()}},
                       losses))
    assert states[0][1] == states[1][1], (states[0][1], states[1][1])
    for k in states[0][0]:
        assert torch.equal(states[0][0][k], states[1][0][k]), k


def test_scheduled_step_equals_reference_order_without_the_fused_kernels(pcg, hgold):
    """train_step(branch=...) — no autograd graph, fused residual block, manual critic passes — must also hold when the nets fall
    back to their per-op paths (configurations the fused kernels are not built for): same state as the reference-order autograd
    step, bit for bit."""
    H = pcg.house
    states = []
    for sched in (False, True):
        G, D, C = _load_golden_nets(pcg, hgold)
        G.use_fused = D.use_fused = C.use_fused = False
        opt_g, opt_d = H.make_optimizers(G, D)
        norm = H.cat_norm_maps(G, H.CONFIG, torch.device(DEV))
        losses = []
        for seed in (4, 5):
            x, y, t, m, gumbel = HR.synthetic_batch(96, seed=seed)
            noise = G.pack_noise({f: _dev(v) for f, v in gumbel.items()})
            out = H.train_step(G, D, C, opt_g, opt_d, _dev(x), _dev(y), _dev(t), _dev(m), norm, gumbel=noise,
                               branch=torch.cuda.Stream() if sched else None)
            torch.cuda.synchronize()
            losses.append((out["D_loss"].item(), out["G_loss"].item(), out["g_cls"].item(), out["reg"].item()))
        states.append(({**{f"G.{k}": v.clone() for k, v in G.state_dict().items()}, **{f"D.{k}": v.clone() for k, v in D.state_dict().items()}}, losses))
    assert states[0][1] == states[1][1], (states[0][1], states[1][1])
    for k in states[0][0]:
        assert torch.equal(states[0][0][k], states[1][0][k]), k


def test_trained_checkpoints_eval_forward(pcg, golden_dir):
    """The checkpoints the reference ships, eval mode, hard Gumbel-softmax (eval_utils.py:76-77), reference-module outputs."""
    H = pcg.house
    gold = dict(np.load(os.path.join(golden_dir, "house_trained_eval.npz")))
    G = H.ResidualGenerator(17, 32, 4, H.CONFIG["continuous_idx"], H.CONFIG["categorical_info"], tau=0.5)
    C = H.NNClassifier(17, 4)
    G.load_state_dict(torch.load(os.path.join(golden_dir, "house_generator_trained.pt"), map_location="cpu", weights_only=True))
    C.load_state_dict(torch.load(os.path.join(golden_dir, "house_classifier_trained.pt"), map_location="cpu", weights_only=True))
    G, C = G.to(DEV).eval(), C.to(DEV).eval()
    x, t, m = _dev(torch.from_numpy(gold["x"])), _dev(torch.from_numpy(gold["target_y"])), _dev(torch.from_numpy(gold["mask"]))
    gumbel = {int(k[7:]): _dev(torch.from_numpy(v)) for k, v in gold.items() if k.startswith("gumbel.")}
    with torch.no_grad():
        cont, logits, samples = G(x, pcg.ops.onehot(t, 4), mask=m, temperature=0.5, hard=True, gumbel=gumbel)
        cl = C(x)
    _close(cont, torch.from_numpy(gold["cont"]), 1e-4, 2e-5)
    _close(cl, torch.from_numpy(gold["clf_logits"]), 1e-4, 1e-4)         # BatchNorm folded into the next Linear: fp32 re-association
    for f in gumbel:
        _close(logits[f], torch.from_numpy(gold[f"logits.{f}"]), 1e-4, 1e-4)
        assert torch.equal(samples[f].cpu(), torch.from_numpy(gold[f"samples.{f}"])), f    # one-hot: exact
    # classifier input gradient (what the G step back-propagates through) against autograd on the oracle module, float64
    _, _, oC = HR.build(seed=0)
    oC.load_state_dict(torch.load(os.path.join(golden_dir, "house_classifier_trained.pt"), map_location="cpu", weights_only=True))
    oC = oC.double().eval()
    xr = torch.from_numpy(gold["x"]).double().requires_grad_(True)
    F.cross_entropy(oC(xr), t.cpu()).backward()
    xg = x.clone().requires_grad_(True)
    H.CrossEntropyLoss()(C(xg), t).backward()
    _close(xg.grad, xr.grad, 1e-4, 1e-5 * float(xr.grad.abs().max()))


def test_device_draws(pcg):
    """trainer.py:248-255 + generator.py:90 on the device: targets differ from the source class, immutable columns are
    never modifiable, the rest is Bernoulli(1/2); Gumbel(0,1) noise has mean 0.5772 and variance pi^2/6; deterministic."""
    H, ops = pcg.house, pcg.ops
    G = H.ResidualGenerator(17, 32, 4, H.CONFIG["continuous_idx"], H.CONFIG["categorical_info"]).to(DEV)
    y = torch.randint(0, 4, (8192,), device=DEV)
    t, mask, noise = H.draw_batch_randoms(ops.DeviceRNG(5), G, y, H.CONFIG, torch.device(DEV))
    assert bool((t != y).all()) and int(t.min()) == 0 and int(t.max()) == 3
    cnt = torch.bincount(t, minlength=4).float() / t.numel()
    assert float((cnt - 0.25).abs().max()) < 0.02
    mc = mask.mean(0).cpu()
    imm = H.CONFIG["immutable_idx"]
    assert float(mask[:, imm].abs().sum()) == 0.0 and set(mask.unique().tolist()) == {0.0, 1.0}
    free = [i for i in range(17) if i not in imm]
    assert float((mc[free] - 0.5).abs().max()) < 0.03
    assert noise.shape == (8192, 70) and abs(float(noise.mean()) - 0.5772) < 0.01 and abs(float(noise.var()) - np.pi ** 2 / 6) < 0.03
    t2, mask2, noise2 = H.draw_batch_randoms(ops.DeviceRNG(5), G, y, H.CONFIG, torch.device(DEV))
    assert torch.equal(t, t2) and torch.equal(mask, mask2) and torch.equal(noise, noise2)
    # the full loop runs on device draws and moves the losses
    D, C = H.Discriminator(17, 32, 4).to(DEV), H.NNClassifier(17, 4).to(DEV).eval()
    cfg = dict(H.CONFIG, epochs=2)
    xs = torch.rand(4 * 128, 17)
    ys = torch.randint(0, 4, (4 * 128,))
    loader = [(xs[i * 128:(i + 1) * 128], ys[i * 128:(i + 1) * 128]) for i in range(4)]
    hist = H.train_countergan_loop(G, D, C, loader, cfg, torch.device(DEV), rng=ops.DeviceRNG(1), log_every=10 ** 9)
    assert len(hist) == 2 and all(np.isfinite(v) for e in hist for v in e)


# ---- the reference-shaped trainer: train_countergan(generator, config, X_train, y_train, clf_model) -------------------------------
class _Scaler:
    def __init__(self, lo, hi):
        self.data_min_, self.data_max_ = lo, hi


def _loop_setup(pcg, golden_dir, tmp_path, **over):
    H = pcg.house
    g = dict(np.load(os.path.join(golden_dir, "house_loop.npz")))
    cfg = dict(H.CONFIG)
    cfg["categorical_info"] = {f: {"n": len(g[f"raw_values.{f}"]), "raw_values": g[f"raw_values.{f}"].tolist()} for f in H.CONFIG["categorical_info"]}
    cfg.update({"scaler": _Scaler(g["scaler.data_min"], g["scaler.data_max"]), "epochs": int(g["meta.epochs"]), "batch_size": int(g["meta.bs"]),
                "seed": int(g["meta.seed"]), "cuda": DEV, "generator_path": str(tmp_path / "sub" / "generator_model.pt")})
    cfg.update(over)
    torch.manual_seed(0)                                     # make_golden.py: classifier, then generator, seed 0
    C = H.NNClassifier(17, 4)
    G = H.ResidualGenerator(17, 32, 4, cfg["continuous_idx"], cfg["categorical_info"], tau=0.5)
    for k, v in C.state_dict().items():
        np.testing.assert_array_equal(_digest(v.float()), g[f"init.C.{k}"], err_msg=f"C.{k}")
    G.load_state_dict({k[7:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("init.G.")})
    C.eval()
    for p in C.parameters():
        p.requires_grad = False
    return H, g, cfg, G, C


def _digest(t, nsamples=64):          # make_golden.py: tensor_digest
    a = np.asarray(t.detach().cpu().numpy(), dtype=np.float64).ravel()
    idx = np.linspace(0, a.size - 1, num=min(nsamples, a.size)).astype(np.int64)
    return np.concatenate([[a.sum(), np.abs(a).sum(), (a * a).sum()], a[idx]])


def test_train_countergan_is_the_reference_trainer(pcg, golden_dir, tmp_path, monkeypatch, capsys):
    """house.train_countergan(generator, config, X_train, y_train, clf_model) against a run of the reference's own function
    (tests/golden/house_loop.npz: 2 epochs x 3 batches of 64 out of 209 rows): seeding -> the critic built inside has the
    reference's initial state and the first epoch's row order is the DataLoader's; cat_norm_maps from config['scaler']; per iteration
    D_loss / G_loss and the four diagnostics (read from the device accumulators); epoch means; G_grad / D_grad (D's includes the
    generator step's critic weight gradients); the saved generator; the log lines' format.  The draws the reference made are supplied."""
    H, g, cfg, G, C = _loop_setup(pcg, golden_dir, tmp_path)
    E, S, bs = int(g["meta.epochs"]), int(g["meta.nbatches"]), int(g["meta.bs"])
    torch.manual_seed(cfg["seed"])
    D0 = H.Discriminator(17, 32, 4)
    for k, v in D0.state_dict().items():
        assert torch.equal(v, torch.from_numpy(g[f"init.D.{k}"])), f"critic init {k}"
    real_perm = H.epoch_permutation
    epoch_no = [0]

    def perm_hook(n):
        e = epoch_no[0]
        epoch_no[0] += 1
        want = torch.from_numpy(g["it.rows"][e].reshape(-1))
        if e == 0:               # same seed, same draws before the loop: the first epoch's order must BE the DataLoader's
            got = real_perm(n)[:S * bs]
            assert torch.equal(got, want)
        return want              # later epochs: the reference drew its CPU-device targets / masks from the same generator in between
    monkeypatch.setattr(H, "epoch_permutation", perm_hook)
    per_it = []

    def draws(epoch, batch_idx, y):
        assert torch.equal(y.cpu(), torch.from_numpy(g["data.y"][g["it.rows"][epoch, batch_idx]]))
        return (torch.from_numpy(g["it.target_y"][epoch, batch_idx]), torch.from_numpy(g["it.mask"][epoch, batch_idx]),
                torch.from_numpy(g["it.gumbel"][epoch, batch_idx]))
    # per-iteration values: wrap train_step to read the device scalars (test only; the trainer itself reads once per epoch)
    real_step = H.train_step

    def step_hook(*a, **k):
        out = real_step(*a, **k)
        per_it.append([out["D_loss"].item(), out["G_loss"].item()] + out["diag"].cpu().tolist())
        return out
    monkeypatch.setattr(H, "train_step", step_hook)
    hist = H.train_countergan(G, cfg, g["data.X"], g["data.y"], C, draws=draws)
    per_it = np.array(per_it).reshape(E, S, 6)
    names = ("d_loss", "g_loss", "pred_gain", "sparsity", "l2_reg", "class_flip_rate")
    # fp32, six Adam steps deep: the oracle restatement meets the same bounds on the CPU (tests/test_oracle_golden.py)
    tol = {"d_loss": 5e-5, "g_loss": 5e-4, "pred_gain": 2e-6, "sparsity": 4e-3, "l2_reg": 5e-4, "class_flip_rate": 0.035}
    for i, k in enumerate(names):
        np.testing.assert_allclose(per_it[:, :, i], g[f"it.{k}"], rtol=0, atol=tol[k], err_msg=k)
    for k_h, k_g in (("d_losses", "d_loss"), ("g_losses", "g_loss"), ("pred_gain", "pred_gain"), ("sparsity", "sparsity"),
                     ("l2_reg", "l2_reg"), ("class_flip_rate", "class_flip_rate")):
        np.testing.assert_allclose(hist[k_h], g[f"it.{k_g}"].mean(1), rtol=0, atol=tol[k_g], err_msg=f"epoch mean {k_h}")
        np.testing.assert_allclose(hist[k_h], per_it[:, :, names.index(k_g)].mean(1), rtol=1e-6, atol=1e-9)     # device accumulators
    np.testing.assert_allclose(hist["G_grad"], g["epoch.G_grad"], rtol=5e-3)
    np.testing.assert_allclose(hist["D_grad"], g["epoch.D_grad"], rtol=5e-3)
    saved = torch.load(cfg["generator_path"], map_location="cpu", weights_only=True)
    assert list(saved) == [k[8:] for k in g if k.startswith("saved.G.")]
    for k, v in saved.items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(g[f"saved.G.{k}"]) == E * S
            continue
        assert np.abs(v.numpy() - g[f"saved.G.{k}"]).max() <= 6 * 2.2 * cfg["lr_G"], k
        assert torch.equal(v, G.state_dict()[k].cpu())
    # the log: same lines, same fields, numbers to the printed precision
    skel = lambda s: re.sub(r"-?[0-9]+\.[0-9]+", "#", s)
    ours = [l for l in capsys.readouterr().out.splitlines() if l.startswith("[")]
    ref = [l for l in str(g["log"]).splitlines() if l.startswith("[")]
    assert [skel(l) for l in ours] == [skel(l) for l in ref]
    for lo, lr_ in zip(ours, ref):
        a, b = [float(v) for v in re.findall(r"-?[0-9]+\.[0-9]+", lo)], [float(v) for v in re.findall(r"-?[0-9]+\.[0-9]+", lr_)]
        np.testing.assert_allclose(a, b, rtol=6e-3, atol=2e-3, err_msg=lo)


def test_train_countergan_graph_replays_equal_eager_iterations(pcg, golden_dir, tmp_path):
    """The same trainer with its own device draws, (a) every iteration one HIP-graph replay — batch taken, targets / mask / noise
    drawn, step, diagnostics, accumulators, all inside the captured launches — and (b) eager iterations on index_select batches
    with draw_batch_randoms: same Philox counters, same rows, same kernels => the histories and the trained generator are
    BIT-identical, and so is a second graph run (determinism)."""
    res = []
    for graph in (True, False, True):
        H, g, cfg, G, C = _loop_setup(pcg, golden_dir, tmp_path, epochs=3, batch_size=32)
        hist = H.train_countergan(G, cfg, g["data.X"], g["data.y"], C, graph=graph, verbose=False, save=False)
        D = hist.pop("discriminator")
        res.append((hist, G.flat_params.clone(), D.flat_params.clone()))
    for other in res[1:]:
        assert other[0] == res[0][0]
        assert torch.equal(other[1], res[0][1]) and torch.equal(other[2], res[0][2])
    assert len(res[0][0]["d_losses"]) == 3 and all(np.isfinite(v) for k in res[0][0] for v in res[0][0][k])


def test_house_batch_draws_take_the_batch(pcg):
    """pcg_house_batch_draws_counter: rows perm[cursor : cursor + B) of the resident training set land in the static buffers
    (bit-exact copies), the draws are those of pcg_house_draws_counter on the same y, the cursor and the Philox offset advance."""
    H, ops = pcg.house, pcg.ops
    dev = torch.device(DEV)
    G = H.ResidualGenerator(17, 32, 4, H.CONFIG["continuous_idx"], H.CONFIG["categorical_info"]).to(DEV)
    N, B, T = 1000, 96, G.total_cat
    g = torch.Generator().manual_seed(2)
    X, Y = torch.rand(N, 17, generator=g).to(DEV), torch.randint(0, 4, (N,), generator=g).to(DEV)
    perm = torch.randperm(N, generator=g).to(DEV)
    imm = torch.tensor(H.CONFIG["immutable_idx"], dtype=torch.int32, device=DEV)
    rng_a, rng_b = ops.DeviceRNG(9), ops.DeviceRNG(9)
    ctr = rng_a.device_counter(dev, cursor=True)
    bufs = (torch.empty(B, 17, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV),
            torch.empty(B, 17, device=DEV), torch.empty(B, T, device=DEV))
    oh = (torch.empty(B, 4, device=DEV), torch.empty(B, 4, device=DEV))
    src = torch.empty(B, dtype=torch.int64, device=DEV)
    for it in range(3):
        rng_a.house_batch_draws(X, Y, perm, 4, T, imm, bufs, oh, ctr, src_out=src)
        rows = perm[it * B:(it + 1) * B]
        assert torch.equal(src, rows) and torch.equal(bufs[0], X[rows]) and torch.equal(bufs[1], Y[rows])
        t, m, nz = H.draw_batch_randoms(rng_b, G, Y[rows].contiguous(), H.CONFIG, dev)
        assert torch.equal(bufs[2], t) and torch.equal(bufs[3], m) and torch.equal(bufs[4], nz)
        assert torch.equal(oh[0], F.one_hot(t, 4).float()) and torch.equal(oh[1], F.one_hot(Y[rows], 4).float())
        assert ctr.tolist() == [rng_b.offset, 0, (it + 1) * B, 0]


def test_house_diag_against_torch(pcg):
    """pcg_house_diag against the reference's expressions (trainer.py:318-343) in float64."""
    ops = pcg.ops
    g = torch.Generator().manual_seed(4)
    B, N = 777, 2000
    lc, lo_all = torch.randn(B, 4, generator=g) * 3, torch.randn(N, 4, generator=g) * 3
    src = torch.randint(0, N, (B,), generator=g)
    t = torch.randint(0, 4, (B,), generator=g)
    m = torch.randn(B, 17, generator=g) * (torch.rand(B, 17, generator=g) < 0.4) * 0.01
    acc = torch.zeros(8, dtype=torch.float64, device=DEV)
    out = ops.house_diag(_dev(lc), _dev(lo_all), _dev(t), _dev(m), src_rows=_dev(src), acc=acc)
    out2 = ops.house_diag(_dev(lc), _dev(lo_all[src]), _dev(t), _dev(m), acc=acc)
    assert torch.equal(out, out2)
    ar = torch.arange(B)
    pc, po = F.softmax(lc.double(), 1), F.softmax(lo_all[src].double(), 1)
    want = [(pc[ar, t] - po[ar, t]).mean().item(), 1.0 - (m.abs() > 1e-3).double().mean().item(),
            m.double().norm(dim=1).mean().item(), (lc.argmax(1) == t).double().mean().item()]
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(acc.cpu().numpy()[2:6], 2 * out.cpu().double().numpy(), rtol=1e-12)


# ---- evaluation path ------------------------------------------------------------------------------


def _eval_setup(pcg, golden_dir):
    H = pcg.house
    gold = dict(np.load(os.path.join(golden_dir, "house_eval.npz")))
    cfg = dict(H.CONFIG)
    cfg["categorical_info"] = {f: {"n": len(gold[f"raw_values.{f}"]), "raw_values": gold[f"raw_values.{f}"].tolist()} for f in H.CONFIG["categorical_info"]}
    cfg["scaler"] = _Scaler(gold["scaler.data_min"], gold["scaler.data_max"])
    G = H.ResidualGenerator(17, 32, 4, cfg["continuous_idx"], cfg["categorical_info"], tau=0.5)
    C = H.NNClassifier(17, 4)
    G.load_state_dict(torch.load(os.path.join(golden_dir, "house_generator_trained.pt"), map_location="cpu", weights_only=True))
    C.load_state_dict(torch.load(os.path.join(golden_dir, "house_classifier_trained.pt"), map_location="cpu", weights_only=True))
    return gold, cfg, G.to(DEV).eval(), C.to(DEV).eval()


def test_eval_metrics_exact_case(pcg, golden_dir):
    """compute_metrics_per_target on the first rows of the real (scaled) test split, one batch per target class, with the hard
    Gumbel-softmax draws the reference made: its metrics and its counterfactual rows."""
    H = pcg.house
    gold, cfg, G, C = _eval_setup(pcg, golden_dir)
    n = int(gold["meta.exact_rows"])
    noise = [G.pack_noise({f: _dev(torch.from_numpy(gold[f"exact.gumbel.{t}.{f}"])) for f in G.cat_idx}) for t in range(4)]
    res, orig, cfs = H.compute_metrics_per_target(G, C, gold["X_test"][:n], gold["y_test"][:n], dict(cfg, batch_size=n),
                                                  gumbel_per_call=noise, max_vis=10 ** 9)
    got = np.array([[r["class_flip"], r["prediction_gain"], r["avg_actionability"]] for r in res])
    # flips are counts / B (a sample whose top-2 logits are within fp32 noise could change it by 1/B: none here); gains and
    # actionability are fp32 means
    np.testing.assert_allclose(got, gold["exact.metrics"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(cfs, gold["exact.x_cf"], rtol=1e-4, atol=2e-5)


def test_eval_full_run_agrees_with_shipped_metrics(pcg, golden_dir):
    """End-to-end anchor: the shipped generator + classifier on the reference's own test split, device-drawn Gumbel noise,
    against results/countergan_metrics.csv as shipped (different random draws: agreement is statistical, 4323 rows)."""
    H = pcg.house
    gold, cfg, G, C = _eval_setup(pcg, golden_dir)
    res, _, _ = H.compute_metrics_per_target(G, C, gold["X_test"], gold["y_test"], dict(cfg, batch_size=int(gold["meta.batch_size"])),
                                             rng=pcg.ops.DeviceRNG(7))
    got = np.array([[r["class_flip"], r["prediction_gain"], r["avg_actionability"]] for r in res])
    assert np.abs(got - gold["shipped.metrics"]).max() < 0.01, got
    assert np.abs(got - gold["full.metrics"]).max() < 0.01, got


# ---- classifier pre-training -------------------------------------------------------------------
def test_classifier_pretraining_steps_match_reference(pcg, golden_dir):
    """Three iterations of the reference's train_classifier (trainer.py:78-87: weighted CrossEntropyLoss, AdamW) with the
    batches its shuffling loader produced and the Dropout draws it made.  First-layer units whose pre-activations keep one
    sign over a batch have an exactly-zero bias gradient; any fp32 run holds noise there, Adam turns it into +-lr moves, and
    the next batch amplifies them — so trajectories of different fp32 implementations separate by O(lr) per step.  Hence:
    (a) per step, starting from the float64 oracle's current parameters, loss and every gradient against that oracle
    (tolerance: 1e-4 of the tensor's scale or 3x the float32 oracle's own distance);  (b) the free-running 3-step result
    against the reference's final state, every entry within the total possible Adam move."""
    from test_oracle_golden import _house_clf_gold
    H = pcg.house
    gold, cw = _house_clf_gold(golden_dir)
    init = {k[5:]: torch.from_numpy(v.copy()) for k, v in gold.items() if k.startswith("init.")}
    X, y = torch.from_numpy(gold["X"]), torch.from_numpy(gold["y"])
    o64, o32 = HR.NNClassifier(17, 4), HR.NNClassifier(17, 4)
    o64.load_state_dict(init); o32.load_state_dict(init)
    o64 = o64.double()
    opt64 = torch.optim.AdamW(o64.parameters(), lr=1e-3, weight_decay=1e-4)
    mine = H.NNClassifier(17, 4)
    mine.load_state_dict(init)
    mine = mine.to(DEV).train()
    crit = H.WeightedCrossEntropyLoss(_dev(cw))
    for i in range(int(gold["meta.steps"])):
        r = torch.from_numpy(gold[f"step{i}.rows"])
        masks = [torch.from_numpy(gold[f"step{i}.mask{j}"]) for j in range(3)]
        sd = {k: v.float() if v.is_floating_point() else v.clone() for k, v in o64.state_dict().items()}
        o32.load_state_dict(sd); mine.load_state_dict(sd)                       # teacher forcing: same starting point
        o32.train(); o32.zero_grad()
        l32 = F.cross_entropy(HR.classifier_forward_train(o32, X[r], masks), y[r], weight=cw)
        l32.backward()
        l64 = HR.classifier_train_step(o64, opt64, cw.double(), X[r].double(), y[r], masks)   # advances the teacher
        mine.dropout_masks = [_dev(t) for t in masks]
        mine.zero_grad()
        lm = crit(mine(_dev(X[r])), _dev(y[r]))
        lm.backward()
        assert abs(lm.item() - l64) <= max(2e-5, 3 * abs(l32.item() - l64)), (i, lm.item(), l64)
        for (n, p), (_, q32), (_, q64) in zip(mine.named_parameters(), o32.named_parameters(), o64.named_parameters()):
            truth = q64.grad
            tol = max(1e-4 * float(truth.abs().max()), 3 * float((q32.grad.double() - truth).abs().max()), 1e-8)
            err = float((p.grad.cpu().double() - truth).abs().max())
            assert err <= tol, (f"step {i} grad {n}", err, tol)
    # (b) free run from the initial state
    mine.load_state_dict(init)
    for b_ in mine.modules():
        if isinstance(b_, torch.nn.BatchNorm1d):
            b_.num_batches_tracked.zero_()
    opt = pcg.optim.AdamW(mine.parameters(), lr=1e-3, weight_decay=1e-4)
    for i in range(int(gold["meta.steps"])):
        r = torch.from_numpy(gold[f"step{i}.rows"])
        mine.dropout_masks = [_dev(torch.from_numpy(gold[f"step{i}.mask{j}"])) for j in range(3)]
        opt.zero_grad()
        crit(mine(_dev(X[r])), _dev(y[r])).backward()
        opt.step()
    for k, v in mine.state_dict().items():
        ref = gold[f"final.{k}"]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref)
        elif "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=2e-3, atol=2e-3, err_msg=k)
        else:
            assert float(np.abs(v.cpu().numpy() - ref).max()) <= 2.2 * 1e-3 * 3 + 1e-5, k


def test_train_classifier_loop_learns(pcg):
    """house.train_classifier (trainer.py:18-180 without plots) end to end on a separable synthetic problem, device-drawn
    Dropout masks: validation accuracy well above chance, BatchNorm folded eval path consistent with training-mode weights."""
    H = pcg.house
    rs = np.random.RandomState(0)
    y = rs.randint(0, 4, size=1200)
    centers = rs.random_sample((4, 17))
    X = np.clip(centers[y] + 0.05 * rs.standard_normal((1200, 17)), 0, 1)
    cfg = dict(H.CONFIG, clf_epochs=6, batch_size=128, seed=1)
    model = H.train_classifier(X[:1000], X[1000:], y[:1000], y[1000:], None, cfg, device=DEV, verbose=False)
    assert model.history[-1][3] > 0.9, model.history
    model.eval()
    with torch.no_grad():
        acc = pcg.ops.cf_metrics(model(_dev(torch.tensor(X[1000:], dtype=torch.float32))).contiguous(), _dev(torch.tensor(y[1000:])),
                                 other=_dev(torch.tensor(y[1000:])))[0].item()
    assert acc > 0.9


def test_graphed_step_with_the_draws_inside_equals_eager_draws_and_steps(pcg, hgold):
    """GraphedTrainStep(rng=...): the per-iteration draws (target class, feature mask, Gumbel noise, one-hot rows) are the first launch
    of the replayed graph, their Philox offsets read from a device counter the launch advances itself.  Four replays against four
    eager iterations that draw with the host-side offsets (draw_batch_randoms) and then step: same draws, same losses, same parameters,
    and the host mirror of the counter in step."""
    H, ops = pcg.house, pcg.ops
    dev = torch.device(DEV)
    B = 192
    data = [HR.synthetic_batch(B, seed=s)[:2] for s in (11, 12, 13, 14)]
    states = []
    for graphed in (False, True):
        G, D, C = _load_golden_nets(pcg, hgold)
        opt_g, opt_d = H.make_optimizers(G, D)
        norm = H.cat_norm_maps(G, H.CONFIG, dev)
        rng = ops.DeviceRNG(21)
        rng.rand((7,), dev)                                   # the stream does not start at offset 0
        gs = H.GraphedTrainStep(G, D, C, opt_g, opt_d, norm, B, rng=rng) if graphed else None
        logs = []
        for x, y in data:
            if graphed:
                gs.load_batch(_dev(x), _dev(y))
                out = gs.replay()
                t, m, noise = gs.target_y, gs.mask, gs.noise
            else:
                t, m, noise = H.draw_batch_randoms(rng, G, _dev(y), H.CONFIG, dev)
                out = H.train_step(G, D, C, opt_g, opt_d, _dev(x), _dev(y), t, m, norm, gumbel=noise)
            logs.append((out["D_loss"].item(), out["G_loss"].item(), out["g_cls"].item(), t.clone(), m.clone(), noise.clone()))
        states.append(({**{f"G.{k}": v.clone() for k, v in G.state_dict().items()}, **{f"D.{k}": v.clone() for k, v in D.state_dict().items()}},
                       logs, rng.offset))
    assert states[0][2] == states[1][2]
    for a, b in zip(states[0][1], states[1][1]):
        assert a[:3] == b[:3], (a[:3], b[:3])
        for u, v in zip(a[3:], b[3:]):
            assert torch.equal(u, v)
    for k in states[0][0]:
        assert torch.equal(states[0][0][k], states[1][0][k]), k

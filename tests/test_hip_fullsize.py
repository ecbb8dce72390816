"""GPU, BASELINE.json sizes of the secondary configurations: size-independent properties — two runs from identical state and inputs
are bit-identical (fixed summation orders: what data-parallel replicas and checkpoint/resume rely on), every loss is finite,
parameters move.  The ORACLE comparisons at these sizes live next to the small-size ones (r04; they cost seconds, not minutes):
test_hip_countergan.py::test_step_vs_oracle_float64[1024], test_hip_house.py::test_step_vs_oracle_float64[4096],
test_hip_wgan.py::test_critic_and_generator_steps_vs_oracle_float64[1024-32-True] (full width; the batch-256 iteration below is
the bench shape), test_hip_groups.py::test_paired_d_step_equals_two_passes[None-512-True] and test_hip_benchshape.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _state(*nets):
    return {f"{i}.{k}": v.clone() for i, n in enumerate(nets) for k, v in n.state_dict().items()}


def _same(a, b):
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_countergan_mnist_batch_1024_step():
    import pcgan_amd  # noqa: F401
    from pcgan_amd import countergan as K
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(1024, 1, 28, 28, generator=g) * 2 - 1).to(DEV)
    y, t = torch.randint(0, 10, (1024,), generator=g).to(DEV), torch.randint(0, 10, (1024,), generator=g).to(DEV)
    mask = pcgan_amd.ops.DeviceRNG(1).patch_mask(1024, 28, 28, 7, 10, DEV)
    runs = []
    for _ in range(2):
        torch.manual_seed(1)
        G, Dn, C = K.ResidualGenerator().to(DEV), K.Discriminator().to(DEV), K.CNNClassifier().to(DEV).eval()
        before = _state(G)
        opt_g, opt_d, bce, ce = K.make_optimizers(G, Dn)
        out = K.train_step(G, Dn, C, opt_g, opt_d, bce, ce, x, y, t, mask)
        runs.append(([out[k].item() for k in ("d_loss", "g_loss", "g_adv", "g_cls", "reg_l1", "mask_pen")], _state(G, Dn)))
        assert all(np.isfinite(v) for v in runs[-1][0])
        assert any(not torch.equal(before[k], runs[-1][1][k]) for k in before)
    assert runs[0][0] == runs[1][0]
    _same(runs[0][1], runs[1][1])


def test_wgan_gp_width_1024_batch_256_iteration():
    import pcgan_amd  # noqa: F401
    from pcgan_amd import wgan as W, ops
    hp = W.Hyperparameter(batchsize=256)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(256, 1, 28, 28, generator=g) * 2 - 1).to(DEV)
    lab = torch.eye(10)[torch.randint(0, 10, (256,), generator=g)].to(DEV)
    z, alpha = torch.randn(256, 32, generator=g).to(DEV), torch.rand(256, 1, generator=g).to(DEV)
    runs = []
    for _ in range(2):
        critic, generator = W.build(DEV, hp, seed=1)
        c_opt, g_opt = W.make_optimizers(critic, generator)
        o1 = W.critic_step(critic, generator, c_opt, hp, x, lab, z, alpha)
        o2 = W.generator_step(critic, generator, g_opt, lab, z)
        runs.append(([o1["critic_loss"].item(), o1["gradient_penalty"].item(), o2["generator_loss"].item()], _state(critic, generator)))
        assert all(np.isfinite(v) for v in runs[-1][0])
    assert runs[0][0] == runs[1][0]
    _same(runs[0][1], runs[1][1])


def test_house_sales_batch_4096_step_eager_and_graphed():
    import pcgan_amd  # noqa: F401
    from pcgan_amd import house as H, ops
    B = 4096
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 17, generator=g).to(DEV)
    y = torch.randint(0, 4, (B,), generator=g).to(DEV)
    runs = []
    for graphed in (False, False, True):
        G, Dn, C = H.build(DEV, seed=0)
        opt_g, opt_d = H.make_optimizers(G, Dn)
        norm = H.cat_norm_maps(G, H.CONFIG, DEV)
        t, mask, noise = H.draw_batch_randoms(ops.DeviceRNG(4), G, y, H.CONFIG, DEV)
        if graphed:
            gs = H.GraphedTrainStep(G, Dn, C, opt_g, opt_d, norm, B)
            gs.load(x, y, t, mask, noise)
            out = gs.replay()
        else:
            out = H.train_step(G, Dn, C, opt_g, opt_d, x, y, t, mask, norm, gumbel=noise)
        runs.append(([out[k].item() for k in ("D_loss", "G_loss", "g_adv", "g_cls", "reg", "mask_pen")], _state(G, Dn)))
        assert all(np.isfinite(v) for v in runs[-1][0])
    assert runs[0][0] == runs[1][0] == runs[2][0]
    _same(runs[0][1], runs[1][1])
    _same(runs[0][1], runs[2][1])

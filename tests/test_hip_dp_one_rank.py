"""GPU, first-contact hardening of the data-parallel paths that have never met a second GPU (r04): the CounteRGAN and WGAN-GP steps
as graph-SEGMENT programs (cut at the gradient exchanges) with the exchanges issued through the library's own RCCL communicator
(pcg_dp_*, csrc/dp_rccl.hip) on a ONE-rank group — `always_exchange=True` makes every all-reduce, side-stream hand-over and event
wait really happen.  Eager data-parallel step == replayed segment program, bit for bit; the bucket the communicator averages is the
one Adam reads.  (DCGAN's counterpart: tests/test_hip_dcgan.py::test_graph_replay_is_bit_identical_to_eager[native].)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def dp_group():
    import torch.distributed as dist
    import pcgan_amd
    from pcgan_amd import parallel
    pcgan_amd.load()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29557")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        dp = parallel.GradSync(always_exchange=True)
        assert dp.native and dp.rccl_ranks() == 1
        yield dp
        dp.wait_all()
        torch.cuda.synchronize()
    finally:
        parallel.shutdown()
        dist.destroy_process_group()


def _state(*nets):
    return {f"{i}.{k}": v.clone() for i, n in enumerate(nets) for k, v in n.state_dict().items()}


def test_countergan_segment_program_equals_eager_dp_step(dp_group):
    import pcgan_amd
    from pcgan_amd import countergan as K, ops
    from pcgan_amd.nn import GraphedStep
    dp = dp_group
    dev = torch.device(DEV)
    B = 64
    rng = ops.DeviceRNG(seed=21)
    cfg = K.Config
    batches = []
    for _ in range(3):
        x = rng.rand((B, 1, 28, 28), dev).mul_(2.0).sub_(1.0)
        batches.append((x, rng.randint(0, cfg.num_classes, B, dev), rng.randint(0, cfg.num_classes, B, dev),
                        rng.patch_mask(B, 28, 28, cfg.patch_size, cfg.num_modifiable_patches, dev)))

    def fresh():
        torch.manual_seed(4)
        G, D, C = K.ResidualGenerator().to(dev), K.Discriminator().to(dev), K.CNNClassifier().to(dev)
        C.eval()
        for p in C.parameters():
            p.requires_grad = False
        return (G, D, C) + tuple(K.make_optimizers(G, D))

    G, D, C, opt_g, opt_d, bce, ce = fresh()
    for b in batches:
        o = K.train_step(G, D, C, opt_g, opt_d, bce, ce, *b, dp=dp)
    dp.wait_all()
    want = ([o[k].item() for k in ("d_loss", "g_loss")], _state(G, D))

    G, D, C, opt_g, opt_d, bce, ce = fresh()
    x, y, t, m = (v.clone() for v in batches[0])
    gs = GraphedStep(lambda d: K.train_step(G, D, C, opt_g, opt_d, bce, ce, x, y, t, m, dp=d), {"x": x, "y": y, "t": t, "m": m},
                     [G, D], [opt_g, opt_d], dp=dp)
    assert len(gs.program) >= 3                       # cut at wait(G), sync_now(D), sync_then(G)
    for i, b in enumerate(batches):
        if i == 1:                                    # a mixed eager step in between, as the benches' event-sampled steps are
            o = K.train_step(G, D, C, opt_g, opt_d, bce, ce, *b, dp=dp)
        else:
            gs.load(x=b[0], y=b[1], t=b[2], m=b[3])
            o = gs.replay()
    dp.wait_all()
    got = ([o[k].item() for k in ("d_loss", "g_loss")], _state(G, D))
    assert all(np.isfinite(v) for v in got[0]) and got[0] == want[0]
    for k in want[1]:
        assert torch.equal(got[1][k], want[1][k]), k


def test_wgan_segment_programs_equal_eager_dp_updates(dp_group):
    import pcgan_amd
    from pcgan_amd import wgan as W, ops
    dp = dp_group
    dev = torch.device(DEV)
    hp = W.Hyperparameter(critic_size=32, generator_size=32, critic_hidden_size=32, batchsize=16, n_critic=2)
    B = hp.batchsize

    def fresh():
        critic, generator = W.build(dev, hp, seed=5)
        return (critic, generator) + W.make_optimizers(critic, generator)

    ce, ge, coe, goe = fresh()
    cg, gg, cog, gog = fresh()
    gs = W.GraphedSteps(cg, gg, cog, gog, hp, B, dev, dp=dp)
    rng = ops.DeviceRNG(13)
    for it in range(3):
        x = rng.rand((B, 1, 28, 28), dev).mul_(2.0).sub_(1.0)
        lab = ops.onehot(rng.randint(0, 10, B, dev), 10)
        noise, alpha = rng.randn((B, hp.latent_size), dev), rng.rand((B, 1), dev)
        oe = W.critic_step(ce, ge, coe, hp, x, lab, noise, alpha, dp=dp)
        dp.wait_all()
        og = gs.critic_step(x, lab, noise, alpha)
        dp.wait_all()
        for k in ("critic_loss", "gradient_penalty"):
            assert torch.equal(oe[k], og[k]), (it, k)
        fake, noise = ops.onehot(rng.randint(0, 10, B, dev), 10), rng.randn((B, hp.latent_size), dev)
        le = W.generator_step(ce, ge, goe, fake, noise, dp=dp)["generator_loss"]
        dp.wait_all()
        lg = gs.generator_step(fake, noise)["generator_loss"]
        dp.wait_all()
        assert torch.equal(le, lg), it
        for m_e, m_g in ((ce, cg), (ge, gg)):
            for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
                assert torch.equal(a, b), (it, k)


def test_conv_scratch_is_never_allocated_under_capture():
    """ops._conv_scratch hands the stream-K arrival counters to the library for the life of the process: they must come from memory
    that outlives any graph pool.  The pool is filled when the library loads (per visible device); a capture that meets an empty pool
    raises instead of allocating inside the graph's private pool."""
    import pcgan_amd
    from pcgan_amd import ops
    pcgan_amd.load()
    assert len(ops._sk_pool.get(0, [])) == ops._SK_SPARES, "the spare scratch of device 0 is set aside at load time"
    s = torch.cuda.Stream()
    before = torch.cuda.memory_allocated()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ops._conv_scratch()                       # a fresh stream's first conv call inside a capture: served from the spares
            t = ops.fill(torch.empty(16, device=DEV), 1.0)      # (a capture must not be empty)
    assert (0, s.cuda_stream) in ops._sk_streams and ops._sk_streams[(0, s.cuda_stream)] is not None
    assert len(ops._sk_pool[0]) == ops._SK_SPARES - 1
    ops._sk_pool[0].clear()                           # no spare left: the next new stream under capture must refuse, not allocate
    s2 = torch.cuda.Stream()
    try:
        with torch.cuda.stream(s2):
            g2 = torch.cuda.CUDAGraph()
            with pytest.raises(pcgan_amd.PcgError, match="spare stream-K scratch"):
                with torch.cuda.graph(g2, stream=s2):
                    ops.fill(torch.empty(16, device=DEV), 1.0)
                    ops._conv_scratch()
    finally:
        ops.prepare_conv_scratch(0)
    assert len(ops._sk_pool[0]) == ops._SK_SPARES

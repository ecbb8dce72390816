"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (one flat bucket per net, averaged across ranks,
replicas stay identical).  The GPU path uses the same GradSync object with backend nccl (= RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeNet:
    def __init__(self, n, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.flat_grads = torch.randn(n, generator=g)
        self.flat_params = torch.full((n,), float(rank))
        self._bufs = [torch.full((3,), float(rank))]

    def buffers(self):
        return self._bufs


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcgan_amd.parallel import GradSync, broadcast_parameters
    netD, netG = _FakeNet(1001, rank), _FakeNet(77, rank)
    expD = sum(_FakeNet(1001, r).flat_grads for r in range(world)) / world
    expG = sum(_FakeNet(77, r).flat_grads for r in range(world)) / world
    dp = GradSync()
    dp.sync_now(netD)
    stepped = []
    dp.sync_then(netG, lambda: stepped.append(netG.flat_grads.clone()))
    dp.wait(netG)
    dp.wait_all()
    broadcast_parameters(netD, src=0)
    ok = (torch.allclose(netD.flat_grads, expD, atol=1e-6) and torch.allclose(netG.flat_grads, expG, atol=1e-6)
          and len(stepped) == 1 and torch.equal(stepped[0], netG.flat_grads)
          and float(netD.flat_params.abs().sum()) == 0.0 and float(netD.buffers()[0].abs().sum()) == 0.0)
    # replicas are bit-identical after the exchange
    gathered = [torch.empty_like(netD.flat_grads) for _ in range(world)]
    dist.all_gather(gathered, netD.flat_grads)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)], res

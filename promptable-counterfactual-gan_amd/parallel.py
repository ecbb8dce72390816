"""Data-parallel replicas: one process per GPU, one flat gradient bucket per net, one RCCL all-reduce per bucket.

The reference is single-process; this is the only exchange step the data-parallel path adds: after
the backward pass(es) of a net, its flat fp32 gradient buffer (DCGAN: D 11.05 MB, G 14.30 MB) is averaged across
ranks (torch.distributed, backend "nccl" = RCCL over xGMI).  D's bucket is needed immediately (Adam(D) precedes the
G step's D forward), so it is reduced in stream order; G's bucket and Adam(G) run on a side HIP stream and overlap
with the next iteration's D(real) forward/backward, which does not touch G.  BatchNorm statistics stay per replica
(the PyTorch-DDP convention).
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, group=None, always_exchange=False):
        """always_exchange: issue the collectives even in a one-rank group (exercises the RCCL path on a single GPU)."""
        self.always_exchange = always_exchange
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self._side = None
        self._pending = {}

    def _allreduce_mean(self, flat):
        if self.world == 1 and not self.always_exchange:
            return
        if self.backend == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:  # gloo (CPU rehearsal): no AVG
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)

    def sync_now(self, net):
        """Average net.flat_grads across ranks in stream order (the next kernel on the current stream sees the result)."""
        self._allreduce_mean(net.flat_grads)

    def sync_then(self, net, fn):
        """Average net.flat_grads and then run fn() (the optimizer step) — on a side stream when on the GPU, so both
        overlap with whatever the main stream does next.  Call wait(net) before the next use of the net's parameters."""
        flat = net.flat_grads
        if not flat.is_cuda:
            self._allreduce_mean(flat)
            fn()
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=flat.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(flat.device))
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            self._allreduce_mean(flat)
            fn()
            done = torch.cuda.Event()
            done.record(self._side)
        self._pending[id(net)] = done

    def wait(self, net):
        done = self._pending.pop(id(net), None)
        if done is not None:
            torch.cuda.current_stream().wait_event(done)

    def wait_all(self):
        for done in self._pending.values():
            torch.cuda.current_stream().wait_event(done)
        self._pending.clear()


def broadcast_parameters(net, src=0, group=None):
    """Make every replica start from rank `src`'s weights and BatchNorm buffers."""
    dist.broadcast(net.flat_params, src=src, group=group)
    for b in net.buffers():
        dist.broadcast(b, src=src, group=group)

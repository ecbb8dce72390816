"""Data-parallel replicas: one process per GPU, one flat gradient bucket per net, one RCCL all-reduce per bucket.

The reference is single-process; this is the only exchange step the data-parallel path adds: after
the backward pass(es) of a net, its flat fp32 gradient buffer (DCGAN: D 11.05 MB, G 14.30 MB) is averaged across
ranks.  D's bucket is needed immediately (Adam(D) precedes the G step's D forward), so it is reduced in stream order;
G's bucket and Adam(G) run on a side HIP stream and overlap with the next iteration's D(real) forward/backward, which
does not touch G.

On the GPU the exchange goes through the C ABI (`pcg_dp_*`, csrc/dp_rccl.hip): libpcgan_hip.so owns the RCCL
communicator, the side stream and the ordering events; `torch.distributed` is only the control plane (rendezvous: its
key-value store carries the 128-byte RCCL id; barriers; the bench's digest all-gather).  On the CPU (the gloo rehearsal
of the exchange logic in tests/) the same object runs the collectives through `torch.distributed`.

BatchNorm under data parallelism: per-replica statistics by default (the PyTorch-DDP convention);
`GradSync(sync_bn=True)` makes them exact — the per-channel sums (sum x, sum x^2; backward: sum dy, sum dy*xhat) are
all-reduced before the statistics are finalised, so N ranks x B/N images compute what one process computes on B
(the reference's BatchNorm spans the whole batch: mnist_dcgan.py:77-87, SURVEY.md §8e option ii).
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check


_native_generation = 0      # how many library communicators this process has created (every rank counts the same way)


def _cur_stream(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class GradSync:
    def __init__(self, group=None, always_exchange=False, native=None, sync_bn=False):
        """always_exchange: issue the collectives even in a one-rank group (exercises the RCCL path on a single GPU).
        native: run the GPU collectives through the C ABI (default: yes whenever CUDA is available; `False` keeps them on
        torch.distributed's own RCCL communicator — an A/B switch).  sync_bn: exact global-batch BatchNorm (see module doc)."""
        self.always_exchange = always_exchange
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised torch.distributed process group")
        if group is not None and native:
            raise RuntimeError("the native RCCL communicator spans the default group only")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.sync_bn = bool(sync_bn)
        self._side = None
        self._pending = {}
        self._slots = {}
        self.native = (torch.cuda.is_available() and group is None) if native is None else bool(native)
        self._lib = None
        if self.sync_bn and torch.cuda.is_available() and not self.native:
            raise RuntimeError("sync_bn needs the library's own communicator (native=True): the statistic sums are all-reduced "
                               "inside the BatchNorm entry points")
        if self.native:
            self._init_native()

    # ---- C-ABI communicator --------------------------------------------------------------------------------------
    def _init_native(self):
        lib = _lib.load()
        if lib.pcg_dp_world() == 0:
            global _native_generation
            store = dist.distributed_c10d._get_default_store()
            # one key per communicator generation: a second communicator (after parallel.shutdown()) must not read the first one's id
            key = f"pcgan_hip/rccl_unique_id/{_native_generation}"
            _native_generation += 1
            if self.rank == 0:
                buf = ctypes.create_string_buffer(128)
                check(lib.pcg_dp_unique_id(buf), "pcg_dp_unique_id")
                store.set(key, bytes(buf.raw))
            uid = store.get(key)
            check(lib.pcg_dp_init(ctypes.create_string_buffer(bytes(uid), 128), self.rank, self.world), "pcg_dp_init")
        elif lib.pcg_dp_world() != self.world or lib.pcg_dp_rank() != self.rank:
            raise RuntimeError("libpcgan_hip's RCCL communicator was initialised for a different group")
        self._lib = lib
        self._side = torch.cuda.ExternalStream(lib.pcg_dp_side_stream())
        check(lib.pcg_dp_sync_batchnorm(1 if self.sync_bn else 0), "pcg_dp_sync_batchnorm")

    def rccl_ranks(self):
        """Ranks the communicator that carries the gradient exchange spans."""
        return int(self._lib.pcg_dp_world()) if self.native else self.world

    def rccl_version(self):
        """ncclGetVersion of the RCCL carrying the exchange (e.g. 22105), or None."""
        if self.native:
            v = int(self._lib.pcg_dp_rccl_version())
            return v or None
        try:
            v = torch.cuda.nccl.version()
            return int(v[0]) * 10000 + int(v[1]) * 100 + int(v[2]) if isinstance(v, tuple) else int(v)
        except Exception:
            return None

    def barrier(self):
        """All ranks have reached this point and their queued GPU work is done.  On the GPU with the library's communicator: a
        4-byte all-reduce on THAT communicator (pcg_dp_barrier) + a stream synchronise — the timed region of the benches then uses
        one communicator only; otherwise torch.distributed.barrier()."""
        if self.native and torch.cuda.is_available():
            check(self._lib.pcg_dp_barrier(_cur_stream()), "pcg_dp_barrier")
            torch.cuda.current_stream().synchronize()
        else:
            dist.barrier(group=self.group)

    def _slot(self, net):
        s = self._slots.get(id(net))
        if s is None:
            s = len(self._slots)
            if s >= 8:
                raise RuntimeError("GradSync: more than 8 nets with overlapped reductions")
            self._slots[id(net)] = s
        return s

    def _exchange(self):
        return self.world > 1 or self.always_exchange

    # ---- collectives -----------------------------------------------------------------------------------------------
    def _allreduce_mean(self, flat):
        if not self._exchange():
            return
        if self.native and flat.is_cuda:
            check(self._lib.pcg_dp_allreduce(ctypes.c_void_p(flat.data_ptr()), flat.numel(), _cur_stream(flat.device)), "pcg_dp_allreduce")
        elif self.backend == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:  # gloo (CPU rehearsal): no AVG
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)

    def allreduce_sum_f64_(self, t):
        """In place, in stream order: sum of a float64 tensor over the ranks (exact-BatchNorm statistic sums)."""
        if not self._exchange():
            return t
        if self.native and t.is_cuda:
            check(self._lib.pcg_dp_allreduce_sum_f64(ctypes.c_void_p(t.data_ptr()), t.numel(), _cur_stream(t.device)),
                  "pcg_dp_allreduce_sum_f64")
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

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
        if self.native:
            slot = self._slot(net)
            if self._exchange():
                check(self._lib.pcg_dp_allreduce_begin(ctypes.c_void_p(flat.data_ptr()), flat.numel(), slot, _cur_stream(flat.device)),
                      "pcg_dp_allreduce_begin")
                with torch.cuda.stream(self._side):
                    fn()
                check(self._lib.pcg_dp_record(slot), "pcg_dp_record")
                self._pending[id(net)] = slot
            else:
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

    def _wait_one(self, token):
        if self.native and isinstance(token, int):
            check(self._lib.pcg_dp_allreduce_wait(token, _cur_stream()), "pcg_dp_allreduce_wait")
        else:
            torch.cuda.current_stream().wait_event(token)

    def wait(self, net):
        token = self._pending.pop(id(net), None)
        if token is not None:
            self._wait_one(token)

    def wait_all(self):
        for token in self._pending.values():
            self._wait_one(token)
        self._pending.clear()

    def broadcast_(self, t, src=0):
        """In place, in stream order: every rank gets rank `src`'s bytes."""
        if self.native and t.is_cuda:
            check(self._lib.pcg_dp_broadcast(ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size(), src, _cur_stream(t.device)),
                  "pcg_dp_broadcast")
        else:
            dist.broadcast(t, src=src, group=self.group)
        return t


def broadcast_parameters(net, src=0, group=None, dp=None):
    """Make every replica start from rank `src`'s weights and BatchNorm buffers (through `dp`'s communicator when given)."""
    if dp is not None:
        dp.broadcast_(net.flat_params, src)
        for b in net.buffers():
            dp.broadcast_(b, src)
        return
    dist.broadcast(net.flat_params, src=src, group=group)
    for b in net.buffers():
        dist.broadcast(b, src=src, group=group)


def shutdown():
    """Destroy the library's RCCL communicator (before torch.distributed.destroy_process_group)."""
    lib = _lib.load()
    if lib.pcg_dp_world():
        check(lib.pcg_dp_shutdown(), "pcg_dp_shutdown")

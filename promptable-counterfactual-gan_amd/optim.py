"""torch.optim.Adam / AdamW drop-ins on the fused HIP kernel (mnist_dcgan.py:126-127; mnist/trainer.py:77-78;
mnist_wgan_conditional.py:118-119).

Parameters of a `FlatModule` are views of one flat buffer, so the whole net is updated by ONE launch; parameters
that are adjacent in memory are coalesced into segments generically (any dense fp32 parameter works).  The step
counter lives on the device and the bias corrections are computed there, so `step()` can be captured in a HIP graph.
"""
import torch

from . import ops
from ._lib import PcgError


class Adam(torch.optim.Optimizer):
    _decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise PcgError("amsgrad is not implemented (the reference never enables it)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._segments = None  # per group: list of dicts(param_flat, grad_flat, exp_avg, exp_avg_sq, step, hyper)

    # -- segments: maximal runs of parameters that are adjacent in memory (with identical grad layout) -----
    @staticmethod
    def _dense_span(t):
        """(ptr, numel) if the tensor's elements occupy one dense span of memory (any permutation), else None."""
        if t.numel() == 0:
            return None
        dims = sorted(((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1), key=lambda d: d[0])
        expect = 1
        for st, sz in dims:
            if st != expect:
                return None
            expect *= sz
        return t.data_ptr(), t.numel()

    def _build(self):
        self._segments = []
        for group in self.param_groups:
            items = []
            for p in group["params"]:
                if not p.requires_grad and p.grad is None:
                    continue
                if p.grad is None:
                    raise PcgError("Adam.step(): a parameter has no gradient yet (FlatModule attaches .grad views at its "
                                   "first forward/zero_grad; call step() after backward)")
                sp, sg = self._dense_span(p.data), self._dense_span(p.grad)
                if sp is None or sg is None or p.grad.stride() != p.data.stride():
                    raise PcgError("Adam: parameter/gradient memory must be dense with identical layout")
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise PcgError("Adam: parameters must be float32 tensors on the GPU")
                items.append((sp[0], sg[0], sp[1], p))
            items.sort(key=lambda it: it[0])
            segs = []
            for pptr, gptr, n, p in items:
                if segs:
                    s = segs[-1]
                    gap = (pptr - s["pend"]) // 4
                    # FlatModule pads every parameter to a multiple of 4 floats; the padding is zero in both buffers
                    if 0 <= gap < 4 and (pptr - s["pend"]) % 4 == 0 and gptr - s["gptr"] == pptr - s["pptr"] and p.device == s["dev"]:
                        s["pend"] = pptr + 4 * n
                        s["params"].append(p)
                        continue
                segs.append({"pptr": pptr, "gptr": gptr, "pend": pptr + 4 * n, "params": [p], "dev": p.device})
            built = []
            for s in segs:
                n = (s["pend"] - s["pptr"]) // 4
                p0 = s["params"][0]
                # typed views over the raw spans (no copy): storage-level views of the first parameter / gradient
                pflat = torch.as_strided(p0.data, (n,), (1,), storage_offset=p0.data.storage_offset())
                gflat = torch.as_strided(p0.grad, (n,), (1,), storage_offset=p0.grad.storage_offset())
                assert pflat.data_ptr() == s["pptr"] and gflat.data_ptr() == s["gptr"]
                dev = s["dev"]
                st = {
                    "param": pflat, "grad": gflat, "n": n, "params": s["params"],
                    "exp_avg": ops.fill(torch.empty(n, dtype=torch.float32, device=dev), 0.0),
                    "exp_avg_sq": ops.fill(torch.empty(n, dtype=torch.float32, device=dev), 0.0),
                    "step": torch.zeros(1, dtype=torch.int64, device=dev),
                    "hyper": torch.zeros(12, dtype=torch.float32, device=dev),       # 48 bytes of kernel-side scratch (ticket + cached corrections)
                }
                built.append(st)
            self._segments.append(built)

    def _stale(self):
        if self._segments is None:
            return True
        for built in self._segments:
            for st in built:
                p0 = st["params"][0]
                if p0.grad is None or p0.data_ptr() != st["param"].data_ptr() or p0.grad.data_ptr() != st["grad"].data_ptr():
                    return True
        return False

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._stale():
            if self._segments is not None and any(int(st["step"].item()) for b in self._segments for st in b):
                raise PcgError("Adam: parameter storage changed after optimisation started (module moved or re-flattened)")
            self._build()
        for group, built in zip(self.param_groups, self._segments):
            b1, b2 = group["betas"]
            for st in built:
                ops.adam_step_capturable(st["param"], st["grad"], st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), float(b1),
                                         float(b2), float(group["eps"]), float(group["weight_decay"]), self._decoupled,
                                         st["step"], st["hyper"])
        return loss

    def zero_grad(self, set_to_none=False):
        """Zero gradients in place (one launch per segment); `.grad` views are kept."""
        if self._stale():
            have = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            if not have:
                return
            if self._segments is None or not any(int(st["step"].item()) for b in self._segments for st in b):
                self._build()
        for built in self._segments:
            for st in built:
                ops.fill(st["grad"], 0.0)

    # -- state snapshot (used around HIP-graph warm-up, which has to execute real steps before capture) ---------------
    def snapshot(self):
        """Copies of (exp_avg, exp_avg_sq, step) per segment, or None if no step has built the state yet."""
        if self._segments is None:
            return None
        return [[(st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"].clone()) for st in built] for built in self._segments]

    def restore(self, snap):
        """Put the state back (snap=None: back to 'never stepped' — zero moments, step 0 — keeping the buffers, so a
        captured graph that references them stays valid)."""
        if self._segments is None:
            return
        for gi, built in enumerate(self._segments):
            for si, st in enumerate(built):
                if snap is None:
                    ops.fill(st["exp_avg"], 0.0); ops.fill(st["exp_avg_sq"], 0.0); st["step"].zero_()
                else:
                    ea, es, stp = snap[gi][si]
                    st["exp_avg"].copy_(ea); st["exp_avg_sq"].copy_(es); st["step"].copy_(stp)

    # number of kernel launches a step() issues (for tests / DESIGN.md)
    def num_segments(self):
        if self._stale():
            self._build()
        return sum(len(b) for b in self._segments)


class AdamW(Adam):
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)

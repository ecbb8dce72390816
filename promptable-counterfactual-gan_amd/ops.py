"""Tensor-level wrappers over the C ABI: PyTorch-ROCm tensors in, raw pointers across the boundary.

Activations are *physically* NHWC: tensors of shape [B, H, W, C], contiguous.  Conv weights are physically
OHWI [Cout, KH, KW, Cin] (ConvTranspose2d: [Cin, KH, KW, Cout]); `ohwi(p)` gives that view of a PyTorch
parameter of logical shape [Cout, Cin, KH, KW] kept in channels_last memory format.  Nothing here computes
on the CPU and nothing falls back to ATen kernels: every function launches kernels of libpcgan_hip.so on
torch's current HIP stream.
"""
import ctypes
import threading

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, ConvGeom, check  # noqa: F401

_ws_cache = {}

# Optional per-launch timing hook (bench.py): when set, every MFMA-family conv call is bracketed by two
# torch.cuda.Event records on the current stream and reported as hook(label, flops, start_event, end_event).
_conv_hook = None


def set_conv_hook(hook):
    global _conv_hook
    _conv_hook = hook


def _conv_label(g, op):
    """Kernel the C dispatcher picks for this geometry (mirrors conv_igemm.hip / thin_conv.hip)."""
    if g.Cin <= 3 or g.Cout <= 3:
        return f"thin_{op}"
    if op == "wgrad":
        n = g.KH * g.KW * g.Cin
        if g.Cout <= 64:
            return "conv_wgrad192_kernel<64x192>" if (n % 192 == 0 and n % 128 != 0) else "conv_wgrad_kernel<64x128>"
        return "conv_wgrad_kernel<128x128>"
    n = g.Cout if op == "fwd" else g.Cin
    if n > 64:
        return f"conv_{op}_kernel<128x128>"
    swz = op == "fwd" or g.stride == 1          # three-per-CU swizzled config: forward and single-phase grad-input
    return f"conv_{op}_kernel<128x64{'/swz3' if swz else ''}>"


def _conv_flops(g):
    return 2.0 * g.B * g.OH * g.OW * g.Cout * g.KH * g.KW * g.Cin


class _Timed:
    def __init__(self, g, op):
        _conv_scratch()                      # every conv entry point passes here: the stream-K launches get their scratch
        self.on = _conv_hook is not None
        if self.on:
            self.g, self.op = g, op
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.t0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.t1.record()
            _conv_hook(_conv_label(self.g, self.op), _conv_flops(self.g), self.t0, self.t1)
        return False


_sk_streams = {}      # (device index, stream handle) -> (parts, arrivals): scratch of the stream-K conv launches, alive for the process
_sk_pool = {}         # device index -> [(parts, arrivals), ...] spare scratch sets, allocated and zeroed OUTSIDE any capture
_sk_arrival_pool = _sk_pool      # (name kept for the tests)
_SK_SPARES = 2
_sk_last = None       # the stream key seen by the previous conv call (eager launches: skips the dictionary work)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def prepare_conv_scratch(device_index=None):
    """Keep `_SK_SPARES` spare stream-K scratch sets (64 MiB of partial tiles + zeroed arrival counters each) for the device:
    the library holds these pointers for the life of the process, so they must never come out of a graph's private pool, and
    the counters' zero-fill must never become a captured fill node.  Called when the library loads (current device) and by every
    conv call outside a capture; a conv call INSIDE a capture on a stream that has no scratch yet is served from the spares."""
    if device_index is None:
        device_index = torch.cuda.current_device()
    if torch.cuda.is_current_stream_capturing():
        return
    lib = _lib.load()
    dev = torch.device("cuda", device_index)
    nparts, narr = lib.pcg_conv_scratch_parts_bytes(), lib.pcg_conv_scratch_arrivals_bytes()
    pool = _sk_pool.setdefault(device_index, [])
    while len(pool) < _SK_SPARES:
        pool.append((torch.empty(nparts, dtype=torch.uint8, device=dev), torch.zeros(narr, dtype=torch.uint8, device=dev)))


def _conv_scratch():
    """First conv call on a stream: give the hybrid stream-K launches their scratch for it (include/pcgan_hip.h,
    pcg_conv_set_scratch).  The scratch comes from the spares of prepare_conv_scratch — never from an allocation made while a stream
    captures (r03 allocated the counters inside the capture when the pool was empty: memory of the graph's private pool handed to
    the library for the life of the process).  An empty pool under capture raises.  A stream the library has no slot left for (it
    keeps 64) simply runs the plain launches — same results up to the order of the K sum."""
    global _sk_last
    s = torch.cuda.current_stream()
    key = (s.device_index, s.cuda_stream)
    if key == _sk_last:
        return
    if key not in _sk_streams:
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            prepare_conv_scratch(s.device_index)
        pool = _sk_pool.setdefault(s.device_index, [])
        if not pool:
            raise _lib.PcgError("conv call on a new stream inside a HIP-graph capture with no spare stream-K scratch left: call "
                                "pcgan_amd.ops.prepare_conv_scratch() on this device before capturing (the scratch must not come "
                                "from the graph's private memory pool)")
        parts, arrivals = pool.pop()
        lib = _lib.load()
        if lib.pcg_conv_set_scratch(ctypes.c_void_p(s.cuda_stream), _p(parts), parts.numel(), _p(arrivals), arrivals.numel()) == 0:
            _sk_streams[key] = (parts, arrivals)
        else:
            pool.append((parts, arrivals))
            _sk_streams[key] = None
        if not capturing:
            prepare_conv_scratch(s.device_index)      # refill: the next capture stream finds a spare
    _sk_last = key


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _chk(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.PcgError(f"{name}: expected a tensor on the GPU (libpcgan_hip has no CPU path), got {t.device}")
    if t.dtype != dtype:
        raise _lib.PcgError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.PcgError(f"{name}: expected a contiguous tensor, got strides {t.stride()} for shape {tuple(t.shape)}")
    return t


_ws_retired = []   # scratch buffers that were outgrown: kept alive for the life of the process (see workspace())


def _scratch(kind, nbytes, device):
    """A cached scratch buffer per (device, HIP stream, kind), grown on demand.

    Stream-ordered use: the buffer belongs to the stream that is current at the call, so work issued on a side stream
    (GradSync.sync_then callbacks, graph capture streams) never shares scratch with the main stream.  A buffer that is outgrown
    is RETIRED, never freed: a captured HIP graph (nn.GraphedStep, house.GraphedTrainStep) has its address baked into split-K
    slabs / BatchNorm partial rows / weight-gradient slabs, and returning it to the caching allocator would let a later replay
    write into memory that meanwhile belongs to another tensor (tests/test_hip_graph.py)."""
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (device.type, device.index, stream, kind)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def workspace(nbytes, device):
    """Scratch for the current stream (split-K slabs, BatchNorm partial rows, reductions)."""
    return _scratch(0, nbytes, device)


def workspace2(nbytes, device):
    """A second scratch buffer (slab partials of linear_wgrad), separate from `workspace` so the two never alias."""
    return _scratch(2, nbytes, device)


_ticket_pool = {}


def _ticket_buffer(device):
    """Zero-initialised ticket counters of the linear weight-gradient kernels (they leave them zero), one set per stream.
    New sets come from a small pool zeroed OUTSIDE any graph capture: a `torch.zeros` issued while a stream is capturing would
    become a fill node that replays with every launch of the graph."""
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (device.type, device.index, stream, "tickets")
    tk = _ws_cache.get(key)
    if tk is None:
        n = _lib.load().pcg_linear_wgrad_ticket_count()
        pool = _ticket_pool.setdefault((device.type, device.index), [])
        capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if not pool and not capturing:
            pool.extend(torch.zeros((8, n), dtype=torch.int32, device=device).unbind(0))                  # setup: zeroed once
        tk = pool.pop() if pool else torch.zeros(n, dtype=torch.int32, device=device)
        _ws_cache[key] = tk
    return tk


def ohwi(w):
    """Physical [O, KH, KW, I] view of a conv parameter of logical shape [O, I, KH, KW] (channels_last memory)."""
    v = w.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise _lib.PcgError("conv weight is not in channels_last (OHWI) memory layout; FlatModule keeps it that way")
    return v


def conv_geom(B, IH, IW, Cin, Cout, KH, KW, stride, pad):
    OH = (IH + 2 * pad - KH) // stride + 1
    OW = (IW + 2 * pad - KW) // stride + 1
    return ConvGeom(B, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad)


# ---- convolution family --------------------------------------------------------------------------
class InputXform:
    """An activation that exists only as (pre-BatchNorm tensor z, folded BatchNorm coef[2][C], activation): consumers read
    act(z * coef[0][c] + coef[1][c]) inside their gathers (pcg_in_xform, include/pcgan_hip.h)."""

    def __init__(self, coef, act, slope):
        C = coef.numel() // 2
        self.coef, self.act, self.slope, self.C = coef, int(act), float(slope), C
        self.c_struct = _lib.InXform(coef.data_ptr(), coef.data_ptr() + 4 * C, int(act), float(slope))

    def ref(self):
        return ctypes.byref(self.c_struct)


def xform_ok(g, operand):
    """Can the conv kernels of geometry g take an input transform?  (MFMA path only; operand: 'x' -> Cin channels, 'dy' -> Cout.)"""
    return g.Cin > 3 and g.Cout > 3 and g.Cin % 4 == 0 and g.Cout % 4 == 0 and g.stride <= 2


class BnInput:
    """An activation that exists only as (pre-BatchNorm tensor z, the producing layer's batch statistics mean / invstd [groups][C], its
    gamma / beta, activation): a full-window one-channel convolution reads act(bn(z)) in its own loads (pcg_conv2d_*_bnin_full).  Accepted
    where an InputXform is: conv2d_fwd(xf=), conv2d_wgrad(xf_x=)."""

    def __init__(self, mean, invstd, gamma, beta, act, slope, groups=1):
        self.mean, self.invstd, self.gamma, self.beta = mean, invstd, gamma, beta
        self.act, self.slope, self.groups = int(act), float(slope), int(groups)


def bnin_full_ok(g, groups=1):
    return bool(_lib.load().pcg_conv2d_bnin_full_ok(ctypes.byref(g), int(groups)))


def xform_thin_ok(g):
    """A Cin = 1 layer whose dy-side operand may carry an input transform (pcg_conv2d_xf_thin_ok): conv2d_dgrad(xf=) / conv2d_wgrad(xf_dy=)."""
    return bool(_lib.load().pcg_conv2d_xf_thin_ok(ctypes.byref(g)))


def _xref(xf):
    return xf.ref() if xf is not None else None


def conv2d_fwd(g, x, w, bias=None, out=None, act=0, slope=0.0, xf=None):
    """y[B,OH,OW,Cout] = act(conv(x[B,IH,IW,Cin], w OHWI) (+ bias)); xf: input transform on x (InputXform)."""
    _chk(x, "x"); _chk(w, "w")
    assert x.numel() == g.B * g.IH * g.IW * g.Cin and w.numel() == g.Cout * g.KH * g.KW * g.Cin
    y = out if out is not None else torch.empty((g.B, g.OH, g.OW, g.Cout), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    need = lib.pcg_conv2d_fwd_workspace_bytes(ctypes.byref(g))
    ws = workspace(need, x.device) if need else None
    with _Timed(g, "fwd"):
        if isinstance(xf, BnInput):
            check(lib.pcg_conv2d_fwd_bnin_full(ctypes.byref(g), _p(x), _p(xf.mean), _p(xf.invstd), _p(xf.gamma), _p(xf.beta), xf.act, xf.slope,
                                               xf.groups, _p(w), _p(bias), int(act), float(slope), _p(y), _stream()), "pcg_conv2d_fwd_bnin_full")
        elif xf is not None:
            assert xf.C == g.Cin
            check(lib.pcg_conv2d_fwd_xf(ctypes.byref(g), _p(x), xf.ref(), _p(w), _p(bias), int(act), float(slope), _p(y), _p(ws),
                                        ws.numel() if need else 0, _stream()), "pcg_conv2d_fwd_xf")
        else:
            check(lib.pcg_conv2d_fwd_act(ctypes.byref(g), _p(x), _p(w), _p(bias), int(act), float(slope), _p(y), _p(ws),
                                         ws.numel() if need else 0, _stream()), "pcg_conv2d_fwd_act")
    return y


def conv2d_dgrad(g, dy, w, bias_x=None, out=None, act=0, slope=0.0, xf=None):
    """dx[B,IH,IW,Cin] = act(conv_transpose(dy[B,OH,OW,Cout], w OHWI) (+ bias_x per Cin)); xf: input transform on dy (the forward
    input of a ConvTranspose2d layer)."""
    _chk(dy, "dy"); _chk(w, "w")
    assert dy.numel() == g.B * g.OH * g.OW * g.Cout and w.numel() == g.Cout * g.KH * g.KW * g.Cin
    dx = out if out is not None else torch.empty((g.B, g.IH, g.IW, g.Cin), dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    need = lib.pcg_conv2d_dgrad_workspace_bytes(ctypes.byref(g))
    ws = workspace(need, dy.device) if need else None
    with _Timed(g, "dgrad"):
        if xf is not None:
            assert xf.C == g.Cout
            check(lib.pcg_conv2d_dgrad_xf(ctypes.byref(g), _p(dy), xf.ref(), _p(w), _p(bias_x), int(act), float(slope), _p(dx), _p(ws),
                                          ws.numel() if need else 0, _stream()), "pcg_conv2d_dgrad_xf")
        else:
            check(lib.pcg_conv2d_dgrad_act(ctypes.byref(g), _p(dy), _p(w), _p(bias_x), int(act), float(slope), _p(dx), _p(ws),
                                           ws.numel() if need else 0, _stream()), "pcg_conv2d_dgrad_act")
    return dx


def conv_bn_train(g, a, w, bias, transposed, eps, momentum, running_mean, running_var, num_batches_tracked, xf=None, gamma=None,
                  beta=None):
    """Convolution (transposed=False: fwd; True: dgrad = ConvTranspose2d forward) + training-mode BatchNorm statistics of
    its output.  One fused call on MFMA layers (statistics come out of the conv epilogue), two calls otherwise.
    xf: input transform on the activation operand `a`.  gamma + beta given: also return the folded scale / shift [2][C] of this
    layer's BatchNorm (written by the statistics finalize — the next consumer's transform).
    Returns (z, save_mean, save_invstd) or (z, save_mean, save_invstd, coef)."""
    _chk(a, "a"); _chk(w, "w")
    lib = _lib.load()
    C = g.Cin if transposed else g.Cout
    shape = (g.B, g.IH, g.IW, g.Cin) if transposed else (g.B, g.OH, g.OW, g.Cout)
    need = (lib.pcg_conv2d_dgrad_bn_workspace_bytes if transposed else lib.pcg_conv2d_fwd_bn_workspace_bytes)(ctypes.byref(g))
    want_coef = gamma is not None
    if need == 0:
        z = conv2d_dgrad(g, a, w, bias, xf=xf) if transposed else conv2d_fwd(g, a, w, bias, xf=xf)
        return (z,) + tuple(bn_train_stats(z, C, eps, momentum, running_mean, running_var, num_batches_tracked, gamma=gamma, beta=beta))
    z = torch.empty(shape, dtype=torch.float32, device=a.device)
    mean = torch.empty(C, dtype=torch.float32, device=a.device)
    invstd = torch.empty(C, dtype=torch.float32, device=a.device)
    coef = torch.empty(2 * C, dtype=torch.float32, device=a.device) if want_coef else None
    ws = workspace(need, a.device)
    fn = lib.pcg_conv2d_dgrad_bn_xf if transposed else lib.pcg_conv2d_fwd_bn_xf
    with _Timed(g, "dgrad" if transposed else "fwd"):
        check(fn(ctypes.byref(g), _p(a), _xref(xf), _p(w), _p(bias), _p(z), eps, momentum, _p(mean), _p(invstd), _p(running_mean),
                 _p(running_var), _p(num_batches_tracked), _p(gamma), _p(beta), _p(coef), _p(ws), ws.numel(), _stream()),
              "pcg_conv2d_*_bn_xf")
    return (z, mean, invstd, coef) if want_coef else (z, mean, invstd)


def conv_bwd_data_fused(g, d, w, transposed, below_act, below_slope, a_below=None, z_below=None, bn=None):
    """Gradient w.r.t. a layer's input with the activation derivative of the layer BELOW applied in the epilogue.
    transposed=False: Conv2d (grad-input kernel); True: ConvTranspose2d (its grad-input is a forward convolution).
    z_below + bn=(mean, invstd, gamma, beta): the layer below is Conv -> BatchNorm(train) -> act; returns (dm, partial, nparts)
    for bn_bwd_partial.  a_below: the layer below is Conv -> act without BatchNorm; returns (dx_masked, None, 0).
    Returns None when the layer is not eligible (thin layers, split-K): the caller then takes the unfused path."""
    lib = _lib.load()
    if (not transposed and bn is None and a_below is not None and g.Cout <= 3
            and lib.pcg_conv2d_dgrad_mask_thin_ok(ctypes.byref(g))):       # a one-channel layer's grad-input: mask in the thin expand kernel
        _chk(d, "d"); _chk(w, "w"); _chk(a_below, "a_below")
        out = torch.empty((g.B, g.IH, g.IW, g.Cin), dtype=torch.float32, device=d.device)
        assert a_below.numel() == out.numel()
        check(lib.pcg_conv2d_dgrad_mask(ctypes.byref(g), _p(d), _p(w), _p(a_below), int(below_act), float(below_slope), _p(out), None, 0, _stream()),
              "pcg_conv2d_dgrad_mask(thin)")
        return out, None, 0
    if g.Cin <= 3 or g.Cout <= 3 or g.Cin % 4 or g.Cout % 4 or g.stride > 2:
        return None
    shape = (g.B, g.OH, g.OW, g.Cout) if transposed else (g.B, g.IH, g.IW, g.Cin)
    if transposed and lib.pcg_conv2d_fwd_workspace_bytes(ctypes.byref(g)) != 0:   # split-K forward: sums are incomplete per slab
        return None
    _chk(d, "d"); _chk(w, "w")
    out = torch.empty(shape, dtype=torch.float32, device=d.device)
    if bn is None:
        _chk(a_below, "a_below")
        assert a_below.numel() == out.numel()
        fn = lib.pcg_conv2d_fwd_mask if transposed else lib.pcg_conv2d_dgrad_mask
        with _Timed(g, "fwd" if transposed else "dgrad"):
            check(fn(ctypes.byref(g), _p(d), _p(w), _p(a_below), int(below_act), float(below_slope), _p(out), None, 0, _stream()),
                  "pcg_conv2d_*_mask")
        return out, None, 0
    _chk(z_below, "z_below")
    assert z_below.numel() == out.numel()
    mean, invstd, gamma, beta = bn
    need = (lib.pcg_conv2d_fwd_bn_workspace_bytes if transposed else lib.pcg_conv2d_dgrad_bn_workspace_bytes)(ctypes.byref(g))
    if need == 0:
        return None
    nparts = (lib.pcg_conv2d_fwd_bn_partial_rows if transposed else lib.pcg_conv2d_dgrad_bn_partial_rows)(ctypes.byref(g))
    partial = torch.empty(need // 4, dtype=torch.float32, device=d.device)   # own tensor: lives until bn_bwd_partial has read it
    fn = lib.pcg_conv2d_fwd_bnbwd if transposed else lib.pcg_conv2d_dgrad_bnbwd
    with _Timed(g, "fwd" if transposed else "dgrad"):
        check(fn(ctypes.byref(g), _p(d), _p(w), _p(z_below), _p(mean), _p(invstd), _p(gamma), _p(beta), int(below_act), float(below_slope),
                 _p(out), _p(partial), need, _stream()), "pcg_conv2d_*_bnbwd")
    return out, partial, nparts


def conv_weight_adjoint(w):
    """OHWI weight of the adjoint convolution of a stride-1 layer: w[co][kh][kw][ci] -> [ci][KH-1-kh][KW-1-kw][co]."""
    _chk(w, "w")
    Cout, KH, KW, Cin = w.shape
    wa = torch.empty((Cin, KH, KW, Cout), dtype=torch.float32, device=w.device)
    check(_lib.load().pcg_conv_weight_adjoint(_p(w), _p(wa), Cout, KH, KW, Cin, _stream()), "pcg_conv_weight_adjoint")
    return wa


def conv_weight_adjoint_many(ws):
    """conv_weight_adjoint of up to 16 OHWI weights of one shape in ONE launch; returns the list of adjoint weights (views of one buffer)."""
    n = len(ws)
    Cout, KH, KW, Cin = ws[0].shape
    for w in ws:
        _chk(w, "w")
        assert tuple(w.shape) == (Cout, KH, KW, Cin)
    buf = torch.empty((n, Cin, KH, KW, Cout), dtype=torch.float32, device=ws[0].device)
    src = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    dst = (ctypes.c_void_p * n)(*[buf[i].data_ptr() for i in range(n)])
    check(_lib.load().pcg_conv_weight_adjoint_many(src, dst, n, Cout, KH, KW, Cin, _stream()), "pcg_conv_weight_adjoint_many")
    return [buf[i] for i in range(n)]


def adjoint_geom(g):
    """Geometry of the grad-input of a stride-1 convolution seen as a forward convolution of dy (see conv_weight_adjoint)."""
    assert g.stride == 1
    return conv_geom(g.B, g.OH, g.OW, g.Cout, g.Cin, g.KH, g.KW, 1, g.KH - 1 - g.pad)


def conv2d_dgrad_add_mask(g, dy, w, addend, a_below, act, slope, out=None, transposed=False):
    """dx = (conv_dgrad(dy, w) + addend) * act'(a_below) in one epilogue (out may be `addend`); transposed: the forward-kernel form."""
    _chk(dy, "dy"); _chk(w, "w"); _chk(addend, "addend"); _chk(a_below, "a_below")
    shape = (g.B, g.OH, g.OW, g.Cout) if transposed else (g.B, g.IH, g.IW, g.Cin)
    n = shape[0] * shape[1] * shape[2] * shape[3]
    assert addend.numel() == n and a_below.numel() == n
    dx = out if out is not None else torch.empty(shape, dtype=torch.float32, device=dy.device)
    fn = _lib.load().pcg_conv2d_fwd_add_mask if transposed else _lib.load().pcg_conv2d_dgrad_add_mask
    with _Timed(g, "fwd" if transposed else "dgrad"):
        check(fn(ctypes.byref(g), _p(dy), _p(w), _p(addend), _p(a_below), int(act), float(slope), _p(dx), _stream()), "pcg_conv2d_*_add_mask")
    return dx


def conv2d_dgrad_add(g, dy, w, addend, out=None, bnsum=None, transposed=False):
    """dx = conv_dgrad(dy, w) + addend (out may be `addend`: in place) — the add of a skip connection in the grad-input epilogue.
    bnsum = (z_next, mean, invstd, scale): also leave the BatchNorm-backward column sums of scale*dx for the BatchNorm whose
    pre-normalisation output is z_next (the next one down the skip chain); returns (dx, partial, nparts) for bn_bwd_partial(dm_scale=scale).
    transposed=True: the forward-kernel form (g, w describe a forward convolution whose input is `dy`: ConvTranspose2d layers, or a
    stride-1 layer through adjoint_geom / conv_weight_adjoint).  addend=None (with bnsum only): the plain grad-input and its sums."""
    _chk(dy, "dy"); _chk(w, "w")
    shape = (g.B, g.OH, g.OW, g.Cout) if transposed else (g.B, g.IH, g.IW, g.Cin)
    if addend is None:
        assert bnsum is not None, "conv2d_dgrad_add: addend=None only together with bnsum"
    else:
        _chk(addend, "addend")
        assert addend.numel() == shape[0] * shape[1] * shape[2] * shape[3]
    dx = out if out is not None else torch.empty(shape, dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    if bnsum is not None:
        z_next, mean, invstd, scale = bnsum
        _chk(z_next, "z_next")
        assert z_next.numel() == dx.numel()
        need = (lib.pcg_conv2d_fwd_bn_workspace_bytes if transposed else lib.pcg_conv2d_dgrad_bn_workspace_bytes)(ctypes.byref(g))
        if need == 0:
            raise _lib.PcgError("conv2d_dgrad_add(bnsum=...): layer not eligible for the fused column sums")
        nparts = (lib.pcg_conv2d_fwd_bn_partial_rows if transposed else lib.pcg_conv2d_dgrad_bn_partial_rows)(ctypes.byref(g))
        partial = torch.empty(need // 4, dtype=torch.float32, device=dy.device)
        fn = lib.pcg_conv2d_fwd_add_bnsum if transposed else lib.pcg_conv2d_dgrad_add_bnsum
        with _Timed(g, "fwd" if transposed else "dgrad"):
            check(fn(ctypes.byref(g), _p(dy), _p(w), _p(addend), _p(z_next), _p(mean), _p(invstd), float(scale),
                     _p(dx), _p(partial), need, _stream()), "pcg_conv2d_*_add_bnsum")
        return dx, partial, nparts
    if transposed:
        with _Timed(g, "fwd"):
            check(lib.pcg_conv2d_fwd_add(ctypes.byref(g), _p(dy), _p(w), _p(addend), _p(dx), None, 0, _stream()), "pcg_conv2d_fwd_add")
        return dx
    with _Timed(g, "dgrad"):
        check(_lib.load().pcg_conv2d_dgrad_add(ctypes.byref(g), _p(dy), _p(w), _p(addend), _p(dx), None, 0, _stream()), "pcg_conv2d_dgrad_add")
    return dx


def bn_bwd_partial(dm, x, C, mean, invstd, gamma, partial, nparts, dgamma, dbeta, accumulate, out=None, dcol=None, accumulate_col=False,
                   dm_scale=1.0):
    """BatchNorm backward from the column sums a fused grad-input epilogue left in `partial` (dm is already masked).
    dcol: also accumulate the column sums of the returned dx there (the bias gradient of the conv in front of the BatchNorm).
    dm_scale: the sums include this factor and the apply pass multiplies dm by it (conv2d_dgrad_add(bnsum=...))."""
    _chk(dm, "dm"); _chk(x, "x")
    rows = x.numel() // C
    dx = out if out is not None else torch.empty_like(x)
    lib = _lib.load()
    if dcol is None and dm_scale == 1.0:
        ws = workspace(lib.pcg_bn_bwd_partial_workspace_bytes(C), x.device)
        check(lib.pcg_bn_bwd_partial(_p(dm), _p(x), rows, C, _p(mean), _p(invstd), _p(gamma), _p(partial), int(nparts), _p(dx), _p(dgamma),
                                     _p(dbeta), int(bool(accumulate)), _p(ws), ws.numel(), _stream()), "pcg_bn_bwd_partial")
    else:
        ws = workspace(lib.pcg_bn_bwd_partial_db_workspace_bytes(C), x.device)
        check(lib.pcg_bn_bwd_partial_db(_p(dm), _p(x), rows, C, _p(mean), _p(invstd), _p(gamma), _p(partial), int(nparts), float(dm_scale),
                                        _p(dx), _p(dgamma), _p(dbeta), int(bool(accumulate)), _p(dcol), int(bool(accumulate_col)), _p(ws),
                                        ws.numel(), _stream()), "pcg_bn_bwd_partial_db")
    return dx


# ---- deferred slab reductions (pcg_slab_defer_*): one reduction launch per backward sweep instead of one per weight gradient -------
_slab_defer = threading.local()      # per thread, like the library's record (autograd runs a net's backward on its own thread)
_SLAB_SLOT0 = 16                      # scratch kinds 16, 17, ...: one slab buffer per deferred weight gradient of the sweep


class slab_reductions_deferred:
    """with ops.slab_reductions_deferred(): ... conv2d_wgrad calls ... — the weight gradients (dw) are complete when the block exits.
    Inside, every split-K weight gradient keeps its slabs in its own scratch slot; the exit sums all of them in one launch,
    bit-identical to the per-call reductions.  Nothing inside the block may read a dw written inside it.  Not re-entrant; on the
    current stream of the entry."""

    def __init__(self, enabled=True):
        self.enabled = enabled and not getattr(_slab_defer, "on", False)

    def __enter__(self):
        if self.enabled:
            check(_lib.load().pcg_slab_defer_begin(_stream()), "pcg_slab_defer_begin")
            _slab_defer.on, _slab_defer.n = True, 0
        return self

    def __exit__(self, et, ev, tb):
        if self.enabled:
            _slab_defer.on = False
            rc = _lib.load().pcg_slab_defer_flush(_stream())
            if et is None:
                check(rc, "pcg_slab_defer_flush")
        return False


def conv2d_wgrad(g, x, dy, dw, accumulate, xf_x=None, xf_dy=None):
    """dw (OHWI, written in place) (+)= sum over pixels of dy (x) gathered x.  xf_x / xf_dy: input transform on the operand that
    is an activation (x for Conv2d; dy for ConvTranspose2d, whose forward input sits on the dy side of the adjoint geometry)."""
    _chk(x, "x"); _chk(dy, "dy"); _chk(dw, "dw")
    assert dw.numel() == g.Cout * g.KH * g.KW * g.Cin
    lib = _lib.load()
    need = lib.pcg_conv2d_wgrad_workspace_bytes(ctypes.byref(g))
    if getattr(_slab_defer, "on", False):
        ws = _scratch(_SLAB_SLOT0 + _slab_defer.n, need, x.device)       # its own slot: the slabs live until the sweep's flush
        _slab_defer.n += 1
    else:
        ws = workspace(need, x.device)
    with _Timed(g, "wgrad"):
        if isinstance(xf_x, BnInput):
            check(lib.pcg_conv2d_wgrad_bnin_full(ctypes.byref(g), _p(x), _p(xf_x.mean), _p(xf_x.invstd), _p(xf_x.gamma), _p(xf_x.beta), xf_x.act,
                                                 xf_x.slope, xf_x.groups, _p(dy), _p(dw), int(bool(accumulate)), _p(ws), ws.numel(), _stream()),
                  "pcg_conv2d_wgrad_bnin_full")
        elif xf_x is None and xf_dy is None:
            check(lib.pcg_conv2d_wgrad(ctypes.byref(g), _p(x), _p(dy), _p(dw), int(bool(accumulate)), _p(ws), ws.numel(), _stream()),
                  "pcg_conv2d_wgrad")
        else:
            check(lib.pcg_conv2d_wgrad_xf(ctypes.byref(g), _p(x), _xref(xf_x), _p(dy), _xref(xf_dy), _p(dw), int(bool(accumulate)), _p(ws),
                                          ws.numel(), _stream()), "pcg_conv2d_wgrad_xf")
    return dw


def colsum(dy2d_rows, C, dy, db, accumulate):
    lib = _lib.load()
    need = lib.pcg_colsum_workspace_bytes(dy2d_rows, C)
    ws = workspace(need, dy.device)
    check(lib.pcg_colsum(_p(dy), dy2d_rows, C, _p(db), int(bool(accumulate)), _p(ws), ws.numel(), _stream()), "pcg_colsum")
    return db


# ---- BatchNorm + activation -------------------------------------------------------------------------
def bn_train_stats(x, C, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None, gamma=None, beta=None):
    """(save_mean, save_invstd); with gamma + beta also the folded scale / shift [2][C] (see conv_bn_train)."""
    _chk(x, "x")
    rows = x.numel() // C
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    coef = torch.empty(2 * C, dtype=torch.float32, device=x.device) if gamma is not None else None
    lib = _lib.load()
    ws = workspace(lib.pcg_bn_workspace_bytes(rows, C), x.device)
    check(lib.pcg_bn_train_stats_coef(_p(x), rows, C, eps, momentum, _p(mean), _p(invstd), _p(running_mean), _p(running_var),
                                      _p(num_batches_tracked), _p(gamma), _p(beta), _p(coef), _p(ws), ws.numel(), _stream()),
          "pcg_bn_train_stats")
    return (mean, invstd, coef) if gamma is not None else (mean, invstd)


def bn_apply_act(x, C, mean, invstd_or_var, gamma, beta, act, slope=0.0, var_eps=-1.0, out=None, residual=None, alpha=1.0):
    """y = residual + alpha * act(bn(x))  (residual None, alpha 1: plain BatchNorm + activation)."""
    _chk(x, "x")
    y = out if out is not None else torch.empty_like(x)
    check(_lib.load().pcg_bn_apply_act(_p(x), x.numel() // C, C, _p(mean), _p(invstd_or_var), var_eps, _p(gamma), _p(beta),
                                       act, slope, _p(residual), alpha, _p(y), _stream()), "pcg_bn_apply_act")
    return y


def bn_act_bwd(dy, x, y, C, mean, invstd, gamma, act, slope, dgamma, dbeta, accumulate, out=None, dy_scale=1.0, beta=None, dcol=None,
               accumulate_col=False):
    """BatchNorm(train) + activation backward.  y=None with ReLU / LeakyReLU and `beta` given: the mask is recomputed from x
    (pcg_bn_act_bwd_premask) instead of read from the saved activation.  dcol: also accumulate the column sums of the returned dx
    there (the bias gradient of the conv in front of the BatchNorm) — taken in the apply pass, no separate reduction."""
    _chk(dy, "dy"); _chk(x, "x")
    if y is not None:
        _chk(y, "y")
    rows = x.numel() // C
    dx = out if out is not None else torch.empty_like(x)
    lib = _lib.load()
    if dcol is not None:
        premask = y is None and act in (ACT_RELU, ACT_LRELU)
        if premask and beta is None:
            raise _lib.PcgError("bn_act_bwd: pass y, or beta to recompute the activation mask")
        ws = workspace(lib.pcg_bn_db_workspace_bytes(rows, C), x.device)
        check(lib.pcg_bn_act_bwd_db(_p(dy), _p(x), _p(y), rows, C, _p(mean), _p(invstd), _p(gamma), _p(beta if premask else None), act, slope,
                                    dy_scale, _p(dx), _p(dgamma), _p(dbeta), int(bool(accumulate)), _p(dcol), int(bool(accumulate_col)),
                                    _p(ws), ws.numel(), _stream()), "pcg_bn_act_bwd_db")
        return dx
    ws = workspace(lib.pcg_bn_workspace_bytes(rows, C), x.device)
    if y is None and act in (ACT_RELU, ACT_LRELU):
        if beta is None:
            raise _lib.PcgError("bn_act_bwd: pass y, or beta to recompute the activation mask")
        check(lib.pcg_bn_act_bwd_premask(_p(dy), _p(x), rows, C, _p(mean), _p(invstd), _p(gamma), _p(beta), act, slope, dy_scale, _p(dx),
                                         _p(dgamma), _p(dbeta), int(bool(accumulate)), _p(ws), ws.numel(), _stream()), "pcg_bn_act_bwd_premask")
        return dx
    check(lib.pcg_bn_act_bwd(_p(dy), _p(x), _p(y), rows, C, _p(mean), _p(invstd), _p(gamma), act, slope, dy_scale, _p(dx), _p(dgamma),
                             _p(dbeta), int(bool(accumulate)), _p(ws), ws.numel(), _stream()), "pcg_bn_act_bwd")
    return dx


def act_fwd(x, act, slope=0.0, out=None):
    _chk(x, "x")
    y = out if out is not None else torch.empty_like(x)
    check(_lib.load().pcg_act_fwd(_p(x), x.numel(), act, slope, _p(y), _stream()), "pcg_act_fwd")
    return y


def act_bwd(dy, y, act, slope=0.0, out=None):
    _chk(dy, "dy"); _chk(y, "y")
    dx = out if out is not None else torch.empty_like(dy)
    check(_lib.load().pcg_act_bwd(_p(dy), _p(y), dy.numel(), act, slope, _p(dx), _stream()), "pcg_act_bwd")
    return dx


# ---- losses -------------------------------------------------------------------------------------------
def bce_fwd_bwd(p, target, target_const, grad_scale=1.0, need_loss=True, need_grad=True, grad_out=None):
    """nn.BCELoss(mean).  Returns (loss[1] or None, dp or None).  `target` tensor or None (=> constant);
    `grad_out`: optional one-element device tensor multiplied into dp (autograd's grad_output)."""
    _chk(p, "p")
    loss = torch.empty(1, dtype=torch.float32, device=p.device) if need_loss else None
    dp = torch.empty_like(p) if need_grad else None
    check(_lib.load().pcg_bce_fwd_bwd(_p(p), _p(target), float(target_const), p.numel(), grad_scale, _p(grad_out), _p(loss),
                                      _p(dp), _stream()), "pcg_bce_fwd_bwd")
    return loss, dp


def bce_logits_fwd_bwd(z, target_const, grad_scale=1.0, need_loss=True, need_grad=True, grad_out=None):
    _chk(z, "z")
    loss = torch.empty(1, dtype=torch.float32, device=z.device) if need_loss else None
    dz = torch.empty_like(z) if need_grad else None
    check(_lib.load().pcg_bce_logits_fwd_bwd(_p(z), float(target_const), z.numel(), grad_scale, _p(grad_out), _p(loss), _p(dz),
                                             _stream()), "pcg_bce_logits_fwd_bwd")
    return loss, dz


def bce_pair(p, n_half, target0, target1, need_loss=True, need_grad=False, cotangents=(None, None, None)):
    """Two nn.BCELoss(mean) over p[:n_half] / p[n_half:] in one launch (pcg_bce_pair).  Returns (loss3 or None, dp or None);
    loss3 = [BCE(first half, target0), BCE(second half, target1), their sum].  cotangents: one-element device tensors or None."""
    _chk(p, "p")
    assert p.numel() == 2 * n_half
    loss = torch.empty(3, dtype=torch.float32, device=p.device) if need_loss else None
    dp = torch.empty_like(p) if need_grad else None
    g0, g1, g2 = cotangents
    check(_lib.load().pcg_bce_pair(_p(p), int(n_half), float(target0), float(target1), _p(g0), _p(g1), _p(g2), _p(loss), _p(dp), _stream()),
          "pcg_bce_pair")
    return loss, dp


# ---- grouped batches: G independent batches side by side along the batch axis (include/pcgan_hip.h, "grouped batches") -------------
def _fast_channels(C):
    q = C // 4
    return C % 4 == 0 and 0 < q <= 256 and (q & (q - 1)) == 0


def group_fwd_ok(g, groups):
    """Can conv_bn_train_g run geometry g (B = all groups' images) with `groups` groups?"""
    lib = _lib.load()
    return (g.B % groups == 0 and ((g.B // groups) * g.OH * g.OW) % 128 == 0 and _fast_channels(g.Cout)
            and lib.pcg_conv2d_fwd_bn_workspace_bytes(ctypes.byref(g)) > 0)


def group_dgrad_ok(g, groups):
    """Can conv_bwd_data_fused_g run the grad-input of geometry g on `groups` groups (equal phases, whole tiles per group)?"""
    if g.B % groups or g.Cin <= 3 or g.Cout <= 3 or g.Cin % 4 or g.Cout % 4 or g.stride > 2 or not _fast_channels(g.Cin):
        return False
    if _lib.load().pcg_conv2d_dgrad_bn_workspace_bytes(ctypes.byref(g)) == 0:
        return False
    s = g.stride
    if g.IH % s or g.IW % s:
        return False
    return ((g.B // groups) * (g.IH // s) * (g.IW // s)) % 128 == 0


def conv_bn_train_g(g, a, w, bias, eps, momentum, running_mean, running_var, num_batches_tracked, groups):
    """Conv2d forward over `groups` side-by-side batches + per-group training-mode BatchNorm statistics (pcg_conv2d_fwd_bn_g).
    Returns (z, save_mean [groups, C], save_invstd [groups, C])."""
    _chk(a, "a"); _chk(w, "w")
    lib = _lib.load()
    C = g.Cout
    need = lib.pcg_conv2d_fwd_bn_workspace_bytes(ctypes.byref(g))
    z = torch.empty((g.B, g.OH, g.OW, C), dtype=torch.float32, device=a.device)
    mean = torch.empty((groups, C), dtype=torch.float32, device=a.device)
    invstd = torch.empty((groups, C), dtype=torch.float32, device=a.device)
    ws = workspace(need, a.device)
    with _Timed(g, "fwd"):
        check(lib.pcg_conv2d_fwd_bn_g(ctypes.byref(g), _p(a), _p(w), _p(bias), _p(z), eps, momentum, _p(mean), _p(invstd), _p(running_mean),
                                      _p(running_var), _p(num_batches_tracked), int(groups), _p(ws), ws.numel(), _stream()), "pcg_conv2d_fwd_bn_g")
    return z, mean, invstd


def bn_apply_act_g(x, C, mean, invstd, gamma, beta, act, slope, groups, out=None):
    _chk(x, "x")
    y = out if out is not None else torch.empty_like(x)
    check(_lib.load().pcg_bn_apply_act_g(_p(x), x.numel() // C, C, _p(mean), _p(invstd), _p(gamma), _p(beta), int(act), float(slope), _p(y),
                                         int(groups), _stream()), "pcg_bn_apply_act_g")
    return y


def conv_bwd_data_fused_g(g, d, w, below_act, below_slope, z_below, bn, groups):
    """Grouped pcg_conv2d_dgrad_bnbwd: returns (dm, partial, nparts, nphases) for bn_bwd_partial_g."""
    lib = _lib.load()
    _chk(d, "d"); _chk(w, "w"); _chk(z_below, "z_below")
    out = torch.empty((g.B, g.IH, g.IW, g.Cin), dtype=torch.float32, device=d.device)
    assert z_below.numel() == out.numel()
    mean, invstd, gamma, beta = bn
    need = lib.pcg_conv2d_dgrad_bn_workspace_bytes(ctypes.byref(g))
    nparts = lib.pcg_conv2d_dgrad_bn_partial_rows(ctypes.byref(g))
    nphases = lib.pcg_conv2d_dgrad_bn_phases(ctypes.byref(g))
    partial = torch.empty(need // 4, dtype=torch.float32, device=d.device)
    with _Timed(g, "dgrad"):
        check(lib.pcg_conv2d_dgrad_bnbwd_g(ctypes.byref(g), _p(d), _p(w), _p(z_below), _p(mean), _p(invstd), _p(gamma), _p(beta), int(below_act),
                                           float(below_slope), _p(out), _p(partial), need, int(groups), _stream()), "pcg_conv2d_dgrad_bnbwd_g")
    return out, partial, nparts, nphases


def bn_bwd_partial_g(dm, x, C, mean, invstd, gamma, partial, nparts, nphases, dgamma, dbeta, accumulate, groups, out=None):
    _chk(dm, "dm"); _chk(x, "x")
    dx = out if out is not None else torch.empty_like(x)
    lib = _lib.load()
    ws = workspace(lib.pcg_bn_bwd_partial_g_workspace_bytes(C, int(groups)), x.device)
    check(lib.pcg_bn_bwd_partial_g(_p(dm), _p(x), x.numel() // C, C, _p(mean), _p(invstd), _p(gamma), _p(partial), int(nparts), int(nphases),
                                   _p(dx), _p(dgamma), _p(dbeta), int(bool(accumulate)), int(groups), _p(ws), ws.numel(), _stream()),
          "pcg_bn_bwd_partial_g")
    return dx


def bn_act_bwd_g(dy, x, C, mean, invstd, gamma, beta, act, slope, dgamma, dbeta, accumulate, groups, out=None):
    """BatchNorm(train) + ReLU / LeakyReLU backward of `groups` side-by-side batches (mask recomputed from x)."""
    _chk(dy, "dy"); _chk(x, "x")
    dx = out if out is not None else torch.empty_like(x)
    lib = _lib.load()
    rows = x.numel() // C
    ws = workspace(lib.pcg_bn_act_bwd_g_workspace_bytes(rows, C, int(groups)), x.device)
    check(lib.pcg_bn_act_bwd_premask_g(_p(dy), _p(x), rows, C, _p(mean), _p(invstd), _p(gamma), _p(beta), int(act), float(slope), _p(dx),
                                       _p(dgamma), _p(dbeta), int(bool(accumulate)), int(groups), _p(ws), ws.numel(), _stream()),
          "pcg_bn_act_bwd_premask_g")
    return dx


def full_dgrad_bn_bwd_ok(g, groups=1):
    return bool(_lib.load().pcg_conv2d_dgrad_bnbwd_full_ok(ctypes.byref(g), int(groups)))


def full_dgrad_bn_bwd(g, dy, w, z_below, mean, invstd, gamma, beta, act, slope, dgamma, dbeta, accumulate, groups=1):
    """conv2d_dgrad(g, dy, w) of a full-window Cout = 1 layer pushed through the BatchNorm + activation backward of the layer below
    without being written (pcg_conv2d_dgrad_bnbwd_full).  Returns dz [B, IH, IW, Cin]; mean / invstd [groups][C]."""
    _chk(dy, "dy"); _chk(w, "w"); _chk(z_below, "z_below")
    assert z_below.numel() == g.B * g.IH * g.IW * g.Cin and dy.numel() == g.B
    dz = torch.empty_like(z_below)
    lib = _lib.load()
    ws = workspace(lib.pcg_conv2d_dgrad_bnbwd_full_workspace_bytes(ctypes.byref(g), int(groups)), dy.device)
    check(lib.pcg_conv2d_dgrad_bnbwd_full(ctypes.byref(g), _p(dy), _p(w), _p(z_below), _p(mean), _p(invstd), _p(gamma), _p(beta), int(act),
                                          float(slope), _p(dz), _p(dgamma), _p(dbeta), int(bool(accumulate)), int(groups), _p(ws), ws.numel(),
                                          _stream()), "pcg_conv2d_dgrad_bnbwd_full")
    return dz


def thin_fwd_bn_bwd_ok(g):
    return bool(_lib.load().pcg_conv2d_fwd_bnbwd_thin_ok(ctypes.byref(g)))


def thin_fwd_bn_bwd(g, x, w, z_below, mean, invstd, gamma, beta, act, slope, dgamma, dbeta, accumulate):
    """conv2d_fwd(g, x, w) with Cin = 1 taken as the gradient w.r.t. act(bn(z_below)) and pushed through that BatchNorm + activation
    backward without being written (pcg_conv2d_fwd_bnbwd_thin).  Returns dz [B, OH, OW, Cout]."""
    _chk(x, "x"); _chk(w, "w"); _chk(z_below, "z_below")
    assert z_below.numel() == g.B * g.OH * g.OW * g.Cout
    dz = torch.empty_like(z_below)
    lib = _lib.load()
    ws = workspace(lib.pcg_conv2d_fwd_bnbwd_thin_workspace_bytes(ctypes.byref(g)), x.device)
    check(lib.pcg_conv2d_fwd_bnbwd_thin(ctypes.byref(g), _p(x), _p(w), _p(z_below), _p(mean), _p(invstd), _p(gamma), _p(beta), int(act),
                                        float(slope), _p(dz), _p(dgamma), _p(dbeta), int(bool(accumulate)), _p(ws), ws.numel(), _stream()),
          "pcg_conv2d_fwd_bnbwd_thin")
    return dz


# ---- optimizer / helpers ------------------------------------------------------------------------------
def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, decoupled, step):
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, n)
    check(_lib.load().pcg_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, beta1, beta2, eps,
                                    weight_decay, int(bool(decoupled)), int(step), _stream()), "pcg_adam_step")


def adam_step_capturable(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, decoupled, step_dev, hyper_dev):
    check(_lib.load().pcg_adam_step_capturable(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, beta1, beta2,
                                               eps, weight_decay, int(bool(decoupled)), _p(step_dev), _p(hyper_dev), _stream()),
          "pcg_adam_step_capturable")


def fill(t, value):
    _chk(t, "t")
    if t.numel():
        check(_lib.load().pcg_fill(_p(t), t.numel(), float(value), _stream()), "pcg_fill")
    return t


def add_bias_rows(x, C, bias):
    """x[..., c] += bias[c] in place."""
    _chk(x, "x"); _chk(bias, "bias")
    check(_lib.load().pcg_add_bias_rows(_p(x), x.numel() // C, C, _p(bias), _stream()), "pcg_add_bias_rows")
    return x


def sumsq(t, out, accumulate=False):
    _chk(t, "t")
    check(_lib.load().pcg_sumsq(_p(t), t.numel(), _p(out), int(bool(accumulate)), _stream()), "pcg_sumsq")
    return out


# ---- CounteRGAN pieces ---------------------------------------------------------------------------------------
def norm_sum(flat, seg, out=None):
    """out[0] = sum_s ||flat[seg[s,0] : seg[s,0] + seg[s,1]]||_2 (seg: int64 [n, 2] on the device) in one launch."""
    _chk(flat, "flat"); _chk(seg, "seg", torch.int64)
    out = out if out is not None else torch.empty(1, dtype=torch.float32, device=flat.device)
    check(_lib.load().pcg_norm_sum(_p(flat), _p(seg), seg.shape[0], _p(out), _stream()), "pcg_norm_sum")
    return out


def _chk_idx(idx, K, name="idx"):
    if idx.dtype != torch.int64 or not idx.is_cuda or not idx.is_contiguous():
        raise _lib.PcgError(f"{name}: expected a contiguous int64 tensor on the GPU")
    return idx


def embed_concat_fwd(x, idx, table, mask):
    """[B,H,W,C] with channels (x, table[idx], [mask]); x, mask: [B,1,H,W] or [B,H,W]; table: [K, H*W]."""
    _chk(x, "x"); _chk(table, "table"); _chk_idx(idx, table.shape[0])
    B, HW = idx.numel(), table.shape[1]
    C = 3 if mask is not None else 2
    out = torch.empty((B, HW, C), dtype=torch.float32, device=x.device)
    check(_lib.load().pcg_embed_concat_fwd(_p(x), _p(idx), _p(table), _p(mask), _p(out), B, HW, C, table.shape[0], _stream()),
          "pcg_embed_concat_fwd")
    return out


def embed_concat_bwd(dinp, idx, C, K, dtable=None, accumulate=False, need_dx=False):
    _chk(dinp, "dinp")
    B = idx.numel()
    HW = dinp.numel() // (B * C)
    dx = torch.empty((B, HW), dtype=torch.float32, device=dinp.device) if need_dx else None
    check(_lib.load().pcg_embed_concat_bwd(_p(dinp), _p(idx), _p(dtable), _p(dx), B, HW, C, K, int(bool(accumulate)), _stream()),
          "pcg_embed_concat_bwd")
    return dx


def embed_table_grad(dinp, idx, C, ch, K, dtable, accumulate=False):
    """dtable[k][p] (+)= sum over the rows b with idx[b] == k of dinp[b][p][ch] (dinp: [B, HW, C])."""
    _chk(dinp, "dinp"); _chk(dtable, "dtable")
    B = idx.numel()
    HW = dinp.numel() // (B * C)
    check(_lib.load().pcg_embed_table_grad(_p(dinp), _p(idx), _p(dtable), B, HW, C, int(ch), K, int(bool(accumulate)), _stream()),
          "pcg_embed_table_grad")


def gather_channel(t, ch):
    """[..., C] -> [..., 1]: channel `ch` of the last axis as a contiguous tensor (one tiny launch)."""
    _chk(t, "t")
    C = t.shape[-1]
    out = torch.empty(t.shape[:-1] + (1,), dtype=torch.float32, device=t.device)
    check(_lib.load().pcg_gather_channel(_p(t), _p(out), t.numel() // C, C, int(ch), _stream()), "pcg_gather_channel")
    return out


def axpby(a, x, b=0.0, y=None, out=None):
    _chk(x, "x")
    out = out if out is not None else torch.empty_like(x)
    check(_lib.load().pcg_axpby(_p(out), float(a), _p(x), float(b), _p(y), x.numel(), _stream()), "pcg_axpby")
    return out


def scale_mask_fwd(c, mask, scale):
    _chk(c, "c")
    raw, masked = torch.empty_like(c), torch.empty_like(c)
    check(_lib.load().pcg_scale_mask_fwd(_p(c), _p(mask), float(scale), _p(raw), _p(masked), c.numel(), _stream()), "pcg_scale_mask_fwd")
    return raw, masked


def scale_mask_bwd(d_raw, d_masked, mask, scale, like):
    dc = torch.empty_like(like)
    check(_lib.load().pcg_scale_mask_bwd(_p(d_raw), _p(d_masked), _p(mask), float(scale), _p(dc), like.numel(), _stream()),
          "pcg_scale_mask_bwd")
    return dc


def clamp_add_fwd(x, r, lo, hi):
    _chk(x, "x"); _chk(r, "r")
    y = torch.empty_like(x)
    check(_lib.load().pcg_clamp_add_fwd(_p(x), _p(r), float(lo), float(hi), _p(y), x.numel(), _stream()), "pcg_clamp_add_fwd")
    return y


def clamp_add_bwd(dy, x, r, lo, hi):
    _chk(dy, "dy")
    dr = torch.empty_like(x)
    check(_lib.load().pcg_clamp_add_bwd(_p(dy), _p(x), _p(r), float(lo), float(hi), _p(dr), x.numel(), _stream()), "pcg_clamp_add_bwd")
    return dr


def abs_mean_fwd(a, m=None, one_minus=False):
    _chk(a, "a")
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    ws = workspace(lib.pcg_abs_mean_workspace_bytes(), a.device)
    check(lib.pcg_abs_mean_fwd(_p(a), _p(m), int(bool(one_minus)), a.numel(), _p(out), _p(ws), ws.numel(), _stream()), "pcg_abs_mean_fwd")
    return out


def abs_mean_bwd(a, m, one_minus, grad_out, scale=1.0, out=None, accumulate=False):
    da = out if out is not None else torch.empty_like(a)
    check(_lib.load().pcg_abs_mean_bwd(_p(a), _p(m), int(bool(one_minus)), a.numel(), _p(grad_out), float(scale), _p(da),
                                       int(bool(accumulate)), _stream()), "pcg_abs_mean_bwd")
    return da


def avgpool_fwd(x, B, HW, C):
    _chk(x, "x")
    y = torch.empty((B, C), dtype=torch.float32, device=x.device)
    check(_lib.load().pcg_avgpool_fwd(_p(x), _p(y), B, HW, C, _stream()), "pcg_avgpool_fwd")
    return y


def avgpool_bwd(dy, B, HW, C):
    _chk(dy, "dy")
    dx = torch.empty((B, HW, C), dtype=torch.float32, device=dy.device)
    check(_lib.load().pcg_avgpool_bwd(_p(dy), _p(dx), B, HW, C, _stream()), "pcg_avgpool_bwd")
    return dx


def cross_entropy_fwd_bwd(logits, target, need_loss=True, need_grad=True, grad_out=None, grad_scale=1.0):
    _chk(logits, "logits"); _chk_idx(target, logits.shape[-1], "target")
    B, K = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device) if need_loss else None
    dz = torch.empty_like(logits) if need_grad else None
    check(_lib.load().pcg_cross_entropy_fwd_bwd(_p(logits), _p(target), B, K, float(grad_scale), _p(grad_out), _p(loss), _p(dz),
                                                _stream()), "pcg_cross_entropy_fwd_bwd")
    return loss, dz


# ---- tabular CounteRGAN building blocks (csrc/tabular.hip) ----------------------------------------------------------------
def gemm(A, B, M, N, K, transA=False, transB=False, lda=None, ldb=None, out=None, ldc=None, bias=None, accumulate=False, act=0, slope=0.0):
    """out[M][N] (+)= opA . opB (+ bias); A/B/out may be column slices of wider row-major buffers (ld* = row stride)."""
    for t, n in ((A, "A"), (B, "B")):
        if not t.is_cuda or t.dtype != torch.float32:
            raise _lib.PcgError(f"gemm: {n} must be an fp32 tensor on the GPU")
    lda = lda if lda is not None else (M if transA else K)
    ldb = ldb if ldb is not None else (K if transB else N)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    ldc = ldc if ldc is not None else N
    check(_lib.load().pcg_gemm_act(int(transA), int(transB), M, N, K, _p(A), lda, _p(B), ldb, _p(out), ldc, _p(bias), int(bool(accumulate)),
                                   int(act), float(slope), _stream()), "pcg_gemm_act")
    return out


def linear_wgrad(dy, x, B, O, I, dW, db=None, ldy=None, ldx=None, accumulate_w=False, accumulate_b=False):
    """dW[O][I] (+)= dy^T x and db[O] (+)= colsum(dy) in one launch (deterministic slab split over the batch)."""
    lib = _lib.load()
    dev = x.device
    tk = _ticket_buffer(dev)
    nbytes = lib.pcg_linear_wgrad_workspace_bytes(B, O, I)
    ws = workspace2(nbytes, dev) if nbytes else None
    check(lib.pcg_linear_wgrad(_p(dy), ldy if ldy is not None else O, _p(x), ldx if ldx is not None else I, B, O, I, _p(dW), _p(db),
                               int(bool(accumulate_w)), int(bool(accumulate_b)), _p(ws), nbytes, _p(tk), _stream()), "pcg_linear_wgrad")


def linear_wgrad_grouped(items, B, device):
    """items: list of (dy, x, O, I, dW, db or None, ldy, ldx, accumulate_w, accumulate_b) — layers that reduce over the same B rows,
    all in one launch (the library tiles them for the matrix cores)."""
    lib = _lib.load()
    n = len(items)
    arr = (_lib.WgradItem * n)()
    for k, (dy, x, O, I, dW, db, ldy, ldx, aw, ab) in enumerate(items):
        arr[k].dy, arr[k].x, arr[k].dW, arr[k].db = dy.data_ptr(), x.data_ptr(), dW.data_ptr(), (db.data_ptr() if db is not None else None)
        arr[k].ldy, arr[k].ldx, arr[k].O, arr[k].I = ldy, ldx, O, I
        arr[k].accumulate_w, arr[k].accumulate_b = int(bool(aw)), int(bool(ab))
        arr[k].tile_x, arr[k].tile_y = 0, 0
    tk = _ticket_buffer(device)
    nbytes = lib.pcg_linear_wgrad_grouped_workspace_bytes(B, arr, n)
    ws = workspace2(nbytes, device)
    check(lib.pcg_linear_wgrad_grouped(arr, n, B, _p(ws), nbytes, _p(tk), _stream()), "pcg_linear_wgrad_grouped")


def onehot(idx, K, out=None):
    _chk_idx(idx, K)
    if out is None:
        out = torch.empty((idx.numel(), K), dtype=torch.float32, device=idx.device)
    check(_lib.load().pcg_onehot(_p(idx), idx.numel(), K, _p(out), _stream()), "pcg_onehot")
    return out


def concat_cols(a, b):
    _chk(a, "a"); _chk(b, "b")
    rows = a.shape[0]
    out = torch.empty((rows, a.shape[1] + b.shape[1]), dtype=torch.float32, device=a.device)
    check(_lib.load().pcg_concat_cols(_p(a), a.shape[1], _p(b), b.shape[1], rows, _p(out), _stream()), "pcg_concat_cols")
    return out


def split_cols(d, ca, cb, need_a=True, need_b=True):
    _chk(d, "d")
    rows = d.shape[0]
    da = torch.empty((rows, ca), dtype=torch.float32, device=d.device) if need_a else None
    db = torch.empty((rows, cb), dtype=torch.float32, device=d.device) if need_b else None
    check(_lib.load().pcg_split_cols(_p(d), ca, cb, rows, _p(da), _p(db), _stream()), "pcg_split_cols")
    return da, db


def film_fwd(g, h, b):
    _chk(g, "gamma"); _chk(h, "h"); _chk(b, "beta")
    y = torch.empty_like(h)
    check(_lib.load().pcg_film_fwd(_p(g), _p(h), _p(b), _p(y), h.numel(), _stream()), "pcg_film_fwd")
    return y


def film_bwd(dy, g, h):
    _chk(dy, "dy")
    dg, dh = torch.empty_like(h), torch.empty_like(h)
    check(_lib.load().pcg_film_bwd(_p(dy), _p(g), _p(h), _p(dg), _p(dh), h.numel(), _stream()), "pcg_film_bwd")
    return dg, dh


def gumbel_softmax_fwd(logits, noise, seg_offsets, tau, hard=False):
    """(y_soft, y_hard or None)"""
    _chk(logits, "logits"); _chk(noise, "noise"); _chk(seg_offsets, "seg_offsets", torch.int32)
    B, T = logits.shape
    y = torch.empty_like(logits)
    yh = torch.empty_like(logits) if hard else None
    check(_lib.load().pcg_gumbel_softmax_fwd(_p(logits), _p(noise), _p(seg_offsets), seg_offsets.numel() - 1, T, B, float(tau), _p(y),
                                             _p(yh), _stream()), "pcg_gumbel_softmax_fwd")
    return y, yh


def gumbel_softmax_bwd(dy, y, seg_offsets, tau):
    _chk(dy, "dy")
    B, T = y.shape
    dl = torch.empty_like(y)
    check(_lib.load().pcg_gumbel_softmax_bwd(_p(dy), _p(y), _p(seg_offsets), seg_offsets.numel() - 1, T, B, float(tau), _p(dl),
                                             _stream()), "pcg_gumbel_softmax_bwd")
    return dl


def assemble_residual_fwd(cont, cont_idx, samples, seg_offsets, cat_idx, norm_vals, x):
    _chk(cont, "cont"); _chk(samples, "samples"); _chk(x, "x"); _chk(norm_vals, "norm_vals")
    B, D = x.shape
    res = torch.empty_like(x)
    check(_lib.load().pcg_assemble_residual_fwd(_p(cont), cont.shape[1], _p(cont_idx), _p(samples), _p(seg_offsets), cat_idx.numel(),
                                                samples.shape[1], _p(cat_idx), _p(norm_vals), _p(x), D, B, _p(res), _stream()),
          "pcg_assemble_residual_fwd")
    return res


def house_residual_fwd(cont, samples, seg_offsets, norm_vals, x, mask, col_src, sn=None):
    """(residual_full, masked_residual, x_cf, mask_penalty, am) of the tabular step in one launch: assemble_residual_fwd +
    scale_mask_fwd + axpby + 2 x abs_mean_fwd, bit for bit.  col_src: host list, per feature column the continuous index (>= 0) or
    -(head + 1).  sn = (w_origs, us, vs, eps, reps): spectral_norm_fwd_batched_reps of these matrices rides in the same launch (the
    two do not depend on each other); its result is appended to the returned tuple."""
    import ctypes
    _chk(cont, "cont"); _chk(samples, "samples"); _chk(x, "x"); _chk(mask, "mask"); _chk(norm_vals, "norm_vals")
    B, D = x.shape
    res, masked, x_cf = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    scal = torch.empty(2, dtype=torch.float32, device=x.device)
    part = torch.empty(512, dtype=torch.float32, device=x.device)
    tk = _ticket_buffer(x.device)
    src = (ctypes.c_int32 * D)(*[int(v) for v in col_src])
    args = (_p(cont), cont.shape[1], _p(samples), _p(seg_offsets), samples.shape[1], _p(norm_vals), _p(x), _p(mask), src, D, B, _p(res),
            _p(masked), _p(x_cf), _p(part), tk.data_ptr() + 4 * 1024, _p(scal), scal.data_ptr() + 4)
    if sn is None:
        check(_lib.load().pcg_house_residual_fwd(*args, _stream()), "pcg_house_residual_fwd")
        return res, masked, x_cf, scal[0], scal[1]
    outs, sn_args = sn_fwd_reps_args(*sn)
    check(_lib.load().pcg_house_residual_fwd_sn(*args, *sn_args, _stream()), "pcg_house_residual_fwd_sn")
    return res, masked, x_cf, scal[0], scal[1], outs


def house_residual_bwd(res, masked, mask, gx_a, gx_b, w_pen, w_am, ncont, cont_idx, seg_offsets, T, cat_idx, norm_vals, losses=None, diag=None):
    """(d_cont, d_samples): the tabular step's backward from dLoss/dx_cf = gx_a + gx_b and the two penalty weights down to the
    generator's outputs, in one launch (see pcg_house_residual_bwd).  losses = (d_real, d_fake, d_fake_g, g_cls, am, pen, lambda_cls,
    w_reg, lambda_mask, w_reg_log): pcg_house_losses rides in the same launch; its scalars (six floats: the five of pcg_house_losses
    and g_cls) are appended to the returned tuple.  g_cls: the cross-entropy value (a device scalar), or the [B] row terms the fused
    classifier forward left (then their mean is formed here, in the cross-entropy kernel's order).
    diag (with losses) = (logits_cf, logits_orig, src_rows or None, target_y, eps, acc or None): the trainer's four per-iteration
    diagnostics ride in the launch too (house_diag); their [4] tensor is appended after the scalars."""
    for t, nme in ((res, "res"), (masked, "masked"), (mask, "mask"), (gx_a, "gx_a"), (gx_b, "gx_b")):
        _chk(t, nme)
    B, D = res.shape
    dcont = torch.empty((B, ncont), dtype=torch.float32, device=res.device)
    dsamples = torch.empty((B, T), dtype=torch.float32, device=res.device)
    args = (_p(res), _p(masked), _p(mask), _p(gx_a), _p(gx_b), float(w_pen), float(w_am), ncont, _p(cont_idx), _p(seg_offsets), cat_idx.numel(),
            T, _p(cat_idx), _p(norm_vals), D, B, _p(dcont), _p(dsamples))
    if losses is None:
        check(_lib.load().pcg_house_residual_bwd(*args, _stream()), "pcg_house_residual_bwd")
        return dcont, dsamples
    d_real, d_fake, d_fake_g, g_cls, am, pen, l_cls, w_reg, l_mask, w_reg_log = losses
    out6 = torch.empty(6, dtype=torch.float32, device=res.device)
    rows = g_cls.numel() > 1
    if diag is not None:
        logits_cf, logits_orig, src_rows, target_y, eps, acc = diag
        _chk(logits_cf, "logits_cf"); _chk(logits_orig, "logits_orig"); _chk(target_y, "target_y", torch.int64)
        if src_rows is not None:
            _chk(src_rows, "src_rows", torch.int64)
        elif logits_orig.shape[0] < B:
            raise _lib.PcgError("house_residual_bwd(diag=...): logits_orig has fewer rows than the batch and no src_rows were given")
        if acc is not None:
            _chk(acc, "acc", torch.float64)
            assert acc.numel() >= 8
        nc = logits_cf.shape[1]
        assert logits_cf.shape[0] == B and logits_orig.shape[1] == nc and target_y.numel() == B
        out4 = torch.empty(4, dtype=torch.float32, device=res.device)
        check(_lib.load().pcg_house_residual_bwd_losses_diag(
            *args, _p(d_real), _p(d_fake), _p(d_fake_g), d_real.numel(), None if rows else _p(g_cls), _p(am), _p(pen), float(l_cls),
            float(w_reg), float(l_mask), float(w_reg_log), _p(g_cls) if rows else None, g_cls.numel() if rows else 0, _p(out6),
            _p(logits_cf), _p(logits_orig), _p(src_rows), _p(target_y), nc, float(eps), _p(out4), _p(acc), _stream()),
            "pcg_house_residual_bwd_losses_diag")
        return dcont, dsamples, out6, out4
    check(_lib.load().pcg_house_residual_bwd_losses(*args, _p(d_real), _p(d_fake), _p(d_fake_g), d_real.numel(), None if rows else _p(g_cls),
                                                    _p(am), _p(pen), float(l_cls), float(w_reg), float(l_mask), float(w_reg_log),
                                                    _p(g_cls) if rows else None, g_cls.numel() if rows else 0, _p(out6), _stream()),
          "pcg_house_residual_bwd_losses")
    return dcont, dsamples, out6


def house_diag(logits_cf, logits_orig, target_y, masked, eps=1e-3, src_rows=None, acc=None):
    """[pred_gain, sparsity, reg_loss_l2, class_flip_rate] of house_sales_kc_usa/trainer.py:318-343 in one launch (pcg_house_diag).
    logits_orig: the frozen classifier on the original rows ([B, nc], or [N, nc] for the whole training set with src_rows [B]);
    acc: float64[8] epoch accumulators (acc[2..5] += the four values)."""
    _chk(logits_cf, "logits_cf"); _chk(logits_orig, "logits_orig"); _chk(masked, "masked"); _chk(target_y, "target_y", torch.int64)
    B, nc = logits_cf.shape
    if src_rows is not None:
        _chk(src_rows, "src_rows", torch.int64)
    elif logits_orig.shape[0] < B:
        raise _lib.PcgError("house_diag: logits_orig has fewer rows than the batch and no src_rows were given")
    if acc is not None:
        _chk(acc, "acc", torch.float64)
    out = torch.empty(4, dtype=torch.float32, device=logits_cf.device)
    check(_lib.load().pcg_house_diag(_p(logits_cf), _p(logits_orig), _p(src_rows), _p(target_y), _p(masked), B, nc, masked.shape[1], float(eps),
                                     _p(out), _p(acc), _stream()), "pcg_house_diag")
    return out


def assemble_residual_bwd(dres, ncont, cont_idx, seg_offsets, T, cat_idx, norm_vals):
    _chk(dres, "dres")
    B, D = dres.shape
    dcont = torch.empty((B, ncont), dtype=torch.float32, device=dres.device)
    dsamples = torch.empty((B, T), dtype=torch.float32, device=dres.device)
    check(_lib.load().pcg_assemble_residual_bwd(_p(dres), ncont, _p(cont_idx), _p(seg_offsets), cat_idx.numel(), T, _p(cat_idx),
                                                _p(norm_vals), D, B, _p(dcont), _p(dsamples), _stream()), "pcg_assemble_residual_bwd")
    return dcont, dsamples


def mean_fwd(x):
    _chk(x, "x")
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = workspace(lib.pcg_mean_workspace_bytes(), x.device)
    check(lib.pcg_mean_fwd(_p(x), x.numel(), _p(out), _p(ws), ws.numel(), _stream()), "pcg_mean_fwd")
    return out


def mean_bwd(grad_out, scale, like):
    dx = torch.empty_like(like)
    check(_lib.load().pcg_mean_bwd(_p(grad_out), float(scale), like.numel(), _p(dx), _stream()), "pcg_mean_bwd")
    return dx


def spectral_norm_fwd(w_orig, u, v, eps, power_iteration):
    _chk(w_orig, "weight_orig"); _chk(u, "weight_u"); _chk(v, "weight_v")
    O, I = w_orig.shape
    w_bar = torch.empty_like(w_orig)
    sigma = torch.empty(1, dtype=torch.float32, device=w_orig.device)
    uu, vu = torch.empty_like(u), torch.empty_like(v)
    check(_lib.load().pcg_spectral_norm_fwd(_p(w_orig), O, I, _p(u), _p(v), float(eps), int(bool(power_iteration)), _p(w_bar), _p(sigma),
                                            _p(uu), _p(vu), _stream()), "pcg_spectral_norm_fwd")
    return w_bar, sigma, uu, vu


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def spectral_norm_fwd_batched(w_origs, us, vs, eps, power_iteration):
    """[(w_bar, sigma, u_used, v_used)] for all layers in one launch."""
    n = len(w_origs)
    outs = [(torch.empty_like(w), torch.empty(1, dtype=torch.float32, device=w.device), torch.empty_like(u), torch.empty_like(v))
            for w, u, v in zip(w_origs, us, vs)]
    I32 = ctypes.c_int32 * n
    check(_lib.load().pcg_spectral_norm_fwd_batched(n, _ptr_array(w_origs), I32(*[w.shape[0] for w in w_origs]), I32(*[w.shape[1] for w in w_origs]),
                                                    _ptr_array(us), _ptr_array(vs), float(eps), int(bool(power_iteration)),
                                                    _ptr_array([o[0] for o in outs]), _ptr_array([o[1] for o in outs]),
                                                    _ptr_array([o[2] for o in outs]), _ptr_array([o[3] for o in outs]), _stream()),
          "pcg_spectral_norm_fwd_batched")
    return outs


def _sn_outputs(w_origs, us, vs, reps):
    outs = [[(torch.empty_like(w), torch.empty(1, dtype=torch.float32, device=w.device), torch.empty_like(u), torch.empty_like(v))
             for w, u, v in zip(w_origs, us, vs)] for _ in range(reps)]
    return outs, [o for call in outs for o in call]


def spectral_norm_fwd_batched_reps(w_origs, us, vs, eps, reps):
    """`reps` successive training-mode calls in one launch: [[(w_bar, sigma, u_used, v_used)] per layer] per call."""
    n = len(w_origs)
    outs, flat = _sn_outputs(w_origs, us, vs, reps)
    I32 = ctypes.c_int32 * n
    check(_lib.load().pcg_spectral_norm_fwd_batched_reps(n, reps, _ptr_array(w_origs), I32(*[w.shape[0] for w in w_origs]),
                                                         I32(*[w.shape[1] for w in w_origs]), _ptr_array(us), _ptr_array(vs), float(eps), 1,
                                                         _ptr_array([o[0] for o in flat]), _ptr_array([o[1] for o in flat]),
                                                         _ptr_array([o[2] for o in flat]), _ptr_array([o[3] for o in flat]), _stream()),
          "pcg_spectral_norm_fwd_batched_reps")
    return outs


def sn_bwd_seq_args(passes, dw_origs, accumulate, bias_adds=None):
    """The argument list of pcg_spectral_norm_bwd_batched_seq (without the stream), for the launch itself or for a rider launch that
    carries it (pcg_house_classifier_fwd_snbwd).  The tensors must stay alive until the launch."""
    n = len(dw_origs)
    flat = [e for call in passes for e in call]
    I32 = ctypes.c_int32 * n
    null = ctypes.c_void_p * n
    dst = null(*[(b[0].data_ptr() if b is not None else None) for b in (bias_adds or [None] * n)])
    src = null(*[(b[1].data_ptr() if b is not None else None) for b in (bias_adds or [None] * n)])
    return (n, len(passes), _ptr_array([e[0] for e in flat]), _ptr_array([e[1] for e in flat]), I32(*[w.shape[0] for w in dw_origs]),
            I32(*[w.shape[1] for w in dw_origs]), _ptr_array([e[2] for e in flat]), _ptr_array([e[3] for e in flat]),
            _ptr_array([e[4] for e in flat]), _ptr_array(dw_origs), I32(*[int(bool(a)) for a in accumulate]), dst, src)


def sn_fwd_reps_args(w_origs, us, vs, eps, reps):
    """(outputs, argument list) of a training-mode pcg_spectral_norm_fwd_batched_reps as a rider launch passes them (no
    power_iteration flag, no stream)."""
    n = len(w_origs)
    outs, flat = _sn_outputs(w_origs, us, vs, reps)
    I32 = ctypes.c_int32 * n
    return outs, (n, reps, _ptr_array(w_origs), I32(*[w.shape[0] for w in w_origs]), I32(*[w.shape[1] for w in w_origs]), _ptr_array(us),
                  _ptr_array(vs), float(eps), _ptr_array([o[0] for o in flat]), _ptr_array([o[1] for o in flat]),
                  _ptr_array([o[2] for o in flat]), _ptr_array([o[3] for o in flat]))


def spectral_norm_bwd_batched_seq(passes, dw_origs, accumulate, bias_adds=None):
    """passes: [[(dw_bar, w_bar, u, v, sigma)] per layer] per call, applied in this order into dw_origs[l]; bias_adds: per layer
    (dst, src) or None — dst += src afterwards."""
    check(_lib.load().pcg_spectral_norm_bwd_batched_seq(*sn_bwd_seq_args(passes, dw_origs, accumulate, bias_adds), _stream()),
          "pcg_spectral_norm_bwd_batched_seq")


def spectral_norm_bwd_batched(items):
    """items: [(dw_bar, w_bar, u, v, sigma, dw_orig, accumulate)] — all layers in one launch."""
    n = len(items)
    I32 = ctypes.c_int32 * n
    check(_lib.load().pcg_spectral_norm_bwd_batched(n, _ptr_array([i[0] for i in items]), _ptr_array([i[1] for i in items]),
                                                    I32(*[i[1].shape[0] for i in items]), I32(*[i[1].shape[1] for i in items]),
                                                    _ptr_array([i[2] for i in items]), _ptr_array([i[3] for i in items]),
                                                    _ptr_array([i[4] for i in items]), _ptr_array([i[5] for i in items]),
                                                    I32(*[int(bool(i[6])) for i in items]), _stream()), "pcg_spectral_norm_bwd_batched")


def spectral_norm_bwd(dw_bar, w_bar, u, v, sigma, dw_orig, accumulate):
    O, I = w_bar.shape
    check(_lib.load().pcg_spectral_norm_bwd(_p(dw_bar), _p(w_bar), O, I, _p(u), _p(v), _p(sigma), _p(dw_orig), int(bool(accumulate)),
                                            _stream()), "pcg_spectral_norm_bwd")


def cross_entropy_weighted_fwd_bwd(logits, target, class_weight, need_loss=True, need_grad=True, grad_out=None, grad_scale=1.0):
    _chk(logits, "logits"); _chk_idx(target, logits.shape[-1], "target"); _chk(class_weight, "class_weight")
    B, K = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device) if need_loss else None
    dz = torch.empty_like(logits) if need_grad else None
    check(_lib.load().pcg_cross_entropy_weighted_fwd_bwd(_p(logits), _p(target), _p(class_weight), B, K, float(grad_scale), _p(grad_out),
                                                         _p(loss), _p(dz), _stream()), "pcg_cross_entropy_weighted_fwd_bwd")
    return loss, dz


def dropout_apply(x, mask, p, inner=1, C=None, out=None):
    """y = x * mask / (1 - p); mask one entry per element (inner=1) or per (sample, channel) of an NHWC activation (inner=HW)."""
    _chk(x, "x"); _chk(mask, "mask")
    C = C if C is not None else x.shape[-1]
    y = out if out is not None else torch.empty_like(x)
    check(_lib.load().pcg_dropout_apply(_p(x), _p(mask), x.numel(), inner, C, 1.0 / (1.0 - p), _p(y), _stream()), "pcg_dropout_apply")
    return y


def weighted_sum_fwd(terms, weights):
    n = len(terms)
    out = torch.empty(1, dtype=torch.float32, device=terms[0].device)
    check(_lib.load().pcg_weighted_sum_fwd(n, _ptr_array(terms), (ctypes.c_float * n)(*[float(w) for w in weights]), _p(out), _stream()),
          "pcg_weighted_sum_fwd")
    return out


def weighted_sum_bwd(weights, grad_out, needs):
    """[w_i * grad_out as a one-element tensor, or None where needs[i] is False]"""
    n = len(weights)
    dev = grad_out.device
    outs = [torch.empty(1, dtype=torch.float32, device=dev) if need else None for need in needs]
    ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() if o is not None else None for o in outs])
    check(_lib.load().pcg_weighted_sum_bwd(n, (ctypes.c_float * n)(*[float(w) for w in weights]), _p(grad_out), ptrs, _stream()),
          "pcg_weighted_sum_bwd")
    return outs


def cf_metrics(logits_cf, target, logits_ref=None, other=None):
    """[class-flip rate, prediction gain] as a 2-element device tensor (see pcg_cf_metrics)."""
    _chk(logits_cf, "logits_cf"); _chk_idx(target, logits_cf.shape[1], "target")
    B, K = logits_cf.shape
    out = torch.empty(2, dtype=torch.float32, device=logits_cf.device)
    check(_lib.load().pcg_cf_metrics(_p(logits_cf), _p(logits_ref), _p(target), _p(other), B, K, _p(out), _stream()), "pcg_cf_metrics")
    return out


# ---- WGAN-GP critic pieces (csrc/instnorm.hip) ----------------------------------------------------------------------------
def instnorm_fwd(x, B, HW, C, gamma, beta, eps, act=0, slope=0.0):
    _chk(x, "x")
    y = torch.empty_like(x)
    mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(B * C, dtype=torch.float32, device=x.device)
    check(_lib.load().pcg_instnorm_fwd(_p(x), B, HW, C, _p(gamma), _p(beta), float(eps), act, float(slope), _p(y), _p(mean), _p(invstd),
                                       _stream()), "pcg_instnorm_fwd")
    return y, mean, invstd


def instnorm_bwd(dy, x, B, HW, C, mean, invstd, gamma, need_dx=True, need_params=True):
    """(dx, dgamma_partial [B,C], dbeta_partial [B,C])"""
    _chk(dy, "dy"); _chk(x, "x")
    dx = torch.empty_like(x) if need_dx else None
    dgp = torch.empty((B, C), dtype=torch.float32, device=x.device) if need_params else None
    dbp = torch.empty((B, C), dtype=torch.float32, device=x.device) if need_params else None
    check(_lib.load().pcg_instnorm_bwd(_p(dy), _p(x), B, HW, C, _p(mean), _p(invstd), _p(gamma), _p(dx), _p(dgp), _p(dbp), _stream()),
          "pcg_instnorm_bwd")
    return dx, dgp, dbp


def instnorm_bwd_fused(dy, x, B, HW, C, mean, invstd, gamma, act_y=None, slope=0.0, keep_dn=False, addend=None, need_params=True,
                       need_dxsum=False):
    """LeakyReLU' -> InstanceNorm' (-> + addend) of a critic stage in one launch: (dx, dn or None, dgamma_partial, dbeta_partial,
    dxsum_partial), partials [B, C] (rowsum3 reduces them over B); see pcg_instnorm_bwd_fused in include/pcgan_hip.h."""
    _chk(dy, "dy"); _chk(x, "x")
    dx = torch.empty_like(x)
    dn = torch.empty_like(x) if (keep_dn and act_y is not None) else None
    mk = lambda on: torch.empty((B, C), dtype=torch.float32, device=x.device) if on else None
    dgp, dbp, dsp = mk(need_params), mk(need_params), mk(need_dxsum)
    check(_lib.load().pcg_instnorm_bwd_fused(_p(dy), _p(act_y), float(slope), _p(x), B, HW, C, _p(mean), _p(invstd), _p(gamma), _p(dn), _p(addend),
                                             _p(dx), _p(dgp), _p(dbp), _p(dsp), _stream()), "pcg_instnorm_bwd_fused")
    return dx, dn, dgp, dbp, dsp


def rowsum3(items, rows, C):
    """items: up to three (src [rows, C], dst [C], accumulate) — dst (+)= column sums of src, all in one launch."""
    assert 1 <= len(items) <= 3
    a = []
    for k in range(3):
        src, dst, acc = items[k] if k < len(items) else (None, None, False)
        a += [_p(src), _p(dst), int(bool(acc))]
    check(_lib.load().pcg_rowsum3(len(items), *a, int(rows), int(C), _stream()), "pcg_rowsum3")


def instnorm_bwd_bwd(r, dy, x, B, HW, C, mean, invstd, gamma, need_ddy=True, need_ez=True, need_gamma=True, act_y=None, slope=0.0):
    """(ddy, ez, dgamma_partial [B,C]); act_y: ddy continues through the stage's LeakyReLU (its output) in the same launch"""
    _chk(r, "r"); _chk(dy, "dy"); _chk(x, "x")
    ddy = torch.empty_like(x) if need_ddy else None
    ez = torch.empty_like(x) if need_ez else None
    dgp = torch.empty((B, C), dtype=torch.float32, device=x.device) if need_gamma else None
    check(_lib.load().pcg_instnorm_bwd_bwd_act(_p(r), _p(dy), _p(x), B, HW, C, _p(mean), _p(invstd), _p(gamma), _p(act_y), float(slope), _p(ddy),
                                               _p(ez), _p(dgp), _stream()), "pcg_instnorm_bwd_bwd_act")
    return ddy, ez, dgp


def nhwc_to_nchw_flat(src, B, HW, C, inverse=False):
    _chk(src, "src")
    dst = torch.empty_like(src)
    check(_lib.load().pcg_nhwc_to_nchw_flat(_p(src), _p(dst), B, HW, C, int(bool(inverse)), _stream()), "pcg_nhwc_to_nchw_flat")
    return dst


def interpolate(alpha, real, fake):
    _chk(alpha, "alpha"); _chk(real, "real"); _chk(fake, "fake")
    B = alpha.numel()
    out = torch.empty_like(real)
    check(_lib.load().pcg_interpolate(_p(alpha), _p(real), _p(fake), _p(out), B, real.numel() // B, _stream()), "pcg_interpolate")
    return out


def interpolate_stack(alpha, real, fake):
    """[real | fake | alpha*real + (1-alpha)*fake] as one [3B, ...] tensor, one launch (the batched critic pass's input)."""
    _chk(alpha, "alpha"); _chk(real, "real"); _chk(fake, "fake")
    B = alpha.numel()
    assert real.shape == fake.shape and real.shape[0] == B
    x3 = torch.empty((3 * B,) + tuple(real.shape[1:]), dtype=torch.float32, device=real.device)
    check(_lib.load().pcg_interpolate_stack(_p(alpha), _p(real), _p(fake), _p(x3), B, real.numel() // B, _stream()), "pcg_interpolate_stack")
    return x3


def gradient_penalty_fwd(grads, B, lam):
    _chk(grads, "grads")
    norms = torch.empty(B, dtype=torch.float32, device=grads.device)
    pen = torch.empty(1, dtype=torch.float32, device=grads.device)
    check(_lib.load().pcg_gradient_penalty_fwd(_p(grads), B, grads.numel() // B, float(lam), _p(norms), _p(pen), _stream()),
          "pcg_gradient_penalty_fwd")
    return pen, norms


def gradient_penalty_bwd(grads, norms, grad_out, B, lam):
    dg = torch.empty_like(grads)
    check(_lib.load().pcg_gradient_penalty_bwd(_p(grads), _p(norms), _p(grad_out), B, grads.numel() // B, float(lam), _p(dg), _stream()),
          "pcg_gradient_penalty_bwd")
    return dg


# ---- device-side batch synthesis ---------------------------------------------------------------------------------
class DeviceRNG:
    """Counter-based stream: every draw advances `offset`, so a (seed, call sequence) pair is reproducible."""

    def __init__(self, seed=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0

    def _advance(self, n):
        # the Philox offset is a kernel ARGUMENT held on the host: a draw captured into a HIP graph would replay the same numbers
        # on every launch.  Draw outside the captured step and feed the result in as a static input (GraphedStep.load), or use the
        # device-counter form of the launch where there is one (house_draws(counter=...): the launch reads and advances the offset itself).
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise _lib.PcgError("DeviceRNG draw during HIP-graph capture: the captured kernel would replay identical random numbers; "
                                "draw before the step and pass the tensors in (GraphedStep inputs)")
        off = self.offset
        self.offset += int(n)
        return off

    def patch_mask(self, B, H, W, patch_size, num_selected, device):
        """build_mask (trainer.py:45-72) on the device: [B, 1, H, W] of {0, 1}."""
        out = torch.empty((B, 1, H, W), dtype=torch.float32, device=device)
        check(_lib.load().pcg_patch_mask(_p(out), B, H, W, patch_size, num_selected, self.seed, self._advance(16 * B), _stream()),
              "pcg_patch_mask")
        return out

    def randint(self, low, high, n, device, exclude=None, out=None):
        out = out if out is not None else torch.empty((n,), dtype=torch.int64, device=device)
        check(_lib.load().pcg_randint(_p(out), n, low, high, _p(exclude), self.seed, self._advance((n + 3) // 4), _stream()),
              "pcg_randint")
        return out

    def randn(self, shape, device, mean=0.0, std=1.0):
        out = torch.empty(shape, dtype=torch.float32, device=device)
        n = out.numel()
        check(_lib.load().pcg_randn(_p(out), n, mean, std, self.seed, self._advance((n + 3) // 4), _stream()), "pcg_randn")
        return out

    def gumbel(self, shape, device, out=None):
        """Gumbel(0,1) noise, the draw F.gumbel_softmax makes (house_sales_kc_usa/models/generator.py:90)."""
        out = out if out is not None else torch.empty(shape, dtype=torch.float32, device=device)
        n = out.numel()
        check(_lib.load().pcg_rand_gumbel(_p(out), n, self.seed, self._advance((n + 3) // 4), _stream()), "pcg_rand_gumbel")
        return out

    def house_draws(self, y, num_classes, D, T, zero_cols, out, onehots=None, counter=None):
        """target class != y, feature mask, Gumbel noise [B, T] in ONE launch, drawn into out = (target_y, mask, noise): the values
        randint(exclude=y), feature_mask, gumbel give when called in this order (same counter offsets).  onehots = (onehot(target_y),
        onehot(y)) float [B, num_classes] buffers: filled by the same launch.  counter: a device counter (device_counter()) that
        supplies the offsets and is advanced by the launch itself — the form a HIP graph can capture (every replay draws the next
        numbers of the stream); the host-side offset is not touched (see house.GraphedTrainStep(rng=...))."""
        o_t, o_m, o_n = out
        B = y.shape[0]
        nz = 0 if zero_cols is None else zero_cols.numel()
        oh_t, oh_y = onehots if onehots is not None else (None, None)
        if counter is not None:
            check(_lib.load().pcg_house_draws_counter(_p(o_t), B, num_classes, _p(y), _p(o_m), D, _p(zero_cols), nz, _p(o_n), T, self.seed,
                                                      _p(oh_t), _p(oh_y), _p(counter), _stream()), "pcg_house_draws_counter")
            return o_t, o_m, o_n
        off_t = self._advance((B + 3) // 4)
        off_m = self._advance((B * D + 3) // 4)
        off_n = self._advance((B * T + 3) // 4)
        check(_lib.load().pcg_house_draws(_p(o_t), B, num_classes, _p(y), off_t, _p(o_m), D, _p(zero_cols), nz, off_m, _p(o_n), T, off_n, self.seed,
                                          _p(oh_t), _p(oh_y), _stream()), "pcg_house_draws")
        return o_t, o_m, o_n

    def house_batch_draws(self, X, Y, perm, num_classes, T, zero_cols, out, onehots, counter, src_out=None):
        """house_draws(counter=...) that also takes the batch: rows perm[cursor .. cursor+B) of the resident training set (X [N, D]
        float32, Y [N] int64) are copied into out = (x, y, target_y, mask, noise), their source rows into src_out; the device counter
        (device_counter(device, cursor=True): int64[4] = [Philox offset, ticket, row cursor, 0]) is advanced by the launch."""
        o_x, o_y, o_t, o_m, o_n = out
        B, D = o_x.shape
        nz = 0 if zero_cols is None else zero_cols.numel()
        oh_t, oh_y = onehots if onehots is not None else (None, None)
        _chk(X, "X"); _chk(Y, "Y", torch.int64); _chk(perm, "perm", torch.int64)
        if counter.numel() < 4 or X.shape[1] != D or Y.numel() != X.shape[0] or perm.numel() < B:
            raise _lib.PcgError("house_batch_draws: counter must be int64[4] (device_counter(cursor=True)), X [N, D], Y [N], perm >= B entries")
        check(_lib.load().pcg_house_batch_draws_counter(_p(o_t), B, num_classes, _p(X), _p(Y), _p(perm), perm.numel(), X.shape[0], _p(o_x), _p(o_y),
                                                        _p(src_out), _p(o_m), D, _p(zero_cols), nz, _p(o_n), T, self.seed, _p(oh_t), _p(oh_y),
                                                        _p(counter), _stream()), "pcg_house_batch_draws_counter")
        return out

    @staticmethod
    def house_draws_span(B, D, T):
        """Counter values one house_draws call consumes."""
        return (B + 3) // 4 + (B * D + 3) // 4 + (B * T + 3) // 4

    def device_counter(self, device, cursor=False):
        """int64[2] on the device: [this stream's current offset, ticket] — the counter argument of house_draws; cursor=True:
        int64[4] = [offset, ticket, row cursor, 0] for house_batch_draws."""
        return torch.tensor([self.offset, 0, 0, 0] if cursor else [self.offset, 0], dtype=torch.int64, device=device)

    def feature_mask(self, B, D, device, zero_cols=None, out=None):
        """Bernoulli(1/2) modifiable-feature mask with immutable columns zeroed (house_sales_kc_usa/trainer.py:253-255);
        zero_cols: int32 device tensor."""
        out = out if out is not None else torch.empty((B, D), dtype=torch.float32, device=device)
        nz = 0 if zero_cols is None else zero_cols.numel()
        check(_lib.load().pcg_feature_mask(_p(out), B, D, _p(zero_cols), nz, self.seed, self._advance((B * D + 3) // 4), _stream()),
              "pcg_feature_mask")
        return out

    def rand(self, shape, device):
        """torch.rand: uniform [0, 1) (WGAN-GP interpolation coefficients, mnist_wgan_conditional.py:146)."""
        out = torch.empty(shape, dtype=torch.float32, device=device)
        n = out.numel()
        check(_lib.load().pcg_rand_uniform(_p(out), n, self.seed, self._advance((n + 3) // 4), _stream()), "pcg_rand_uniform")
        return out

    def bernoulli(self, shape, device, keep_prob):
        """0/1 mask with P(1) = keep_prob — the noise nn.Dropout / nn.Dropout2d draw."""
        out = torch.empty(shape, dtype=torch.float32, device=device)
        n = out.numel()
        check(_lib.load().pcg_rand_bernoulli(_p(out), n, float(keep_prob), self.seed, self._advance((n + 3) // 4), _stream()),
              "pcg_rand_bernoulli")
        return out


def tune(name, value):
    """A/B switch of a launch-planning choice (pcg_tune_set): 'korder', 'wgrad_order', 'dgrad_interleave'; -1 = built-in."""
    check(_lib.load().pcg_tune_set(name.encode(), int(value)), "pcg_tune_set")


# ---- calibration (bench lines; not on the step's path) -------------------------------------------------------------------
def calibrate(device, mfma_ms=20.0, copy_mb=512, rounds=2, warm_ms=30.0):
    """What this box's fp32 matrix pipe and HBM sustain NOW: {"mfma_tflops", "mfma_clock_mhz", "hbm_gbs", ...}.

    A bare v_mfma_f32_32x32x2_f32 loop on every CU for ~mfma_ms (pcg_calib_mfma; the in-kernel shader clock comes from its
    s_memtime / s_memrealtime stamps) and a copy_mb-MiB device copy (pcg_calib_copy), each timed with events on the current
    stream.  bench.py runs it before and after the timed steps, outside the timed bracket, so that a step measured on a box
    with a slower clock can be told from a slower kernel (roofline.achieved / calib.mfma_tflops is box-independent)."""
    lib = _lib.load()
    nbytes = lib.pcg_calib_mfma_workspace_bytes(rounds)
    blocks = lib.pcg_calib_mfma_blocks(rounds)
    if blocks <= 0:
        raise _lib.PcgError("calibrate: no GPU visible to libpcgan_hip")
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    ws[:8192].view(torch.float32).copy_(torch.linspace(-1.0, 1.0, 2048))
    flop = ctypes.c_double(0.0)
    stamps = ctypes.c_uint64(0)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def run(iters):
        e[0].record()
        check(lib.pcg_calib_mfma(int(iters), rounds, _p(ws), nbytes, ctypes.byref(flop), ctypes.byref(stamps), _stream()), "pcg_calib_mfma")
        e[1].record()
        e[1].synchronize()
        return e[0].elapsed_time(e[1])

    probe_iters = 2048
    run(probe_iters)                                   # code object load, clocks up
    ms = run(probe_iters)
    iters = max(probe_iters, int(probe_iters * mfma_ms / max(ms, 1e-3)))
    if warm_ms > 0:                                    # untimed: the reading that follows is not a cold-clock one (r03: "before" read
        run(int(iters * warm_ms / mfma_ms))            # 2273 MHz against 2365 after the timed steps)
    ms = run(iters)
    st = ws[8192 + blocks * 1024:].view(torch.int64).view(blocks, 2).cpu().double()
    mhz = (st[:, 0] / st[:, 1].clamp_min(1.0) * 100.0)
    out = {"mfma_tflops": round(flop.value / (ms * 1e-3) / 1e12, 2), "mfma_ms": round(ms, 3),
           "mfma_clock_mhz": round(float(mhz.median()), 0), "mfma_clock_mhz_min": round(float(mhz.min()), 0),
           "mfma_cycles_per_inst": round(float((st[:, 0] / (iters * 16.0)).median()), 2)}
    n = int(copy_mb) << 20
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty(n, dtype=torch.uint8, device=device)
    fill(src.view(torch.float32), 1.0)
    times = []
    for _ in range(4):
        e[0].record()
        check(lib.pcg_calib_copy(_p(src), _p(dst), n, _stream()), "pcg_calib_copy")
        e[1].record()
        e[1].synchronize()
        times.append(e[0].elapsed_time(e[1]))
    out["hbm_gbs"] = round(2.0 * n / (min(times[1:]) * 1e-3) / 1e9, 1)
    return out

"""MI355X-native (gfx950) engine for the Generator/Discriminator training step of
flash4242/Promptable-Counterfactual-GAN: hand-written HIP kernels behind a C ABI (csrc/, include/pcgan_hip.h)
and PyTorch nn.Module / optimizer drop-ins that call them.

The directory name carries a hyphen; import it as `pcgan_amd` (the shim `pcgan_amd.py` at the repo root).
"""
from . import _lib, ops  # noqa: F401
from ._lib import LIB_PATH, PcgError  # noqa: F401


def load():
    """Load libpcgan_hip.so (once).  On a GPU box this also sets aside the spare stream-K scratch of the CURRENT device
    (ops.prepare_conv_scratch): a conv call that first meets a stream inside a HIP-graph capture is then served without allocating."""
    lib = _lib.load()
    import torch
    if torch.cuda.is_available():
        ops.prepare_conv_scratch()
    return lib

__all__ = ["ops", "load", "LIB_PATH", "PcgError"]

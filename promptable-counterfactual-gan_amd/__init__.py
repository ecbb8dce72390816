"""MI355X-native (gfx950) engine for the Generator/Discriminator training step of
flash4242/Promptable-Counterfactual-GAN: hand-written HIP kernels behind a C ABI (csrc/, include/pcgan_hip.h)
and PyTorch nn.Module / optimizer drop-ins that call them.

The directory name carries a hyphen; import it as `pcgan_amd` (the shim `pcgan_amd.py` at the repo root).
"""
from . import _lib, ops  # noqa: F401
from ._lib import LIB_PATH, PcgError, load  # noqa: F401

__all__ = ["ops", "load", "LIB_PATH", "PcgError"]

"""CPU: the C-ABI library loads (no GPU needed) and exports every symbol include/pcgan_hip.h declares; the ctypes
prototype table covers the same set.  No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pcgan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcg_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("pcg_conv2d_fwd", "pcg_conv2d_dgrad", "pcg_conv2d_wgrad", "pcg_bn_train_stats", "pcg_bn_act_bwd",
                 "pcg_bce_fwd_bwd", "pcg_adam_step", "pcg_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import pcgan_amd
    lib = pcgan_amd.load()
    raw = ctypes.CDLL(pcgan_amd.LIB_PATH)
    for name in _declared():
        assert hasattr(raw, name), f"{name} declared in pcgan_hip.h but not exported by libpcgan_hip.so"
    assert lib.pcg_abi_version() == 5
    assert lib.pcg_target_arch() == b"gfx950"


def test_ctypes_table_matches_header():
    from pcgan_amd import _lib
    assert sorted(_lib.PROTOTYPES) == _declared()


def test_no_compute_without_gpu_and_no_fallback():
    """The product path must fail loudly when it cannot run on the GPU: no CPU / PyTorch fallback."""
    import torch
    import pcgan_amd
    from pcgan_amd import dcgan
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = dcgan.Discriminator({"d_hidden": 8})
    with pytest.raises(pcgan_amd.PcgError, match="no CPU path"):
        net(torch.zeros(2, 1, 64, 64))
    with pytest.raises(pcgan_amd.PcgError, match="GPU"):
        pcgan_amd.ops.act_fwd(torch.zeros(8), pcgan_amd.ops.ACT_RELU)


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "promptable-counterfactual-gan_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src, f"{fn} mentions the oracle: it is test infrastructure only"

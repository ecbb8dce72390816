"""GPU: the DCGAN input transform on the device (pcgan_amd.data.ResizeNormalize -> pcg_resize8_normalize) against Pillow +
the torchvision tensor arithmetic (tests/golden/mnist_resize.npz): bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_resize_normalize_is_bit_exact_against_pillow(golden_dir):
    import pcgan_amd  # noqa: F401
    from pcgan_amd import data
    gold = np.load(os.path.join(golden_dir, "mnist_resize.npz"))
    tf = data.ResizeNormalize((28, 28), (64, 64), 0.5, 0.5, device="cuda:0")
    out = tf(torch.from_numpy(gold["images"]).to("cuda:0"))
    assert out.shape == (gold["images"].shape[0], 1, 64, 64)
    assert np.array_equal(out.cpu().numpy(), gold["out"])
    # a full training batch keeps the per-image result (one block per image)
    big = torch.from_numpy(np.tile(gold["images"], (22, 1, 1))[:512]).to("cuda:0")
    out2 = tf(big)
    assert torch.equal(out2[:24], out) and torch.equal(out2[480:504], out)

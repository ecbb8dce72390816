"""CPU: the input-transform restatement (oracle/data_ref.py) and the host-side coefficient tables the HIP kernel consumes
(pcgan_amd.data) against Pillow's own output (tests/golden/mnist_resize.npz)."""
import os

import numpy as np

from oracle import data_ref as DR


def test_resize_restatement_is_bit_exact_against_pillow(golden_dir):
    gold = np.load(os.path.join(golden_dir, "mnist_resize.npz"))
    got = DR.resize_u8(gold["images"], (64, 64))
    assert np.array_equal(got, gold["resized_u8"])
    assert np.array_equal(DR.to_tensor_normalize(got).reshape(gold["out"].shape), gold["out"])


def test_host_coefficient_tables_match_the_restatement():
    import pcgan_amd  # noqa: F401
    from pcgan_amd import data
    for insz, outsz in ((28, 64), (28, 32), (64, 28), (5, 17)):
        bounds, ik, ksize = data._pillow_bilinear_coeffs(insz, outsz)
        ref, ks = DR.bilinear_coeffs(insz, outsz)
        assert ks == ksize
        for i, (x0, k) in enumerate(ref):
            assert bounds[i, 0] == x0 and bounds[i, 1] == len(k) and list(ik[i, :len(k)]) == k


def test_idx_readers(tmp_path):
    import struct
    import pcgan_amd  # noqa: F401
    from pcgan_amd import data
    imgs = (np.arange(3 * 28 * 28) % 251).astype(np.uint8).reshape(3, 28, 28)
    (tmp_path / "img").write_bytes(struct.pack(">IIII", 2051, 3, 28, 28) + imgs.tobytes())
    (tmp_path / "lab").write_bytes(struct.pack(">II", 2049, 3) + bytes([7, 0, 9]))
    assert np.array_equal(data.read_idx_images(tmp_path / "img"), imgs)
    assert np.array_equal(data.read_idx_labels(tmp_path / "lab"), np.array([7, 0, 9]))


def test_house_sales_preprocessing_matches_reference_bit_for_bit(golden_dir):
    """`pcgan_amd.data.load_house_sales` (numpy only) against the reference's own
    `data_utils.load_and_preprocess` (pandas qcut + sklearn train_test_split + MinMaxScaler), run unmodified by
    tests/golden/make_golden.py on the first 3000 rows of the dataset it ships: same quartile edges, same labels, same shuffled
    split, and bit-identical scaled features (float64)."""
    import os
    import numpy as np
    from pcgan_amd import data
    gold = np.load(os.path.join(golden_dir, "house_preprocess.npz"))
    cfg = {}
    Xtr, Xte, ytr, yte = data.load_house_sales(os.path.join(golden_dir, "kc_house_head3000.csv"), cfg)
    assert np.array_equal(cfg["bins"], gold["bins"])
    assert np.array_equal(ytr, gold["y_train"]) and np.array_equal(yte, gold["y_test"])
    assert Xtr.dtype == np.float64 and Xtr.shape == gold["X_train"].shape and Xte.shape == gold["X_test"].shape
    assert np.array_equal(Xtr, gold["X_train"]) and np.array_equal(Xte, gold["X_test"])
    sc = cfg["scaler"]
    assert np.array_equal(sc.data_min_, gold["data_min"]) and np.array_equal(sc.data_max_, gold["data_max"])
    back = sc.inverse_transform(Xte)
    assert np.allclose(sc.transform(back), Xte, rtol=0, atol=1e-12)
    assert len(cfg["feature_names"]) == 17 and cfg["feature_names"][0] == "bedrooms"


def test_mnist_loader_pieces_host_logic(tmp_path):
    """conditional_counteRGAN/mnist/data_utils.py:6-32 restated for device-resident data (checked here on CPU tensors: the
    pieces are tensor plumbing, no HIP kernel): ToTensor + Normalize arithmetic, the stratified 90/10 split, DataLoader-like
    batching (last short batch kept, a fresh permutation per epoch, every sample exactly once)."""
    import gzip, struct
    import numpy as np
    import torch
    from pcgan_amd import data
    rng = np.random.RandomState(0)
    imgs = rng.randint(0, 256, size=(203, 28, 28)).astype(np.uint8)
    labels = rng.randint(0, 10, size=203).astype(np.uint8)
    pi, pl = tmp_path / "img-idx3-ubyte.gz", tmp_path / "lab-idx1-ubyte"
    with gzip.open(pi, "wb") as f:
        f.write(struct.pack(">IIII", 2051, 203, 28, 28) + imgs.tobytes())
    with open(pl, "wb") as f:
        f.write(struct.pack(">II", 2049, 203) + labels.tobytes())
    x = data.normalize_mnist(data.read_idx_images(pi), "cpu")
    want = (torch.from_numpy(imgs).float().div(255).unsqueeze(1) - 0.5) / 0.5            # ToTensor, then Normalize((0.5,), (0.5,))
    assert x.shape == (203, 1, 28, 28) and torch.equal(x, want) and x.min() >= -1 and x.max() <= 1
    tr, va = data.stratified_split(labels, 0.1, seed=3)
    assert len(va) == 21 and len(tr) == 182 and len(np.intersect1d(tr, va)) == 0
    cnt, cva = np.bincount(labels, minlength=10), np.bincount(labels[va], minlength=10)
    assert np.all(np.abs(cva - cnt * 0.1) < 1.0)                                             # every class keeps its share
    tl, vl, sl, (xf, yf) = data.get_dataloaders(pi, pl, pi, pl, batch_size=64, device="cpu", seed=1)
    assert len(tl) == 3 and len(vl) == 1 and len(sl) == 4 and xf.shape[0] == 203
    seen = []
    for e in range(2):
        got = [(xb, yb) for xb, yb in tl]
        assert [b[0].shape[0] for b in got] == [64, 64, 54] and all(b[0].shape[1:] == (1, 28, 28) for b in got)
        seen.append(torch.cat([b[1] for b in got]))
    assert not torch.equal(seen[0], seen[1]) and torch.equal(seen[0].sort().values, seen[1].sort().values)
    for xb, yb in sl:                                                                       # unshuffled: file order
        pass
    assert torch.equal(torch.cat([b[1] for b in sl]), torch.from_numpy(labels.astype(np.int64)))

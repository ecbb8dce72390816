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

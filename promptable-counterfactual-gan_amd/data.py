"""Input pipeline pieces of the reference's training scripts.

    reference                                                        here
    -------------------------------------------------------------    -------------------------------------------------
    dset.MNIST(root, train=True) raw idx files                        read_idx_images / read_idx_labels (host, numpy)
    transforms.Resize(64) + ToTensor + Normalize((0.5,), (0.5,))      ResizeNormalize (device, bit-exact Pillow bilinear)
      dconv_gan/mnist/mnist_dcgan.py:42-46
    data_utils.load_and_preprocess(csv, config)  (house_sales_kc_usa/data_utils.py:5-41)   load_house_sales / MinMax (host, numpy)
    data_utils.get_dataloaders(...)  (conditional_counteRGAN/mnist/data_utils.py:6-32)      get_dataloaders / normalize_mnist /
                                                                                          stratified_split / DeviceLoader
"""
import gzip
import math
import struct

import numpy as np
import torch

from . import ops, _lib
from ._lib import PcgError

_PRECISION_BITS = 32 - 8 - 2


def read_idx_images(path):
    """MNIST `*-images-idx3-ubyte[.gz]`: big-endian magic 2051, count, rows, cols, then uint8 pixels -> [N, rows, cols] uint8."""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        magic, n, h, w = struct.unpack(">IIII", f.read(16))
        if magic != 2051:
            raise PcgError(f"{path}: not an idx3 image file (magic {magic})")
        return np.frombuffer(f.read(n * h * w), dtype=np.uint8).reshape(n, h, w)


def read_idx_labels(path):
    """MNIST `*-labels-idx1-ubyte[.gz]`: magic 2049, count, then uint8 labels -> [N] int64."""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        magic, n = struct.unpack(">II", f.read(8))
        if magic != 2049:
            raise PcgError(f"{path}: not an idx1 label file (magic {magic})")
        return np.frombuffer(f.read(n), dtype=np.uint8).astype(np.int64)


def _pillow_bilinear_coeffs(in_size, out_size):
    """[Pillow] Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter:
    per output coordinate the source window and its integer weights (22 fractional bits)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)])
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, (-0.5 + kk * (1 << _PRECISION_BITS)).astype(np.int64), (0.5 + kk * (1 << _PRECISION_BITS)).astype(np.int64))
    return bounds, ik.astype(np.int32), ksize


class ResizeNormalize:
    """transforms.Compose([Resize(size), ToTensor(), Normalize((mean,), (std,))]) for uint8 [N, H, W] batches, on the device.
    Square-image Resize(size) as the reference uses it (28x28 -> 64x64)."""

    def __init__(self, in_hw, out_hw, mean=0.5, std=0.5, device="cuda:0"):
        self.in_hw, self.out_hw, self.mean, self.std = tuple(in_hw), tuple(out_hw), float(mean), float(std)
        dev = torch.device(device)
        yb, yk, self.yks = _pillow_bilinear_coeffs(in_hw[0], out_hw[0])
        xb, xk, self.xks = _pillow_bilinear_coeffs(in_hw[1], out_hw[1])
        self.tables = tuple(torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in (xb, xk, yb, yk))

    def __call__(self, images_u8):
        if images_u8.dtype != torch.uint8 or not images_u8.is_cuda or not images_u8.is_contiguous():
            raise PcgError("ResizeNormalize: expected a contiguous uint8 tensor [N, H, W] on the GPU")
        N = images_u8.shape[0]
        if tuple(images_u8.shape[-2:]) != self.in_hw:
            raise PcgError(f"ResizeNormalize: built for {self.in_hw} images, got {tuple(images_u8.shape[-2:])}")
        out = torch.empty((N, 1) + self.out_hw, dtype=torch.float32, device=images_u8.device)
        xb, xk, yb, yk = self.tables
        ops.check(_lib.load().pcg_resize8_normalize(ops._p(images_u8), N, self.in_hw[0], self.in_hw[1], self.out_hw[0], self.out_hw[1],
                                                    ops._p(xb), ops._p(xk), self.xks, ops._p(yb), ops._p(yk), self.yks, self.mean, self.std,
                                                    ops._p(out), ops._stream()), "pcg_resize8_normalize")
        return out


# ---- house_sales_kc_usa/data_utils.py:5-41 — CSV -> quartile price classes -> 80/20 split -> MinMax (host side, as in the reference) ----
class MinMax:
    """The part of sklearn's MinMaxScaler (feature_range (0, 1)) the reference uses: fit on the training split, `transform`,
    `inverse_transform`, and the attributes trainer.py:207-217 reads (`data_min_`, `data_max_`, `scale_`, `min_`).  Same arithmetic
    and order of operations as sklearn (X * scale_ + min_, zero ranges scaled by 1), so results are bit-identical."""

    def fit(self, X):
        X = np.asarray(X, np.float64)
        self.data_min_, self.data_max_ = X.min(axis=0), X.max(axis=0)
        self.data_range_ = self.data_max_ - self.data_min_
        rng = self.data_range_.copy()
        rng[rng < 10 * np.finfo(rng.dtype).eps] = 1.0          # sklearn _handle_zeros_in_scale
        self.scale_ = 1.0 / rng
        self.min_ = 0.0 - self.data_min_ * self.scale_
        self.n_features_in_ = X.shape[1]
        return self

    def transform(self, X):
        X = np.array(X, np.float64, copy=True)
        X *= self.scale_
        X += self.min_
        return X

    def fit_transform(self, X):
        return self.fit(X).transform(X)

    def inverse_transform(self, X):
        X = np.array(X, np.float64, copy=True)
        X -= self.min_
        X /= self.scale_
        return X


def _split_indices(n, test_size=0.2, seed=42):
    """sklearn.model_selection.train_test_split(shuffle=True): ShuffleSplit's one permutation of a RandomState(seed) — the first
    ceil(test_size * n) indices are the test rows, the next floor((1 - test_size) * n) the training rows."""
    n_test = int(math.ceil(test_size * n))
    n_train = int(math.floor((1.0 - test_size) * n))
    perm = np.random.RandomState(seed).permutation(n)
    return perm[n_test:n_test + n_train], perm[:n_test]


def load_house_sales(data_path, config, verbose=False):
    """house_sales_kc_usa/data_utils.py:5-41 `load_and_preprocess(data_path, config)` without pandas / scikit-learn:
    drop id / date / zipcode (:9), bedrooms clipped at 8 (:10), price -> 4 quartile classes (pd.qcut: linear-interpolated
    quantiles, right-closed bins, lowest edge included, :12-13), features = every remaining column but the price (:26), 80/20
    shuffle split with seed 42 (:37), MinMax fitted on the training split (:39-41).  Stores `bins` and `scaler` in `config`
    like the reference.  Returns (X_train_scaled, X_test_scaled, y_train, y_test) as float64 / int64 arrays."""
    import csv
    with open(data_path, newline="") as f:
        rd = csv.reader(f)
        header = next(rd)
        rows = [r for r in rd if r]
    drop = {"id", "date", "zipcode"}
    keep = [i for i, h in enumerate(header) if h not in drop]
    names = [header[i] for i in keep]
    data = np.array([[float(r[i]) for i in keep] for r in rows], np.float64)
    if "bedrooms" in names:
        b = names.index("bedrooms")
        data[data[:, b] > 8, b] = 8.0
    price = data[:, names.index("price")]
    bins = np.unique(np.quantile(price, [0.0, 0.25, 0.5, 0.75, 1.0]))               # duplicates='drop'
    y = np.clip(np.searchsorted(bins, price, side="left") - 1, 0, len(bins) - 2).astype(np.int64)   # (lo, hi] bins, lowest included
    config["bins"] = bins
    feat = [i for i, nme in enumerate(names) if nme != "price"]
    X = data[:, feat]
    if verbose:
        for i in range(len(bins) - 1):
            print(f"Class {i}: ${bins[i]:,.0f} - ${bins[i + 1]:,.0f}")
    tr, te = _split_indices(len(X))
    scaler = MinMax()
    X_train = scaler.fit_transform(X[tr])
    X_test = scaler.transform(X[te])
    config["scaler"] = scaler
    config["feature_names"] = [names[i] for i in feat]
    return X_train, X_test, y[tr], y[te]


# ---- conditional_counteRGAN/mnist/data_utils.py:6-32 — MNIST idx -> ToTensor + Normalize -> stratified 90/10 split -> loaders ----------
def normalize_mnist(images_u8, device):
    """transforms.ToTensor() + Normalize((0.5,), (0.5,)) (data_utils.py:9-12) for a whole uint8 image array at once:
    [N, H, W] uint8 -> [N, 1, H, W] float32 in [-1, 1] on `device`, with torchvision's two roundings (x / 255, then (v - 0.5) / 0.5).
    One-off upload of the data set (plumbing: plain tensor arithmetic, not a hot-path kernel)."""
    x = torch.from_numpy(np.array(images_u8, dtype=np.uint8, copy=True)).to(device)     # frombuffer arrays are read-only
    return x.to(torch.float32).div_(255.0).sub_(0.5).div_(0.5).unsqueeze(1)


def stratified_split(labels, test_size=0.1, seed=None):
    """train_test_split(indices, test_size=0.1, stratify=targets) (data_utils.py:19): every class keeps its share in both parts
    (largest-remainder rounding of the per-class test counts, like sklearn's StratifiedShuffleSplit).  The reference call is
    unseeded, so only the stratification is reproducible, not the particular indices."""
    labels = np.asarray(labels)
    rng = np.random.RandomState(seed)
    n = len(labels)
    n_test = int(math.ceil(test_size * n))
    classes, counts = np.unique(labels, return_counts=True)
    want = counts * (n_test / n)
    take = np.floor(want).astype(int)
    for c in np.argsort(-(want - take), kind="stable")[: n_test - take.sum()]:
        take[c] += 1
    train, test = [], []
    for c, k in zip(classes, take):
        idx = rng.permutation(np.nonzero(labels == c)[0])
        test.append(idx[:k]); train.append(idx[k:])
    return rng.permutation(np.concatenate(train)), rng.permutation(np.concatenate(test))


class DeviceLoader:
    """What `train_countergan(generator, discriminator, classifier, train_loader, cfg, device)` iterates over (trainer.py:89:
    `for x, y in train_loader`): batches of a data set that already lives on the device — DataLoader(shuffle=True) without the
    per-batch host collation and PCIe copy (60000 x 784 floats = 188 MB of 288 GB).  Like DataLoader, the last short batch is
    kept, and each epoch draws a new permutation."""

    def __init__(self, x, y, batch_size, shuffle=True, seed=None):
        if x.shape[0] != y.shape[0]:
            raise PcgError(f"DeviceLoader: {x.shape[0]} samples but {y.shape[0]} labels")
        self.x, self.y, self.batch_size, self.shuffle = x, y, int(batch_size), shuffle
        self._gen = torch.Generator(device="cpu")
        if seed is not None:
            self._gen.manual_seed(seed)

    def __len__(self):
        return (self.x.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.x.shape[0]
        order = torch.randperm(n, generator=self._gen).to(self.x.device) if self.shuffle else None
        for i in range(0, n, self.batch_size):
            if order is None:
                yield self.x[i:i + self.batch_size], self.y[i:i + self.batch_size]
            else:
                idx = order[i:i + self.batch_size]
                yield self.x.index_select(0, idx), self.y.index_select(0, idx)


def get_dataloaders(images_path, labels_path, test_images_path, test_labels_path, batch_size=128, device="cuda", valid_size=0.1, seed=None):
    """data_utils.py:6-32 on the raw idx files: (train_loader, valid_loader, test_loader, (x_full, y_full)) with the data set
    normalised once and resident on `device`."""
    x = normalize_mnist(read_idx_images(images_path), device)
    y_np = read_idx_labels(labels_path)
    y = torch.from_numpy(y_np).to(device)
    tr, va = stratified_split(y_np, valid_size, seed)
    tr_t, va_t = torch.from_numpy(tr).to(device), torch.from_numpy(va).to(device)
    xt = normalize_mnist(read_idx_images(test_images_path), device)
    yt = torch.from_numpy(read_idx_labels(test_labels_path)).to(device)
    return (DeviceLoader(x.index_select(0, tr_t), y.index_select(0, tr_t), batch_size, True, seed),
            DeviceLoader(x.index_select(0, va_t), y.index_select(0, va_t), batch_size, False),
            DeviceLoader(xt, yt, batch_size, False), (x, y))

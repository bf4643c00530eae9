"""Datamodules with the reference's constructor surface (src/tinyedm/datamodules/*.py), re-designed for a GPU with
288 GB of HBM: the whole dataset is loaded ONCE, kept resident on the device as uint8 (CIFAR-10: 150 MB, MNIST:
47 MB) or as fp32 latents, and every batch is one gather kernel (`edm_u8_gather_normalize`: index -> x/255 ->
flip -> normalise, csrc/data.hip) -- no worker processes, no pinned staging buffers, no H2D copy per step.
The on-disk formats are read directly (torchvision is not a dependency): `cifar-10-batches-py` pickles, MNIST
idx-ubyte files, per-sample `.npy` latents.

``SyntheticImageDataModule`` is the throughput / plumbing dataset; ``RandomNoiseDataModule`` reproduces the
reference's predict-time dataset (datamodules/random_datamodule.py:6-18: N(0,1) noise + one random label of shape
(1,) per sample)."""
from __future__ import annotations

import gzip
import os
import pickle
import struct
from pathlib import Path

import numpy as np
import torch


class _DeviceBatches:
    def __init__(self, make, n_batches):
        self.make, self.n = make, n_batches

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.make(i)


class SyntheticImageDataModule:
    """x = 0.5*randn(B,C,H,W) (sigma_data = 0.5, like +-1-normalised images), labels = randint(num_classes);
    one fixed batch generated on the device and reused (no dataloader in the timed region)."""

    def __init__(self, batch_size: int, image_shape=(3, 32, 32), num_classes: int = 10, num_samples: int = 50000,
                 seed: int = 42, device: str | None = None):
        self.batch_size, self.image_shape = batch_size, tuple(image_shape)
        self.num_classes, self.num_samples, self.seed = num_classes, num_samples, seed
        self.device = device
        self._batch = None

    def prepare_data(self):
        pass

    def setup(self, stage=None):
        dev = torch.device(self.device) if self.device else torch.device(
            "cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        g = torch.Generator().manual_seed(self.seed)
        x = 0.5 * torch.randn(self.batch_size, *self.image_shape, generator=g)
        y = torch.randint(0, self.num_classes, (self.batch_size,), generator=g)
        self._batch = (x.to(dev), y.to(dev))

    def _loader(self, n):
        if self._batch is None:
            self.setup()
        return _DeviceBatches(lambda i: self._batch, n)

    def train_dataloader(self):
        return self._loader(max(1, self.num_samples // self.batch_size))

    def val_dataloader(self):
        return self._loader(2)


class RandomNoiseDataModule:
    """datamodules/random_datamodule.py:21-45, same positional order (batch_size, num_workers, image_size,
    num_samples, num_classes).  Extensions are keyword-only: `in_channels` (the reference hard-codes 3),
    `image_shape` (overrides in_channels/image_size), `seed`, `device`.  num_workers is accepted and ignored."""

    def __init__(self, batch_size: int, num_workers: int, image_size: int, num_samples: int,
                 num_classes: int | None, *, in_channels: int = 3, image_shape=None, seed: int = 0,
                 device: str | None = None):
        self.batch_size, self.num_workers, self.image_size = batch_size, num_workers, image_size
        self.num_samples, self._num_classes = num_samples, num_classes
        self.image_shape = tuple(image_shape) if image_shape is not None else (in_channels, image_size, image_size)
        self.seed, self.device = seed, device

    @property
    def num_classes(self):
        return self._num_classes

    def prepare_data(self):
        pass

    def setup(self, stage=None):
        pass

    def predict_dataloader(self):
        dev = torch.device(self.device) if self.device else torch.device("cuda", torch.cuda.current_device())
        n_batches = (self.num_samples + self.batch_size - 1) // self.batch_size

        def make(i):
            g = torch.Generator().manual_seed(self.seed + i)
            b = min(self.batch_size, self.num_samples - i * self.batch_size)
            x = torch.randn(b, *self.image_shape, generator=g)
            k = self.num_classes if self.num_classes else 1
            y = torch.randint(0, k, (b, 1), generator=g)
            return x.to(dev), y.to(dev)

        return _DeviceBatches(make, n_batches)


# --------------------------------------------------------------------------------------------------------------
# resident image datasets
# --------------------------------------------------------------------------------------------------------------
def read_cifar10(data_dir: str, train: bool):
    """`cifar-10-batches-py` as written by the CIFAR-10 python tarball (what torchvision.datasets.CIFAR10 reads)."""
    d = os.path.join(data_dir, "cifar-10-batches-py")
    names = [f"data_batch_{i}" for i in range(1, 6)] if train else ["test_batch"]
    xs, ys = [], []
    for n in names:
        path = os.path.join(d, n)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: place the extracted CIFAR-10 python archive under {data_dir} "
                                    "(this build never downloads)")
        with open(path, "rb") as f:
            rec = pickle.load(f, encoding="bytes")
        xs.append(np.asarray(rec[b"data"], dtype=np.uint8).reshape(-1, 3, 32, 32))
        ys.append(np.asarray(rec.get(b"labels", rec.get(b"fine_labels")), dtype=np.int64))
    return np.concatenate(xs), np.concatenate(ys)


def _open_maybe_gz(path):
    if os.path.exists(path):
        return open(path, "rb")
    if os.path.exists(path + ".gz"):
        return gzip.open(path + ".gz", "rb")
    raise FileNotFoundError(f"{path}[.gz] not found (this build never downloads)")


def read_mnist(data_dir: str, train: bool):
    """`MNIST/raw/*-idx?-ubyte[.gz]` (what torchvision.datasets.MNIST reads): big-endian idx headers."""
    d = os.path.join(data_dir, "MNIST", "raw")
    stem = "train" if train else "t10k"
    with _open_maybe_gz(os.path.join(d, f"{stem}-images-idx3-ubyte")) as f:
        magic, n, h, w = struct.unpack(">IIII", f.read(16))
        if magic != 2051:
            raise ValueError(f"bad MNIST image magic {magic}")
        x = np.frombuffer(f.read(n * h * w), dtype=np.uint8).reshape(n, 1, h, w)
    with _open_maybe_gz(os.path.join(d, f"{stem}-labels-idx1-ubyte")) as f:
        magic, n2 = struct.unpack(">II", f.read(8))
        if magic != 2049 or n2 != n:
            raise ValueError("bad MNIST label file")
        y = np.frombuffer(f.read(n), dtype=np.uint8).astype(np.int64)
    return x, y


def _rank_world(rank=None, world=None):
    """this process's (rank, world size): explicit arguments, else torch.distributed, else the launcher's env"""
    if rank is not None and world is not None:
        return int(rank), int(world)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_len(n: int, world: int) -> int:
    """samples per rank: ceil(n / world) -- every rank runs the same number of steps (short shards wrap around)"""
    return (n + world - 1) // world


def epoch_order(n: int, shuffle: bool, seed: int, epoch: int, rank: int, world: int, device):
    """Index tensor of the samples rank `rank` sees in epoch `epoch`: ONE permutation shared by all ranks (seeded by
    (seed, epoch) only), padded by wrapping to a multiple of `world`, then strided rank::world -- the index sets of
    the ranks are disjoint (up to the <= world-1 wrapped samples) and together cover the dataset, which is what
    Lightning's DistributedSampler gives the reference's DataLoaders."""
    if shuffle:
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + epoch)
        order = torch.randperm(n, device=device, generator=g)
    else:
        order = torch.arange(n, device=device)
    if world > 1:
        total = shard_len(n, world) * world
        if total > n:
            order = torch.cat([order, order[: total - n]])
        order = order[rank::world]
    return order


class _ResidentLoader:
    """Batches gathered on the device from a resident uint8 dataset; shuffled per epoch with torch.randperm
    (same permutation on every rank), each rank taking its `rank::world` share (DistributedSampler semantics)."""

    def __init__(self, data_u8, labels, batch_size, shuffle, flip, seed, mean, std, rank=None, world=None):
        self.data, self.labels, self.batch_size = data_u8, labels, batch_size
        self.shuffle, self.flip, self.seed, self.mean, self.std = shuffle, flip, seed, mean, std
        self.rank, self.world = _rank_world(rank, world)
        self.epoch = 0

    def __len__(self):
        return (shard_len(self.data.shape[0], self.world) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        from . import ops
        order = epoch_order(self.data.shape[0], self.shuffle, self.seed, self.epoch, self.rank, self.world,
                            self.data.device)
        for bi in range(len(self)):
            idx = order[bi * self.batch_size:(bi + 1) * self.batch_size].contiguous()
            x = ops.u8_gather_normalize(self.data, idx, self.mean, self.std, flip=self.flip,
                                        seed=self.seed + 7919 * self.rank, epoch=self.epoch * 65536 + bi)
            yield x, self.labels[idx]
        self.epoch += 1


class AbstractDataModule:
    """Constructor / property surface of datamodules/abstract_datamodule.py:6-67 (num_workers is accepted and
    ignored: there are no worker processes)."""

    def __init__(self, data_dir, batch_size: int, num_workers: int = 0, device: str | None = None, seed: int = 42):
        self.data_dir, self.batch_size, self.num_workers = data_dir, batch_size, num_workers
        self.device, self.seed = device, seed
        self.train_dataset = self.val_dataset = self.test_dataset = None
        self.mean, self.std, self.flip = 0.5, 0.5, False

    def _dev(self):
        return torch.device(self.device) if self.device else torch.device("cuda", torch.cuda.current_device())

    def prepare_data(self):
        pass

    def _resident(self, x, y):
        dev = self._dev()
        return torch.from_numpy(np.ascontiguousarray(x)).to(dev), torch.from_numpy(y).to(dev)

    def _loader(self, ds, shuffle, flip):
        if ds is None:
            raise RuntimeError("call setup() first")
        return _ResidentLoader(ds[0], ds[1], self.batch_size, shuffle, flip, self.seed, self.mean, self.std)

    def train_dataloader(self):
        return self._loader(self.train_dataset, True, self.flip)

    def val_dataloader(self):
        return self._loader(self.val_dataset, False, self.flip)

    def test_dataloader(self):
        return self._loader(self.test_dataset, False, self.flip)

    def denormalize(self, x):
        """(x*127.5 + 128).clip(0,255).to(uint8) (cifar10datamodule.py:34-35), on the device."""
        from . import ops
        return ops.denormalize_u8(x.to(torch.float32).contiguous())


class CIFAR10DataModule(AbstractDataModule):
    """datamodules/cifar10datamodule.py:8-49 (image_size must be the native 32: the reference's Resize is then the
    identity).  The flip is applied to every loader, as the reference's single transform does."""
    classes = ["airplane", "automobile", "bird", "cat", "deer", "dog", "frog", "horse", "ship", "truck"]

    def __init__(self, data_dir: str = "datasets/cifar", image_size: int = 32, batch_size: int = 16, num_workers: int = 16,
                 device: str | None = None, seed: int = 42):
        super().__init__(data_dir, batch_size, num_workers, device, seed)
        if image_size != 32:
            raise ValueError("CIFAR10DataModule: only the native image_size 32 is supported")
        self.img_size, self.flip = image_size, True

    def prepare_data(self):
        read_cifar10(self.data_dir, True)  # existence check only: never downloads

    def setup(self, stage=None):
        if stage == "fit" or stage is None:
            self.train_dataset = self._resident(*read_cifar10(self.data_dir, True))
            self.val_dataset = self._resident(*read_cifar10(self.data_dir, False))
        if stage == "test":
            self.test_dataset = self._resident(*read_cifar10(self.data_dir, False))

    @property
    def num_classes(self) -> int:
        return 10


class MNISTDataModule(AbstractDataModule):
    """datamodules/mnistdatamodule.py:9-47 (native image_size 28, no flip)."""
    classes = [str(i) for i in range(10)]

    def __init__(self, batch_size: int, num_workers: int = 0, image_size: int = 28, data_dir: str = "datasets/mnist",
                 device: str | None = None, seed: int = 42):
        super().__init__(data_dir, batch_size, num_workers, device, seed)
        if image_size != 28:
            raise ValueError("MNISTDataModule: only the native image_size 28 is supported")

    def prepare_data(self):
        read_mnist(self.data_dir, True)

    def setup(self, stage=None):
        if stage == "fit" or stage is None:
            self.train_dataset = self._resident(*read_mnist(self.data_dir, True))
            self.val_dataset = self._resident(*read_mnist(self.data_dir, False))
        if stage == "test":
            self.test_dataset = self._resident(*read_mnist(self.data_dir, False))

    @property
    def num_classes(self) -> int:
        return 10


class _LatentLoader:
    def __init__(self, lat, lab, batch_size, shuffle, seed, rank=None, world=None):
        self.lat, self.lab, self.batch_size, self.shuffle, self.seed, self.epoch = lat, lab, batch_size, shuffle, seed, 0
        self.rank, self.world = _rank_world(rank, world)

    def __len__(self):
        return (shard_len(self.lat.shape[0], self.world) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = epoch_order(self.lat.shape[0], self.shuffle, self.seed, self.epoch, self.rank, self.world, self.lat.device)
        for bi in range(len(self)):
            idx = order[bi * self.batch_size:(bi + 1) * self.batch_size]
            yield self.lat[idx], self.lab[idx]
        self.epoch += 1


class ImageNetLatentsDataModule(AbstractDataModule):
    """datamodules/imagenet_latents_datamodule.py:8-50: `<data_dir>/{train,val}/{latents,labels}/<i>.npy`, read once
    into resident fp32 (1.28 M latents of 4x64x64 = 84 GB: fits one MI355X; 4x32x32 = 21 GB)."""

    def __init__(self, data_dir, image_size, batch_size, num_workers: int = 0, device: str | None = None, seed: int = 42):
        super().__init__(Path(data_dir), batch_size, num_workers, device, seed)
        self.image_size = image_size

    def _read(self, split):
        root = Path(self.data_dir) / split
        n = len(list((root / "latents").glob("*.npy")))
        if n == 0:
            raise FileNotFoundError(f"no latents under {root / 'latents'}")
        first = np.load(root / "latents" / "0.npy")
        lat = torch.empty((n,) + first.shape, dtype=torch.float32, device=self._dev())
        lab = torch.empty(n, dtype=torch.int64, device=self._dev())
        for i in range(n):
            lat[i] = torch.from_numpy(np.load(root / "latents" / f"{i}.npy").astype(np.float32))
            lab[i] = int(np.load(root / "labels" / f"{i}.npy"))
        return lat, lab

    def setup(self, stage=None):
        if stage == "fit" or stage is None:
            self.train_dataset = self._read("train")
            self.val_dataset = self._read("val")

    def train_dataloader(self):
        return _LatentLoader(*self.train_dataset, self.batch_size, True, self.seed)

    def val_dataloader(self):
        return _LatentLoader(*self.val_dataset, self.batch_size, False, self.seed)

    @property
    def num_classes(self) -> int:
        return 1000

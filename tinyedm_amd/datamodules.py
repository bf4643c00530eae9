"""Device-resident synthetic data for throughput runs and plumbing tests (the reference's real
datamodules -- torchvision CIFAR/MNIST, .npy latents -- are host I/O and out of the hot-path scope).
``RandomNoiseDataModule`` reproduces the reference's predict-time dataset semantics
(datamodules/random_datamodule.py:6-18: N(0,1) noise + one random label of shape (1,) per sample)."""
from __future__ import annotations

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
    def __init__(self, batch_size: int, num_samples: int, image_shape=(3, 32, 32), num_classes: int | None = None,
                 seed: int = 0, device: str | None = None):
        self.batch_size, self.num_samples, self.image_shape = batch_size, num_samples, tuple(image_shape)
        self.num_classes, self.seed, self.device = num_classes, seed, device

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

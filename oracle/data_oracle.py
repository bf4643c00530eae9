"""CPU restatement of the reference's image <-> tensor conversions.  TEST INFRASTRUCTURE ONLY (see edm_oracle.py).

Pinned by construction: every function is the reference's own expression evaluated with torch CPU fp32 ops in the
same order (the reference's transforms are themselves torch CPU ops; torchvision is absent here, so
``v2.ToDtype(float32, scale=True)`` and ``v2.Normalize`` are restated from their documented arithmetic
x/255 and (x-mean)/std), plus the all-256-byte-values round trip checked in tests/test_data_cpu.py.
"""
import gzip
import os
import pickle
import struct

import numpy as np
import torch


def normalize_u8(x_u8: torch.Tensor, mean: float = 0.5, std: float = 0.5, flip: torch.Tensor = None) -> torch.Tensor:
    """datamodules/cifar10datamodule.py:18-32 / mnistdatamodule.py:18-30 on a uint8 (B,C,H,W) batch:
    ToDtype(float32, scale=True) -> x/255; RandomHorizontalFlip (``flip`` = bool per sample); Normalize -> (x-mean)/std."""
    x = x_u8.to(torch.float32) / 255.0
    if flip is not None:
        x = torch.where(flip.view(-1, 1, 1, 1), x.flip(-1), x)
    return (x - mean) / std


def denormalize(x: torch.Tensor) -> torch.Tensor:
    """cifar10datamodule.py:34-35."""
    return (x.to(torch.float32) * 127.5 + 128).clip(0, 255).to(torch.uint8)


def prediction_to_u8_nhwc(pred: torch.Tensor, mean, std) -> torch.Tensor:
    """callbacks.py:141-153 (PreditionWriter.write_on_batch_end) up to the PNG encode."""
    mean = torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1)
    std = torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1)
    images = pred * std * 2 + mean
    images = torch.clamp(images, 0, 1).permute(0, 2, 3, 1) * 255
    return images.to(torch.uint8)


# ---- on-disk formats the reference reads through torchvision (restated: torchvision is not installed) ----------------
def write_cifar10_batches(root: str, images_u8: np.ndarray, labels: np.ndarray, n_train: int):
    """Writes the standard `cifar-10-batches-py` layout (data_batch_1..5 + test_batch, pickled dicts with
    b'data' uint8 [n,3072] in CHW order and b'labels') -- used by the tests to fabricate a tiny dataset."""
    d = os.path.join(root, "cifar-10-batches-py")
    os.makedirs(d, exist_ok=True)
    per = (n_train + 4) // 5
    for i in range(5):
        sl = slice(i * per, min((i + 1) * per, n_train))
        with open(os.path.join(d, f"data_batch_{i + 1}"), "wb") as f:
            pickle.dump({b"data": images_u8[sl].reshape(-1, 3072), b"labels": [int(v) for v in labels[sl]]}, f)
    with open(os.path.join(d, "test_batch"), "wb") as f:
        pickle.dump({b"data": images_u8[n_train:].reshape(-1, 3072), b"labels": [int(v) for v in labels[n_train:]]}, f)


def write_mnist_idx(root: str, images_u8: np.ndarray, labels: np.ndarray, n_train: int):
    """Writes MNIST/raw/{train,t10k}-{images-idx3,labels-idx1}-ubyte(.gz) (big-endian idx headers)."""
    d = os.path.join(root, "MNIST", "raw")
    os.makedirs(d, exist_ok=True)
    for name, sl in (("train", slice(0, n_train)), ("t10k", slice(n_train, None))):
        im, lb = images_u8[sl], labels[sl]
        with gzip.open(os.path.join(d, f"{name}-images-idx3-ubyte.gz"), "wb") as f:
            f.write(struct.pack(">IIII", 2051, im.shape[0], im.shape[1], im.shape[2]) + im.tobytes())
        with open(os.path.join(d, f"{name}-labels-idx1-ubyte"), "wb") as f:
            f.write(struct.pack(">II", 2049, lb.shape[0]) + lb.astype(np.uint8).tobytes())

"""Sampling-side callbacks with the reference's surface (src/tinyedm/callbacks.py), without lightning / torchvision /
wandb: ``GenerateCallback`` (in-training sampling through ``swap_ema_weights``, callbacks.py:12-58) and
``PreditionWriter`` (sic -- the reference's spelling; per-image PNG dump of predict outputs, callbacks.py:126-156).
Image conversion runs on the GPU (csrc/data.hip); only the final bytes cross PCIe.  ``LatentsGenerateCallback``
needs the third-party SD-VAE weights (no network here) and is not provided."""
from __future__ import annotations

from pathlib import Path
from typing import Literal, Sequence

import numpy as np
import torch

from . import ops


def make_grid_u8(images_u8_nchw: torch.Tensor, nrow: int = 8, padding: int = 2) -> np.ndarray:
    """uint8 (B,C,H,W) -> one uint8 (Hg,Wg,C) mosaic (torchvision.utils.make_grid layout: row-major, black padding)."""
    x = images_u8_nchw.cpu().numpy()
    B, C, H, W = x.shape
    ncol = min(nrow, B)
    nr = (B + ncol - 1) // ncol
    grid = np.zeros((C, nr * (H + padding) + padding, ncol * (W + padding) + padding), dtype=np.uint8)
    for i in range(B):
        r, c = divmod(i, ncol)
        y0, x0 = padding + r * (H + padding), padding + c * (W + padding)
        grid[:, y0:y0 + H, x0:x0 + W] = x[i]
    return np.transpose(grid, (1, 2, 0))


def _save_image(arr_hwc: np.ndarray, path: Path):
    from PIL import Image
    a = arr_hwc[:, :, 0] if arr_hwc.shape[2] == 1 else arr_hwc
    Image.fromarray(a).save(path)


class GenerateCallback:
    """Every ``every_n_epochs`` epochs: Heun-sample ``num_samples`` images from fixed noise with the EMA weights
    swapped in, denormalise to uint8 on the GPU and write a PNG mosaic (the reference logs it to wandb)."""

    def __init__(self, solver, img_shape: tuple[int, int, int], num_samples: int = 8, every_n_epochs=5,
                 output_dir: str = "generated"):
        self.solver, self.num_samples, self.img_shape = solver, num_samples, tuple(img_shape)
        self.every_n_epochs, self.output_dir = every_n_epochs, Path(output_dir)
        self.class_labels, self.x0, self.last_grid = None, None, None

    def on_train_start(self, trainer, pl_module):
        if getattr(trainer, "global_rank", 0) != 0:
            return
        dev = pl_module.device
        self.class_labels = (torch.arange(0, pl_module.num_classes, device=dev, dtype=torch.long)
                             if pl_module.conditional else None)
        n = self.num_samples if self.class_labels is None else self.class_labels.shape[0]
        self.x0 = torch.randn(n, *self.img_shape, device=dev)

    def on_train_epoch_end(self, trainer, pl_module):
        if getattr(trainer, "global_rank", 0) != 0 or self.x0 is None:
            return
        if trainer.current_epoch % self.every_n_epochs != 0:
            return
        was_training = pl_module.training
        pl_module.eval()
        with torch.no_grad():
            if pl_module.use_ema:
                with pl_module.swap_ema_weights(trainer):
                    xT = self.solver.solve(pl_module, self.x0, self.class_labels)
            else:
                xT = self.solver.solve(pl_module, self.x0, self.class_labels)
            dm = getattr(trainer, "datamodule", None)
            images = dm.denormalize(xT) if dm is not None and hasattr(dm, "denormalize") else ops.denormalize_u8(
                xT.to(torch.float32).contiguous())
        self.last_grid = make_grid_u8(images)
        self.output_dir.mkdir(parents=True, exist_ok=True)
        _save_image(self.last_grid, self.output_dir / f"epoch_{trainer.current_epoch:05d}.png")
        if was_training:
            pl_module.train()


class PreditionWriter:
    def __init__(self, output_dir: str, write_interval: Literal["batch", "epoch", "batch_and_epoch"], mean: Sequence,
                 std: Sequence, first_index: int = 0):
        self.output_dir = Path(output_dir)
        self.write_interval = write_interval
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self._mean_t = self._std_t = None
        self._count = first_index          # replicas write disjoint global index ranges
        self.output_dir.mkdir(parents=True, exist_ok=True)

    def setup(self, trainer, pl_module, stage: str = "predict"):
        dev = pl_module.device
        self._mean_t = torch.tensor(self.mean, device=dev, dtype=torch.float32)
        self._std_t = torch.tensor(self.std, device=dev, dtype=torch.float32)

    def write_on_batch_end(self, trainer, pl_module, prediction, batch_indices, batch, batch_idx, dataloader_idx) -> None:
        if self._mean_t is None:
            self.setup(trainer, pl_module)
        images = ops.prediction_to_u8_nhwc(prediction.to(torch.float32).contiguous(), self._mean_t, self._std_t)
        images = images.cpu().numpy()
        if batch_indices is None:
            batch_indices = range(self._count, self._count + images.shape[0])
        for batch_index, image in zip(batch_indices, images):
            _save_image(image, self.output_dir / f"{batch_index}.png")
        self._count += images.shape[0]

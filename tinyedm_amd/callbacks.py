"""Sampling-side callbacks with the reference's surface (src/tinyedm/callbacks.py), without lightning / torchvision /
wandb: ``GenerateCallback`` (in-training sampling through ``swap_ema_weights``, callbacks.py:12-58) and
``PreditionWriter`` (sic -- the reference's spelling; per-image PNG dump of predict outputs, callbacks.py:126-156).
Image conversion runs on the GPU (csrc/data.hip); only the final bytes cross PCIe.  ``LatentsGenerateCallback``
(callbacks.py:61-123) samples and de-normalises latents the same way; decoding them needs the third-party SD-VAE
(`diffusers`, weights from the network): used when importable, otherwise the latents themselves are written.
``ModelCheckpoint`` stands in for `lightning.pytorch.callbacks.ModelCheckpoint`, which the reference's YAMLs name
(conf/cifar10.yaml:58-67): top-k by a monitored metric + `last.ckpt`, in the reference's checkpoint key layout."""
from __future__ import annotations

import os
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


class _eval_dtype:
    """`with _eval_dtype(pl_module, "f32"):` -- the denoiser's evaluation precision for the duration of a sampling call"""

    def __init__(self, pl_module, dtype):
        self.den, self.dtype = getattr(pl_module, "denoiser", None), dtype
        self.prev = None

    def __enter__(self):
        if self.den is not None and hasattr(self.den, "set_eval_dtype"):
            self.prev = self.den.eval_dtype
            self.den.set_eval_dtype(self.dtype)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            self.den.set_eval_dtype(self.prev)
        return False


class GenerateCallback:
    """Every ``every_n_epochs`` epochs: Heun-sample ``num_samples`` images from fixed noise with the EMA weights
    swapped in, denormalise to uint8 on the GPU and write a PNG mosaic (the reference logs it to wandb)."""

    def __init__(self, solver, img_shape: tuple[int, int, int], num_samples: int = 8, every_n_epochs=5,
                 output_dir: str = "generated", network_dtype: str = "f32x3"):
        self.solver, self.num_samples, self.img_shape = solver, num_samples, tuple(img_shape)
        self.every_n_epochs, self.output_dir = every_n_epochs, Path(output_dir)
        # the reference's callback samples outside Lightning's autocast context, i.e. in fp32 (callbacks.py:41-49): the default
        # here is fp32-accurate too ("f32x3": split-bf16 convs; "f32": exact fp32 products); "bf16" = the training path's
        # kernels (fast mode)
        self.network_dtype = network_dtype
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
        with torch.no_grad(), _eval_dtype(pl_module, self.network_dtype):
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


class LatentsGenerateCallback:
    """callbacks.py:61-123: every `every_n_epochs` validation epochs, Heun-sample `num_samples_per_class` latents for
    each of `num_classes` random labels with the EMA weights swapped in and de-normalise them (x*std*2 + mean).
    The reference then decodes with `diffusers.AutoencoderKL("stabilityai/sd-vae-ft-ema")`: done here only when
    `diffusers` and its weights are available; otherwise the de-normalised latents are saved (`.npy`)."""

    def __init__(self, solver, img_shape: tuple[int, int, int], mean: tuple, std: tuple,
                 value_range: tuple[float, float] = (0, 1), num_samples_per_class: int = 8, num_classes=10,
                 every_n_epochs=100, output_dir: str = "generated", network_dtype: str = "f32x3"):
        self.network_dtype = network_dtype          # fp32-accurate like the reference's callback (callbacks.py:95-104); "bf16" = fast mode
        self.solver, self.img_shape = solver, tuple(img_shape)
        self.num_samples_per_class, self.num_classes, self.every_n_epochs = num_samples_per_class, num_classes, every_n_epochs
        self.value_range, self.mean, self.std = tuple(value_range), mean, std
        self.output_dir = Path(output_dir)
        self.class_labels = self.x0 = self.vae = self.last = None

    def on_fit_start(self, trainer, pl_module):
        if getattr(trainer, "global_rank", 0) != 0:
            return
        dev = pl_module.device
        n_cls = getattr(trainer.datamodule, "num_classes", None) or pl_module.num_classes or self.num_classes
        labels = torch.randint(0, n_cls, (self.num_classes,), device=dev, dtype=torch.long)
        self.x0 = torch.randn(self.num_samples_per_class * self.num_classes, *self.img_shape, device=dev)
        self.class_labels = labels.repeat(self.num_samples_per_class)
        self._std = torch.tensor(self.std, device=dev, dtype=torch.float32).view(1, -1, 1, 1)
        self._mean = torch.tensor(self.mean, device=dev, dtype=torch.float32).view(1, -1, 1, 1)
        try:
            from diffusers.models import AutoencoderKL            # third-party decoder: optional
            self.vae = AutoencoderKL.from_pretrained("stabilityai/sd-vae-ft-ema").to(dev).eval()
        except Exception:                                          # not installed / no weights / no network
            self.vae = None

    def on_validation_epoch_end(self, trainer, pl_module):
        if getattr(trainer, "global_rank", 0) != 0 or self.x0 is None:
            return
        if trainer.current_epoch % self.every_n_epochs != 0:
            return
        with torch.no_grad(), _eval_dtype(pl_module, self.network_dtype):
            labels = self.class_labels if pl_module.conditional else None
            if pl_module.use_ema:
                with pl_module.swap_ema_weights(trainer):
                    xT = self.solver.solve(pl_module, self.x0, labels)
            else:
                xT = self.solver.solve(pl_module, self.x0, labels)
            XT = xT.float() * self._std * 2 + self._mean
            self.last = XT
            self.output_dir.mkdir(parents=True, exist_ok=True)
            if self.vae is not None:
                images = torch.clamp(self.vae.decode(XT).sample, *self.value_range)
                u8 = (images * 255).to(torch.uint8)
                _save_image(make_grid_u8(u8, nrow=self.num_classes), self.output_dir / f"epoch_{trainer.current_epoch:05d}.png")
            else:
                np.save(self.output_dir / f"latents_epoch_{trainer.current_epoch:05d}.npy", XT.cpu().numpy())


class ModelCheckpoint:
    """Stand-in for `lightning.pytorch.callbacks.ModelCheckpoint` with the keyword surface the reference's YAMLs use
    (`monitor, mode, save_top_k, save_last, verbose, every_n_epochs, save_on_train_epoch_end`, plus `dirpath` /
    `filename`): after validation (or at train-epoch end when `save_on_train_epoch_end`) every `every_n_epochs`
    epochs, writes `<dirpath>/epoch=E-step=S.ckpt` through `Trainer.save_checkpoint`, keeps the best `save_top_k`
    by `monitor` and, with `save_last`, a copy named `last.ckpt`."""

    def __init__(self, dirpath=None, filename=None, monitor=None, verbose: bool = False, save_last=None,
                 save_top_k: int = 1, mode: str = "min", every_n_epochs=None, save_on_train_epoch_end=None,
                 every_n_train_steps=None, **_ignored):
        if mode not in ("min", "max"):
            raise ValueError("ModelCheckpoint: mode must be 'min' or 'max'")
        self.dirpath = Path(dirpath) if dirpath is not None else Path("checkpoints")
        self.filename, self.monitor, self.verbose = filename, monitor, verbose
        self.save_last, self.save_top_k, self.mode = bool(save_last), save_top_k, mode
        self.every_n_epochs = 1 if every_n_epochs is None else every_n_epochs
        self.save_on_train_epoch_end = bool(save_on_train_epoch_end)
        self.best_k_models: dict = {}
        self.best_model_path, self.best_model_score, self.last_model_path = "", None, ""

    def _maybe_save(self, trainer, pl_module):
        epoch = trainer.current_epoch
        if self.every_n_epochs < 1 or (epoch + 1) % self.every_n_epochs != 0:
            return
        score = None
        if self.monitor is not None:
            score = trainer.callback_metrics.get(self.monitor)
            if score is None:
                return                       # nothing logged under that name yet (e.g. no validation ran)
            score = float(score)
        name = (self.filename or "epoch={epoch}-step={step}").format(epoch=epoch, step=trainer.global_step)
        path = self.dirpath / f"{name}.ckpt"
        if trainer.global_rank == 0:
            self.dirpath.mkdir(parents=True, exist_ok=True)
        keep = self.save_top_k != 0
        if keep and self.monitor is not None and self.save_top_k > 0 and len(self.best_k_models) >= self.save_top_k:
            worst = (max if self.mode == "min" else min)(self.best_k_models, key=self.best_k_models.get)
            better = score < self.best_k_models[worst] if self.mode == "min" else score > self.best_k_models[worst]
            keep = better
            if better:
                self.best_k_models.pop(worst)
                if trainer.global_rank == 0 and os.path.exists(worst):
                    os.remove(worst)
        if keep:
            trainer.save_checkpoint(str(path), pl_module)
            if self.monitor is not None:
                self.best_k_models[str(path)] = score
                best = (min if self.mode == "min" else max)(self.best_k_models, key=self.best_k_models.get)
                self.best_model_path, self.best_model_score = best, self.best_k_models[best]
            if self.verbose and trainer.global_rank == 0:
                print(f"[ModelCheckpoint] epoch {epoch}: saved {path}" + (f" ({self.monitor}={score:.5f})" if score is not None else ""))
        if self.save_last:
            last = self.dirpath / "last.ckpt"
            trainer.save_checkpoint(str(last), pl_module)
            self.last_model_path = str(last)

    def on_validation_end(self, trainer, pl_module):
        if not self.save_on_train_epoch_end:
            self._maybe_save(trainer, pl_module)

    def on_train_epoch_end(self, trainer, pl_module):
        if self.save_on_train_epoch_end:
            self._maybe_save(trainer, pl_module)

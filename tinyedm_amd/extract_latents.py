"""Latent extraction for the ImageNet-256 latent-diffusion configuration (reference:
datamodules/extract_latents.py:14-125): an image folder (`<data_dir>/<class>/<image>`) -> centre-cropped, randomly
flipped, [-1, 1]-normalised batches -> VAE encoder -> latents normalised with the reference's per-channel constants
((z - mean) / (2 std)) -> `<out_dir>/latents/<i>.npy` + `<out_dir>/labels/<i>.npy`, the directory layout
`ImageNetLatentsDataModule` reads.

The encoder is third-party (`diffusers.AutoencoderKL("stabilityai/sd-vae-ft-{ema,mse}")`, weights fetched from the hub):
it is used when `diffusers` and its weights are present; any callable `encoder(x: (B,3,S,S) float in [-1,1]) -> (B,4,S/8,S/8)`
can be passed instead (tests use a fixed linear stand-in).  Everything around it -- folder walk in torchvision's
ImageFolder order, ADM centre crop, flip, normalisation, file layout, drop_last batching -- is this module's.
    python -m tinyedm_amd.extract_latents --data_dir <imagenet/train> --out_dir <latents/train> --image_size 256
"""
from __future__ import annotations

import argparse
import os
from pathlib import Path
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

# per-channel statistics of SD-VAE latents the reference normalises with (extract_latents.py:73-79) and
# LatentsGenerateCallback inverts (conf/imagenet.yaml mean / std)
LATENT_MEAN = (5.81, 3.25, 0.12, -2.15)
LATENT_STD = (4.17, 4.62, 3.71, 3.28)
IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def center_crop_arr(pil_image, image_size: int):
    """ADM's centre crop (guided-diffusion image_datasets.py): halve with a box filter while the short side is at least
    twice the target, resize the short side to the target bicubically, cut the central square."""
    from PIL import Image
    while min(pil_image.size) >= 2 * image_size:
        pil_image = pil_image.resize((pil_image.size[0] // 2, pil_image.size[1] // 2), resample=Image.BOX)
    s = image_size / min(pil_image.size)
    pil_image = pil_image.resize((round(pil_image.size[0] * s), round(pil_image.size[1] * s)), resample=Image.BICUBIC)
    a = np.asarray(pil_image)
    y0, x0 = (a.shape[0] - image_size) // 2, (a.shape[1] - image_size) // 2
    return Image.fromarray(a[y0:y0 + image_size, x0:x0 + image_size])


def list_image_folder(root) -> Tuple[List[Tuple[str, int]], List[str]]:
    """(path, class index) samples in torchvision.datasets.ImageFolder order: classes = sorted sub-directories, files
    sorted within a class (walked recursively), image extensions only."""
    root = Path(root)
    classes = sorted(d.name for d in root.iterdir() if d.is_dir())
    if not classes:
        raise FileNotFoundError(f"no class folders under {root}")
    samples = []
    for ci, c in enumerate(classes):
        for dirpath, _dirs, files in sorted(os.walk(root / c, followlinks=True)):
            for f in sorted(files):
                if f.lower().endswith(IMG_EXTENSIONS):
                    samples.append((os.path.join(dirpath, f), ci))
    return samples, classes


def load_batch(samples, image_size: int, flips: np.ndarray) -> torch.Tensor:
    """uint8 images -> (B,3,S,S) float32 in [-1, 1] (ToTensor + Normalize(0.5, 0.5)), centre-cropped, flipped where asked"""
    from PIL import Image
    out = np.empty((len(samples), 3, image_size, image_size), dtype=np.float32)
    for k, ((path, _), flip) in enumerate(zip(samples, flips)):
        with Image.open(path) as im:
            a = np.asarray(center_crop_arr(im.convert("RGB"), image_size), dtype=np.float32)
        if flip:
            a = a[:, ::-1]
        out[k] = (a / 255.0 - 0.5).transpose(2, 0, 1) / 0.5
    return torch.from_numpy(out)


def normalize_latents(z: torch.Tensor) -> torch.Tensor:
    mean = torch.tensor(LATENT_MEAN, device=z.device, dtype=z.dtype).view(1, -1, 1, 1)
    std2 = torch.tensor(LATENT_STD, device=z.device, dtype=z.dtype).view(1, -1, 1, 1) * 2
    return (z - mean) / std2


def denormalize_latents(x: torch.Tensor) -> torch.Tensor:
    """inverse of normalize_latents: what LatentsGenerateCallback applies before decoding (callbacks.py:103-105)"""
    mean = torch.tensor(LATENT_MEAN, device=x.device, dtype=x.dtype).view(1, -1, 1, 1)
    std2 = torch.tensor(LATENT_STD, device=x.device, dtype=x.dtype).view(1, -1, 1, 1) * 2
    return x * std2 + mean


def sd_vae_encoder(vae: str = "ema", device=None) -> Callable[[torch.Tensor], torch.Tensor]:
    """The reference's encoder: AutoencoderKL(stabilityai/sd-vae-ft-<vae>).encode(x).latent_dist.sample()"""
    try:
        from diffusers.models import AutoencoderKL
    except ImportError as e:
        raise RuntimeError("extract_latents: the SD-VAE encoder needs `diffusers` (and its hub weights); pass your own "
                           "`encoder=` callable instead") from e
    net = AutoencoderKL.from_pretrained(f"stabilityai/sd-vae-ft-{vae}").to(device).eval()

    @torch.no_grad()
    def encode(x):
        return net.encode(x).latent_dist.sample()
    return encode


@torch.no_grad()
def extract(data_dir, out_dir, image_size: int = 256, batch_size: int = 32, encoder: Optional[Callable] = None,
            seed: int = 42, device=None, vae: str = "ema", flip: bool = True) -> int:
    """Writes `<out_dir>/latents/<i>.npy` (float32 (4, S/8, S/8), normalised) and `<out_dir>/labels/<i>.npy` (int64 class
    index) for every image of every FULL batch (the reference's loader drops the last partial batch); returns the count."""
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    if encoder is None:
        encoder = sd_vae_encoder(vae, device)
    samples, _classes = list_image_folder(data_dir)
    out_dir = Path(out_dir)
    (out_dir / "latents").mkdir(parents=True, exist_ok=True)
    (out_dir / "labels").mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    n = 0
    for b0 in range(0, len(samples) - batch_size + 1, batch_size):
        chunk = samples[b0:b0 + batch_size]
        flips = rng.random(len(chunk)) < 0.5 if flip else np.zeros(len(chunk), dtype=bool)
        x = load_batch(chunk, image_size, flips).to(device)
        z = normalize_latents(encoder(x).float())
        for k, (_, label) in enumerate(chunk):
            np.save(out_dir / "latents" / f"{b0 + k}.npy", z[k].cpu().numpy().astype(np.float32))
            np.save(out_dir / "labels" / f"{b0 + k}.npy", np.asarray(label, dtype=np.int64))
        n += len(chunk)
    return n


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--data_dir", type=str, required=True)
    ap.add_argument("--out_dir", type=str, required=True)
    ap.add_argument("--image_size", type=int, default=256)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--num_workers", type=int, default=0, help="accepted for CLI compatibility (images are read in-process)")
    ap.add_argument("--vae", type=str, choices=["ema", "mse"], default="ema")
    a = ap.parse_args(argv)
    n = extract(a.data_dir, a.out_dir, a.image_size, a.batch_size, seed=a.seed, vae=a.vae)
    print(f"wrote {n} latents to {a.out_dir}")


if __name__ == "__main__":
    main()

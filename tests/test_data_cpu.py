"""CPU-side checks of the data formats either side of the hot path: the byte<->float oracle on
every byte value, the reference's closed-form expressions, and the on-disk readers (CIFAR-10 pickles, MNIST idx)
on tiny fabricated datasets."""
import os

import numpy as np
import torch

from oracle import data_oracle as DO


def test_normalize_denormalize_all_byte_values_roundtrip():
    u8 = torch.arange(256, dtype=torch.uint8).view(1, 1, 16, 16)
    x = DO.normalize_u8(u8)
    assert x.min().item() == -1.0 and x.max().item() == 1.0
    # reference expression, written out: ((v/255) - 0.5)/0.5
    v = torch.arange(256, dtype=torch.float32)
    assert torch.equal(x.flatten(), (v / 255.0 - 0.5) / 0.5)
    # denormalize(normalize(v)) == v for all 256 values: (2v/255-1)*127.5+128 = v + 0.5 -> truncation
    assert torch.equal(DO.denormalize(x), u8)


def test_denormalize_clips_and_truncates():
    x = torch.tensor([-5.0, -1.0, -0.999, 0.0, 0.003, 0.99, 1.0, 7.0])
    # x*127.5+128 = -509.5, 0.5, 0.627, 128, 128.38, 254.225, 255.5, 1020.5 -> clip -> truncate
    assert DO.denormalize(x).tolist() == [0, 0, 0, 128, 128, 254, 255, 255]


def test_flip_is_per_sample():
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (4, 3, 5, 7), dtype=torch.uint8, generator=g)
    flip = torch.tensor([True, False, True, False])
    x = DO.normalize_u8(u8, flip=flip)
    ref = DO.normalize_u8(u8)
    assert torch.equal(x[0], ref[0].flip(-1)) and torch.equal(x[1], ref[1])


def test_prediction_to_u8_matches_written_out_expression():
    g = torch.Generator().manual_seed(1)
    pred = torch.randn(2, 3, 4, 5, generator=g)
    out = DO.prediction_to_u8_nhwc(pred, [0.5, 0.4, 0.6], [0.25, 0.2, 0.3])
    assert out.shape == (2, 4, 5, 3) and out.dtype == torch.uint8
    c = 1
    ref = (torch.clamp(pred[:, c] * 0.2 * 2 + 0.4, 0, 1) * 255).to(torch.uint8)
    assert torch.equal(out[..., c], ref)


def test_cifar_and_mnist_readers_roundtrip(tmp_path):
    from tinyedm_amd import datamodules as DM
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (70, 3, 32, 32), dtype=np.uint8)
    lab = rng.integers(0, 10, 70)
    DO.write_cifar10_batches(str(tmp_path), img, lab, n_train=50)
    xtr, ytr = DM.read_cifar10(str(tmp_path), True)
    xte, yte = DM.read_cifar10(str(tmp_path), False)
    assert np.array_equal(xtr, img[:50]) and np.array_equal(ytr, lab[:50])
    assert np.array_equal(xte, img[50:]) and np.array_equal(yte, lab[50:])
    m = rng.integers(0, 256, (30, 28, 28), dtype=np.uint8)
    ml = rng.integers(0, 10, 30)
    DO.write_mnist_idx(str(tmp_path), m, ml, n_train=20)
    xm, ym = DM.read_mnist(str(tmp_path), True)
    assert xm.shape == (20, 1, 28, 28) and np.array_equal(xm[:, 0], m[:20]) and np.array_equal(ym, ml[:20])
    xt, yt = DM.read_mnist(str(tmp_path), False)
    assert np.array_equal(xt[:, 0], m[20:]) and np.array_equal(yt, ml[20:])


def test_missing_dataset_raises_instead_of_downloading(tmp_path):
    import pytest
    from tinyedm_amd import datamodules as DM
    with pytest.raises(FileNotFoundError):
        DM.CIFAR10DataModule(str(tmp_path), 32, 4, device="cpu").prepare_data()


def test_make_grid_layout():
    from tinyedm_amd.callbacks import make_grid_u8
    x = torch.arange(5 * 1 * 2 * 3, dtype=torch.uint8).view(5, 1, 2, 3)
    g = make_grid_u8(x, nrow=4, padding=1)
    assert g.shape == (1 + 2 * 3, 1 + 4 * 4, 1)
    assert np.array_equal(g[1:3, 1:4, 0], x[0, 0].numpy()) and np.array_equal(g[4:6, 1:4, 0], x[4, 0].numpy())


def test_extract_latents_layout_and_normalisation(tmp_path):
    """tinyedm_amd.extract_latents (reference datamodules/extract_latents.py:14-125) with a stand-in encoder: ImageFolder
    order, ADM centre crop to the target size, [-1, 1] range, the reference's latent normalisation constants, one .npy per
    latent / label in the layout ImageNetLatentsDataModule reads, last partial batch dropped."""
    import numpy as np
    import torch
    from PIL import Image
    from tinyedm_amd import extract_latents as E
    rng = np.random.default_rng(0)
    sizes = {"n01": [(70, 50), (40, 90), (33, 33)], "n02": [(64, 64), (130, 70)]}
    for cls, ss in sizes.items():
        (tmp_path / "img" / cls).mkdir(parents=True)
        for k, (w, h) in enumerate(ss):
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / "img" / cls / f"im{k}.png")
    samples, classes = E.list_image_folder(tmp_path / "img")
    assert classes == ["n01", "n02"] and [c for _, c in samples] == [0, 0, 0, 1, 1]
    assert [os.path.basename(p) for p, _ in samples] == ["im0.png", "im1.png", "im2.png", "im0.png", "im1.png"]
    seen = []

    def encoder(x):                       # a fixed "VAE": 8x8 average pooling of the three channels + their mean
        seen.append(x.clone())
        p = torch.nn.functional.avg_pool2d(x, 8)
        return torch.cat([p, p.mean(1, keepdim=True)], 1)

    n = E.extract(tmp_path / "img", tmp_path / "lat", image_size=32, batch_size=2, encoder=encoder, seed=1, device="cpu")
    assert n == 4                         # 5 images, batch 2: the partial last batch is dropped (drop_last=True)
    x = torch.cat(seen)
    assert x.shape == (4, 3, 32, 32) and x.min() >= -1 and x.max() <= 1
    # sample 2 is already 33x33: crop only -> pixel-exact centre crop (possibly flipped)
    with Image.open(samples[2][0]) as im:
        ref = (np.asarray(E.center_crop_arr(im.convert("RGB"), 32), dtype=np.float32) / 255 - 0.5) / 0.5
    got = x[2].permute(1, 2, 0).numpy()
    assert np.allclose(got, ref, atol=1e-6) or np.allclose(got, ref[:, ::-1], atol=1e-6)
    z = np.load(tmp_path / "lat" / "latents" / "2.npy")
    assert z.shape == (4, 4, 4) and z.dtype == np.float32
    raw = encoder(x[2:3])[0].numpy()
    mean, std = np.array(E.LATENT_MEAN).reshape(4, 1, 1), np.array(E.LATENT_STD).reshape(4, 1, 1)
    assert np.allclose(z, (raw - mean) / (2 * std), atol=1e-6)
    assert np.allclose(E.denormalize_latents(torch.from_numpy(z)[None])[0].numpy(), raw, atol=1e-5)
    labels = [int(np.load(tmp_path / "lat" / "labels" / f"{i}.npy")) for i in range(4)]
    assert labels == [0, 0, 0, 1]
    assert not (tmp_path / "lat" / "latents" / "4.npy").exists()

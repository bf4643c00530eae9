"""Bit-exact parity of the byte<->float kernels (csrc/data.hip) with the CPU oracle, the size-independent round trip
at the full CIFAR-10 size, and the resident datamodule / PNG-writer plumbing around the sampler."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data_oracle as DO

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tinyedm_amd import ops as o
    return o


@pytest.mark.parametrize("shape", [(7, 3, 32, 32), (5, 1, 28, 28), (3, 4, 5, 9)])
def test_gather_normalize_bit_exact(ops, shape):
    g = torch.Generator().manual_seed(shape[1])
    N = 23
    data = torch.randint(0, 256, (N,) + shape[1:], dtype=torch.uint8, generator=g)
    idx = torch.randint(0, N, (shape[0],), generator=g)
    out = ops.u8_gather_normalize(data.to(DEV), idx.to(DEV))
    assert torch.equal(out.cpu(), DO.normalize_u8(data[idx]))
    # flips: each sample is either the plain or the mirrored image, and both occur over a larger batch
    idx2 = torch.randint(0, N, (64,), generator=g)
    f = ops.u8_gather_normalize(data.to(DEV), idx2.to(DEV), flip=True, seed=5, epoch=3).cpu()
    ref = DO.normalize_u8(data[idx2])
    plain = torch.tensor([torch.equal(f[i], ref[i]) for i in range(64)])
    mirrored = torch.tensor([torch.equal(f[i], ref[i].flip(-1)) for i in range(64)])
    assert bool((plain | mirrored).all()) and 12 <= int(mirrored.sum()) <= 52
    f2 = ops.u8_gather_normalize(data.to(DEV), idx2.to(DEV), flip=True, seed=5, epoch=3).cpu()
    assert torch.equal(f, f2)                                  # counter-based: replayable


def test_denormalize_bit_exact_including_edges(ops):
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(100000, generator=g) * 0.7, torch.linspace(-1.2, 1.2, 20001),
                   (torch.arange(256).float() / 255 - 0.5) / 0.5, torch.tensor([-1e9, 1e9, 0.0, -0.0])])
    assert torch.equal(ops.denormalize_u8(x.to(DEV)).cpu(), DO.denormalize(x))


def test_prediction_to_u8_bit_exact(ops):
    g = torch.Generator().manual_seed(2)
    pred = torch.randn(6, 3, 32, 32, generator=g)
    mean, std = [0.5, 0.45, 0.55], [0.25, 0.2, 0.3]
    got = ops.prediction_to_u8_nhwc(pred.to(DEV), torch.tensor(mean, device=DEV), torch.tensor(std, device=DEV))
    assert torch.equal(got.cpu(), DO.prediction_to_u8_nhwc(pred, mean, std))


def test_full_size_roundtrip_is_identity(ops):
    """All 50 000 x 3 x 32 x 32 bytes of a CIFAR-10-sized set: denormalize(normalize(u8)) == u8."""
    g = torch.Generator(device=DEV).manual_seed(0)
    data = torch.randint(0, 256, (50000, 3, 32, 32), dtype=torch.uint8, device=DEV, generator=g)
    for lo in range(0, 50000, 12500):
        idx = torch.arange(lo, lo + 12500, device=DEV)
        x = ops.u8_gather_normalize(data, idx)
        assert torch.equal(ops.denormalize_u8(x), data[lo:lo + 12500])


def test_resident_cifar_datamodule_and_png_writer(ops, tmp_path):
    from tinyedm_amd import datamodules as DM
    from tinyedm_amd.callbacks import PreditionWriter
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (48, 3, 32, 32), dtype=np.uint8)
    lab = rng.integers(0, 10, 48)
    DO.write_cifar10_batches(str(tmp_path), img, lab, n_train=40)
    dm = DM.CIFAR10DataModule(str(tmp_path), 32, batch_size=16, device=DEV)
    dm.prepare_data()
    dm.setup("fit")
    seen = []
    for x, y in dm.train_dataloader():
        assert x.dtype == torch.float32 and x.shape[1:] == (3, 32, 32) and y.dtype == torch.int64
        u8 = dm.denormalize(x).cpu().numpy()
        for i in range(u8.shape[0]):
            # every emitted sample is one of the training images (possibly mirrored) with its own label
            cand = [j for j in range(40) if lab[j] == int(y[i]) and (np.array_equal(u8[i], img[j])
                                                                      or np.array_equal(u8[i], img[j][:, :, ::-1]))]
            assert cand
            seen.append(cand[0])
    assert sorted(seen) == list(range(40))                     # a permutation: every image exactly once per epoch
    w = PreditionWriter(str(tmp_path / "png"), "batch", [0.5] * 3, [0.25] * 3)

    class M:
        device = torch.device(DEV)
    pred = torch.from_numpy(img[:4].astype(np.float32) / 127.5 - 1.0).to(DEV)
    w.write_on_batch_end(None, M(), pred, None, None, 0, 0)
    from PIL import Image
    back = np.asarray(Image.open(tmp_path / "png" / "2.png"))
    assert back.shape == (32, 32, 3)
    assert np.abs(back.astype(int) - np.transpose(img[2], (1, 2, 0)).astype(int)).max() <= 1


def test_generate_callback_inside_fit_and_checkpoint_roundtrip(ops, tmp_path):
    """in-training sampling through swap_ema_weights (callbacks.py:12-58) restores the training
    weights; a saved checkpoint reloads (optionally with its EMA weights) into a model that samples identically."""
    import tinyedm_amd as T
    from tinyedm_amd.callbacks import GenerateCallback
    from tinyedm_amd.datamodules import SyntheticImageDataModule
    from oracle.make_golden import tiny_cfgs
    ecfg, dcfg = tiny_cfgs()
    T.manual_seed(3)
    torch.manual_seed(3)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                     tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                     tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), 0.0, dcfg.sigma_data,
                     dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim, dcfg.num_heads)
    model = T.EDM(diffuser=T.Diffuser(-1.2, 1.2), embedding=emb, denoiser=den, use_ema=True, use_uncertainty=False,
                  steady_steps=10, rampup_steps=2, scheduler_interval="step", lr=1e-3, ema_length=0.13)
    cb = GenerateCallback(T.DeterministicSolver(num_steps=3), (3, 16, 16), num_samples=4, every_n_epochs=1,
                          output_dir=str(tmp_path / "gen"))
    dm = SyntheticImageDataModule(8, (3, 16, 16), num_classes=ecfg.num_classes, num_samples=24)
    tr = T.Trainer(max_epochs=1, callbacks=[cb], check_val_every_n_epoch=100)
    seen = {}
    orig = cb.on_train_epoch_end

    def wrapped(trainer, pl_module):
        before = [p.detach().clone() for p in pl_module.parameters()]
        orig(trainer, pl_module)
        seen["same"] = all(torch.equal(a, b) for a, b in zip(before, pl_module.parameters()))
    cb.on_train_epoch_end = wrapped
    tr.fit(model, datamodule=dm)
    assert seen["same"], "swap_ema_weights must restore the training weights"
    assert cb.last_grid is not None and cb.last_grid.dtype == np.uint8
    assert os.path.exists(tmp_path / "gen" / "epoch_00000.png")
    assert model.training
    # checkpoint in the reference's key layout -> load_from_checkpoint (edm.py:159-203), with and without EMA weights
    path = str(tmp_path / "last.ckpt")
    ck = tr.save_checkpoint(path)
    assert {"state_dict", "hyper_parameters", "optimizer_states"} <= set(ck) and "ema" in ck["optimizer_states"][0]
    x0 = torch.randn(4, 3, 16, 16, device=DEV)
    lab = torch.arange(4, device=DEV) % ecfg.num_classes
    sol = T.DeterministicSolver(num_steps=3)
    model.eval()
    with torch.no_grad():
        ref_plain = sol.solve(model, x0, lab)
        with model.swap_ema_weights(tr):
            ref_ema = sol.solve(model, x0, lab)
    m1 = T.EDM.load_from_checkpoint(path).to(DEV).eval()
    m2 = T.EDM.load_from_checkpoint(path, load_ema=True).to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(sol.solve(m1, x0, lab), ref_plain)
        assert torch.equal(sol.solve(m2, x0, lab), ref_ema)
    assert not torch.equal(ref_plain, ref_ema)


@pytest.mark.parametrize("size", [32, 64])
def test_latent_pipeline_datamodule_fit_and_generate_callback(ops, tmp_path, size):
    """SURVEY 8f row 4 (reference datamodules/imagenet_latents_datamodule.py:8-50, callbacks.py:61-123): a latent directory
    in the layout extract_latents writes -> ImageNetLatentsDataModule (resident, rank-sharded loader) -> two optimisation
    steps of a latent-space EDM (4 channels, class-conditional) -> LatentsGenerateCallback at the validation epoch end:
    Heun-samples with the EMA weights swapped in and de-normalises the latents (x * 2 std + mean); the SD-VAE decode is
    skipped (diffusers is not installed) and the de-normalised latents are written instead.  4x32x32 and 4x64x64."""
    import tinyedm_amd as T
    from tinyedm_amd import extract_latents as E
    from tinyedm_amd.callbacks import LatentsGenerateCallback
    from tinyedm_amd.datamodules import ImageNetLatentsDataModule
    g = torch.Generator().manual_seed(size)
    n_train, n_val, ncls = 12, 4, 1000
    lat = {}
    for split, n in (("train", n_train), ("val", n_val)):
        (tmp_path / "lat" / split / "latents").mkdir(parents=True)
        (tmp_path / "lat" / split / "labels").mkdir(parents=True)
        lat[split] = (0.5 * torch.randn(n, 4, size, size, generator=g)).numpy().astype(np.float32)
        for i in range(n):
            np.save(tmp_path / "lat" / split / "latents" / f"{i}.npy", lat[split][i])
            np.save(tmp_path / "lat" / split / "labels" / f"{i}.npy", np.asarray((37 * i + 5) % ncls, dtype=np.int64))
    dm = ImageNetLatentsDataModule(str(tmp_path / "lat"), size, batch_size=6, num_workers=0)
    dm.setup("fit")
    assert dm.num_classes == 1000
    seen = []
    for xb, yb in dm.train_dataloader():
        assert xb.is_cuda and xb.dtype == torch.float32 and tuple(xb.shape[1:]) == (4, size, size) and yb.dtype == torch.int64
        for x, y in zip(xb.cpu().numpy(), yb.cpu().numpy()):
            j = [k for k in range(n_train) if np.array_equal(x, lat["train"][k])]
            assert len(j) == 1 and y == (37 * j[0] + 5) % ncls          # latent and label stay paired through the shuffle
            seen.append(j[0])
    assert sorted(seen) == list(range(n_train))
    # a small latent-space denoiser: two resolutions below the input so that attention runs on <= 256 tokens
    T.manual_seed(5)
    torch.manual_seed(5)
    emb = T.Embedding(32, 64, ncls)
    den = T.Denoiser(4, 4, ("Enc", "EncD", "Enc", "EncD", "EncA"), ("DecA", "Dec", "DecU", "Dec", "DecU", "Dec"),
                     (64, 64, 64, 64, 64), (64, 64, 64, 64, 64, 64), (False, True, True, True, True, True), 0.0, 0.5, 0.3,
                     0.3, 64, 2)
    with torch.no_grad():
        den.gain_out.fill_(0.5)
    model = T.EDM(diffuser=T.Diffuser(-0.4, 1.0), embedding=emb, denoiser=den, use_ema=True, use_uncertainty=False,
                  steady_steps=10, rampup_steps=2, scheduler_interval="step", lr=1e-3, ema_length=0.13)
    mean, std = E.LATENT_MEAN, E.LATENT_STD
    cb = LatentsGenerateCallback(T.DeterministicSolver(num_steps=3), (4, size, size), mean, std, num_samples_per_class=2,
                                 num_classes=3, every_n_epochs=1, output_dir=str(tmp_path / "gen"))
    tr = T.Trainer(max_epochs=1, callbacks=[cb], check_val_every_n_epoch=1)
    tr.fit(model, datamodule=dm)
    assert tr.global_step == 2                                           # 12 latents / batch 6
    assert cb.vae is None                                                # no diffusers here: decode skipped
    out = cb.last
    assert out is not None and tuple(out.shape) == (6, 4, size, size) and torch.isfinite(out).all()
    saved = np.load(tmp_path / "gen" / "latents_epoch_00000.npy")
    assert np.array_equal(saved, out.cpu().numpy())
    # de-normalisation == the inverse of extract_latents' normalisation, on the solver's output for the same x0 / labels.
    # Reference semantics, kept: the EMA callback has ALREADY swapped the EMA weights in for validation (ema.py:83-100) and
    # the callback's own `swap_ema_weights` (callbacks.py:110-113) swaps back, so this sample runs on the training weights
    model.eval()
    assert cb.network_dtype == "f32x3" and model.denoiser.eval_dtype == "bf16"  # the callback samples at fp32 accuracy (the
    model.denoiser.set_eval_dtype("f32x3")                                       # reference's) and restores the module's setting
    with torch.no_grad():
        xT = T.DeterministicSolver(num_steps=3).solve(model, cb.x0, cb.class_labels)
        with model.swap_ema_weights(tr):
            xT_ema = T.DeterministicSolver(num_steps=3).solve(model, cb.x0, cb.class_labels)
    model.denoiser.set_eval_dtype("bf16")
    assert torch.allclose(out, E.denormalize_latents(xT.float()), rtol=1e-5, atol=1e-5)
    assert not torch.allclose(out, E.denormalize_latents(xT_ema.float()), rtol=1e-5, atol=1e-5)
    assert "val_loss" in tr.callback_metrics and np.isfinite(tr.callback_metrics["val_loss"])

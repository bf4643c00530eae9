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

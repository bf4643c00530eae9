"""The Lightning-shaped surface around the hot path, on the GPU: Trainer.fit with checkpoint resume
(reference experiments/train.py:30-33), Trainer.validate with the EMA swap (ema.py:83-123, edm.py:238-248),
Trainer.predict + PreditionWriter (edm.py:288-295, callbacks.py:126-156), the `generate` CLI with the reference's
flags (generate.py:50-96), DenoiserWrapper (networks.py:608-646) and the ModelCheckpoint stand-in."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data_oracle as DO
from oracle import edm_oracle as O
from oracle.make_golden import tiny_cfgs
from parity_log import record

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def build_model(P=None, use_ema=True, interval="step", pdrop=0.0, seed=7):
    import tinyedm_amd as T
    from tinyedm_amd import networks as N
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ecfg, dcfg = tiny_cfgs()
    if P is None:
        P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(seed))
    N._rng_sub_counter[0] = 0
    T.manual_seed(11)
    torch.manual_seed(11)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                     tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                     tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), pdrop, dcfg.sigma_data,
                     dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim, dcfg.num_heads)
    emb.load_state_dict({k[len("embedding."):]: v for k, v in P.items() if k.startswith("embedding.")})
    den.load_state_dict({k[len("denoiser."):]: v for k, v in P.items() if k.startswith("denoiser.")})
    model = T.EDM(diffuser=T.Diffuser(-1.2, 1.2), embedding=emb, denoiser=den, use_ema=use_ema, use_uncertainty=False,
                  steady_steps=2, rampup_steps=2, scheduler_interval=interval, lr=2e-3,
                  ema_length=0.13 if use_ema else None)
    return model, ecfg, dcfg, P


class Batches:
    """deterministic loader: n distinct batches per epoch"""

    def __init__(self, n=4, B=4, hw=8, seed=1):
        g = torch.Generator().manual_seed(seed)
        self.b = [(0.5 * torch.randn(B, 3, hw, hw, generator=g), torch.randint(0, 10, (B,), generator=g)) for _ in range(n)]

    def __len__(self):
        return len(self.b)

    def __iter__(self):
        return iter(self.b)


def _state(trainer):
    from tinyedm_amd.ema import EMAOptimizer
    opt = trainer.optimizers[0]
    assert isinstance(opt, EMAOptimizer)
    base = opt.optimizer
    return {"theta": base.arena.theta.clone(), "m": base.m.clone(), "v": base.v.clone(), "ema": opt.ema_arena.clone(),
            "step": base.step_count, "ema_step": opt.current_step}


@pytest.mark.parametrize("graph", ["0", "1"])
def test_fit_resumes_from_checkpoint(tmp_path, monkeypatch, graph):
    """3 uninterrupted steps == 2 steps, save, fresh process state, resume, 1 step: weights, Adam moments, EMA, LR
    schedule position, step counters and the Philox position all come back (mid-epoch checkpoint)."""
    import tinyedm_amd as T
    monkeypatch.setenv("EDM_GRAPH", graph)
    data = Batches(n=5)
    mA, *_ = build_model()
    tA = T.Trainer(max_epochs=1, max_steps=3)
    tA.fit(mA.to(DEV), train_dataloaders=data)
    A = _state(tA)
    lrA = tA.lr_scheduler.get_last_lr()[0]

    mB, *_ = build_model()
    tB = T.Trainer(max_epochs=1, max_steps=2)
    tB.fit(mB.to(DEV), train_dataloaders=data)
    path = str(tmp_path / "mid.ckpt")
    tB._batch_in_epoch = 2
    tB.save_checkpoint(path)
    ck = torch.load(path, weights_only=False)
    assert {"state_dict", "hyper_parameters", "optimizer_states", "lr_schedulers", "epoch", "global_step"} <= set(ck)
    assert ck["global_step"] == 2 and "ema" in ck["optimizer_states"][0]

    T.manual_seed(999)                                  # the resumed run must not depend on leftover RNG state
    mC, *_ = build_model(seed=8)                        # different initial weights: everything comes from the file
    tC = T.Trainer(max_epochs=1, max_steps=3)
    tC.fit(mC.to(DEV), train_dataloaders=data, ckpt_path=path)
    C = _state(tC)
    assert tC.global_step == 3 and C["step"] == A["step"] == 3 and C["ema_step"] == A["ema_step"]
    assert abs(tC.lr_scheduler.get_last_lr()[0] - lrA) <= 1e-12
    for k, lim in (("theta", 1e-4), ("m", 5e-3), ("v", 5e-3), ("ema", 1e-4)):
        e = rel(C[k], A[k])
        record(f"resume[graph={graph}]/{k}", e, lim)
        assert e <= lim, f"{k}: resumed run deviates from the uninterrupted one by {e:.2e}"


def test_load_state_accepts_torch_adam_layout():
    """reference checkpoints hold torch.optim.Adam's {state: {i: {step, exp_avg, exp_avg_sq}}, param_groups}"""
    model, *_ = build_model()
    model = model.to(DEV)
    base = model.configure_optimizers()["optimizer"]
    params = base.arena.params
    g = torch.Generator().manual_seed(2)
    sd = {"state": {i: {"step": torch.tensor(5.0), "exp_avg": torch.randn(p.shape, generator=g),
                        "exp_avg_sq": torch.rand(p.shape, generator=g)} for i, p in enumerate(params)},
          "param_groups": [{"lr": 0.01, "betas": (0.9, 0.999), "eps": 1e-8, "params": list(range(len(params)))}]}
    base.load_state_dict(sd)
    assert base.step_count == 5 and base.param_groups[0]["lr"] == 0.01
    for i, (p, o) in enumerate(zip(params, base.arena.offsets)):
        assert torch.equal(base.m[o:o + p.numel()].view_as(p).cpu(), sd["state"][i]["exp_avg"])
        assert torch.equal(base.v[o:o + p.numel()].view_as(p).cpu(), sd["state"][i]["exp_avg_sq"])
    with pytest.raises(KeyError):
        base.load_state_dict({"foo": 1, "param_groups": []})


def test_validate_swaps_ema_weights_and_matches_oracle():
    """Trainer.validate: eval mode, EMA weights in place of the masters during validation_step (unless
    validate_original_weights), masters back afterwards, val_loss == the oracle's sigma-weighted MSE of the EMA net."""
    import tinyedm_amd as T
    model, ecfg, dcfg, P = build_model()
    model = model.to(DEV)
    g = torch.Generator().manual_seed(3)
    B = 4
    clean = 0.5 * torch.randn(B, 3, 8, 8, generator=g)
    labels = torch.randint(0, 10, (B,), generator=g)
    eps, noise = torch.randn(B, generator=g), torch.randn(B, 3, 8, 8, generator=g)

    class Fixed(torch.nn.Module):
        def forward(self, x):
            noisy, sigma = O.diffuse(x.cpu(), eps, noise, -1.2, 1.2)
            return noisy.to(DEV), sigma.to(DEV)
    model.diffuser = Fixed()
    trainer = T.Trainer()
    trainer._configure(model)
    trainer._call("on_fit_start", model)                      # EMA callback wraps the optimizer: EMA = current weights = P
    opt = trainer.optimizers[0]
    base = opt.optimizer
    with torch.no_grad():                                     # masters drift away from the EMA
        base.arena.theta.mul_(0.9)
    T.networks.bump_weight_epoch()
    masters = base.arena.theta.clone()
    seen = {}
    orig = model.validation_step

    def spy(batch, idx):
        seen["training"] = model.training or model.denoiser.training
        seen["theta"] = base.arena.theta.clone()
        return orig(batch, idx)
    model.validation_step = spy
    val = trainer.validate(model, [(clean, labels)])
    assert seen["training"] is False
    assert torch.equal(seen["theta"], opt.ema_arena)                   # validation ran on exactly the EMA copy (= P)
    assert not torch.equal(seen["theta"], masters) and rel(seen["theta"], masters / 0.9) <= 1e-6
    assert torch.equal(base.arena.theta, masters)                      # masters restored
    assert model.training
    ref = O.training_loss(P, ecfg, dcfg, clean, eps, noise, -1.2, 1.2, labels, bf16=True, normalize_weights=False).item()
    e = abs(float(val) - ref) / abs(ref)
    record("validate/val_loss_vs_oracle", e, 2e-2)
    assert e <= 2e-2, (float(val), ref)
    assert abs(trainer.callback_metrics["val_loss"] - float(val)) < 1e-12


def test_predict_writes_pngs_bit_exact(tmp_path):
    """Trainer.predict -> EDM.predict_step -> solver.solve -> PreditionWriter: one PNG per sample whose bytes are the
    oracle's uint8 conversion (callbacks.py:141-153) of the returned predictions."""
    from PIL import Image
    import tinyedm_amd as T
    from tinyedm_amd.callbacks import PreditionWriter
    from tinyedm_amd.datamodules import RandomNoiseDataModule
    model, *_ = build_model()
    with torch.no_grad():
        model.denoiser.gain_out.fill_(0.5)
    model = model.to(DEV)
    model.solver = T.DeterministicSolver(num_steps=4)
    dm = RandomNoiseDataModule(3, 0, 8, 7, 10)               # reference positional order
    assert (dm.batch_size, dm.num_workers, dm.image_size, dm.num_samples, dm.num_classes) == (3, 0, 8, 7, 10)
    mean, std = (0.49, 0.48, 0.45), (0.25, 0.24, 0.26)
    writer = PreditionWriter(str(tmp_path), "batch", mean, std)
    outs = T.Trainer(callbacks=[writer]).predict(model, datamodule=dm, distributed=False)
    assert [o.shape[0] for o in outs] == [3, 3, 1]
    files = sorted(glob.glob(str(tmp_path / "*.png")), key=lambda p: int(os.path.basename(p)[:-4]))
    assert [os.path.basename(f) for f in files] == [f"{i}.png" for i in range(7)]
    pred = torch.cat([o.float().cpu() for o in outs])
    exp = DO.prediction_to_u8_nhwc(pred, mean, std).numpy()
    for i, f in enumerate(files):
        assert np.array_equal(np.asarray(Image.open(f)), exp[i]), f"image {i} differs from the oracle conversion"


def test_generate_cli_with_reference_flags(tmp_path):
    """`python experiments/generate.py --ckpt_path ... --load_ema --output_dir ... --num_samples --image_size
    --num_classes --batch_size --num_steps` (reference generate.py:50-96) end to end from a checkpoint file."""
    import tinyedm_amd as T
    model, *_ = build_model()
    model = model.to(DEV)
    trainer = T.Trainer(max_epochs=1, max_steps=2)
    trainer.fit(model, train_dataloaders=Batches(n=3))
    ckpt = str(tmp_path / "m.ckpt")
    trainer.save_checkpoint(ckpt)
    out = tmp_path / "gen"
    cmd = [sys.executable, os.path.join(ROOT, "experiments", "generate.py"), "--ckpt_path", ckpt, "--load_ema",
           "--output_dir", str(out), "--num_samples", "5", "--image_size", "8", "--num_classes", "10", "--batch_size", "4",
           "--num_workers", "0", "--num_steps", "3"]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    from PIL import Image
    files = sorted(os.listdir(out))
    assert files == sorted(f"{i}.png" for i in range(5))
    assert Image.open(out / "0.png").size == (8, 8)


def test_denoiser_wrapper_matches_closed_form():
    """networks.py:608-646: D = c_skip*x + c_out*net(c_in*x, ln(sigma)/4, emb) with the oracle's scalars"""
    import tinyedm_amd as T

    class Net(torch.nn.Module):
        def forward(self, x, c_noise, emb):
            return 0.3 * x + c_noise.view(-1, 1, 1, 1) + (0.0 if emb is None else emb.view(-1, 1, 1, 1))
    w = T.DenoiserWrapper(Net(), sigma_data=0.5).to(DEV)
    assert w.sigma_data == 0.5
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 3, 8, 8, generator=g)
    sigma = torch.randn(5, generator=g).exp()
    emb = torch.randn(5, generator=g)
    got = w(x.to(DEV), sigma.to(DEV), emb.to(DEV)).cpu()
    c_skip, c_out, c_in = O.precond_scalars(sigma, 0.5)
    F_ = 0.3 * (c_in * x) + (sigma.log() / 4).view(-1, 1, 1, 1) + emb.view(-1, 1, 1, 1)
    assert torch.allclose(got, c_skip * x + c_out * F_, rtol=1e-5, atol=1e-6)
    assert torch.allclose(w(x.to(DEV), sigma.to(DEV)).cpu(), c_skip * x + c_out * (F_ - emb.view(-1, 1, 1, 1)), rtol=1e-5, atol=1e-6)


def test_model_checkpoint_callback_in_fit(tmp_path):
    """the reference YAML's checkpoint_callback (monitor val_loss, top-k, save_last) driven by Trainer.fit"""
    import tinyedm_amd as T
    from tinyedm_amd.callbacks import ModelCheckpoint
    model, *_ = build_model(interval="epoch")
    model = model.to(DEV)
    cb = ModelCheckpoint(dirpath=str(tmp_path), monitor="val_loss", mode="min", save_top_k=2, save_last=True,
                         every_n_epochs=1, save_on_train_epoch_end=False)
    tr = T.Trainer(max_epochs=4, check_val_every_n_epoch=1, callbacks=[cb])
    tr.fit(model, train_dataloaders=Batches(n=2), val_dataloaders=Batches(n=1, seed=5))
    names = sorted(os.listdir(tmp_path))
    assert "last.ckpt" in names and len([n for n in names if n.startswith("epoch=")]) == 2
    assert cb.best_model_path in [str(tmp_path / n) for n in names] and cb.best_model_score == min(cb.best_k_models.values())
    # a checkpoint written at an epoch end resumes with the next epoch, and loads through the reference's API
    m2 = T.EDM.load_from_checkpoint(str(tmp_path / "last.ckpt"), load_ema=True)
    assert isinstance(m2, T.EDM)
    tr2 = T.Trainer(max_epochs=5, check_val_every_n_epoch=100)
    model3, *_ = build_model(interval="epoch")
    tr2.fit(model3.to(DEV), train_dataloaders=Batches(n=2), ckpt_path=str(tmp_path / "last.ckpt"))
    assert tr2.current_epoch == 4 and tr2.global_step == 10


def test_train_script_runs_the_mnist_config(tmp_path):
    """BASELINE configs[0] plumbing: `experiments/train.py --config-name=mnist` (reference experiments/train.py:8-36)
    end to end -- Hydra-style compose + overrides, datamodule, model (87 M parameters), callbacks from the YAML
    (ModelCheckpoint stand-in, GenerateCallback), Trainer.fit for a few steps, one validation pass, a checkpoint and a
    sample grid on disk.  (The reference runs this config on CPU in fp32; this build has no CPU path by design.)"""
    out = tmp_path / "run"
    out.mkdir()
    cmd = [sys.executable, os.path.join(ROOT, "experiments", "train.py"), "--config-name=mnist",
           "trainer.max_epochs=1", "+trainer.max_steps=4", "trainer.check_val_every_n_epoch=1", "datamodule.batch_size=16",
           "datamodule.num_samples=64", f"callbacks.checkpoint_callback.dirpath={out / 'ckpt'}",
           "callbacks.checkpoint_callback.every_n_epochs=1", "callbacks.generate_callback.num_samples=4",
           f"+callbacks.generate_callback.output_dir={out / 'gen'}", "callbacks.generate_callback.solver.num_steps=3"]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(out))
    assert r.returncode == 0, r.stderr[-3000:]
    assert (out / "ckpt" / "epoch=0-step=4.ckpt").exists(), os.listdir(out / "ckpt")
    assert (out / "gen" / "epoch_00000.png").exists()
    import tinyedm_amd as T
    m = T.EDM.load_from_checkpoint(str(out / "ckpt" / "epoch=0-step=4.ckpt"))
    assert m.denoiser.in_channels == 1 and m.conditional


def test_the_training_step_learns_a_small_dataset():
    """End-to-end signal that forward, backward, grouped weight gradients, Adam+EMA and the captured step fit together:
    the CIFAR-10 network on a four-pattern dataset (tools/soak.py: 200 eager steps with the weight-gradient side stream,
    then 200 replays of the captured step) must bring the loss from ~1.0 to below 0.6."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "400"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    last = [l for l in r.stdout.splitlines() if l.startswith("SOAK OK")][-1].split()
    first, final = float(last[2]), float(last[4])
    assert first > 0.9 and final < 0.6, (first, final)


def test_generate_cli_reference_precision(tmp_path):
    """`python -m tinyedm.generate ...` (default, = `--network_dtype f32`): the sampler evaluates the denoiser through the
    exact-fp32 kernels (the reference samples in fp32, generate.py:39-44); the images of the opt-in bf16 fast mode differ
    by at most a few grey levels on a handful of pixels, and every run writes every index"""
    import numpy as np
    from PIL import Image
    outs = {}
    for dt in ("bf16", "f32", "f32x3", "default"):
        out = tmp_path / dt
        r = subprocess.run([sys.executable, "-m", "tinyedm.generate", "--config_name", "cifar10", "--output_dir", str(out),
                            "--num_samples", "6", "--image_size", "32", "--num_classes", "10", "--batch_size", "4",
                            "--num_steps", "4"] + ([] if dt == "default" else ["--network_dtype", dt]),
                           capture_output=True, text=True, cwd=ROOT, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
        assert r.returncode == 0, r.stderr[-2000:]
        files = sorted(glob.glob(str(out / "*.png")), key=lambda f: int(os.path.basename(f)[:-4]))
        assert [os.path.basename(f) for f in files] == [f"{i}.png" for i in range(6)]
        outs[dt] = np.stack([np.asarray(Image.open(f)).astype(int) for f in files])
    assert np.array_equal(outs["default"], outs["f32x3"])    # no flag = fp32-accurate evaluation (round 4), and it
    assert np.abs(outs["f32x3"] - outs["f32"]).max() <= 1    # rounds to the exact-fp32 path's bytes (a grey level at most)
    d = np.abs(outs["bf16"] - outs["f32"])
    assert d.max() <= 3 and (d > 0).mean() < 0.2, (d.max(), (d > 0).mean())


def test_fit_probes_eager_against_replay_and_trains_the_same(monkeypatch):
    """Round 5: with EDM_GRAPH unset `Trainer.fit` captures the step, then runs a block of steps through the eager loop and
    a block through the replay and keeps the faster form (`trainer.step_launch`).  Whatever it settles on, and however the
    two forms were mixed on the way, the run equals an all-eager and an all-replay run of the same steps (same counters,
    same Philox positions; weights up to the order of the fp32 atomics)."""
    import tinyedm_amd as T
    from tinyedm_amd import trainer as TR
    monkeypatch.setattr(TR._LaunchProbe, "PROBE_STEPS", 3)
    data = Batches(n=6)
    res = {}
    for mode in ("auto", "0", "1"):
        if mode == "auto":
            monkeypatch.delenv("EDM_GRAPH", raising=False)
        else:
            monkeypatch.setenv("EDM_GRAPH", mode)
        m, *_ = build_model()
        t = T.Trainer(max_epochs=3, max_steps=16)
        t.fit(m.to(DEV), train_dataloaders=data)
        res[mode] = (_state(t), t.step_launch, t.global_step)
    assert res["auto"][1] in ("hipGraph replay", "eager loop") and res["0"][1] == "eager loop" and res["1"][1] == "hipGraph replay"
    assert res["auto"][2] == res["0"][2] == res["1"][2] == 16
    for other in ("0", "1"):
        assert res["auto"][0]["step"] == res[other][0]["step"] and res["auto"][0]["ema_step"] == res[other][0]["ema_step"]
        for k, lim in (("theta", 2e-4), ("ema", 2e-4), ("m", 1e-2), ("v", 1e-2)):
            e = rel(res["auto"][0][k], res[other][0][k])
            record(f"fit_probe/auto_vs_EDM_GRAPH={other}/{k}", e, lim)
            assert e <= lim, (other, k, e)

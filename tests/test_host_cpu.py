"""CPU-only: C-ABI library loads and exports every declared symbol; host logic (config surface, arch tables,
LR schedule, EMA constants, sigma table, hparams round trip); product refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from tinyedm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tinyedm_hip.h")).read()
    declared = set(re.findall(r"\b(edm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(h, name), f"{name} declared in include/tinyedm_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.lib().edm_version() == 1


def test_no_cpu_fallback():
    import tinyedm_amd as T
    den = T.Denoiser(3, 3, ("Enc",), ("Dec", "Dec"), (64,), (64, 64), (True, True), embedding_dim=64, num_heads=1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        den(torch.randn(1, 3, 8, 8), torch.ones(1), torch.randn(1, 64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        T.Diffuser(-1.2, 1.2)(torch.randn(2, 3, 8, 8))
    with pytest.raises(RuntimeError):
        T.DeterministicSolver(4).solve(lambda *a: None, torch.randn(1, 3, 8, 8))


def test_arch_tables_match_reference(golden_dir):
    from tinyedm_amd import networks as N
    g = np.load(os.path.join(golden_dir, "tables.npz"))
    assert list(N.get_encoder_blocks_types()) == list(g["enc_types"])
    assert list(N.get_decoder_blocks_types()) == list(g["dec_types"])
    assert list(N.get_encoder_out_channels()) == list(g["enc_ch"])
    assert list(N.get_decoder_out_channels()) == list(g["dec_ch"])
    assert list(N.get_skip_connections()) == [bool(b) for b in g["skips"]]
    sc = N.get_skip_channels(N.get_encoder_out_channels(), N.get_decoder_out_channels(), N.get_skip_connections())
    assert list(sc) == list(g["skip_ch"])
    # reference tests/test_unet_builder.py:14-30
    assert len(N.get_decoder_out_channels()) == 21 and len(N.get_encoder_out_channels()) == 15
    assert len(N.get_skip_connections()) == 21 and len(sc) == 21


def test_config_surface_and_hparams_roundtrip():
    """reference tests/test_deinstantiate.py:8-16 on this repo's config loader."""
    import tinyedm
    from tinyedm.config import compose, instantiate
    cfg = compose("cifar10", os.path.join(ROOT, "experiments", "conf"), ["model.lr=0.01", "trainer.max_epochs=3"])
    assert cfg.model.denoiser.embedding_dim == 256 and cfg.model.lr == 0.01 and cfg.trainer.max_epochs == 3
    assert cfg.model.embedding.num_classes is None
    model = instantiate(cfg.model)
    assert isinstance(model, tinyedm.EDM) and not model.conditional
    assert sum(p.numel() for p in model.denoiser.parameters()) == 35604390
    assert sum(p.numel() for p in model.embedding.parameters()) == 16384
    d = tinyedm.utils.deinstantiate(model)
    assert d["_target_"] == "tinyedm.edm.EDM" and d["denoiser"]["_target_"] == "tinyedm.networks.Denoiser"
    assert isinstance(d["denoiser"]["encoder_block_types"], list)
    again = instantiate(d)
    again.load_state_dict(model.state_dict(), strict=True)
    keys = set(model.state_dict())
    assert {"embedding.fourier_embed.freqs", "embedding.sigma_embed.weight", "denoiser.gain_out",
            "denoiser.conv_in.weight", "denoiser.encoder_blocks.3.attention.qkv_conv.weight",
            "denoiser.decoder_blocks.2.cat_factor.layer1.weight", "denoiser.decoder_blocks.0.embed.weight"} <= keys
    with pytest.raises(ValueError):
        tinyedm.EDM(diffuser=model.diffuser, embedding=model.embedding, denoiser=model.denoiser, use_ema=True,
                    use_uncertainty=False, steady_steps=1, rampup_steps=1, scheduler_interval="step")
    with pytest.raises(ValueError, match="num_classes is None"):
        model.embedding(torch.ones(2), torch.zeros(2, dtype=torch.long))


def test_lr_schedule_ema_and_sigma_table(golden_dir):
    import tinyedm
    for step, want in [(0, 1e-8), (100, 0.500000005), (199, 0.99500000005), (200, 1), (400, 1), (600, 0.70710678),
                       (1000, 0.5)]:
        assert abs(tinyedm.EDM.lr_lambda(step, 200, 200) - want) < 1e-8
    assert abs(tinyedm.sigma_rel_to_gamma(0.13) - 4.603596781479866) < 1e-9
    with pytest.raises(tinyedm.ema.MisconfigurationException):
        tinyedm.EMA(0.3)
    s = np.load(os.path.join(golden_dir, "solver.npz"))
    assert np.array_equal(tinyedm.DeterministicSolver(32).t_steps.numpy().view(np.uint32), s["t32"].view(np.uint32))
    assert np.array_equal(tinyedm.DeterministicSolver(18).t_steps.numpy().view(np.uint32), s["t18"].view(np.uint32))
    t5 = tinyedm.DeterministicSolver(5, sigma_min=0.01, sigma_max=20.0, rho=5.0).t_steps
    assert np.array_equal(t5.numpy().view(np.uint32), s["t5"].view(np.uint32))


def test_qkv_permutation_is_a_bijection():
    from tinyedm_amd.networks import _qkv_perm
    p = _qkv_perm(256, 4)
    assert sorted(p.tolist()) == list(range(768))
    # packed (head 1, k, dd 5) -> reference channel 1*192 + 5*3 + 1
    assert p[1 * 192 + 1 * 64 + 5].item() == 192 + 15 + 1


# ---------------------------------------------------------------------------------------------- round 2: boundary
def test_random_noise_datamodule_has_the_reference_signature():
    """datamodules/random_datamodule.py:22-29: (batch_size, num_workers, image_size, num_samples, num_classes)"""
    import inspect
    from tinyedm_amd.datamodules import RandomNoiseDataModule
    names = list(inspect.signature(RandomNoiseDataModule.__init__).parameters)[1:6]
    assert names == ["batch_size", "num_workers", "image_size", "num_samples", "num_classes"]
    dm = RandomNoiseDataModule(16, 8, 64, 100, 1000, in_channels=4)
    assert dm.image_shape == (4, 64, 64) and dm.num_classes == 1000 and dm.num_samples == 100


def test_generate_cli_flags_match_the_reference():
    """generate.py:50-96: flag names (argparse would exit on an unknown one)"""
    import io
    import contextlib
    from tinyedm_amd import generate as G
    buf = io.StringIO()
    with pytest.raises(SystemExit), contextlib.redirect_stdout(buf):
        G.main(["--help"])
    text = buf.getvalue()
    for flag in ("--ckpt_path", "--load_ema", "--output_dir", "--num_samples", "--image_size", "--num_classes",
                 "--batch_size", "--num_workers", "--num_steps"):
        assert flag in text, flag
    import tinyedm.generate                                   # the reference's module path
    assert tinyedm.generate.generate is G.generate


def test_epoch_order_shards_like_a_distributed_sampler():
    """every rank used to iterate the same indices.  Shards are disjoint, cover the dataset, have equal
    length on every rank and share one permutation per epoch."""
    from tinyedm_amd.datamodules import _ResidentLoader, epoch_order, shard_len
    n, world = 1003, 4
    for epoch in (0, 1):
        shards = [epoch_order(n, True, 42, epoch, r, world, "cpu") for r in range(world)]
        assert all(len(s) == shard_len(n, world) == 251 for s in shards)
        allidx = torch.cat(shards)
        assert set(allidx.tolist()) == set(range(n))                      # covers the dataset
        assert len(allidx) - len(set(allidx.tolist())) == world * 251 - n  # only the wrapped tail repeats
        single = epoch_order(n, True, 42, epoch, 0, 1, "cpu")
        assert torch.equal(torch.stack([s[:250] for s in shards], 1).flatten(), single[:1000])   # rank::world striding
    assert not torch.equal(epoch_order(n, True, 42, 0, 0, 1, "cpu"), epoch_order(n, True, 42, 1, 0, 1, "cpu"))
    assert torch.equal(epoch_order(10, False, 0, 0, 1, 2, "cpu"), torch.tensor([1, 3, 5, 7, 9]))
    data = torch.zeros(1003, 3, 4, 4, dtype=torch.uint8)
    lens = {len(_ResidentLoader(data, torch.zeros(1003), 32, True, False, 1, 0.5, 0.5, rank=r, world=4)) for r in range(4)}
    assert lens == {(251 + 31) // 32}                                     # == ceil(ceil(N / world) / B) on every rank


def test_model_checkpoint_keeps_top_k():
    from tinyedm_amd.callbacks import ModelCheckpoint
    import tempfile

    class FakeTrainer:
        global_rank, global_step, current_epoch = 0, 0, 0
        callback_metrics = {}

        def save_checkpoint(self, path, model=None):
            open(path, "w").write("x")
    with tempfile.TemporaryDirectory() as d:
        cb = ModelCheckpoint(dirpath=d, monitor="val_loss", mode="min", save_top_k=2, save_last=True, every_n_epochs=2)
        tr = FakeTrainer()
        for epoch, loss in enumerate([5.0, 4.0, 3.0, 9.0, 8.0, 1.0, 7.0, 2.0]):
            tr.current_epoch, tr.global_step = epoch, 10 * (epoch + 1)
            tr.callback_metrics = {"val_loss": loss}
            cb.on_validation_end(tr, None)
        kept = sorted(os.listdir(d))
        # epochs 1,3,5,7 are eligible (every 2): losses 4, 9, 1, 2 -> top-2 = epochs 5 and 7
        assert kept == ["epoch=5-step=60.ckpt", "epoch=7-step=80.ckpt", "last.ckpt"], kept
        assert cb.best_model_score == 1.0 and cb.best_model_path.endswith("epoch=5-step=60.ckpt")


def test_reference_yaml_sections_instantiate():
    """Every `_target_` of the shipped configs -- the same keys as the reference's YAMLs, including
    `lightning.pytorch.callbacks.ModelCheckpoint` -- resolves; when the reference tree is present (build container
    only) its own three YAMLs are loaded unchanged too."""
    from tinyedm.config import compose, instantiate
    import tinyedm
    dirs = [os.path.join(ROOT, "experiments", "conf")]
    ref = "/root/reference/experiments/conf"
    if os.path.isdir(ref):
        dirs.append(ref)
    for d in dirs:
        for name in sorted(f[:-5] for f in os.listdir(d) if f.endswith(".yaml")):
            cfg = compose(name, d)
            cbs = instantiate(cfg.callbacks)
            kinds = {type(c).__name__ for c in cbs.values()}
            assert "ModelCheckpoint" in kinds and kinds & {"GenerateCallback", "LatentsGenerateCallback"}, (d, name, kinds)
            dm = instantiate(cfg.datamodule)
            assert hasattr(dm, "train_dataloader") or hasattr(dm, "setup")
            tr = tinyedm.Trainer(callbacks=list(cbs.values()), **cfg.trainer)
            assert tr.max_epochs == cfg.trainer.max_epochs
            if name != "imagenet":                       # 272.9 M parameters on the CPU: constructed in the GPU tests
                model = instantiate(cfg.model)
                assert isinstance(model, tinyedm.EDM) and model.hparams["_target_"] == "tinyedm.edm.EDM"


def test_flat_arena_layout3_and_older_checkpoint_migration():
    """FlatArena layout 3 keeps every parameter whose gradient is final only at the END of the backward pass in a tail
    region -- tensors flagged `_edm_late` (the blocks' embed Linears, the Embedding module), then the 0-dim parameters -- so
    that no data-parallel bucket of the body has to wait for them; optimizer states written with the sequential layout of
    rounds 1-2 (no "layout" key) or with layout 2 (rounds 3-4: only the 0-dim parameters in the tail) are moved slice by
    slice; torch-Adam states keep mapping by parameter INDEX."""
    import torch
    from tinyedm_amd.ema import FlatArena, FusedAdam, layout_offsets, sequential_offsets
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.tensor(0.5)), torch.nn.Parameter(torch.randn(70)),
          torch.nn.Parameter(torch.tensor(-1.5)), torch.nn.Parameter(torch.randn(2, 2, 3, 3))]
    ps[0]._edm_late = True                       # e.g. a block's embed.weight
    vals = [p.detach().clone() for p in ps]
    opt = FusedAdam(ps, lr=1e-3)
    a = opt.arena
    # body: ps[2] (128 aligned), ps[4] (64); late region: ps[0] (64), then the scalars ps[1], ps[3] (64 each)
    assert a.LAYOUT == 3 and a.scalar_lo == 128 + 64 and a.numel == a.scalar_lo + 64 + 128
    assert a.offsets == [192, 256, 0, 320, 128]
    for p, v, o in zip(ps, vals, a.offsets):
        assert torch.equal(p.detach(), v) and p.data_ptr() == a.theta.data_ptr() + 4 * o and p.grad.data_ptr() == a.grad.data_ptr() + 4 * o
    # older states: layout 1 (m / v laid out sequentially in list order) and layout 2 (only the scalars in the tail)
    for layout in (1, 2):
        old, _, total = layout_offsets(ps, layout)
        if layout == 1:
            assert (old, total) == sequential_offsets(ps)
        else:
            assert old == [0, 256, 64, 320, 192] and total == 384
        m1, v1 = torch.zeros(total), torch.zeros(total)
        for k, (p, o) in enumerate(zip(ps, old)):
            m1[o:o + p.numel()] = k + 1
            v1[o:o + p.numel()] = 10 * (k + 1)
        sd_old = {"m": m1, "v": v1, "step": 7, "param_groups": [{"lr": 1e-3, "betas": (0.9, 0.999), "eps": 1e-8}]}
        if layout > 1:
            sd_old["layout"] = layout
        opt.load_state_dict(sd_old)
        assert opt.step_count == 7
        for k, (p, o) in enumerate(zip(ps, a.offsets)):
            assert (opt.m[o:o + p.numel()] == k + 1).all() and (opt.v[o:o + p.numel()] == 10 * (k + 1)).all()
    # its own (layout 3) state round-trips unchanged
    sd = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in opt.state_dict().items()}
    assert sd["layout"] == 3
    ps2 = [torch.nn.Parameter(v.clone()) for v in vals]
    ps2[0]._edm_late = True
    opt2 = FusedAdam(ps2, lr=1e-3)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    # a model whose late flags differ from the writer's (flag lost on a re-created Parameter) has other offsets
    # for the same "layout 3": the stored offsets move every slice to its new home instead of scrambling the moments
    ps3 = [torch.nn.Parameter(v.clone()) for v in vals]
    opt3 = FusedAdam(ps3, lr=1e-3)
    assert opt3.arena.offsets != a.offsets and sd["offsets"] == a.offsets
    opt3.load_state_dict(sd)
    for p, o3, o in zip(ps3, opt3.arena.offsets, a.offsets):
        n = p.numel()
        assert torch.equal(opt3.m[o3:o3 + n], opt.m[o:o + n]) and torch.equal(opt3.v[o3:o3 + n], opt.v[o:o + n])
    bad = dict(sd, numels=[1] + sd["numels"][1:])
    with pytest.raises(ValueError):
        opt3.load_state_dict(bad)


def test_launch_probe_verdict_follows_the_timed_blocks_only():
    """the fit() probe times PROBE_STEPS steps per launch form AFTER untimed steps of that form (a cold first
    eager step must not hand the verdict to the replay), and keeps the faster.  Stubbed clock: a step costs what the
    table says; the first eager step (cold side stream) costs 50x."""
    from tinyedm_amd import trainer as TR

    class Captured:
        _graphs = {((4, 3, 8, 8), None, None): object()}

    class Clock:
        def __init__(self):
            self.t = 0.0

        def mark(self):
            return self.t

        def elapsed(self, a, b):
            return b - a

    batch = (torch.zeros(4, 3, 8, 8), None)
    for eager_ms, graph_ms, want_graph in ((10.0, 10.5, False), (10.0, 9.5, True)):
        clock = Clock()
        probe = TR._LaunchProbe(clock)
        forms, first_eager = [], True
        for _ in range(64):
            use_graph = probe.choose(Captured, batch)
            if probe.done:
                break
            forms.append(use_graph)
            cost = graph_ms if use_graph else eager_ms
            if not use_graph and first_eager:
                cost, first_eager = 50 * eager_ms, False
            clock.t += cost
        assert probe.done and probe.graph_wins is want_graph
        n, we, wg = probe.PROBE_STEPS, probe.WARM_STEPS["eager"], probe.WARM_STEPS["graph"]
        assert forms == [False] * (we + n) + [True] * (wg + n)
        assert abs(probe.times["eager"] - n * eager_ms) < 1e-9 and abs(probe.times["graph"] - n * graph_ms) < 1e-9
        # a ragged batch in the middle of a probe is passed through untimed
        probe2 = TR._LaunchProbe(Clock())
        probe2.choose(Captured, batch)
        assert probe2.choose(Captured, (torch.zeros(2, 3, 8, 8), None)) is False and probe2.n == 1 - probe2.WARM_STEPS["eager"]


def test_capture_token_releases_slots_and_plan_pins_once():
    """what a captured graph holds on to -- launch-table slots and pinned weight-prep plans -- goes back when
    ops.release_capture() gets its token, exactly once (host logic only: no GPU call)."""
    from tinyedm_amd import ops

    class Plan:
        pins = 0

    t = ops._LaunchTables()
    t.dev[0] = {"free": [], "key": 0}
    a, b = Plan(), Plan()
    t.deferring, t.pins = True, []
    saved = ops._tables
    ops._tables = t
    try:
        assert ops.note_capture_pin(a) and ops.note_capture_pin(a) and ops.note_capture_pin(b)     # a plan is pinned once per capture
        assert (a.pins, b.pins) == (1, 1)
        tok = ops._CaptureToken(((0, 3), (0, 5)), tuple(t.pins))
        t.deferring = False
        assert not ops.note_capture_pin(a) and a.pins == 1          # outside a capture_begin() / capture_end() pair: not counted
        assert list(tok) == [(0, 3), (0, 5)] and len(tok) == 2
        ops.release_capture(tok)
        assert (a.pins, b.pins) == (0, 0) and sorted(t.dev[0]["free"]) == [3, 5]
        ops.release_capture(tok)                                    # a second release is a no-op
        assert (a.pins, b.pins) == (0, 0) and sorted(t.dev[0]["free"]) == [3, 5]
        ops.release_capture(None)
    finally:
        ops._tables = saved


def test_backward_hook_feeds_a_cached_marked_unit_gradient():
    """LightningModule.backward (Lightning's hook of the same name): the root gradient is ONE cached scalar carrying the
    `_edm_unit` mark, the loss function's backward returns its saved gradient untouched when it sees the mark (metric._scaled)
    and multiplies otherwise -- same parameter gradients as a plain loss.backward(), no ones_like / mul per step."""
    import torch
    from tinyedm_amd import metric, trainer

    class Toy(trainer.LightningModule):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.arange(6.0).view(2, 3))

        def loss(self, x):
            return ((self.w * x) ** 2).sum()

    seen = []

    class Spy(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            seen.append(g)
            return metric._scaled(torch.ones_like(g), g)

    m = Toy()
    x = torch.tensor([[1.0, -2.0, 0.5], [3.0, 0.25, -1.0]])
    m.loss(x).backward()
    ref = m.w.grad.clone()
    m.w.grad = None
    m.backward(Spy.apply(m.loss(x)))
    assert torch.equal(m.w.grad, ref)
    one = trainer.unit_gradient(torch.zeros(()))
    assert getattr(one, "_edm_unit", False) and one.item() == 1.0 and trainer.unit_gradient(torch.zeros(())) is one
    assert len(seen) == 1 and getattr(seen[0], "_edm_unit", False)        # the mark survives the trip through the engine
    d = torch.full((2, 2), 3.0)
    assert metric._scaled(d, one) is d                                     # marked: no multiplication
    assert torch.equal(metric._scaled(d, torch.tensor(0.5)), d * 0.5)     # any other incoming gradient: the product
    # non-scalar losses and explicit gradients take the ordinary path
    m.w.grad = None
    m.backward(m.w * 2.0, torch.ones(2, 3))
    assert torch.equal(m.w.grad, torch.full((2, 3), 2.0))


def test_wgrad3_plan_splits_k_only_where_team_members_would_idle():
    """conv_wgrad3.hip make_plan (host logic): layers whose 128 x 64 tiles do not fill a team of eight workgroups split
    their K range into shares; the 256 / 512-channel layers of the CIFAR-10 net (2 x 4, 2 x 8 tiles) never do, its conv_in does."""
    from tinyedm_amd import ops
    cifar = [(128, 32, 32, 256, 256), (128, 32, 32, 512, 256), (128, 16, 16, 256, 256), (128, 8, 8, 512, 256)]
    assert ops.wgrad3_plan_ksplit(cifar) == [1, 1, 1, 1]
    # ... but its conv_in does (4 input channels padded to 32 -> 256: 2 x 1 tiles; 2 of 8 members busy without the shares)
    assert ops.wgrad3_plan_ksplit([(128, 32, 32, 32, 256)] + cifar[:1]) == [4, 1]
    # MNIST 28x28 / 14x14 128-channel layers: 1 x 2 tiles -> 4 shares fill the team
    assert ops.wgrad3_plan_ksplit([(128, 28, 28, 128, 128), (128, 14, 14, 128, 128)]) == [4, 4]
    # ImageNet-64 192-channel layers (2 x 3 tiles, 6 of 8 members): 4 shares = 3 full groups of a quarter of the stages
    assert ops.wgrad3_plan_ksplit([(32, 64, 64, 192, 192)]) == [4]
    # too few stages to pay for the extra partial tiles: not split
    assert ops.wgrad3_plan_ksplit([(1, 8, 8, 64, 64), (2, 8, 8, 128, 128)]) == [1, 1]
    ks = ops.wgrad3_plan_ksplit([(64, 32, 32, 384, 384), (64, 16, 16, 576, 576), (64, 64, 64, 64, 64)][:2])
    assert all(1 <= k <= 8 for k in ks)
    # the group size the Python side queues up to is the kernel table's (48 layers: every 3x3 layer of the CIFAR-10 net)
    from tinyedm_amd import _lib
    assert ops.W3_MAX_LAYERS == _lib.call("edm_wgrad3_max_layers") == 48
    assert ops.wgrad3_plan_ksplit([(128, 8, 8, 256, 256)] * 48) == [1] * 48

"""hipGraph-captured training step (tinyedm_amd/graph.py) against the eager Python step: same model, same seeds, same
batches -> same losses, weights, Adam moments and EMA after several steps with a changing learning rate, and the
host-side counters (Philox step, Adam step, EMA step) advance identically.  Replays must draw fresh noise/dropout
(a frozen Philox step would give identical losses on identical batches)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from parity_log import record

DEV = "cuda"


def _build(seed=11, pdrop=0.1):
    import tinyedm_amd as T
    from oracle.make_golden import tiny_cfgs
    from tinyedm_amd import networks as N
    ecfg, dcfg = tiny_cfgs()
    N._rng_sub_counter[0] = 0
    T.manual_seed(seed)
    torch.manual_seed(seed)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                     tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                     tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), pdrop, dcfg.sigma_data,
                     dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim, dcfg.num_heads)
    with torch.no_grad():
        den.gain_out.fill_(0.7)
    model = T.EDM(diffuser=T.Diffuser(-1.2, 1.2), embedding=emb, denoiser=den, use_ema=True, use_uncertainty=False,
                  steady_steps=3, rampup_steps=3, scheduler_interval="step", lr=2e-3, ema_length=0.13)
    return model.to(DEV).train(), ecfg


def _opt(model):
    import tinyedm_amd as T
    from tinyedm_amd.ema import EMAOptimizer
    cfg = model.configure_optimizers()
    base, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    return EMAOptimizer(base, device=DEV, gamma=T.sigma_rel_to_gamma(0.13)), base, sched


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def test_captured_step_matches_eager_step():
    import tinyedm_amd as T
    from tinyedm_amd import networks as N
    from tinyedm_amd.graph import CapturedTrainStep
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    g = torch.Generator().manual_seed(5)
    batches = [((0.5 * torch.randn(8, 3, 16, 16, generator=g)).to(DEV), torch.randint(0, 10, (8,), generator=g).to(DEV))
               for _ in range(6)]
    # ---- eager
    model_e, _ = _build()
    opt_e, base_e, sched_e = _opt(model_e)
    opt_e.zero_grad()
    losses_e = []
    for b in batches:
        loss = model_e.training_step(b, 0)
        loss.backward()
        opt_e.step()
        opt_e.zero_grad()
        sched_e.step()
        losses_e.append(float(loss))
    counters_e = (base_e.step_count, opt_e.current_step, N.rng.step)
    # ---- captured (2 warm-up steps run eagerly, the rest replays one graph)
    model_g, _ = _build()
    opt_g, base_g, sched_g = _opt(model_g)
    opt_g.zero_grad()
    step = CapturedTrainStep(model_g, opt_g)
    losses_g = []
    for b in batches:
        loss = step(b)
        sched_g.step()
        losses_g.append(float(loss))
    assert len(step._graphs) == 1
    assert (base_g.step_count, opt_g.current_step, N.rng.step) == counters_e
    worst = max(abs(a - b) / abs(b) for a, b in zip(losses_g, losses_e))
    record("captured_step/loss_vs_eager", worst, 2e-3)
    assert worst <= 2e-3, (losses_g, losses_e)
    assert len(set(round(l, 6) for l in losses_g)) == len(losses_g)
    for name, a, b, lim in (("theta", base_g.arena.theta, base_e.arena.theta, 2e-3), ("adam_m", base_g.m, base_e.m, 2e-2),
                            ("adam_v", base_g.v, base_e.v, 2e-2), ("ema", opt_g.ema_arena, opt_e.ema_arena, 2e-3)):
        e = rel(a, b)
        record(f"captured_step/{name}_vs_eager", e, lim)
        assert e <= lim, f"{name}: rel {e:.3e}"
    assert float(base_g.arena.grad.abs().max()) == 0.0          # zero_grad rides in the optimizer kernel


def test_replays_draw_fresh_noise():
    """the same batch replayed twice must see different sigma / noise / dropout (Philox step comes from the device
    record, not from a frozen kernel argument): the losses differ, and equal a fresh eager run step for step"""
    from tinyedm_amd.graph import CapturedTrainStep
    g = torch.Generator().manual_seed(9)
    b = ((0.5 * torch.randn(8, 3, 16, 16, generator=g)).to(DEV), torch.randint(0, 10, (8,), generator=g).to(DEV))
    model, _ = _build(seed=3)
    opt, base, _ = _opt(model)
    for pg in base.param_groups:
        pg["lr"] = 0.0                                          # weights frozen: only the random draws change
    opt.zero_grad()
    step = CapturedTrainStep(model, opt)
    losses = [float(step(b)) for _ in range(5)]
    assert len(set(round(l, 5) for l in losses)) == 5, losses


def test_replay_after_stream_sync_is_not_corrupted():
    """ROCm 7.2 runtime bug (tinyedm_amd/_runtime_env.py): the first hipGraph replay after a stream / device
    synchronisation ran with clobbered kernel arguments.  tests/graph_sync_probe.py is the sequence that failed every
    time; it runs in its own process so that the flag is set by the package import alone."""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "graph_sync_probe.py")], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    record("graph/replay_after_sync_clean", 0.0 if last.startswith("CLEAN") else 1.0, 0.0)
    assert last.startswith("CLEAN"), last
    # informational: the same sequence with the runtime's default path (documents whether the bug is still there)
    env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(here, "graph_sync_probe.py")], env=env, capture_output=True,
                         text=True, timeout=300)
    lines = out.stdout.strip().splitlines()
    print("default runtime path:", lines[-1][:200] if lines else out.stderr[-300:])


def test_graph_paths_refuse_when_the_runtime_flag_is_unsafe(monkeypatch):
    from tinyedm_amd import _runtime_env
    monkeypatch.setattr(_runtime_env, "GRAPH_REPLAY_SAFE", False)
    with pytest.raises(RuntimeError, match="hipGraph replay is unsafe"):
        _runtime_env.require_graph_replay_safe("test")

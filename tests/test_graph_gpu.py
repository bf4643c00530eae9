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
    synchronisation ran with clobbered kernel arguments under the runtime's default packet-capture path.
    tests/graph_sync_probe.py replays across synchronisations with in-graph probes; it runs in its own process so that
    the safe setting is the one the package import alone puts in place.  Only the safe setting is ever exercised: the
    failing configuration launches kernels with garbage pointers and is never provoked on purpose."""
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


def test_graph_paths_refuse_when_the_runtime_flag_is_unsafe(monkeypatch):
    from tinyedm_amd import _runtime_env
    monkeypatch.setattr(_runtime_env, "GRAPH_REPLAY_SAFE", False)
    with pytest.raises(RuntimeError, match="hipGraph replay is unsafe"):
        _runtime_env.require_graph_replay_safe("test")


def test_package_import_before_gpu_init_is_graph_safe_and_late_import_is_not():
    """fail-closed witness (_runtime_env.hip_runtime_live): importing the package before the first HIP call reports
    safe; after the runtime is up (torch.cuda.is_available() initialises it WITHOUT setting torch's lazy-init flag) an
    import without the inherited variable must report unsafe"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    early = "import sys; sys.path.insert(0, %r); import torch, tinyedm_amd; print(tinyedm_amd._runtime_env.GRAPH_REPLAY_SAFE)" % root
    late = ("import sys; sys.path.insert(0, %r); import torch; torch.cuda.is_available(); torch.zeros(1, device='cuda'); "
            "import tinyedm_amd; print(tinyedm_amd._runtime_env.GRAPH_REPLAY_SAFE)" % root)
    out_e = subprocess.run([sys.executable, "-c", early], env=env, capture_output=True, text=True, timeout=300)
    out_l = subprocess.run([sys.executable, "-c", late], env=env, capture_output=True, text=True, timeout=300)
    assert out_e.stdout.strip().splitlines()[-1] == "True", out_e.stderr[-1000:]
    assert out_l.stdout.strip().splitlines()[-1] == "False", out_l.stderr[-1000:]
    env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"         # inherited safe value: safe however late the import is
    out_i = subprocess.run([sys.executable, "-c", late], env=env, capture_output=True, text=True, timeout=300)
    assert out_i.stdout.strip().splitlines()[-1] == "True", out_i.stderr[-1000:]


def test_health_sentinel_trips_on_a_poisoned_replay():
    """a NaN fed into the REPLAYED step (what a replay with corrupted arguments amounts to) leaves the in-graph sentinel
    bit behind and the host read point raises; a clean replay does not"""
    from tinyedm_amd import ops
    from tinyedm_amd.graph import CapturedTrainStep
    g = torch.Generator().manual_seed(2)
    b = ((0.5 * torch.randn(8, 3, 16, 16, generator=g)).to(DEV), torch.randint(0, 10, (8,), generator=g).to(DEV))
    model, _ = _build(seed=4)
    opt, base, _ = _opt(model)
    for pg in base.param_groups:
        pg["lr"] = 0.0
    opt.zero_grad()
    step = CapturedTrainStep(model, opt)
    for _ in range(CapturedTrainStep.WARMUP + 2):
        step(b)
    ops.check_health(DEV, "clean replays")                  # does not raise
    bad = (b[0].clone(), b[1])
    bad[0][3, 1, 5, 5] = float("nan")
    theta = base.arena.theta.clone()
    step(bad)                                               # replay of the SAME graph with a poisoned input
    with pytest.raises(ops.GraphCorruptionError, match="non-finite gradient"):
        ops.check_health(DEV, "poisoned replay")
    ops.check_health(DEV, "cleared")                        # the word is cleared by the failed check
    base.arena.theta.copy_(theta)                           # (the poisoned step wrote NaN weights; put them back)


def test_trainer_fit_raises_on_nonfinite_step(tmp_path):
    """Trainer.fit reads the sentinel at its log interval: a batch with a NaN makes fit() raise instead of training on"""
    import tinyedm_amd as T
    from tinyedm_amd import ops
    model, _ = _build(seed=6)
    g = torch.Generator().manual_seed(8)
    xs = 0.5 * torch.randn(4, 8, 3, 16, 16, generator=g)
    xs[2, 0, 0, 0, 0] = float("nan")
    loader = [(xs[i], torch.randint(0, 10, (8,), generator=g)) for i in range(4)]
    tr = T.Trainer(max_epochs=1, log_every_n_steps=1)
    with pytest.raises(ops.GraphCorruptionError):
        tr.fit(model, train_dataloaders=loader)
    ops.health(DEV).zero_()


def test_sampler_sentinel_trips_on_nonfinite_state():
    import tinyedm_amd as T
    from tinyedm_amd import ops
    model, _ = _build(seed=5)
    model.eval()
    solver = T.DeterministicSolver(num_steps=4)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(4, 3, 16, 16, generator=g).to(DEV)
    lab = torch.randint(0, 10, (4,), generator=g).to(DEV)
    out = solver.solve(model, x0, lab, graph=True)
    assert torch.isfinite(out).all()
    x0b = x0.clone()
    x0b[1, 0, 2, 2] = float("inf")
    with pytest.raises(ops.GraphCorruptionError, match="non-finite sampler state"):
        solver.solve(model, x0b, lab, graph=True)


def test_fused_clear_is_not_trusted_after_an_eval_mode_backward():
    """FusedAdam(fuse_zero_grad): step() clears the arena and the next zero_grad() is free ONLY if no gradient was written
    since.  An eval-mode forward with autograd on (fine-tuning, parity runs with dropout off) writes gradients too: the
    sequence step(); eval fwd/bwd; zero_grad() must leave the arena all-zero."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    model, _ = _build(pdrop=0.0)
    opt, base, _ = _opt(model)
    base.fuse_zero_grad = True
    g = torch.Generator().manual_seed(6)
    bt = ((0.5 * torch.randn(8, 3, 16, 16, generator=g)).to(DEV), torch.randint(0, 10, (8,), generator=g).to(DEV))
    opt.zero_grad()
    model.training_step(bt, 0).backward()
    opt.step()                                  # clears the arena in the Adam pass
    torch.cuda.synchronize()
    assert float(base.arena.grad.abs().max()) == 0.0
    model.eval()
    with torch.enable_grad():
        model.training_step(bt, 1).backward()   # eval-mode forward, gradients are written all the same
    torch.cuda.synchronize()
    assert float(base.arena.grad.abs().max()) > 0.0
    opt.zero_grad()
    torch.cuda.synchronize()
    assert float(base.arena.grad.abs().max()) == 0.0
    model.train()


def test_copy_free_concat_matches_the_standalone_concat_kernels():
    """FUSE_CAT (round 4): the producer of a decoder block's input writes it (and mp_silu of it) straight into the next
    block's concatenated operands and the 1x1 dgrad splits d loss / d cat -- same values as the standalone concat kernels
    (networks.py:311 and its autograd): the evaluation forward bit for bit, every parameter gradient up to the order of
    the fp32 atomics both paths share."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import tinyedm_amd as T
    from tinyedm_amd import networks as N
    g = torch.Generator().manual_seed(9)
    x = (0.5 * torch.randn(8, 3, 16, 16, generator=g)).to(DEV)
    y = torch.randint(0, 10, (8,), generator=g).to(DEV)
    sigma = torch.randn(8, generator=g).exp().to(DEV)
    res = {}
    old = N.FUSE_CAT
    try:
        for mode in (False, True):
            N.FUSE_CAT = mode
            model, _ = _build(pdrop=0.1)
            opt, base, _ = _opt(model)
            opt.zero_grad()
            model.eval()
            with torch.no_grad():
                D = model(x, sigma, y).clone()
            model.train()
            T.manual_seed(11)
            loss = model.training_step((x, y), 0)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (D, float(loss), base.arena.grad.clone())
    finally:
        N.FUSE_CAT = old
    assert torch.equal(res[True][0], res[False][0])
    assert abs(res[True][1] - res[False][1]) <= 1e-6 * abs(res[False][1])
    assert rel(res[True][2], res[False][2]) <= 1e-5, rel(res[True][2], res[False][2])


def test_captured_launch_table_pool_grows_outside_capture():
    """ops._LaunchTables: the device slots of captured launches are never reused (a graph keeps reading its tables), so a
    process that captures again and again must not run out -- capture_begin() reserves room for the next capture by adding
    a chunk of slots (outside the capture: memory allocated inside one belongs to the graph's pool)"""
    from tinyedm_amd import ops
    t = ops._tables
    st = t._state(torch.cuda.current_device())
    before_chunks, before_i, before_free = len(st["pool_devb"]), st["pool_i"], list(st["free"])
    try:
        st["pool_i"] = len(st["pool_devb"]) * t.POOL - 3          # three slots left ...
        st["free"][:] = []                                        # ... and none handed back by graphs that are gone
        ops.capture_begin()
        assert len(st["pool_devb"]) == before_chunks + 1
        assert len(st["pool_devb"]) * t.POOL - st["pool_i"] >= 64
        assert st["pool_devb"][-1].device.type == "cuda" and st["pool_devb"][-1].shape == (t.POOL, st["nb"])
    finally:
        ops.capture_end()
        st["pool_i"] = max(before_i, 0)
        st["free"][:] = before_free


def test_launch_table_slots_are_released_and_bare_captures_have_their_own_index():
    """ADVICE r4 (low) + round-4 review: (1) deferred captures (capture_begin / capture_end) and bare captures draw from
    separate indices -- many deferred captures no longer exhaust the pinned pool of the bare path; (2) capture_end() returns
    the slots the capture took and release_capture() makes them reusable: a process that re-captures does not grow."""
    from tinyedm_amd import ops
    t = ops._tables
    dev = torch.cuda.current_device()
    st = t._state(dev)
    bare0, pool0, free0 = st["bare_i"], st["pool_i"], list(st["free"])
    s = torch.cuda.Stream()
    toks = []
    for _ in range(3):
        ops.capture_begin()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            _ = torch.ones(4, device="cuda") + 1
            got = [t.take(dev) for _ in range(5)]
        assert all(x[2] == 1 for x in got)                       # deferred upload
        toks.append(ops.capture_end())
        del g
    assert st["bare_i"] == bare0                                 # the bare pool was not touched
    used = st["pool_i"] - pool0 + (len(free0) - len(st["free"]))
    assert used == 15
    for tok in toks:
        ops.release_capture(tok)
    mark = st["pool_i"]
    ops.capture_begin()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        _ = torch.ones(4, device="cuda") + 1
        for _ in range(15):
            t.take(dev)
    tok = ops.capture_end()
    assert st["pool_i"] == mark                                  # all fifteen came from the free list
    ops.release_capture(tok)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):                         # a bare capture: pinned slot + memcpy node path
        _ = torch.ones(4, device="cuda") + 1
        h, d, defer, _rel = t.take(dev)
    assert defer == 0 and st["bare_i"] == bare0 + 1


def test_plans_are_unpinned_when_their_graphs_go():
    """a weight-prep plan is pinned while a captured graph reads its buffers -- and ONLY that long.  Evicting a
    captured solve (the solver keeps MAX_GRAPHS per model) and releasing a captured training step give the pins back, so the
    plans can be evicted again (before: pinned for the life of the Denoiser, every evicted graph kept its full pack set)."""
    import tinyedm
    from tinyedm_amd.graph import CapturedTrainStep
    model, _ = _build(pdrop=0.0)
    model.eval()
    den = model.denoiser
    solver = tinyedm.DeterministicSolver(num_steps=3)
    solver.MAX_GRAPHS = 1
    g = torch.Generator().manual_seed(3)
    xa = torch.randn(4, 3, 16, 16, generator=g).to(DEV)
    xb = torch.randn(6, 3, 16, 16, generator=g).to(DEV)
    la, lb = torch.randint(0, 10, (4,), generator=g).to(DEV), torch.randint(0, 10, (6,), generator=g).to(DEV)
    out_a = solver.solve(model, xa, la, graph=True)
    pins_a = sum(p.pins for p in den._plans.values())
    assert pins_a >= 1 and not any(p._pinned_forever for p in den._plans.values())
    solver.solve(model, xb, lb, graph=True)                 # evicts the first solve: its pins go, the new solve's come
    assert sum(p.pins for p in den._plans.values()) == pins_a
    out_a2 = solver.solve(model, xa, la, graph=True)        # re-captured: same result
    assert torch.equal(out_a, out_a2)
    # a captured training step pins the training plan and release() unpins it
    model.train()
    opt, base, _ = _opt(model)
    opt.zero_grad()
    step = CapturedTrainStep(model, opt)
    batch = ((0.5 * torch.randn(8, 3, 16, 16, generator=g)).to(DEV), torch.randint(0, 10, (8,), generator=g).to(DEV))
    for _ in range(4):
        step(batch)
    assert step._graphs
    before = sum(p.pins for p in den._plans.values())
    step.release()
    assert sum(p.pins for p in den._plans.values()) < before

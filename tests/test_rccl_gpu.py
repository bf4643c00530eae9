"""The RCCL code path on ONE GPU: a one-rank "nccl" process group (RCCL on ROCm) with the reducer's hooks forced on
(GradReducer(force=True) / EDM_FORCE_REDUCE=1), so the bucket all-reduces are launched from the weight-gradient
hooks DURING the backward pass on the comm stream, behind the side stream of the weight-gradient kernels, exactly as
with N ranks -- a sum over one rank must leave the gradients unchanged.  Also: broadcast of parameters and buffers,
bf16 transport through RCCL, and bench.py / Trainer.fit end to end through init_process_group("nccl").
The N > 1 scaling curve itself can only be measured by the driver on a multi-GPU node (reference:
experiments/conf/cifar10.yaml:4-8 devices -1 / strategy auto = DDP over NCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_graph_gpu import _build
        from tinyedm_amd.ddp import GradReducer
        import tinyedm_amd as T
        g = torch.Generator().manual_seed(5)
        x = (0.5 * torch.randn(8, 3, 16, 16, generator=g)).to(dev)
        y = torch.randint(0, 10, (8,), generator=g).to(dev)
        out = {}
        for mode in ("plain", "fp32", "bf16"):
            model, _ = _build()
            base = model.configure_optimizers()["optimizer"]
            red = GradReducer(base.arena, bucket_bytes=1 << 20, force=(mode != "plain"),
                              transport="bf16" if mode == "bf16" else "fp32")
            red.broadcast_parameters()
            red.broadcast_buffers(model)
            launched = []
            orig = red._launch
            red._launch = lambda b, _o=orig: (launched.append(b["lo"]), _o(b))[1]
            base.zero_grad()
            T.manual_seed(11)
            loss = model.training_step((x, y), 0)
            loss.backward()
            during_backward = len(launched)
            scale = red.finish()
            torch.cuda.synchronize()
            out[mode] = dict(grad=base.arena.grad.cpu().numpy(), n_buckets=len(red.buckets), during=during_backward,
                             total=len(launched), scale=scale, active=red.active)
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", out))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put(("err", traceback.format_exc()))


def test_forced_reduce_runs_rccl_allreduce_from_hooks_on_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=600)
    p.join(120)
    assert status == "ok", out
    assert p.exitcode == 0
    for v in out.values():      # numpy arrays travel through the queue by value (tensors would go by fd: the worker may exit first)
        v["grad"] = torch.from_numpy(v["grad"])
    plain, f32, b16 = out["plain"], out["fp32"], out["bf16"]
    assert not plain["active"] and plain["total"] == 0
    assert f32["active"] and f32["n_buckets"] > 1 and f32["scale"] == 1.0
    # every bucket was all-reduced, most of them from hooks while the backward pass was still running (overlap)
    assert f32["total"] == f32["n_buckets"] and f32["during"] >= f32["n_buckets"] - 1, (f32["during"], f32["n_buckets"])
    rel = ((f32["grad"] - plain["grad"]).norm() / plain["grad"].norm()).item()
    assert rel <= 1e-5, rel                                           # sum over one rank: identity (fp32 atomics noise)
    relb = ((b16["grad"] - plain["grad"]).norm() / plain["grad"].norm()).item()
    assert 1e-5 < relb <= 2 ** -7, relb                               # bf16 wire format: one rounding per element


def _worker_graph(port, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import tinyedm_amd as T             # before the first GPU call: the graph-safe runtime setting
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        import time
        from test_graph_gpu import _build, _opt
        from tinyedm_amd import networks as N
        from tinyedm_amd.ddp import GradReducer
        from tinyedm_amd.graph import CapturedTrainStep
        g = torch.Generator().manual_seed(5)
        batches = [((0.5 * torch.randn(8, 3, 16, 16, generator=g)).to(dev), torch.randint(0, 10, (8,), generator=g).to(dev))
                   for _ in range(6)]
        res = {}
        for mode in ("eager", "graph"):
            model, _ = _build()
            opt, base, sched = _opt(model)
            red = GradReducer(base.arena, bucket_bytes=1 << 18, force=True)
            red.broadcast_parameters()
            assert red.active and red.capturable() and len(red.buckets) > 2
            opt.zero_grad()
            launched = []
            orig = red._launch
            red._launch = lambda b, _o=orig: (launched.append(b["lo"]), _o(b))[1]
            losses, host = [], []
            cap = CapturedTrainStep(model, opt, reducer=red) if mode == "graph" else None
            for bt in batches:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if cap is None:
                    loss = model.training_step(bt, 0)
                    loss.backward()
                    base.grad_scale = red.finish()
                    opt.step()
                    opt.zero_grad()
                else:
                    loss = cap(bt)
                host.append(time.perf_counter() - t0)
                sched.step()
                losses.append(float(loss.detach()))
            torch.cuda.synchronize()
            res[mode] = dict(theta=base.arena.theta.cpu().numpy(), losses=losses, launches=len(launched),
                             buckets=len(red.buckets), host_ms=1e3 * min(host[-2:]), counters=(base.step_count, N.rng.step))
        T.ops.check_health(dev, "captured data-parallel step")
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", res))
    except Exception:          # noqa: BLE001
        import traceback
        q.put(("err", traceback.format_exc()))


def test_captured_data_parallel_step_has_the_collectives_in_the_graph():
    """The collective-bearing step as ONE hipGraph (forced one-rank RCCL group): the bucket all-reduces are issued by the
    hooks while the step is captured (comm stream forked from the capture stream) and then live in the graph -- replays
    issue NO collective from Python, cost the host a graph launch, and give the eager reducer step's losses / weights."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_graph, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=600)
    p.join(120)
    assert status == "ok", out
    assert p.exitcode == 0
    e, g = out["eager"], out["graph"]
    assert e["launches"] == 6 * e["buckets"]                       # eager: every bucket, every step, from Python
    # captured: 2 eager warm-up steps + the capture itself issue them; the 3 replays issue none
    assert g["launches"] == 3 * g["buckets"], (g["launches"], g["buckets"])
    assert g["counters"] == e["counters"]
    worst = max(abs(a - b) / abs(b) for a, b in zip(g["losses"], e["losses"]))
    assert worst <= 2e-3, (g["losses"], e["losses"])
    te, tg = torch.from_numpy(e["theta"]), torch.from_numpy(g["theta"])
    rel = ((tg - te).norm() / te.norm()).item()
    assert rel <= 2e-3, rel
    assert g["host_ms"] <= 5.0, g["host_ms"]                       # a replayed data-parallel step is host-cheap


def test_bench_and_fit_through_the_nccl_backend_on_one_gpu(tmp_path):
    """bench.py with EDM_FORCE_REDUCE=1: init_process_group("nccl"), broadcast, hook-driven bucket all-reduces on
    the comm stream every step (eager step: the collectives are issued from autograd hooks), one JSON line out."""
    env = dict(os.environ, EDM_FORCE_REDUCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "16",
                        "--no-sampler", "--no-cpu-baseline", "--step-launch", "auto"], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["collective"] == "rccl all-reduce (forced, 1 rank)"
    # the collective-bearing step is host-cheap: the all-reduces are nodes of the replayed graph
    assert line["config"]["graph_host_ms_per_step"] <= 5.0, line["config"]
    assert line["config"]["step_launch"] in ("hipGraph replay", "eager")
    # round 5: the eager probe reports what the step could not hide of its all-reduces (events around finish()'s wait)
    assert 0.0 <= line["config"]["exposed_comm_ms"] < 50.0 and line["config"]["grad_buckets"] >= 2, line["config"]


def test_bench_capture_watchdog_reports_the_eager_line(tmp_path):
    """bench.py times the eager step first when the captured step would contain multi-rank collectives, and a timer prints
    that line and ends the process if capture + first replays do not finish (a hang there cannot be rehearsed on one GPU).
    EDM_BENCH_WATCHDOG=1 arms the timer with one rank, EDM_BENCH_CAPTURE_TIMEOUT makes it fire at once."""
    env = dict(os.environ, EDM_FORCE_REDUCE="1", EDM_BENCH_WATCHDOG="1", EDM_BENCH_CAPTURE_TIMEOUT="0.001", WORLD_SIZE="1",
               RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "16",
                        "--no-sampler", "--no-cpu-baseline", "--step-launch", "graph"], capture_output=True, text=True, env=env,
                       timeout=900)
    # a hung capture is a FAILED run (exit 3) that still carries the eager figures, flagged, exactly once
    assert r.returncode == 3, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["capture_hang"] is True
    assert line["steps"] == 3 and line["value"] > 0 and line["ms_per_step"] > 0
    assert line["config"]["step_launch"].startswith("eager (capturing"), line["config"]
    assert line["config"]["capture_hang_where"].startswith("rank 0: capture"), line["config"]
    assert "graph capture watchdog fired -- rank 0" in r.stderr



# ---- two REAL RCCL ranks (one process per GPU).  Skipped on a one-GPU box: torch.cuda.device_count() does not initialise
# the runtime on this image.  The driver's multi-GPU node is where these run.

def _two_rank_worker(rank, world, port, q, captured):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import tinyedm_amd as T             # before the first GPU call: the graph-safe runtime setting
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        from test_graph_gpu import _build, _opt
        from tinyedm_amd.ddp import GradReducer
        from tinyedm_amd.graph import CapturedTrainStep
        g = torch.Generator().manual_seed(5)
        full = [((0.5 * torch.randn(8, 3, 16, 16, generator=g)), torch.randint(0, 10, (8,), generator=g)) for _ in range(5)]
        model, _ = _build()
        opt, base, sched = _opt(model)
        red = GradReducer(base.arena, bucket_bytes=1 << 18)
        assert red.active and red.world == world and red.capturable() and len(red.buckets) > 2
        red.broadcast_parameters()
        red.broadcast_buffers(model)
        opt.zero_grad()
        cap = CapturedTrainStep(model, opt, reducer=red) if captured else None
        losses = []
        for x, y in full:
            n = x.shape[0] // world
            bt = (x[rank * n:(rank + 1) * n].to(dev), y[rank * n:(rank + 1) * n].to(dev))
            T.manual_seed(11)               # p = 0 here: the draw that matters is the Diffuser's, re-seeded per step
            if cap is None:
                loss = model.training_step(bt, 0)
                loss.backward()
                base.grad_scale = red.finish()
                opt.step()
                opt.zero_grad()
            else:
                loss = cap(bt)
            sched.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        T.ops.check_health(dev, "two-rank step")
        theta = base.arena.theta.cpu().numpy()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", theta, losses))
    except Exception:          # noqa: BLE001
        import traceback
        q.put((rank, "err", traceback.format_exc(), None))


def _run_two_ranks(captured):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q, captured)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
    for r in res:
        assert r[1] == "ok", r[2]
    for p in procs:
        assert p.exitcode == 0
    return res


needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (real RCCL ranks)")


@needs_two
@pytest.mark.parametrize("captured", [False, True], ids=["eager", "hipgraph"])
def test_two_real_rccl_ranks_keep_replicas_identical(captured):
    """Two processes, two devices, RCCL: after 5 optimisation steps on disjoint halves of each batch both replicas hold
    bit-identical weights (same summed gradients, same fused Adam), eager hooks and captured collectives alike, and the
    captured run lands on the eager run's weights."""
    res = _run_two_ranks(captured)
    assert np.array_equal(res[0][2], res[1][2])
    assert all(np.isfinite(res[0][3])) and all(np.isfinite(res[1][3]))
    if captured:
        ref = _run_two_ranks(False)
        te, tg = torch.from_numpy(ref[0][2]), torch.from_numpy(res[0][2])
        rel = ((tg - te).norm() / te.norm()).item()
        assert rel <= 2e-3, rel


@needs_two
def test_bench_self_launches_two_rccl_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two RCCL ranks as child processes and relays rank 0's
    line (reference: experiments/conf/cifar10.yaml:4-8, `devices` ranks under Lightning's DDP strategy)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--batch", "32"], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["value"] > 0
    assert line["config"]["collective"].startswith("rccl")


def _worker_tail(port, q):
    """CIFAR-10 net, forced one-rank RCCL reducer, the data-parallel grouping policy (networks.W3_TAIL = 4): log the order
    of grouped 3x3 weight-gradient launches and bucket all-reduce launches of one backward pass."""
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        sys.path.insert(0, ROOT)
        import tinyedm_amd as T
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        import tinyedm
        from tinyedm.config import compose, instantiate
        from tinyedm_amd import networks as N, ops
        from tinyedm_amd.ddp import GradReducer
        cfg = compose("cifar10", os.path.join(ROOT, "experiments", "conf"))
        tinyedm.manual_seed(cfg.seed)
        torch.manual_seed(cfg.seed)
        model = instantiate(cfg.model).to(dev).train()
        base = model.configure_optimizers()["optimizer"]
        base.fuse_zero_grad = True
        red = GradReducer(base.arena, force=True)
        N.W3_TAIL = GradReducer.W3_TAIL                 # what a multi-rank reducer switches on
        g = torch.Generator().manual_seed(1)
        x = (0.5 * torch.randn(16, 3, 32, 32, generator=g)).to(dev)
        log = []
        o_launch, o_w3 = red._launch, ops.wgrad3_group
        red._launch = lambda b: (log.append(("allreduce", (b["hi"] - b["lo"]) * 4)), o_launch(b))[1]
        ops.wgrad3_group = lambda items: (log.append(("wgrad3", len(items))), o_w3(items))[1]
        try:
            for step in range(3):                       # (the layer count of a pass is learned from the previous one)
                del log[:]
                loss = model.training_step((x, None), step)
                loss.backward()
                in_backward = len(log)
                base.grad_scale = red.finish()
                base.step()
                base.zero_grad()
            torch.cuda.synchronize()
        finally:
            ops.wgrad3_group = o_w3
            N.W3_TAIL = 0
        assert torch.isfinite(loss).item()
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", dict(log=list(log), in_backward=in_backward, arena_bytes=base.arena.numel * 4,
                          buckets=[(b["hi"] - b["lo"]) * 4 for b in red.buckets])))
    except Exception:          # noqa: BLE001
        import traceback
        q.put(("err", traceback.format_exc()))


def test_small_final_weight_gradient_group_leaves_little_to_reduce_after_backward():
    """Round-4 review #7: with the data-parallel grouping policy the LAST grouped weight-gradient launch of a pass holds 4 of
    the 43 3x3 layers (16 + 16 + 7 + 4), the buckets over the last-finished part of the arena are 8 MB instead of 32, and at
    least 85 % of the gradient bytes have their all-reduce ISSUED before that last launch -- what remains (<= 15 %) is all an
    8-rank step can have exposed behind its backward pass."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_tail, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=600)
    p.join(120)
    assert status == "ok", out
    log, total = out["log"], out["arena_bytes"]
    groups = [n for kind, n in log if kind == "wgrad3"]
    assert len(groups) == 4 and groups[:2] == [16, 16] and groups[3] == 4 and sum(groups) in (42, 43), groups
    last = max(i for i, (kind, _) in enumerate(log) if kind == "wgrad3")
    before = sum(n for kind, n in log[:last] if kind == "allreduce")
    assert sum(n for kind, n in log if kind == "allreduce") == total          # every byte is reduced exactly once
    frac = before / total
    print(f"all-reduce bytes issued before the last weight-gradient launch: {frac:.3f} of {total / 2**20:.1f} MiB; "
          f"buckets (MiB): {[round(b / 2**20, 1) for b in out['buckets']]}")
    from parity_log import record
    record("ddp/allreduce_bytes_issued_before_last_wgrad_launch (fraction; must be >= limit)", 1.0 - frac, 0.15)
    assert frac >= 0.85, frac

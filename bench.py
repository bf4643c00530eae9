"""Headline benchmark: CIFAR-10 32x32 EDM2 U-Net (35.6 M params) bf16 training images/sec on N MI355X,
plus 32-step Heun sampler images/sec, roofline of the dominant kernel and a CPU baseline.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = Diffuser -> Embedding -> Denoiser fwd -> sigma-weighted MSE -> backward -> bucketed gradient
all-reduce over RCCL (overlapped with backward) -> fused Adam+EMA, on a synthetic device-resident batch
(x = 0.5*randn(B,3,32,32), seed 42; conf/cifar10.yaml model, dropout 0.13, EMA 0.13, lr 0.02).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails in legacy mode); the driver
# exports this already, keep it for hand launches.  Must be set before the HIP runtime initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks as CHILD
    processes -- `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>`, the reference's
    `trainer.devices: N` / `strategy: ddp` (/root/reference/experiments/conf/cifar10.yaml:4-8) -- relay their output
    (stdout is inherited: rank 0's ONE JSON line) and exit with their return code.  This runs before torch is imported:
    the parent never touches the GPU and nothing that has is ever re-exec'ed."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    n = pre.parse_known_args()[0].gpus
    if n <= 1:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd).returncode)


if __name__ == "__main__":
    _self_launch()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import tinyedm_amd  # noqa: E402,F401  (before the first GPU call: sets the HIP runtime flag hipGraph replay needs)


def note(msg):
    """progress line on stderr (the JSON result line is the only thing on stdout)."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Threads for the CPU baseline = the cores this process may actually USE: the scheduler affinity, cut down to the
    cgroup CPU quota when one is set.  A GPU box exposes every logical CPU of the host (256) to each tenant but grants
    a share of them (16 per GPU): without a visible quota the share is taken to be 16 (EDM_CPU_THREADS overrides) --
    running 256 oversubscribed threads there took minutes per step."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()
            if q != "max":
                quota = max(1, -(-int(q) // int(p)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, p = int(f1.read()), int(f2.read())
                if q > 0:
                    quota = max(1, -(-q // p))
        except (OSError, ValueError):
            pass
    if "EDM_CPU_THREADS" in os.environ:
        return max(1, int(os.environ["EDM_CPU_THREADS"]))
    return max(1, min(n, quota if quota is not None else 16))

# work per unit, CIFAR-10 config (SURVEY.md 8(d) / BASELINE.md 3)
TRAIN_GFLOP_PER_IMG = 81.0
FWD_GFLOP_PER_IMG = 27.0
MFMA_PEAK_TFLOPS = 2500.0      # bf16 dense, MI355X_MICROARCH.md chip table
SUSTAINED_MFMA_TFLOPS = 1660.0  # measured: bare v_mfma_f32_16x16x32_bf16 + ds_read loop, all CUs, power-limited clock (tools/mfma_shape)
HBM_PEAK_GBS = 8000.0


def pmc_traffic(kernel):
    """HBM GB per launch of `kernel` from the committed PMC summary (tools/pmc_hbm.py; counters cannot be read from
    inside the process, so this is the offline measurement of the same command), or None."""
    import glob
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))[-1]     # newest round's pass
        with open(path) as f:
            ks = json.load(f)["kernels"]
        rec = (next((v for k, v in ks.items() if k.startswith(kernel + "<5, 0, 4")), None) or ks.get(kernel) or
               next((v for k, v in ks.items() if k.startswith(kernel + "<")), None))
        return None if rec is None else round(rec["hbm_gb_per_launch"], 4)
    except (OSError, ValueError, KeyError, IndexError):
        return None


def pmc_source():
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))
    return ("offline: " + os.path.relpath(paths[-1], ROOT) + " (committed rocprofv3 --pmc passes of this command on the replayed "
            "graph; counters cannot be read from inside the timed process)") if paths else None


def vendor_gemm_tflops(device, M=131072, N=256, K=2304, iters=10):
    """torch.matmul (hipBLASLt) bf16 on the implicit-GEMM shape of the dominant conv, timed with HIP events here"""
    try:
        a = torch.randn(M, K, device=device, dtype=torch.bfloat16)
        b = torch.randn(K, N, device=device, dtype=torch.bfloat16)
        for _ in range(3):
            torch.matmul(a, b)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            torch.matmul(a, b)
        e.record()
        torch.cuda.synchronize()
        return round(2.0 * M * N * K * iters / (s.elapsed_time(e) * 1e-3) / 1e12, 1)
    except Exception:       # noqa: BLE001  (the comparison figure must not sink the bench line)
        return None


def build_model(device, conditional=False):
    import tinyedm
    from tinyedm.config import compose, instantiate
    cfg = compose("cifar10_cond" if conditional else "cifar10", os.path.join(ROOT, "experiments", "conf"))
    tinyedm.manual_seed(cfg.seed)
    torch.manual_seed(cfg.seed)
    model = instantiate(cfg.model).to(device)
    return model, cfg


def timed_region(args, step, world, device, marks=None, last=None):
    """The contract's timed region: EXACTLY args.steps steps between barrier + synchronize on both sides, MAX over ranks.
    marks (optional) receives the host timestamps t0, t1 .. tK; last the final step's loss."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if marks is not None:
        marks.append(t0)
    loss = None
    for i in range(args.steps):
        loss = step(args.warmup + i)
        if marks is not None:
            marks.append(time.perf_counter())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    if last is not None:
        last.append(loss)
    return dt


def train_bench(args, rank, world, device):
    import tinyedm
    from tinyedm_amd import ops
    from tinyedm_amd.ddp import GradReducer
    from tinyedm_amd.ema import EMAOptimizer

    model, cfg = build_model(device, args.conditional)
    model.train()
    if world > 1:
        # as Trainer._setup_distributed: equal weights (same seed + broadcast), DIFFERENT Philox streams per rank -- the
        # ranks must not noise / drop out their shards with identical draws
        from tinyedm_amd import networks as _n
        _n.rng.seed = (_n.rng.seed + 0x9E3779B97F4A7C15 * rank) & 0xFFFFFFFFFFFFFFFF
    opt_cfg = model.configure_optimizers()
    base = opt_cfg["optimizer"]
    base.fuse_zero_grad = True            # as Trainer.fit: the gradient arena is cleared by the Adam pass that reads it
    gamma = tinyedm.sigma_rel_to_gamma(model.ema_length)
    opt = EMAOptimizer(base, device=device, gamma=gamma, every_n_steps=model.every_n_steps)
    reducer = GradReducer(base.arena)
    reducer.broadcast_parameters()
    B = args.batch
    g = torch.Generator().manual_seed(42 + rank)
    x = (0.5 * torch.randn(B, 3, 32, 32, generator=g)).to(device)
    y = torch.randint(0, 10, (B,), generator=g).to(device)
    batch = (x, y)

    def step(i):
        loss = model.training_step(batch, i)
        model.backward(loss)              # (Lightning's hook: what its automatic optimisation calls; trainer.LightningModule)
        base.grad_scale = reducer.finish()
        opt.step()
        opt.zero_grad()
        return loss.detach()

    # The whole step (fwd + bwd + bucketed gradient all-reduce + Adam/EMA) is replayed from a hipGraph; the per-step
    # scalars come from a device record (tinyedm_amd/graph.py).  With N > 1 ranks (or EDM_FORCE_REDUCE=1) the RCCL
    # all-reduces are nodes of the graph: the hooks issue them on the comm stream forked from the capture stream.
    eager_step = step
    from tinyedm_amd import _lib as _L, _runtime_env as _RE
    can_graph = _RE.GRAPH_REPLAY_SAFE and (not reducer.active or reducer.capturable())
    if args.step_launch == "graph" and not can_graph:
        # an explicit request that cannot be honoured is an error, not a silent eager run (a profile labelled "graph replay"
        # must be one): e.g. under `rocprofv3 --pmc` the profiler initialises the GPU before python starts, so the runtime
        # flag must be inherited (export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)
        raise SystemExit("bench.py: --step-launch graph, but hipGraph replay is not available here ("
                         + ("the HIP runtime was initialised before DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 could be set: export it"
                            if not _RE.GRAPH_REPLAY_SAFE else "the reducer's collectives are not capturable") + ")")
    mode = args.step_launch if can_graph else "eager"
    launch_info = {}
    opt.zero_grad()
    note(f"model built, warmup {args.warmup} steps")
    for i in range(args.warmup):
        loss = eager_step(i)
    torch.cuda.synchronize()

    def probe(fn, n=5):
        """(wall ms per step, host enqueue ms per step, entry-point calls per step) over n steps.  The host figure is
        the time one step's enqueue takes with an EMPTY device queue (a synchronisation before each of 5 extra steps):
        in the back-to-back loop the runtime blocks the host once enough work is queued, which measures the GPU."""
        torch.cuda.synchronize()
        c0, t0 = _L.N_CALLS, time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        calls = (_L.N_CALLS - c0) / n
        hs = []
        for i in range(5):
            torch.cuda.synchronize()
            a = time.perf_counter()
            fn(n + i)
            hs.append(time.perf_counter() - a)
        torch.cuda.synchronize()
        return (t2 - t0) / n * 1e3, sorted(hs)[len(hs) // 2] * 1e3, calls
    if rank == 0 or world > 1:
        reducer.measure = reducer.active             # events around finish()'s wait for the comm stream
        e_ms, e_host, e_calls = probe(eager_step, 10)
        reducer.measure = False
        launch_info = {"eager_probe_ms": round(e_ms, 3), "host_enqueue_ms_per_step": round(e_host, 3),
                       "entry_point_calls_per_step": round(e_calls, 1)}
        exp_ms = reducer.exposed_comm_ms()
        if exp_ms is not None:
            # what the eager step could NOT hide of its gradient all-reduces: the time the main stream waited for the comm
            # stream after the backward pass had ended (mean over the probe's steps; slowest rank)
            if world > 1:
                t = torch.tensor([exp_ms], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                exp_ms = t.item()
            launch_info["exposed_comm_ms"] = round(exp_ms, 3)
            launch_info["grad_buckets"] = len(reducer.buckets)
    watchdog = None
    if can_graph and mode in ("auto", "graph") and (world > 1 or os.environ.get("EDM_BENCH_WATCHDOG") == "1"):   # (test hook)
        # Insurance for the one configuration this code cannot rehearse on a one-GPU box: capturing a step whose RCCL
        # all-reduces span several ranks.  A capture that RAISES falls back to the eager step below; one that HANGS would
        # leave the job without its line -- so the eager step is timed first (the contract's K steps, barriers and all), and
        # a timer prints that line and ends every rank if capture + first replays have not finished after 180 s.
        import threading
        e_dt = timed_region(args, eager_step, world, device)
        wd_lock = threading.Lock()              # bail() and the main thread agree on who prints the ONE line
        wd_state = {"done": False, "phase": "capture: warm-up steps on the capture stream"}

        def bail():
            with wd_lock:
                if wd_state["done"]:            # the main thread got past capture + probe while the timer was firing
                    return
                wd_state["done"] = True
                # what was stuck: the phase the main thread announced last, and how far the reducer got in this pass
                fired = [i for i, b in enumerate(reducer.buckets) if b["work"] is not None]
                where = (f"rank {rank}: {wd_state['phase']}; buckets issued this pass: {len(fired)} of {len(reducer.buckets)}"
                         + (f" (last: #{fired[-1]}, arena [{reducer.buckets[fired[-1]]['lo']}, {reducer.buckets[fired[-1]]['hi']}))"
                            if fired else ""))
                note(f"graph capture watchdog fired -- {where}")
                if rank == 0:
                    line = {"metric": "train imgs/sec CIFAR-10 32x32 bf16", "value": round(args.batch * world * args.steps / e_dt, 2),
                            "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                            "ms_per_step": round(e_dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                            "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "capture_hang": True,
                            "config": {"workload": "CIFAR-10 32x32 unconditional EDM2 U-Net (conf/cifar10.yaml, 35.6M params) "
                                                   "full training step: diffuse+embed+denoiser fwd/bwd+loss+grad all-reduce+Adam+EMA",
                                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                                       "step_launch": "eager (capturing the collective-bearing step did not finish in time)",
                                       "capture_hang_where": where,
                                       "collective": "rccl bucketed all-reduce", **launch_info}}
                    print(json.dumps(line), flush=True)
                # the eager figures above are valid, but a hung capture / collective is a FAILURE of this run: say so to the
                # launcher (never re-exec or retry from here: the process has touched the GPU)
                os._exit(3)
        watchdog = threading.Timer(float(os.environ.get("EDM_BENCH_CAPTURE_TIMEOUT", "180")), bail)
        watchdog.daemon = True
        watchdog.start()
    if can_graph and mode in ("auto", "graph"):
        from tinyedm_amd.graph import CapturedTrainStep
        captured, ok = None, 1
        try:
            captured = CapturedTrainStep(model, opt, reducer=reducer)
            for i in range(CapturedTrainStep.WARMUP + 1):    # warm-up on the capture stream, then the capture itself
                if watchdog is not None:
                    wd_state["phase"] = (f"capture: warm-up step {i}" if i < CapturedTrainStep.WARMUP else
                                         "capture: stream capture of the step (training_step, backward, bucket all-reduces, Adam+EMA)")
                captured(batch)
            if captured.fallback:            # the step's own cross-rank agreement: some rank could not capture
                ok = 0
            if watchdog is not None:
                wd_state["phase"] = "cross-rank agreement on the capture (all_reduce MIN)"
        except Exception as e:      # noqa: BLE001  (a runtime that cannot capture the collectives must not sink the line)
            note(f"capturing the step failed ({type(e).__name__}: {str(e)[:200]}); timing the eager step")
            ok = 0
        if world > 1:               # every rank replays, or none does
            flag = torch.tensor([ok], device=device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            can_graph, mode = False, "eager"
            launch_info["graph_capture"] = "failed on at least one rank"
    if can_graph and mode in ("auto", "graph"):
        if watchdog is not None:
            wd_state["phase"] = "first replays of the captured step (graph upload + RCCL nodes)"
        for i in range(3):                                   # the first replays upload the executable graph
            captured(batch)
        if watchdog is not None:
            torch.cuda.synchronize()
            wd_state["phase"] = "10-step replay probe"
        g_ms, g_host, _ = probe(lambda i: captured(batch), 10)
        launch_info.update({"graph_probe_ms": round(g_ms, 3), "graph_host_ms_per_step": round(g_host, 3)})
        if mode == "auto":
            pair = torch.tensor([g_ms, launch_info["eager_probe_ms"]], device=device, dtype=torch.float64)
            if world > 1:           # one decision for the whole job: the slowest rank's figures
                dist.all_reduce(pair, op=dist.ReduceOp.MAX)
            mode = "graph" if float(pair[0]) < float(pair[1]) else "eager"
        note(f"step launch probe: eager {launch_info['eager_probe_ms']:.2f} ms (host enqueue {e_host:.2f} ms), "
             f"hipGraph replay {g_ms:.2f} ms -> timing the {mode} step")
    if watchdog is not None:
        watchdog.cancel()
        with wd_lock:                           # a timer that already started printing finishes (and exits) first
            wd_state["done"] = True
    if mode == "graph":
        def step(i):                            # noqa: F811
            return captured(batch)
    use_graph = mode == "graph"
    launch_info["step_launch"] = "hipGraph replay" if use_graph else "eager"
    note(f"warmup done, timing {args.steps} steps ({launch_info['step_launch']})")
    marks, last = [], []
    dt = timed_region(args, step, world, device, marks, last)
    host = [(b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:])]
    launch_info["host_step_ms_max"] = round(max(host), 2)
    launch_info["host_step_ms_median"] = round(sorted(host)[len(host) // 2], 2)
    loss = last[0]
    final_loss = float(loss.detach())
    wnorm = float(base.arena.theta.norm())
    if not (final_loss == final_loss and abs(final_loss) < 1e4 and wnorm == wnorm and wnorm < 1e6):
        # a diverged / corrupted run draws less power, clocks higher and "wins": never report it
        raise RuntimeError(f"bench: loss {final_loss}, |weights| {wnorm} after the timed region -- the step is broken")
    note(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step")

    # ---- roofline of the dominant kernel (3x3 implicit-GEMM conv): extra, instrumented EAGER steps AFTER the
    # timed region: HIP events around every launch on the launch stream; algorithmic FLOPs from the shapes.
    # Every rank runs the extra steps (they contain the gradient all-reduce); only rank 0 instruments them.  Kernels run
    # back to back on one stream (the weight-gradient side stream is off by default since round 2); when it is enabled
    # (EDM_WGRAD_STREAM=1) a second instrumented step records the per-kernel durations under that overlap too.
    from tinyedm_amd import networks as _nets
    roof = None
    roofs = {}
    INSTR = 5          # instrumented steps per mode, behind 2 uninstrumented ones: a single step right after the
    nstep = [args.warmup + args.steps]   # synchronisation ran the big kernels 8 % slower than their steady state (clock ramp)
    for overlapped in ((False, True) if _nets.WGRAD_STREAM else (False,)):
        saved = _nets.WGRAD_STREAM
        _nets.WGRAD_STREAM = saved and overlapped
        for _ in range(2):
            eager_step(nstep[0]); nstep[0] += 1
        if rank == 0:
            ops.PROFILE = {}
        for _ in range(INSTR):
            eager_step(nstep[0]); nstep[0] += 1
        torch.cuda.synchronize()
        _nets.WGRAD_STREAM = saved
        if rank == 0:
            roofs[overlapped], ops.PROFILE = ops.PROFILE, None
    if rank == 0:
        roof = {}
        for name, recs in roofs[False].items():           # per-step figures: sums over the INSTR steps / INSTR
            ms = sum(s.elapsed_time(e) for s, e, _, _ in recs) / INSTR
            roof[name] = {"launches": len(recs) // INSTR, "ms": ms, "gflop": sum(f for _, _, f, _ in recs) / 1e9 / INSTR,
                          "gbytes": sum(b for _, _, _, b in recs) / 1e9 / INSTR}
        for name, recs in roofs.get(True, {}).items():
            if name in roof:
                roof[name]["ms_overlapped"] = sum(s.elapsed_time(e) for s, e, _, _ in recs) / INSTR
    return model, B * world * args.steps / dt, dt / args.steps * 1e3, final_loss, roof, launch_info


F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md chip table (= the fp32 vector rate)


def sampler_bench(args, model, device, network_dtype="bf16"):
    """32-step Heun solve (63 network evaluations) of the CIFAR-10 net, hipGraph-captured.  network_dtype "bf16": the
    training path's kernels; "f32": the reference-precision evaluation (exact-fp32 kernels, csrc/eval_f32.hip) -- the
    reference itself samples in fp32 (generate.py:39-44)."""
    import tinyedm
    from tinyedm_amd import _runtime_env as _RE
    model.eval()
    model.denoiser.set_eval_dtype(network_dtype)
    solver = tinyedm.DeterministicSolver(num_steps=32)
    f32 = network_dtype == "f32"
    split = network_dtype == "f32x3"
    B = args.sampler_f32_batch if f32 else args.sampler_batch
    iters = 1 if f32 else args.sampler_iters
    graph = _RE.GRAPH_REPLAY_SAFE
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(B, 3, 32, 32, generator=g).to(device)
    note(f"sampler ({network_dtype} network): warm-up" + (" + hipGraph capture" if graph else ""))
    solver.solve(model, x0, None, graph=graph)         # capture + first replay
    torch.cuda.synchronize()
    note(f"sampler ({network_dtype} network): timing")
    t0 = time.perf_counter()
    for _ in range(iters):
        out = solver.solve(model, x0, None, graph=graph)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    if not bool(torch.isfinite(out).all()):
        raise RuntimeError("bench: the sampler produced non-finite images")
    model.denoiser.set_eval_dtype("bf16")
    peak = F32_MFMA_PEAK_TFLOPS if f32 else MFMA_PEAK_TFLOPS
    passes = 3 if split else 1      # split-bf16: every conv product is three bf16 MFMA passes (hi.w_hi + hi.w_lo + lo.w_hi)
    return {"img_per_s": B / dt, "batch": B, "heun_steps": 32, "nfe": 63, "ms_per_solve": dt * 1e3,
            "hipgraph": bool(graph), "state_dtype": "f32", "network_dtype": network_dtype,
            "mfma_frac": round(B / dt * 63 * FWD_GFLOP_PER_IMG * passes / 1e3 / peak, 4),
            "mfma_peak_tflops": peak, "mfma_passes_per_product": passes}


def cpu_baseline(args):
    """Reference algorithm on the host cores: the CPU oracle's training step (plain torch CPU ops: F.conv2d, SDPA
    math, Adam + EMA; the same CIFAR-10 model, dropout 0.13 applied with F.dropout), fp32 and -- as the reference's
    `precision: bf16-mixed` -- under torch.autocast("cpu", bfloat16).  Bounded: ~12 s per leg."""
    from oracle import edm_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    note(f"cpu baseline on {cores} host threads ({os.cpu_count()} logical CPUs visible)")
    ecfg, dcfg = O.cifar10_cfg()
    B = args.cpu_batch

    def leg(autocast, budget_s):
        P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(1), gains_nonzero=True)
        keys = O.trainable_keys(P)
        for k in keys:
            P[k].requires_grad_(True)
        g = torch.Generator().manual_seed(42)
        clean = 0.5 * torch.randn(B, 3, 32, 32, generator=g)
        m = {k: torch.zeros_like(P[k]) for k in keys}
        v = {k: torch.zeros_like(P[k]) for k in keys}
        ema = {k: P[k].detach().clone() for k in keys}
        times = []
        budget_t0 = time.perf_counter()
        for it in range(1 + args.cpu_steps):
            t0 = time.perf_counter()
            eps, noise = torch.randn(B, generator=g), torch.randn(B, 3, 32, 32, generator=g)
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                loss = O.training_loss(P, ecfg, dcfg, clean, eps, noise, -1.2, 1.2)
            grads = torch.autograd.grad(loss.float(), [P[k] for k in keys])
            with torch.no_grad():
                for k, gr in zip(keys, grads):
                    O.adam_step(P[k], gr.float(), m[k], v[k], it + 1, 0.02)
                    O.ema_step(ema[k], P[k], O.ema_beta(it, 4.6036))
            if it > 0:
                times.append(time.perf_counter() - t0)
            note(f"  cpu step {it} ({'bf16 autocast' if autocast else 'fp32'}): {time.perf_counter() - t0:.2f} s")
            if time.perf_counter() - budget_t0 > budget_s and times:
                break
        return sum(times) / len(times), len(times)

    sec, n = leg(False, 12.0)
    out = {"value": B / sec, "unit": "img/s", "cores": cores, "cores_visible": os.cpu_count(), "kind": "port",
           "sample": f"{n} fp32 training steps of the CPU oracle (oracle/edm_oracle.py), batch {B}, same CIFAR-10 model, "
                     f"dropout 0.13, {sec:.2f} s/step"}
    try:
        sec_bf, n_bf = leg(True, 12.0)
        out["bf16_autocast"] = {"value": B / sec_bf, "unit": "img/s",
                                "sample": f"{n_bf} steps under torch.autocast('cpu', bfloat16), {sec_bf:.2f} s/step"}
    except Exception as e:          # noqa: BLE001  (a CPU without a usable bf16 path must not sink the bench line)
        out["bf16_autocast"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 x ~13 ms: the third digit of a 20-step run is noise)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (BASELINE.json configs[1]: 128)")
    ap.add_argument("--conditional", action="store_true")
    ap.add_argument("--sampler-batch", type=int, default=512)
    ap.add_argument("--sampler-iters", type=int, default=2)
    ap.add_argument("--no-sampler", action="store_true")
    ap.add_argument("--no-sampler-fp32", action="store_true", help="skip the reference-precision (fp32 network) sampler leg")
    ap.add_argument("--sampler-f32-batch", type=int, default=256,
                    help="batch of the exact-fp32 sampler leg (the split-bf16 leg runs at --sampler-batch since round 6: at 512 "
                         "its 8x8 / 16x16 layers fill the chip with the larger tiles, +6 %% img/s)")
    ap.add_argument("--step-launch", choices=["auto", "graph", "eager"], default="auto",
                    help="time the hipGraph replay of the step (with N > 1 ranks the RCCL all-reduces are nodes of the "
                         "graph), the eager Python step, or (auto) whichever a 10-step probe finds faster")
    ap.add_argument("--no-graph", action="store_true", help="same as --step-launch eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=20, help="timed CPU-oracle steps (~0.5 s each on 16 threads)")
    args = ap.parse_args()
    if args.no_graph:
        args.step_launch = "eager"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    # Test hook (tests/test_ddp_gpu.py): EDM_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 with the gloo backend so the
    # N>1 code path can be exercised on a one-GPU box (RCCL refuses two ranks on one device).  Never set by the driver.
    one_device = os.environ.get("EDM_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # EDM_FORCE_REDUCE=1 (tests/test_rccl_gpu.py): run the RCCL path -- process group, broadcast, hook-driven bucket
    # all-reduces on the comm stream -- even with one rank, so a one-GPU box executes it
    forced = os.environ.get("EDM_FORCE_REDUCE") == "1"
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints a version banner on STDOUT when its communicator is created; stdout carries the ONE JSON line, so
        # the communicator is brought up (init + a first collective) with fd 1 pointed at stderr
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if one_device:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            dist.all_reduce(torch.zeros(1, device=device))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    model, ips, ms, final_loss, roof, launch_info = train_bench(args, rank, world, device)
    out = None
    if rank == 0:
        # dominant kernel: k_conv3x3_v6<5,0,4> (3x3 implicit GEMM of the 32x32 layers: forward convs, with the modulation
        # epilogue on the first conv of each block, and plain dgrads; the two backward-epilogue instantiations
        # <5,1>/<5,2> are reported beside it in per_kernel as ..._v6_modbwd / ..._v6_silubwd).  per_kernel keys name the
        # kernel that ran (ops.V46: _v6, or _v4 under EDM_V4_MFMA16=0)
        from tinyedm_amd import ops as _ops
        mfma16 = _ops.V46 == "_v6"
        conv = roof.get("conv3x3_igemm" + _ops.V46, {"launches": 0, "ms": 0.0, "gflop": 0.0, "gbytes": 0.0})
        achieved = conv["gflop"] / conv["ms"] if conv["ms"] > 0 else 0.0          # GFLOP/ms == TFLOP/s
        kname = "k_conv3x3_v6" if mfma16 else "k_conv3x3_v4"
        traffic = pmc_traffic(kname)
        out = {
            "metric": "train imgs/sec CIFAR-10 32x32 bf16",
            "value": round(ips, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "CIFAR-10 32x32 unconditional EDM2 U-Net (conf/cifar10.yaml, 35.6M params) "
                                   "full training step: diffuse+embed+denoiser fwd/bwd+loss+grad all-reduce+Adam+EMA",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "conditional": bool(args.conditional), "final_loss": final_loss,
                       **launch_info,
                       "collective": ("rccl all-reduce (forced, 1 rank)" if forced else "rccl bucketed all-reduce")
                       if (world > 1 or forced) else "none"},
            "roofline": {
                "bound": "mfma", "kernel": kname + "<5,0,4> (3x3 implicit-GEMM conv of the 32x32 layers: forward, incl. fused "
                                                   "modulation epilogue, and plain dgrad; "
                                                   + ("v_mfma_f32_16x16x32_bf16" if mfma16 else "v_mfma_f32_32x32x16_bf16") + ")",
                "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": pmc_source(),
                "traffic_unit": "GB of HBM traffic per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc "
                                "passes, newest profiles/r*_pmc_hbm.json)",
                "algorithmic_gbytes_per_launch": round(conv["gbytes"] / max(1, conv["launches"]), 4),
                # the vendor GEMM (torch.matmul -> hipBLASLt) on the conv-as-GEMM shape of this kernel, M = 131072 pixels,
                # N = 256, K = 2304, plain row-major operands with no im2col work -- measured in THIS run
                "library_gemm_tflops_same_shape": vendor_gemm_tflops(device),
                "achieved_overlapped": round(conv["gflop"] / conv["ms_overlapped"], 2) if conv.get("ms_overlapped") else None,
                "launches_per_step": conv["launches"],
                "avg_launch_ms": round(conv["ms"] / max(1, conv["launches"]), 4),
                "algorithmic_gflop_per_step": round(conv["gflop"], 1),
                # what the matrix pipe SUSTAINS on this power-limited part (not the contract's `peak`, kept above): a bare
                # MFMA + LDS loop of the same instruction on all 256 CUs holds 1.80 GHz x 88 % = 1 660 TFLOP/s
                # (tools/mfma_shape); inside this kernel's main loop the chip runs 1.845 GHz with the pipe 81 % busy
                # (tools/v6_timeline.py, profiles/r04_v6_timeline_clock.txt) -- clock x occupancy is what the power cap fixes
                "sustained_mfma_tflops": SUSTAINED_MFMA_TFLOPS,
                "frac_of_sustained": round(achieved / SUSTAINED_MFMA_TFLOPS, 4),
                "whole_step_mfma_frac": round(ips / world * TRAIN_GFLOP_PER_IMG / 1e3 / MFMA_PEAK_TFLOPS, 4),
                "whole_step_hbm_frac": round(ips / world * 108.6e6 / 1e9 / HBM_PEAK_GBS, 4),
                # BASELINE.json's literal "HBM-bound 3x3-conv roofline": 3x3-conv activation bytes only (59.7 MB per
                # training image) over 8 TB/s; 70 % of it would need 7 PFLOP/s of bf16 MFMA (not reachable)
                "northstar_hbm3x3_frac": round(ips / world * 59.7e6 / 1e9 / HBM_PEAK_GBS, 4),
                # the other chip-filling MFMA kernel of the step: the grouped stream-K weight gradient (conv_wgrad3.hip)
                # avg_launch_ms / achieved / frac: the k_wgrad3 LAUNCH alone (events recorded inside the grouped entry point,
                # include/tinyedm_hip_diag.h edm_wgrad3_probe) -- what the rocprofv3 traces in profiles/ show for the kernel;
                # call_ms: the whole edm_wgrad3_group call (launch-table upload + k_wgrad3 + k_wgrad3_finish), the figure
                # rounds 2-5 reported as the kernel's
                "wgrad3": (lambda w, c: {"kernel": f"k_wgrad3<1> (3x3 weight gradients, {w['launches']} grouped launches per step)",
                                         "achieved": round(w["gflop"] / w["ms"], 2) if w["ms"] else 0.0,
                                         "frac": round(w["gflop"] / w["ms"] / MFMA_PEAK_TFLOPS, 4) if w["ms"] else 0.0,
                                         "avg_launch_ms": round(w["ms"] / max(1, w["launches"]), 4),
                                         "call_ms_incl_table_upload_and_finish": round(c["ms"] / max(1, c["launches"]), 4),
                                         "algorithmic_gbytes_per_launch": round(w["gbytes"] / max(1, w["launches"]), 4),
                                         "traffic": pmc_traffic("k_wgrad3")})(
                    roof.get("conv3x3_wgrad_kernel", roof.get("conv3x3_wgrad", {"launches": 0, "ms": 0.0, "gflop": 0.0, "gbytes": 0.0})),
                    roof.get("conv3x3_wgrad", {"launches": 0, "ms": 0.0, "gflop": 0.0, "gbytes": 0.0})),
                "per_kernel": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in roof.items()},
            },
        }
        # sampler + CPU baseline ride on the single-GPU line only (the N>1 lines are the training scaling series;
        # hipGraph capture beside a live RCCL communicator is avoided)
        if not args.no_sampler and world == 1:
            out["sampler"] = sampler_bench(args, model, device)
            if not args.no_sampler_fp32:
                # the reference-faithful figure: the reference samples in fp32; priced against the f32-input MFMA peak
                out["sampler_fp32"] = sampler_bench(args, model, device, "f32")
                # the same precision class (2^-17 per operand; 32-step trajectory 2e-6 from the exact path) on the bf16 kernels:
                # split-bf16, three MFMA passes per product (tinyedm_amd/networks.py _conv_f32)
                out["sampler_fp32_split"] = sampler_bench(args, model, device, "f32x3")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

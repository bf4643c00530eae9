"""Two data-parallel ranks running the REAL HIP training step on one GPU (both on cuda:0; the collective is gloo over
GPU tensors because RCCL refuses two ranks on one device): the bucketed reducer is driven by the weight-gradient
kernels' own hooks (gradients written directly into the arena on the side stream), must not deadlock, and must leave
every rank with the mean of the per-rank gradients, after which the fused Adam step keeps the replicas identical.
RCCL itself is only exercised by the driver's multi-GPU bench."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(device):
    import tinyedm_amd as T
    from oracle.make_golden import tiny_cfgs
    from tinyedm_amd import networks as N
    ecfg, dcfg = tiny_cfgs()
    N._rng_sub_counter[0] = 0               # per-block dropout sub-streams are numbered at construction
    T.manual_seed(11)
    torch.manual_seed(11)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                     tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                     tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), 0.1, dcfg.sigma_data,
                     dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim, dcfg.num_heads)
    with torch.no_grad():
        den.gain_out.fill_(0.7)
    model = T.EDM(diffuser=T.Diffuser(-1.2, 1.2), embedding=emb, denoiser=den, use_ema=False, use_uncertainty=False,
                  steady_steps=10, rampup_steps=2, scheduler_interval="step", lr=1e-3)
    return model.to(device).train(), ecfg


def _batch(ecfg, device):
    g = torch.Generator().manual_seed(5)
    x = 0.5 * torch.randn(8, 3, 16, 16, generator=g)
    y = torch.randint(0, ecfg.num_classes, (8,), generator=g)
    return x.to(device), y.to(device)


def _one_rank_grads(model, base, xs, ys):
    base.zero_grad()
    import tinyedm_amd as T
    T.manual_seed(11)                       # same Philox stream (noise, dropout) as the rank that saw this shard
    loss = model.training_step((xs, ys), 0)
    loss.backward()
    torch.cuda.synchronize()
    return base.arena.grad.clone()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tinyedm_amd.ddp import GradReducer
    model, ecfg = _build(dev)
    base = model.configure_optimizers()["optimizer"]
    red = GradReducer(base.arena, bucket_bytes=1 << 20)          # several buckets for the tiny net
    assert len(red.buckets) > 1 and red.world == 2
    red.broadcast_parameters()
    x, y = _batch(ecfg, dev)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    import tinyedm_amd as T
    out = []
    for it in range(2):                                         # two steps: reducer state must reset
        base.zero_grad()
        T.manual_seed(11)
        loss = model.training_step((xs, ys), 0)
        loss.backward()
        base.grad_scale = red.finish()
        torch.cuda.synchronize()
        out.append((base.arena.grad * base.grad_scale).cpu().numpy())
        if it == 1:
            base.step()
    torch.cuda.synchronize()
    q.put((rank, out[0], out[1], base.arena.theta.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hip_backward_reduces_to_the_mean_of_shard_gradients():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import time
    t0 = time.time()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    t1 = time.time()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    print(f"workers answered after {t1 - t0:.1f} s, exited after {time.time() - t0:.1f} s")
    # single-process reference: the two shard gradients computed one after the other, then averaged
    dev = torch.device("cuda", 0)
    model, ecfg = _build(dev)
    base = model.configure_optimizers()["optimizer"]
    x, y = _batch(ecfg, dev)
    g0 = _one_rank_grads(model, base, x[:4], y[:4])
    # the training forward renormalised the weights in place; rebuild so the second shard starts from the same weights
    model, _ = _build(dev)
    base = model.configure_optimizers()["optimizer"]
    g1 = _one_rank_grads(model, base, x[4:], y[4:])
    ref = (0.5 * (g0 + g1)).cpu().numpy()
    def close(a, b):
        # fp32 atomics (modulation / split-K reductions) make the summation order run-dependent: compare in norm
        return np.linalg.norm(a - b) <= 2e-3 * np.linalg.norm(b) and np.abs(a - b).max() <= 2e-2 * np.abs(b).max()
    for rank, a0, a1, theta in res:
        assert close(a0, ref), (np.linalg.norm(a0 - ref) / np.linalg.norm(ref), np.abs(a0 - ref).max())
        assert close(a1, ref)                                   # second step: same inputs, weights already normalised
    assert np.array_equal(res[0][3], res[1][3])                  # replicas stay bit-identical after the Adam step


def test_bench_runs_with_two_ranks_on_one_gpu():
    """bench.py's N>1 path end to end (rank set-up, barriers, max-over-ranks timing, the instrumented steps on every
    rank, one JSON line from rank 0), launched exactly as the driver does but with both ranks on cuda:0 over gloo."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EDM_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "8"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 16 and out["value"] > 0
    assert "sampler" not in out and "cpu_baseline" not in out and out["roofline"]["bound"] == "mfma"


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (WORLD_SIZE unset) starts the two ranks itself as child
    processes of a parent that never touches the GPU, relays rank 0's ONE line and returns the children's exit code
    (VERDICT r3 #2).  Here both ranks share cuda:0 over gloo (EDM_BENCH_ONE_DEVICE: a one-GPU box cannot host two RCCL
    ranks); tests/test_rccl_gpu.py::test_bench_self_launches_two_rccl_ranks is the same with real RCCL ranks on >= 2 GPUs."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["EDM_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "8"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "starting 2 ranks" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["value"] > 0

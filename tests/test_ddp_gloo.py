"""world_size-2 gloo test of the data-parallel path: the flat-arena bucketed reducer must give every rank the
mean of the per-rank gradients == the single-process gradient of the concatenated batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Gain(torch.nn.Module):
    """a 0-dim parameter between tensor parameters: lands in the arena's scalar tail (FlatArena layout 2), i.e. in the
    reducer's last bucket"""

    def __init__(self):
        super().__init__()
        self.g = torch.nn.Parameter(torch.tensor(1.3))

    def forward(self, x):
        return x * self.g


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), _Gain(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                               _Gain(), torch.nn.Linear(32, 4))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tinyedm_amd.ddp import GradReducer
    from tinyedm_amd.ema import FlatArena
    model = _model()
    if rank == 1:                               # different init on rank 1: broadcast must fix it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    arena = FlatArena(list(model.parameters()))
    red = GradReducer(arena, bucket_bytes=2048)  # several buckets
    assert len(red.buckets) > 2
    assert red.buckets[-1]["lo"] == arena.scalar_lo and len(red.buckets[-1]["params"]) == 2     # the scalar tail
    assert sorted(i for b in red.buckets for i in b["params"]) == list(range(len(arena.params)))
    assert sum(b["hi"] - b["lo"] for b in red.buckets) == arena.numel
    red.broadcast_parameters()
    g = torch.Generator().manual_seed(123)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    out = {}
    for it in range(2):                          # two steps: reducer state must reset
        arena.zero_grad()
        # gradient accumulation: first micro-batch without sync, second with
        red.enabled = False
        (torch.nn.functional.mse_loss(model(xs[:2]), ys[:2]) * 0.5).backward()
        red.enabled = True
        (torch.nn.functional.mse_loss(model(xs[2:]), ys[2:]) * 0.5).backward()
        scale = red.finish()
        out[it] = (arena.grad * scale).clone()
    q.put((rank, out[0].numpy(), out[1].numpy(), arena.theta.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference on the concatenated batch
    from tinyedm_amd.ema import FlatArena
    model = _model()
    arena = FlatArena(list(model.parameters()))
    g = torch.Generator().manual_seed(123)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    arena.zero_grad()
    torch.nn.functional.mse_loss(model(X), Y).backward()
    for rank, g0, g1, theta in res:
        g0, g1, theta = torch.from_numpy(g0), torch.from_numpy(g1), torch.from_numpy(theta)
        assert torch.allclose(theta, arena.theta)                       # broadcast from rank 0
        assert torch.allclose(g0, arena.grad, rtol=1e-5, atol=1e-6)
        assert torch.allclose(g1, arena.grad, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------- bf16 transport
def _worker_bf16(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tinyedm_amd.ddp import GradReducer
    from tinyedm_amd.ema import FlatArena
    res = {}
    for transport in ("fp32", "bf16"):
        model = _model()
        arena = FlatArena(list(model.parameters()))
        red = GradReducer(arena, bucket_bytes=2048, transport=transport)
        red.broadcast_parameters()
        red.broadcast_buffers(model)
        g = torch.Generator().manual_seed(321)
        X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
        arena.zero_grad()
        torch.nn.functional.mse_loss(model(X[rank * 4:(rank + 1) * 4]), Y[rank * 4:(rank + 1) * 4]).backward()
        scale = red.finish()
        res[transport] = (arena.grad * scale).clone().numpy()
    q.put((rank, res["fp32"], res["bf16"]))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_transport_is_equivalent_to_fp32_up_to_bf16_rounding():
    """GradReducer(transport="bf16"): half the bytes on the wire; each element is rounded to bf16 before the sum and
    the sum once more, so it equals the fp32 all-reduce to ~2^-8 relative (and is identical on every rank)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bf16, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    f0, b0 = torch.from_numpy(res[0][1]), torch.from_numpy(res[0][2])
    f1, b1 = torch.from_numpy(res[1][1]), torch.from_numpy(res[1][2])
    assert torch.equal(f0, f1) and torch.equal(b0, b1)                    # replicas agree bit for bit
    rel = ((b0 - f0).norm() / f0.norm()).item()
    assert 0 < rel <= 2 ** -7, rel
    assert (b0 - f0).abs().max() <= 2 ** -6 * f0.abs().max()


# ---------------------------------------------------------------------------------------------- per-rank RNG across a resume
def _worker_resume(rank, world, port, q, ckpt):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from tinyedm_amd import networks
    from tinyedm_amd.trainer import Trainer
    networks.manual_seed(42)
    model = _model()
    model.hparams = {}
    tr = Trainer(max_epochs=1)
    tr._setup_distributed(model)                 # gloo group + per-rank Philox offset
    seed_fresh = networks.rng.seed
    tr._model = model
    networks.rng.step = 17
    tr.save_checkpoint(ckpt, model)              # rank 0 writes; every rank would write the same base seed
    dist.barrier()
    # ---- a new run resumes: same launch sequence as Trainer.fit (setup, then load_checkpoint)
    networks.manual_seed(42)
    tr2 = Trainer(max_epochs=1)
    tr2._setup_distributed(model)
    tr2.load_checkpoint(ckpt, model)
    q.put((rank, seed_fresh, networks.rng.seed, networks.rng.step))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_keep_distinct_philox_seeds_across_a_resume(tmp_path):
    """Trainer._setup_distributed offsets every rank's Philox seed; the checkpoint stores the BASE seed and
    load_checkpoint re-applies the rank offset -- after fit(ckpt_path=...) the ranks must still differ (and each rank
    must continue ITS OWN stream: same seed as before the checkpoint, same step)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ckpt = str(tmp_path / "resume.ckpt")
    procs = [ctx.Process(target=_worker_resume, args=(r, world, port, q, ckpt)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=60) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, fresh0, resumed0, step0), (_, fresh1, resumed1, step1) = res
    assert fresh0 != fresh1 and resumed0 != resumed1
    assert resumed0 == fresh0 and resumed1 == fresh1
    assert step0 == step1 == 17
    assert torch.load(ckpt, weights_only=False)["tinyedm_amd"]["rng_seed"] == 42


def test_bucket_plan_keeps_late_gradients_out_of_the_body_and_shrinks_the_tail():
    """Round 5: a bucket's all-reduce starts when ALL its gradients are final, so (1) the parameters whose
    gradients are written at the very end of a backward pass (flagged `_edm_late`: the blocks' embed Linears, the Embedding
    module; and the 0-dim gains) live in the arena's tail = the reducer's LAST bucket, none of them inside a body bucket,
    and (2) the buckets over the last-finished 16 MB of the body (lowest offsets = first layers) are at most 8 MB while the
    others go up to the requested 32 MB.  Pure host logic: no process group needed."""
    from tinyedm_amd.ddp import GradReducer
    from tinyedm_amd.ema import FlatArena
    torch.manual_seed(0)
    params, late = [], set()
    for blk in range(40):                           # 40 "blocks": a 2.25 MB conv weight, a late embed weight, a late gain
        params.append(torch.nn.Parameter(torch.empty(256, 256, 3, 3)))
        e = torch.nn.Parameter(torch.empty(256, 64))
        e._edm_late = True
        g = torch.nn.Parameter(torch.zeros(()))
        params += [e, g]
        late |= {len(params) - 2, len(params) - 1}
    arena = FlatArena(params)
    red = GradReducer(arena)                        # world 1, inactive: only the plan is built
    assert not red.active
    lo = arena.scalar_lo
    assert all((arena.offsets[i] >= lo) == (i in late) for i in range(len(params)))
    tail = red.buckets[-1]
    assert tail["lo"] == lo and tail["hi"] == arena.numel and set(tail["params"]) == late
    body = red.buckets[:-1]
    assert all(set(b["params"]).isdisjoint(late) for b in body)
    assert sum(b["hi"] - b["lo"] for b in red.buckets) == arena.numel
    region = GradReducer.TAIL_REGION_BYTES // 4
    sizes_tail = [(b["hi"] - b["lo"]) * 4 for b in body if b["hi"] <= region]
    sizes_rest = [(b["hi"] - b["lo"]) * 4 for b in body if b["hi"] > region]
    assert sizes_tail and max(sizes_tail) <= GradReducer.TAIL_BUCKET_BYTES + 4 * 256 * 256 * 9     # (one tensor of slack)
    assert max(sizes_rest) > GradReducer.TAIL_BUCKET_BYTES and max(sizes_rest) <= (32 << 20) + 4 * 256 * 256 * 9
    # buckets are in backward order: the first one holds the LAST body parameters, the last body bucket starts at offset 0
    assert body[0]["hi"] == lo and body[-1]["lo"] == 0

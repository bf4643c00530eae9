"""Probe: the data-parallel training step captured into ONE hipGraph with the RCCL all-reduces inside (one forced rank).
Compares the replayed step's gradients-through-weights with the eager reducer step and times host cost.
    EDM_FORCE_REDUCE=1 python tools/ddp_graph_probe.py"""
import faulthandler
import os
import sys
import time

faulthandler.enable()

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("EDM_FORCE_REDUCE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import tinyedm_amd  # noqa: E402,F401
import tinyedm  # noqa: E402
from bench import build_model  # noqa: E402
from tinyedm_amd.ddp import GradReducer  # noqa: E402
from tinyedm_amd.ema import EMAOptimizer  # noqa: E402
from tinyedm_amd.graph import CapturedTrainStep  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


def make():
    model, cfg = build_model(dev)
    model.train()
    base = model.configure_optimizers()["optimizer"]
    base.fuse_zero_grad = True
    opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps)
    red = GradReducer(base.arena)
    red.broadcast_parameters()
    return model, base, opt, red


g = torch.Generator().manual_seed(42)
B = int(os.environ.get("PROBE_BATCH", "128"))
batch = ((0.5 * torch.randn(B, 3, 32, 32, generator=g)).to(dev), None)
N = 8
# ---- eager reducer step
model, base, opt, red = make()
assert red.active
opt.zero_grad()
for i in range(N):
    loss = model.training_step(batch, i)
    loss.backward()
    base.grad_scale = red.finish()
    opt.step()
    opt.zero_grad()
torch.cuda.synchronize()
theta_e, loss_e = base.arena.theta.clone(), float(loss)
# ---- captured step with the collectives inside
model, base, opt, red = make()
opt.zero_grad()
cap = CapturedTrainStep(model, opt, reducer=red)
for i in range(N):
    print("captured-path call", i, flush=True)
    loss = cap(batch)
    torch.cuda.synchronize()
theta_g, loss_g = base.arena.theta.clone(), float(loss)
rel = ((theta_g - theta_e).norm() / theta_e.norm()).item()
print(f"eager loss {loss_e:.5f} graph loss {loss_g:.5f} theta rel diff {rel:.3e}", flush=True)
t0 = time.perf_counter()
for i in range(20):
    cap(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"captured DDP step: host {1e3 * (t1 - t0) / 20:.3f} ms/step, wall {1e3 * (t2 - t0) / 20:.3f} ms/step", flush=True)
hs = []
for i in range(10):             # host cost of ONE enqueue with an empty queue (no back-pressure from earlier launches)
    torch.cuda.synchronize()
    a = time.perf_counter()
    cap(batch)
    hs.append(1e3 * (time.perf_counter() - a))
torch.cuda.synchronize()
print("unblocked host ms per step:", " ".join(f"{h:.2f}" for h in hs), flush=True)
from tinyedm_amd import ops  # noqa: E402
ops.check_health(dev, "ddp_graph_probe")
dist.barrier()
dist.destroy_process_group()
print("OK", flush=True)

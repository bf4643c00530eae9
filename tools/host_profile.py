"""Host-side view of one training step: which CPU ops launch fill/elementwise kernels, and how long the host
spends enqueueing a step (python tools/host_profile.py).  Diagnostic only."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import tinyedm
    from tinyedm_amd.ddp import GradReducer
    from tinyedm_amd.ema import EMAOptimizer
    dev = torch.device("cuda:0")
    model, cfg = bench.build_model(dev)
    model.train()
    base = model.configure_optimizers()["optimizer"]
    opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps)
    red = GradReducer(base.arena)
    x = 0.5 * torch.randn(128, 3, 32, 32, device=dev)
    y = torch.randint(0, 10, (128,), device=dev)

    def step(i):
        loss = model.training_step((x, y), i)
        loss.backward()
        base.grad_scale = red.finish()
        opt.step()
        opt.zero_grad()
        return loss

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    # host enqueue time vs wall time
    for tag in ("a", "b"):
        t0 = time.perf_counter()
        for i in range(5):
            step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"host enqueue {1e3 * (t1 - t0) / 5:.2f} ms/step, wall {1e3 * (t2 - t0) / 5:.2f} ms/step", flush=True)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(10)
        torch.cuda.synchronize()
    rows = prof.key_averages(group_by_stack_n=6)
    sel = [r for r in rows if r.key in ("aten::zeros", "aten::zero_", "aten::fill_", "aten::zeros_like", "aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::sum", "aten::to", "aten::clone", "aten::empty")]
    sel.sort(key=lambda r: -r.count)
    for r in sel[:45]:
        st = [s for s in r.stack if "tinyedm" in s or "bench" in s or "autograd" in s][:3]
        print(f"{r.key:18s} n={r.count:4d} cpu_us={r.cpu_time_total:8.0f}  {' <- '.join(s.split('/')[-1] for s in st)}", flush=True)
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25), flush=True)


if __name__ == "__main__":
    main()

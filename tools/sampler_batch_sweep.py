"""Captured 32-step Heun solve at several batch sizes: img/s against the batch.
    python tools/sampler_batch_sweep.py [bf16|f32x3|f32] [256 512 1024 2048]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import tinyedm  # noqa: E402

dev = torch.device("cuda:0")
model, cfg = bench.build_model(dev)
model.eval()
args = sys.argv[1:]
dtype = args.pop(0) if args and not args[0].isdigit() else "bf16"
model.denoiser.set_eval_dtype(dtype)
solver = tinyedm.DeterministicSolver(num_steps=32)
print(f"network dtype {dtype}", flush=True)
for B in [int(v) for v in args] or [256, 512, 1024, 2048]:
    x0 = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(7)).to(dev)
    solver.solve(model, x0, None, graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        out = solver.solve(model, x0, None, graph=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"B={B}: {dt * 1e3:.1f} ms per solve, {B / dt:.1f} img/s, finite {bool(torch.isfinite(out).all())}", flush=True)

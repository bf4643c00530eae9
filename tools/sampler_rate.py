"""hipGraph-replayed 32-step Heun solve rate of the CIFAR-10 net:  python tools/sampler_rate.py [B] [bf16|f32|f32x3] [reps]"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt_name = sys.argv[2] if len(sys.argv) > 2 else "f32x3"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
model.denoiser.set_eval_dtype(dt_name)
solver = tinyedm.DeterministicSolver(num_steps=32)
x0 = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(7)).to(dev)
out = solver.solve(model, x0, None, graph=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = solver.solve(model, x0, None, graph=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{dt_name} B={B}: {dt * 1e3:.1f} ms per solve, {B / dt:.1f} img/s, |x|={float(out.norm()):.4f}", flush=True)

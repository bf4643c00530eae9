"""Eager (non-graph) 32-step Heun solve of the CIFAR-10 net at the bench's sampler batch, for rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sampler -- python3 tools/sampler_profile.py [B] [bf16|f32]"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model.denoiser.set_eval_dtype(sys.argv[2] if len(sys.argv) > 2 else "bf16")
solver = tinyedm.DeterministicSolver(num_steps=32)
x0 = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(7)).to(dev)
solver.solve(model, x0, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
solver.solve(model, x0, None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"eager solve: {dt * 1e3:.1f} ms, {B / dt:.1f} img/s", flush=True)

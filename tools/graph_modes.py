"""Diagnostic: is the hipGraph replay time of the training step bimodal per capture or per process?
Captures the step several times in ONE process and times blocks of replays; prints the shader clock / power that sysfs
reports while the block runs."""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tinyedm_amd.ema import EMAOptimizer  # noqa: E402
from tinyedm_amd.graph import CapturedTrainStep  # noqa: E402
import tinyedm  # noqa: E402


def sysfs():
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            cur = [l for l in open(f).read().splitlines() if l.endswith("*")]
            out.append(cur[0].split()[1] if cur else "?")
        except OSError:
            pass
    pw = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) + \
            sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")):
        try:
            pw.append(str(int(open(f).read()) // 1000000))
        except (OSError, ValueError):
            pass
    return "sclk " + ",".join(out) + " W " + ",".join(pw)


dev = torch.device("cuda:0")
model, cfg = bench.build_model(dev)
model.train()
base = model.configure_optimizers()["optimizer"]
opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps)
g = torch.Generator().manual_seed(42)
batch = ((0.5 * torch.randn(128, 3, 32, 32, generator=g)).to(dev), torch.randint(0, 10, (128,), generator=g).to(dev))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cap = CapturedTrainStep(model, opt)
for i in range(CapturedTrainStep.WARMUP + 4):
    cap(batch)
E = os.environ.get
for rep in range(reps):
    if E("GM_SYNC", "1") == "1":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = ""
    for i in range(nrep):
        loss = cap(batch)
        if i == nrep // 2 and E("GM_SYSFS", "1") == "1":
            info = sysfs()
    torch.cuda.synchronize()
    print(f"block {rep}: {(time.perf_counter() - t0) / nrep * 1e3:.3f} ms/step  loss {float(loss):.5f} |w| {float(base.arena.theta.norm()):.4f} |g| {float(base.arena.grad.norm()):.4e} {info}", flush=True)

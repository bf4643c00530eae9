"""Diagnostic: replay the captured training step back to back (no host sync, nothing between replays) with finite-ness
probes recorded INSIDE the graph, and report the first step at which loss / gradients / weights go non-finite together
with the step-parameter record the graph saw."""
import os
import sys

import torch

if os.environ.get("HUNT_SETENV") == "after_import":
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
if os.environ.get("HUNT_SETENV") == "after_init":
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tinyedm_amd import ops  # noqa: E402
from tinyedm_amd.ema import EMAOptimizer  # noqa: E402
from tinyedm_amd.graph import CapturedTrainStep  # noqa: E402
import tinyedm  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 46


class Probed(CapturedTrainStep):
    def _capture(self, batch):
        x, y = batch
        sx = x.clone()
        sy = None if y is None else y.clone()
        snap = self._snapshot()
        self.hist = torch.zeros(N + 8, 15, device=x.device)
        self.ctr = torch.zeros(1, dtype=torch.int64, device=x.device)
        torch.cuda.synchronize()
        ops.capture_begin()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, stream=self.stream):
                dyn_in = self.params.dev.clone()
                loss = self.model.training_step((sx, sy), 0)
                loss.backward()
                gb = (~torch.isfinite(self.base.arena.grad)).sum().float()
                gn = self.base.arena.grad.float().norm()
                self.base.step_dyn(self.params.dev, ema=self.ema.ema_arena if self.ema is not None else None)
                loss = loss.detach()
                wb = (~torch.isfinite(self.base.arena.theta)).sum().float()
                mb = (~torch.isfinite(self.base.m)).sum().float() + (~torch.isfinite(self.base.v)).sum().float()
                eb = (~torch.isfinite(self.ema.ema_arena)).sum().float()
                xin = (~torch.isfinite(sx)).sum().float()
                dyn_out = self.params.dev
                row = torch.cat([torch.stack([loss.float().reshape(()), gb, gn, wb, mb, eb, xin]),
                                 dyn_in[:1].float(), dyn_in.view(torch.float32)[4:9],
                                 dyn_out[:1].float(), (dyn_out != dyn_in).sum().float().reshape(1)])[None]
                self.hist.index_copy_(0, self.ctr, row)
                self.ctr += 1
        finally:
            ops.capture_end()
            self._restore(snap)
        return graph, sx, sy, loss


dev = torch.device("cuda:0")
model, cfg = bench.build_model(dev)
model.train()
base = model.configure_optimizers()["optimizer"]
opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps)
g = torch.Generator().manual_seed(42)
batch = ((0.5 * torch.randn(128, 3, 32, 32, generator=g)).to(dev), torch.randint(0, 10, (128,), generator=g).to(dev))
cap = Probed(model, opt)
import time
for i in range(6):
    cap(batch)
how = os.environ.get("HUNT_SYNC", "device")
if how == "device":
    torch.cuda.synchronize()
elif how == "stream":
    torch.cuda.current_stream().synchronize()
elif how == "sleep":
    time.sleep(1.0)
elif how == "event":
    ev = torch.cuda.Event()
    ev.record()
    ev.synchronize()
for i in range(N - 6):
    loss = cap(batch)
torch.cuda.synchronize()
h = cap.hist.cpu()[:N - 2]
cols = "loss g_bad g_norm w_bad mv_bad ema_bad x_bad step_in lr beta gscale bc1 bc2s step_out dyn_changed".split()
bad = [i for i in range(len(h)) if not torch.isfinite(h[i, :7]).all() or h[i, 1] > 0 or h[i, 3:7].sum() > 0]
if not bad:
    print(f"clean after {N} steps, loss {float(h[-1, 0]):.5f}  dyn_changed rows {int((h[:, 14] > 0).sum())}")
else:
    i = bad[0]
    print(f"FIRST BAD replay {i}: " + " ".join(f"{c}={float(v):.4g}" for c, v in zip(cols[:7], h[i])))

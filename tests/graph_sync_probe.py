"""Helper process of tests/test_graph_gpu.py::test_replay_after_stream_sync_is_not_corrupted (not collected).

Replays the captured CIFAR-10 training step with finite-ness probes recorded INSIDE the graph, with a device
synchronisation between replays.  On ROCm 7.2 with the runtime's default AQL-packet-capture path the first replay
after the synchronisation ran nodes with clobbered kernel arguments;
tinyedm_amd/_runtime_env.py switches that path off at import.  Prints `CLEAN <n>` or `CORRUPT <row>`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyedm_amd  # noqa: E402,F401  (first: HIP runtime flags)
import torch  # noqa: E402

import tinyedm  # noqa: E402
from tinyedm.config import compose, instantiate  # noqa: E402
from tinyedm_amd import ops  # noqa: E402
from tinyedm_amd.ema import EMAOptimizer  # noqa: E402
from tinyedm_amd.graph import CapturedTrainStep  # noqa: E402

N = 14


class Probed(CapturedTrainStep):
    def _capture(self, batch):
        x, y = batch
        sx, sy = x.clone(), None if y is None else y.clone()
        snap = self._snapshot()
        self.hist = torch.zeros(N + 2, 6, device=x.device)
        self.ctr = torch.zeros(1, dtype=torch.int64, device=x.device)
        torch.cuda.synchronize()
        ops.capture_begin()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, stream=self.stream):
                loss = self.model.training_step((sx, sy), 0)
                loss.backward()
                gb = (~torch.isfinite(self.base.arena.grad)).sum().float()
                gn = self.base.arena.grad.float().norm()
                self.base.step_dyn(self.params.dev, ema=self.ema.ema_arena)
                loss = loss.detach()
                wb = (~torch.isfinite(self.base.arena.theta)).sum().float()
                mb = (~torch.isfinite(self.base.m)).sum().float() + (~torch.isfinite(self.base.v)).sum().float()
                eb = (~torch.isfinite(self.ema.ema_arena)).sum().float()
                self.hist.index_copy_(0, self.ctr, torch.stack([loss.float().reshape(()), gb, gn, wb, mb, eb])[None])
                self.ctr += 1
        finally:
            token = ops.capture_end()
            self._restore(snap)
        return graph, sx, sy, loss, token


dev = torch.device("cuda:0")
cfg = compose("cifar10", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments", "conf"))
tinyedm.manual_seed(cfg.seed)
torch.manual_seed(cfg.seed)
model = instantiate(cfg.model).to(dev).train()
base = model.configure_optimizers()["optimizer"]
opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps)
g = torch.Generator().manual_seed(42)
batch = ((0.5 * torch.randn(64, 3, 32, 32, generator=g)).to(dev), None)
cap = Probed(model, opt)
for i in range(N):
    cap(batch)
    if i in (5, 9):
        torch.cuda.synchronize() if i == 5 else torch.cuda.current_stream().synchronize()
torch.cuda.synchronize()
h = cap.hist.cpu()[:N - CapturedTrainStep.WARMUP]
bad = [i for i in range(len(h)) if not torch.isfinite(h[i]).all() or h[i, 1] > 0 or h[i, 3:].sum() > 0 or h[i, 2] > 10]
print(f"CORRUPT {bad[0]} {h[bad[0]].tolist()}" if bad else f"CLEAN {len(h)}")

"""Training-step throughput of any conf/*.yaml model on synthetic data:
    python tools/bench_config.py imagenet 8 [steps] [--graph]      (--graph: replay the hipGraph-captured step)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    use_graph = "--graph" in sys.argv
    argv = [a for a in sys.argv if a != "--graph"]
    name, B = argv[1], int(argv[2])
    steps = int(argv[3]) if len(argv) > 3 else 5
    import tinyedm
    from tinyedm.config import compose, instantiate
    from tinyedm_amd.ddp import GradReducer
    from tinyedm_amd.ema import EMAOptimizer
    cfg = compose(name, os.path.join(ROOT, "experiments", "conf"))
    tinyedm.manual_seed(cfg.seed)
    torch.manual_seed(cfg.seed)
    dev = torch.device("cuda:0")
    model = instantiate(cfg.model).to(dev).train()
    base = model.configure_optimizers()["optimizer"]
    opt = base
    if model.use_ema:
        opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps)
    red = GradReducer(base.arena)
    shape = tuple(cfg.datamodule.image_shape)
    x = 0.5 * torch.randn(B, *shape, device=dev)
    y = torch.randint(0, cfg.datamodule.num_classes, (B,), device=dev)

    def step(i):
        loss = model.training_step((x, y), i)
        loss.backward()
        base.grad_scale = red.finish()
        opt.step()
        opt.zero_grad()
        return loss

    opt.zero_grad()
    if use_graph:
        from tinyedm_amd.graph import CapturedTrainStep
        cap = CapturedTrainStep(model, opt)
        for i in range(CapturedTrainStep.WARMUP + 2):
            cap((x, y))

        def step(i):                        # noqa: F811
            return cap((x, y))
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(2 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name}: batch {B} {shape}: {dt * 1e3:.1f} ms/step, {B / dt:.1f} img/s, loss {float(loss.detach()):.4f}"
          f"{' (hipGraph replay)' if use_graph else ''}", flush=True)


if __name__ == "__main__":
    main()

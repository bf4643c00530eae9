"""cProfile of the host side of training steps (which Python functions the 13-14 ms of enqueue time go to)."""
import cProfile
import os
import pstats
import sys

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

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(5):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()

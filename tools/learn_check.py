"""Does the HIP training path actually learn?  A tiny EDM on a trivially structured dataset (every image is one of ten
smooth class patterns): the sigma-weighted loss starts at ~1 (gain_out = 0 -> D = c_skip*x) and must fall."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(steps=200, B=64, lr=5e-3, seed=0):
    import tinyedm_amd as T
    from tinyedm_amd.ema import EMAOptimizer
    T.manual_seed(seed)
    torch.manual_seed(seed)
    dev = torch.device("cuda:0")
    emb = T.Embedding(32, 64, 10)
    den = T.Denoiser(3, 3, ("Enc", "EncD", "EncA"), ("DecA", "DecA", "DecU", "Dec", "Dec"), (64, 128, 128),
                     (128, 128, 128, 64, 64), (True, True, False, True, True), 0.0, 0.5, 0.3, 0.3, 64, 2)
    model = T.EDM(diffuser=T.Diffuser(-1.2, 1.2), embedding=emb, denoiser=den, use_ema=True, use_uncertainty=False,
                  steady_steps=10 ** 6, rampup_steps=20, scheduler_interval="step", lr=lr, ema_length=0.13).to(dev).train()
    cfg = model.configure_optimizers()
    base, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    opt = EMAOptimizer(base, device=dev, gamma=T.sigma_rel_to_gamma(0.13))
    g = torch.Generator().manual_seed(1)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 16), torch.linspace(-1, 1, 16), indexing="ij")
    protos = torch.stack([torch.stack([torch.sin((k + 1) * xx + c) * torch.cos((k % 3 + 1) * yy) for c in range(3)])
                          for k in range(10)]) * 0.7                      # (10, 3, 16, 16), std ~0.5
    losses = []
    opt.zero_grad()
    for it in range(steps):
        y = torch.randint(0, 10, (B,), generator=g)
        x = protos[y] + 0.02 * torch.randn(B, 3, 16, 16, generator=g)
        loss = model.training_step((x.to(dev), y.to(dev)), it)
        loss.backward()
        opt.step()
        opt.zero_grad()
        sched.step()
        losses.append(loss.item())
    return losses


if __name__ == "__main__":
    ls = run()
    k = 20
    print("first", sum(ls[:k]) / k, "last", sum(ls[-k:]) / k, flush=True)

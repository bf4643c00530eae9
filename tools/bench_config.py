"""Training-step and sampler throughput of any conf/*.yaml model on synthetic data, with whole-step MFMA / HBM fractions:
    python tools/bench_config.py imagenet 176 [steps] [--graph] [--shape 3,64,64] [--sampler bf16|f32x3 B] [--fwd-gflop G]
  --graph             replay the hipGraph-captured step (default: the eager loop, weight gradients on the side stream)
  --shape C,H,W       input shape (sets the denoiser's in / out channels too): 3,64,64 = ImageNet-64 pixel space,
                      4,64,64 = the YAML's latents, 4,32,32 = BASELINE.json configs[4] (ImageNet-256 through the SD-VAE)
  --sampler DT B      also time a captured 32-step Heun solve (63 network evaluations) of the same net at batch B
  --fwd-gflop G       GFLOP of one network evaluation per image (192.9 at 64x64, 48.0 at 32x32 for the default
                      Denoiser; 27.0 CIFAR-10; 20.1 MNIST): prints achieved fractions of the 2.5 PFLOP/s bf16 MFMA peak
                      (training = 3 x forward) -- and of 8 TB/s with --fwd-mb M (conv activation MB per evaluation)
Under rocprofv3 --kernel-trace the kernel breakdown of the same steps: tools/step_breakdown.py / tools/kstats.py."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _opt(argv, name, n=1, default=None):
    if name not in argv:
        return default
    i = argv.index(name)
    vals = argv[i + 1:i + 1 + n]
    del argv[i:i + 1 + n]
    return vals[0] if n == 1 else vals


def main():
    argv = list(sys.argv)
    use_graph = "--graph" in argv
    argv = [a for a in argv if a != "--graph"]
    shape_o = _opt(argv, "--shape")
    sampler = _opt(argv, "--sampler", 2)
    gflop = float(_opt(argv, "--fwd-gflop", default=0) or 0)
    fwd_mb = float(_opt(argv, "--fwd-mb", default=0) or 0)
    no_train = "--no-train" in argv
    argv = [a for a in argv if a != "--no-train"]
    name, B = argv[1], int(argv[2])
    steps = int(argv[3]) if len(argv) > 3 else 5
    import tinyedm
    from tinyedm.config import compose, instantiate
    from tinyedm_amd.ddp import GradReducer
    from tinyedm_amd.ema import EMAOptimizer
    cfg = compose(name, os.path.join(ROOT, "experiments", "conf"))
    if shape_o:
        shp = [int(v) for v in shape_o.split(",")]
        cfg.datamodule.image_shape = shp
        cfg.model.denoiser.in_channels = shp[0]
        cfg.model.denoiser.out_channels = shp[0]
    tinyedm.manual_seed(cfg.seed)
    torch.manual_seed(cfg.seed)
    dev = torch.device("cuda:0")
    model = instantiate(cfg.model).to(dev).train()
    nparam = sum(p.numel() for p in model.parameters())
    shape = tuple(cfg.datamodule.image_shape)
    ncls = getattr(cfg.datamodule, "num_classes", None)
    cond = model.conditional
    tag = f"{name} {shape} ({nparam / 1e6:.1f} M parameters)"
    if not no_train:
        base = model.configure_optimizers()["optimizer"]
        opt = base
        if model.use_ema:
            opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps)
        red = GradReducer(base.arena)
        x = 0.5 * torch.randn(B, *shape, device=dev)
        y = torch.randint(0, ncls, (B,), device=dev) if cond else None

        def step(i):
            loss = model.training_step((x, y), i)
            model.backward(loss)
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
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = step(3 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        line = (f"train {tag}: batch {B}: {dt * 1e3:.1f} ms/step, {B / dt:.1f} img/s, loss {float(loss.detach()):.4f} "
                f"({'hipGraph replay' if use_graph else 'eager loop'})")
        if gflop:
            line += f", whole-step MFMA {B / dt * 3 * gflop / 2.5e6:.3f} of 2.5 PF"
        if fwd_mb:
            # algorithmic bytes of a step: 3 x the conv activation traffic of an evaluation + 12 passes over the fp32 parameters
            gb = (3 * fwd_mb * B + 12 * 4 * nparam / 1e6) / 1e3
            line += f", whole-step HBM {gb / dt / 8e3:.3f} of 8 TB/s ({gb:.2f} GB algorithmic)"
        print(line, flush=True)
        if use_graph:
            cap.release()
        del opt, base, red
    if sampler:
        dtype, Bs = sampler[0], int(sampler[1])
        model.eval()
        model.denoiser.set_eval_dtype(dtype)
        solver = tinyedm.DeterministicSolver(num_steps=32)
        x0 = torch.randn(Bs, *shape, generator=torch.Generator().manual_seed(7)).to(dev)
        lab = torch.randint(0, ncls, (Bs,), generator=torch.Generator().manual_seed(8)).to(dev) if cond else None
        solver.solve(model, x0, lab, graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = solver.solve(model, x0, lab, graph=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        line = (f"sampler {tag}: {dtype} network, batch {Bs}, 32 Heun steps (63 evaluations, hipGraph): {dt * 1e3:.0f} ms per "
                f"solve, {Bs / dt:.1f} img/s, finite {bool(torch.isfinite(out).all())}")
        if gflop:
            passes = 3 if dtype == "f32x3" else 1
            line += f", MFMA {Bs / dt * 63 * gflop * passes / 2.5e6:.3f} of 2.5 PF"
        print(line, flush=True)


if __name__ == "__main__":
    main()

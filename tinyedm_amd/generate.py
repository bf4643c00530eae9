"""Sampling entry point with the reference's surface (src/tinyedm/generate.py:8-47, 50-96): the same `generate(...)`
signature and the same command-line flags (`--ckpt_path --load_ema --output_dir --num_samples --image_size
--num_classes --batch_size --num_workers --num_steps`), running the hipGraph-captured Heun sampler on the HIP path.

    python -m tinyedm.generate --ckpt_path last.ckpt --load_ema --output_dir samples --num_samples 50000 \\
        --image_size 32 --num_classes 10 --batch_size 512

Precision: the denoiser is evaluated at fp32 accuracy BY DEFAULT, as the reference does (generate.py:39-44:
`L.Trainer(accelerator="gpu")`, i.e. 32-bit precision).  `--network_dtype f32x3` (default): fp32 activations, every conv product
as three bf16 MFMA passes over (hi, lo) operand pairs accumulated in fp32 -- 32-step trajectories 2e-6 from the exact-fp32 path
and <= 1e-4 from the fp32 oracle (tests/test_evalf32_gpu.py), 158 img/s on the CIFAR-10 net; `f32`: exact fp32 products
(the f32-input matrix instruction; the checker: 3e-7 from the oracle, 58 img/s); `bf16`: the opt-in fast mode, the training
path's kernels, 540 img/s, 1.3e-3 from the fp32 trajectory.
Extensions (all optional): `--network_dtype`,
`--in_channels` (the reference's noise dataset hard-codes 3; default = the checkpoint's
denoiser.in_channels), `--mean/--std` (default: the reference's CIFAR-10 constants), `--seed`, `--no_graph`, and
`--config_name` to sample from random-init weights of a config instead of a checkpoint (plumbing runs).
Multi-GPU = replicas only: under `python -m torch.distributed.run --nproc-per-node N` every rank samples
its own contiguous index range with its own noise seed and writes `<global index>.png`; there is no collective.
"""
from __future__ import annotations

import argparse
import os

import torch

CIFAR_MEAN = (0.49139968, 0.48215841, 0.44653091)      # generate.py:31-34 ("need to do better" in the reference)
CIFAR_STD = (0.24703223, 0.24348513, 0.26158784)


def generate(ckpt_path, load_ema, output_dir, num_samples, image_size, num_classes, batch_size, num_workers=16,
             num_steps=32, *, in_channels=None, mean=None, std=None, seed=0, graph=True, model=None,
             network_dtype="f32x3") -> None:
    from .callbacks import PreditionWriter
    from .datamodules import RandomNoiseDataModule
    from .edm import EDM
    from .solvers import DeterministicSolver
    from .trainer import Trainer

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dev = torch.device("cuda", torch.cuda.current_device())
    if model is None:
        model = EDM.load_from_checkpoint(ckpt_path, load_ema=load_ema)
    model = model.to(dev)
    model.denoiser.set_eval_dtype(network_dtype)
    model.solver = DeterministicSolver(num_steps=num_steps)
    from . import _runtime_env
    if graph and _runtime_env.GRAPH_REPLAY_SAFE:      # otherwise the eager Heun loop: same values
        solve = model.solver.solve
        model.solver.solve = lambda m, x0, labels=None: solve(m, x0, labels, graph=True)
    C = int(in_channels) if in_channels is not None else int(model.denoiser.in_channels)
    per_rank = (num_samples + world - 1) // world
    first = rank * per_rank
    n_local = max(0, min(per_rank, num_samples - first))
    datamodule = RandomNoiseDataModule(batch_size, num_workers, image_size, n_local, num_classes, in_channels=C,
                                       seed=seed + 1000003 * rank)
    if mean is None or std is None:
        mean, std = (CIFAR_MEAN, CIFAR_STD) if C == 3 else ((0.5,) * C, (0.25,) * C)
    writer = PreditionWriter(output_dir=output_dir, write_interval="batch", mean=mean, std=std, first_index=first)
    trainer = Trainer(accelerator="gpu", strategy="auto", callbacks=[writer])
    if n_local > 0:
        trainer.predict(model, datamodule=datamodule, return_predictions=False, ckpt_path=None, distributed=False)   # generate.py:45-47
    print(f"[rank {rank}] wrote images {first}..{first + n_local - 1} to {output_dir}", flush=True)


def main(argv=None):
    parser = argparse.ArgumentParser(description="Run the model generation")
    parser.add_argument("--ckpt_path", type=str, default=None, help="Path to the checkpoint file")
    parser.add_argument("--load_ema", action="store_true", help="Load the exponential moving average of the weights")
    parser.add_argument("--output_dir", type=str, required=True, help="Directory for output")
    parser.add_argument("--num_samples", type=int, required=True, help="Number of samples to generate")
    parser.add_argument("--image_size", type=int, required=True, help="Image size")
    parser.add_argument("--num_classes", type=int, required=True, help="Number of classes")
    parser.add_argument("--batch_size", type=int, required=True, help="Batch size")
    parser.add_argument("--num_workers", type=int, default=16, help="Number of workers (default: 16)")
    parser.add_argument("--num_steps", type=int, default=32, help="Number of steps (default: 32)")
    # extensions
    parser.add_argument("--in_channels", type=int, default=None)
    parser.add_argument("--mean", type=float, nargs="+", default=None)
    parser.add_argument("--std", type=float, nargs="+", default=None)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--no_graph", action="store_true", help="eager Heun loop instead of the captured hipGraph")
    parser.add_argument("--network_dtype", choices=["bf16", "f32", "f32x3"], default="f32x3",
                        help="denoiser evaluation precision: f32x3 (default) = fp32-accurate (split-bf16, 3 MFMA passes), "
                             "f32 = exact fp32 products (checker, slower), bf16 = fast mode")
    parser.add_argument("--config_name", type=str, default=None,
                        help="sample from random-init weights of experiments/conf/<name>.yaml (no checkpoint)")
    parser.add_argument("--config_path", type=str, default=None)
    args = parser.parse_args(argv)
    model = None
    if args.ckpt_path is None:
        if args.config_name is None:
            parser.error("--ckpt_path is required (or --config_name for a random-init plumbing run)")
        from . import networks
        from .config import compose, instantiate
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cfg = compose(args.config_name, args.config_path or os.path.join(root, "experiments", "conf"))
        networks.manual_seed(cfg.seed)
        torch.manual_seed(cfg.seed)
        model = instantiate(cfg.model)
    generate(args.ckpt_path, args.load_ema, args.output_dir, args.num_samples, args.image_size, args.num_classes,
             args.batch_size, args.num_workers, args.num_steps, in_channels=args.in_channels, mean=args.mean,
             std=args.std, seed=args.seed, graph=not args.no_graph, model=model, network_dtype=args.network_dtype)


if __name__ == "__main__":
    main()

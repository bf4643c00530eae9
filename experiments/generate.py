"""Sampling entry point: load a checkpoint (optionally its EMA weights), draw N(0,1) noise +
random labels like the reference's RandomNoiseDataset, run the hipGraph-captured Heun sampler and write PNGs:

    python experiments/generate.py --ckpt last.ckpt --num-samples 50000 --batch-size 512 --out samples [--load-ema]
    python experiments/generate.py --config-name cifar10 --num-samples 64      # random-init weights (plumbing)

Multi-GPU = replicas only: under `python -m torch.distributed.run --nproc-per-node N` every rank samples
its own contiguous index range with its own noise seed and writes `<global index>.png`; there is no collective.
"""
import argparse
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for multi-process GPU runs (RCCL)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import tinyedm  # noqa: E402
from tinyedm.callbacks import PreditionWriter  # noqa: E402
from tinyedm.config import compose, instantiate  # noqa: E402
from tinyedm.datamodules import RandomNoiseDataModule  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--load-ema", action="store_true")
    ap.add_argument("--config-name", default="cifar10")
    ap.add_argument("--config-path", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf"))
    ap.add_argument("--num-samples", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--num-steps", type=int, default=32)
    ap.add_argument("--out", default="samples")
    ap.add_argument("--mean", type=float, nargs="+", default=None)
    ap.add_argument("--std", type=float, nargs="+", default=None)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    per_rank = (a.num_samples + world - 1) // world
    first = rank * per_rank
    n_local = max(0, min(per_rank, a.num_samples - first))
    if a.ckpt:
        model = tinyedm.EDM.load_from_checkpoint(a.ckpt, load_ema=a.load_ema)
    else:
        cfg = compose(a.config_name, a.config_path)
        tinyedm.manual_seed(cfg.seed)
        torch.manual_seed(cfg.seed)
        model = instantiate(cfg.model)
    model = model.to(torch.device("cuda", torch.cuda.current_device()))
    model.solver = tinyedm.DeterministicSolver(num_steps=a.num_steps)
    C = model.denoiser.in_channels
    H = 28 if C == 1 else 32
    dm = RandomNoiseDataModule(a.batch_size, n_local, image_shape=(C, H, H), num_classes=model.num_classes,
                               seed=a.seed + 1000003 * rank)
    mean = a.mean or [0.5] * C
    std = a.std or [0.25] * C        # pred*std*2 + mean with std 0.25: [-1,1] -> [0,1]
    writer = PreditionWriter(a.out, "batch", mean, std, first_index=first)
    trainer = tinyedm.Trainer(callbacks=[writer])
    if n_local > 0:
        trainer.predict(model, datamodule=dm, distributed=False)
    print(f"[rank {rank}] wrote images {first}..{first + n_local - 1} to {a.out}")


if __name__ == "__main__":
    main()

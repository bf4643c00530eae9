"""Counterpart of the reference's experiments/train.py (hydra.main + lightning Trainer + wandb), on the
stand-ins of tinyedm_amd: ``python experiments/train.py --config-name=cifar10 [key=value ...]``.
Multi-GPU: ``python -m torch.distributed.run --nproc-per-node N experiments/train.py --config-name=...``"""
import argparse
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for multi-process GPU runs (RCCL)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import tinyedm  # noqa: E402
from tinyedm.config import compose, instantiate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-name", default="cifar10")
    ap.add_argument("--config-path", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf"))
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args()
    cfg = compose(a.config_name, a.config_path, a.overrides)
    tinyedm.manual_seed(cfg.seed)
    import torch
    torch.manual_seed(cfg.seed)
    datamodule = instantiate(cfg.datamodule)
    datamodule.prepare_data()
    model = instantiate(cfg.model)
    callbacks = list(instantiate(cfg.get("callbacks") or {}).values())
    trainer_kwargs = {k: v for k, v in cfg.trainer.items()}
    trainer = tinyedm.Trainer(callbacks=callbacks, **trainer_kwargs)
    ckpt_path = cfg.get("ckpt_path")
    trainer.fit(model, datamodule=datamodule, ckpt_path=ckpt_path)


if __name__ == "__main__":
    main()

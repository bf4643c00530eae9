"""Minimal stand-ins for the parts of lightning the reference's hot path touches (lightning is not
installed): a LightningModule base (``log``, ``hparams``, ``lr_schedulers``, ``device``) and a Trainer
(``fit`` / ``predict``, callbacks, gradient accumulation, one process per GPU with the flat-arena
gradient reducer).  Everything that is control plane in Lightning (checkpoint managers, loggers,
progress bars) is out of scope; hooks used by the reference's EMA callback are honoured."""
from __future__ import annotations

import os
import time
from typing import Any, List, Optional

import torch
import torch.distributed as dist

from .ddp import GradReducer
from .ema import EMAOptimizer, FusedAdam


class LightningModule(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.hparams: dict = {}
        self.trainer: Optional["Trainer"] = None
        self._logged: dict = {}

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def log(self, name: str, value: Any, **kwargs) -> None:
        self._logged[name] = value

    def lr_schedulers(self):
        return self.trainer.lr_scheduler if self.trainer is not None else None

    def configure_callbacks(self):
        return []


class _ConstLR:
    def __init__(self, lr):
        self._lr = lr

    def get_last_lr(self):
        return [self._lr]

    def step(self):
        pass


class Trainer:
    """``Trainer(max_epochs=, max_steps=, accumulate_grad_batches=, check_val_every_n_epoch=, callbacks=,
    devices=, accelerator=, strategy=, precision=, logger=)`` -- the kwargs of conf/*.yaml ``trainer:``.
    ``precision`` is accepted for config compatibility: the HIP path always runs the bf16-mixed policy."""

    def __init__(self, max_epochs: int = 1, max_steps: int = -1, accumulate_grad_batches: int = 1,
                 check_val_every_n_epoch: int = 1, callbacks: Optional[List[Any]] = None, devices: Any = 1,
                 accelerator: str = "gpu", strategy: str = "auto", precision: Any = "bf16-mixed", logger: Any = None,
                 log_every_n_steps: int = 50, **_ignored):
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.accumulate_grad_batches = accumulate_grad_batches
        self.check_val_every_n_epoch = check_val_every_n_epoch
        self.callbacks = list(callbacks or [])
        self.precision = precision
        self.logger = logger
        self.log_every_n_steps = log_every_n_steps
        self.optimizers: List[Any] = []
        self.lr_scheduler = None
        self.scheduler_interval = "epoch"
        self.global_step = 0
        self.current_epoch = 0
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.global_rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.reducer: Optional[GradReducer] = None
        self.datamodule = None

    # ------------------------------------------------------------------ setup
    def _setup_distributed(self, model):
        if self.world_size > 1 and not dist.is_initialized():
            backend = "nccl" if torch.cuda.is_available() else "gloo"
            dist.init_process_group(backend)
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
            model.to(torch.device("cuda", self.local_rank))

    def _call(self, hook: str, *args):
        for cb in self.callbacks:
            fn = getattr(cb, hook, None)
            if fn is not None:
                fn(self, *args)

    def _configure(self, model):
        model.trainer = self
        self._model = model
        self.callbacks += [c for c in model.configure_callbacks()]
        cfg = model.configure_optimizers()
        if isinstance(cfg, dict):
            opt = cfg["optimizer"]
            sch = cfg.get("lr_scheduler")
            if isinstance(sch, dict):
                self.scheduler_interval = sch.get("interval", "epoch")
                sch = sch["scheduler"]
            self.lr_scheduler = sch
        else:
            opt = cfg
        self.optimizers = [opt]
        if self.lr_scheduler is None:
            self.lr_scheduler = _ConstLR(opt.param_groups[0]["lr"])
        base = opt.optimizer if isinstance(opt, EMAOptimizer) else opt
        if isinstance(base, FusedAdam):
            self.reducer = GradReducer(base.arena)
            self.reducer.broadcast_parameters()

    # ------------------------------------------------------------------ fit
    def fit(self, model, datamodule=None, train_dataloaders=None, val_dataloaders=None, ckpt_path=None):
        self._setup_distributed(model)
        self._configure(model)
        self.datamodule = datamodule
        if datamodule is not None and hasattr(datamodule, "setup"):
            datamodule.setup("fit")
        self._call("on_fit_start", model)
        self._call("on_train_start", model)
        train = train_dataloaders if train_dataloaders is not None else datamodule.train_dataloader()
        val = val_dataloaders
        if val is None and datamodule is not None and hasattr(datamodule, "val_dataloader"):
            val = datamodule.val_dataloader()
        opt = self.optimizers[0]
        base = opt.optimizer if isinstance(opt, EMAOptimizer) else opt
        model.train()
        opt.zero_grad()
        done = False
        t0, imgs = time.time(), 0
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            for bi, batch in enumerate(train):
                batch = _to_device(batch, model.device)
                last_micro = (bi + 1) % self.accumulate_grad_batches == 0
                if self.reducer is not None:
                    self.reducer.enabled = last_micro
                loss = model.training_step(batch, bi)
                (loss / self.accumulate_grad_batches).backward()
                imgs += batch[0].shape[0] * self.world_size
                if last_micro:
                    base.grad_scale = self.reducer.finish() if self.reducer is not None else 1.0
                    opt.step()
                    opt.zero_grad()
                    self.global_step += 1
                    if self.scheduler_interval == "step":
                        self.lr_scheduler.step()
                    if self.global_rank == 0 and self.global_step % self.log_every_n_steps == 0:
                        print(f"[fit] epoch {epoch} step {self.global_step} loss {float(loss.detach()):.4f} "
                              f"{imgs / (time.time() - t0):.1f} img/s", flush=True)
                    if 0 < self.max_steps <= self.global_step:
                        done = True
                        break
            if self.scheduler_interval == "epoch":
                self.lr_scheduler.step()
            self._call("on_train_epoch_end", model)
            if val is not None and (epoch + 1) % self.check_val_every_n_epoch == 0:
                self.validate(model, val)
            if done:
                break
        self._call("on_fit_end", model)

    @torch.no_grad()
    def validate(self, model, val):
        self._call("on_validation_start", model)
        model.eval()
        if hasattr(model, "val_mse"):
            model.val_mse.reset()
        for bi, batch in enumerate(val):
            model.validation_step(_to_device(batch, model.device), bi)
        model.train()
        self._call("on_validation_end", model)
        return model.val_mse.compute() if hasattr(model, "val_mse") else None

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, path, model=None):
        """Writes the Lightning checkpoint keys the reference's `EDM.load_from_checkpoint` reads (edm.py:159-203):
        `state_dict`, `hyper_parameters` (deinstantiated config, utils.py:5-27) and `optimizer_states[0]["ema"]`."""
        model = model if model is not None else self._model
        ckpt = {"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "hyper_parameters": dict(model.hparams), "epoch": self.current_epoch, "global_step": self.global_step,
                "optimizer_states": [_cpu_tree(o.state_dict()) for o in self.optimizers]}
        if self.global_rank == 0:
            torch.save(ckpt, path)
        return ckpt

    def load_checkpoint(self, path, model):
        """Resume: weights, optimizer (Adam moments, EMA) and counters."""
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["state_dict"], strict=False)
        for o, sd in zip(self.optimizers, ckpt.get("optimizer_states", [])):
            o.load_state_dict(sd)
        self.current_epoch, self.global_step = ckpt.get("epoch", 0), ckpt.get("global_step", 0)

    # ------------------------------------------------------------------ predict
    @torch.no_grad()
    def predict(self, model, datamodule=None, dataloaders=None, ckpt_path=None, distributed: bool = True):
        """``distributed=False``: replicas-only sampling (no process group; each rank runs its own loader)."""
        if distributed:
            self._setup_distributed(model)
        elif torch.cuda.is_available():
            model.to(torch.device("cuda", torch.cuda.current_device()))
        model.trainer = self
        model.eval()
        self.datamodule = datamodule
        if dataloaders is None and hasattr(datamodule, "setup"):
            datamodule.setup("predict")
        loader = dataloaders if dataloaders is not None else datamodule.predict_dataloader()
        outs = []
        for bi, batch in enumerate(loader):
            batch = _to_device(batch, model.device)
            out = model.predict_step(batch, bi)
            for cb in self.callbacks:
                fn = getattr(cb, "write_on_batch_end", None)
                if fn is not None:
                    fn(self, model, out, None, batch, bi, 0)
            outs.append(out)
        return outs


def _cpu_tree(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu()
    if isinstance(x, dict):
        return {k: _cpu_tree(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_cpu_tree(v) for v in x)
    return x


def _to_device(batch, device):
    if isinstance(batch, (list, tuple)):
        return type(batch)(_to_device(b, device) for b in batch)
    if isinstance(batch, torch.Tensor):
        return batch.to(device, non_blocking=True)
    return batch

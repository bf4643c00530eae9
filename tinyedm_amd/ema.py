"""Optimizer side of the step: fused Adam + power-function EMA over flat HBM arenas.

Replaces ``optim.Adam(fused=True)`` (reference edm.py:251-253) and the NeMo-derived
``EMAOptimizer`` / ``EMA`` callback (reference ema.py): parameters, gradients, Adam moments and the
EMA copy each live in ONE contiguous fp32 arena (35.6 M floats for the CIFAR net), parameters and
their ``.grad`` are views into them, and a step is a single HIP launch that reads theta,g,m,v,ema and
writes theta,m,v,ema once -- instead of the multi-tensor Adam launch plus two foreach passes on a
side stream.  The flat gradient arena is also what the data-parallel reducer all-reduces.
"""
from __future__ import annotations

import contextlib
from typing import Iterable, List, Optional

import numpy as np
import torch

from . import ops
from .utils import swap_tensors


def sigma_rel_to_gamma(sigma_rel):
    """reference ema.py:29-32."""
    t = sigma_rel ** -2
    return np.roots([1, 7, 16 - t, 12 - t]).real.max()


class MisconfigurationException(Exception):
    pass


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def sequential_offsets(params):
    """layout 1 (rounds 1-2): every parameter in list order, 64-element aligned"""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += _align(p.numel())
    return offs, off


def _is_late(p) -> bool:
    """the parameter's gradient becomes final only when the backward pass ENDS: 0-dim parameters (block gains: one
    modulation finish per step; gain_out rides along) and the weights flagged by the network (`_edm_late`: the blocks'
    embed Linears and the Embedding module, whose weight gradients are ONE launch behind every block's backward)"""
    return p.dim() == 0 or bool(getattr(p, "_edm_late", False))


def layout_offsets(params, layout: int):
    """-> (offsets in the caller's parameter order, first offset of the late region, total elements) of arena layout 1 / 2 / 3"""
    if layout == 1:
        offs, total = sequential_offsets(params)
        return offs, total, total
    late = (lambda p: p.dim() == 0) if layout == 2 else _is_late
    offs, off = [0] * len(params), 0
    for i, p in enumerate(params):
        if not late(p):
            offs[i] = off
            off += _align(p.numel())
    lo = off
    for tensors_first in (True, False):          # late tensors, then the 0-dim parameters (layout 2: only the latter exist)
        for i, p in enumerate(params):
            if late(p) and (p.dim() > 0) == tensors_first:
                offs[i] = off
                off += _align(p.numel())
    return offs, lo, off


class FlatArena:
    """Re-homes a parameter list into one contiguous fp32 buffer (views keep names/shapes/strides).

    Layout 3 (round 5): the parameters whose gradients are final when their own layer's backward has run, in list order;
    then a TAIL with every parameter whose gradient becomes final only at the END of the backward pass -- the blocks' embed
    Linear weights and the Embedding module (their weight gradients are one launch behind all blocks), then the 0-dim
    parameters (block gains, gain_out).  `self.params` keeps the caller's order (index i is the i-th parameter of the
    optimizer, which is what torch-Adam checkpoints are keyed on), only the offsets differ.  Why: the data-parallel
    reducer launches a bucket's all-reduce when ALL its gradients are final.  In layout 2 (round 3: only the 0-dim parameters
    in the tail) every 32 MB bucket contained some block's embed weight, so NO bucket could start before the backward pass
    ended (tests/test_rccl_gpu.py, round 5: 0.00 of the bytes issued before the last weight-gradient launch); now the
    tail is the reducer's last bucket(s) and the body's buckets follow the weight-gradient launches."""
    LAYOUT = 3

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("FlatArena expects fp32 master parameters")
        self.offsets, self.scalar_lo, off = layout_offsets(self.params, self.LAYOUT)   # [scalar_lo, numel): the late region
        self.numel = off
        self.theta = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.theta[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view_as(p)
                p._edm_direct = True          # weight-gradient kernels may accumulate into p.grad directly
                p._edm_hooks = []

    def rebind_grads(self):
        for p, o in zip(self.params, self.offsets):
            g = self.grad[o:o + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def zero_grad(self):
        self.grad.zero_()
        self.rebind_grads()


class FusedAdam(torch.optim.Optimizer):
    """Adam (no weight decay / amsgrad) with torch.optim.Adam semantics, one launch over the arena."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.arena = FlatArena(params)
        self.m = torch.zeros_like(self.arena.theta)
        self.v = torch.zeros_like(self.arena.theta)
        self.step_count = 0
        self.grad_scale = 1.0

    # fuse_zero_grad = True: `step()` clears the gradient arena in the pass that reads it (it is the last reader) and the
    # `zero_grad()` that follows is free -- instead of a separate 142 MB fill (Trainer.fit and bench.py switch it on;
    # off by default because torch optimizers leave .grad intact after step()).
    fuse_zero_grad = False
    _cleared = False

    def zero_grad(self, set_to_none: bool = False):
        # step() cleared the arena itself (fuse_zero_grad) and nothing has been written into it since: free.  "Since"
        # is tracked by the Denoiser forward (networks.note_backward_pending, via reset_backward_state), the one place
        # every gradient-producing pass goes through -- a step(); backward(); zero_grad() sequence clears for real.
        if self._cleared and not _grads_written_since_clear(self):
            self._cleared = False
            self.arena.rebind_grads()
            return
        self._cleared = False
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, closure=None, ema: Optional[torch.Tensor] = None, ema_beta: float = 0.0):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self.arena.rebind_grads()
        self.step_count += 1
        ops.adam_ema(self.arena.theta, self.arena.grad, self.m, self.v, ema, g["lr"], g["betas"][0], g["betas"][1],
                     g["eps"], self.step_count, ema_beta, self.grad_scale, zero_grad=self.fuse_zero_grad)
        self._cleared = bool(self.fuse_zero_grad)
        from . import networks
        self._fwd_epoch_at_clear = networks.FORWARD_EPOCH
        networks.bump_weight_epoch()
        return loss

    @torch.no_grad()
    def step_dyn(self, dyn: torch.Tensor, ema: Optional[torch.Tensor] = None):
        """The same update with lr / bias corrections / ema_beta / grad_scale read from the device record `dyn`
        (edm_step_params) and the gradient arena cleared in the same pass: the form a captured step replays
        (graph.CapturedTrainStep owns the host-side counters)."""
        g = self.param_groups[0]
        self.arena.rebind_grads()
        ops.adam_ema(self.arena.theta, self.arena.grad, self.m, self.v, ema, g["lr"], g["betas"][0], g["betas"][1],
                     g["eps"], max(1, self.step_count), 0.0, 1.0, dyn=dyn, zero_grad=True)

    def state_dict(self):
        # "offsets": where each parameter's slice sits in m / v.  Layout 3 derives them from a Python attribute of the
        # parameters (`_edm_late`); a checkpoint that only said "layout 3" could not tell a model whose flags differ
        # (Parameter re-created, deepcopy, user re-init) from the one that wrote it
        return {"m": self.m, "v": self.v, "step": self.step_count, "layout": FlatArena.LAYOUT,
                "offsets": list(self.arena.offsets), "numels": [p.numel() for p in self.arena.params],
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        """Accepts this class's own layout ({m, v, step}: flat arenas) or torch.optim.Adam's
        ({state: {i: {step, exp_avg, exp_avg_sq}}, param_groups}: what the reference's checkpoints hold)."""
        if "m" in sd:
            a = self.arena
            if "offsets" in sd:     # round 6 checkpoints carry the slices themselves
                old = [int(o) for o in sd["offsets"]]
                if len(old) != len(a.params) or [int(n) for n in sd["numels"]] != [p.numel() for p in a.params]:
                    raise ValueError("optimizer state: the checkpoint's parameter list (count / sizes) does not match the model")
                total = sd["m"].numel()
                if any(o < 0 or o + p.numel() > total for p, o in zip(a.params, old)):
                    raise ValueError("optimizer state: a stored offset lies outside the stored arena")
            else:                   # older checkpoints: the slices follow from the layout number and the CURRENT flags
                old, _, total = layout_offsets(a.params, int(sd.get("layout", 1)))
                if sd["m"].numel() != total:
                    raise ValueError(f"optimizer state: flat arena of {sd['m'].numel()} elements, expected {total}")
            if old == list(a.offsets) and sd["m"].numel() == self.m.numel():
                self.m.copy_(sd["m"])
                self.v.copy_(sd["v"])
            else:       # another arena layout (an earlier round's, or different late flags): every slice to its new home
                for p, o_new, o_old in zip(a.params, a.offsets, old):
                    n = p.numel()
                    self.m[o_new:o_new + n].copy_(sd["m"][o_old:o_old + n])
                    self.v[o_new:o_new + n].copy_(sd["v"][o_old:o_old + n])
            self.step_count = int(sd["step"])
        elif "state" in sd:
            a = self.arena
            if len(sd["state"]) not in (0, len(a.params)):
                raise ValueError(f"optimizer state has {len(sd['state'])} entries, the model {len(a.params)} parameters")
            steps = []
            for i, (p, o) in enumerate(zip(a.params, a.offsets)):
                st = sd["state"].get(i)
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(p.shape)}")
                self.m[o:o + p.numel()].view_as(p).copy_(st["exp_avg"])
                self.v[o:o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
                steps.append(int(st["step"]))
            self.step_count = max(steps) if steps else 0
        else:
            raise KeyError("optimizer state dict has neither this build's {m, v, step} nor torch Adam's {state, param_groups}")
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k != "params"})


def _grads_written_since_clear(opt: "FusedAdam") -> bool:
    """a training-mode network forward (the start of every gradient-producing pass) ran after the clearing step()"""
    from . import networks
    return networks.FORWARD_EPOCH != getattr(opt, "_fwd_epoch_at_clear", -1)


class EMAOptimizer:
    """Same surface as the reference's EMAOptimizer (ema.py:160-348): wraps an optimizer, keeps an EMA
    copy updated with beta_t = (1 - 1/(t+1))^(gamma+1) after every ``every_n_steps`` optimizer steps,
    in-place weight swapping, and a state dict {opt, ema, current_step, gamma, every_n_steps}.
    With ``FusedAdam`` the EMA update is fused into the optimizer kernel."""

    def __init__(self, optimizer: FusedAdam, device=None, gamma: float = 0.0, every_n_steps: int = 1,
                 current_step: int = 0):
        if not isinstance(optimizer, FusedAdam):
            raise TypeError("tinyedm_amd.EMAOptimizer wraps tinyedm_amd.FusedAdam")
        self.optimizer = optimizer
        self.gamma = gamma
        self.device = device
        self.current_step = current_step
        self.every_n_steps = every_n_steps
        self.ema_arena = optimizer.arena.theta.clone()

    @property
    def ema_params(self):
        a = self.optimizer.arena
        return tuple(self.ema_arena[o:o + p.numel()].view_as(p) for p, o in zip(a.params, a.offsets))

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def all_parameters(self):
        return (p for g in self.param_groups for p in g["params"])

    def _should_update_at_step(self) -> bool:
        return self.current_step % self.every_n_steps == 0

    def step(self, closure=None, **kwargs):
        if self._should_update_at_step():
            decay = (1 - 1 / (self.current_step + 1)) ** (self.gamma + 1)
            loss = self.optimizer.step(closure, ema=self.ema_arena, ema_beta=float(decay))
        else:
            loss = self.optimizer.step(closure)
        self.current_step += 1
        return loss

    def zero_grad(self, set_to_none: bool = False):
        self.optimizer.zero_grad()

    def switch_main_parameter_weights(self):
        swap_tensors(self.optimizer.arena.theta, self.ema_arena)
        from .networks import bump_weight_epoch
        bump_weight_epoch()

    @contextlib.contextmanager
    def swap_ema_weights(self, enabled: bool = True):
        if enabled:
            self.switch_main_parameter_weights()
        try:
            yield
        finally:
            if enabled:
                self.switch_main_parameter_weights()

    def __getattr__(self, name):
        return getattr(self.__dict__["optimizer"], name)

    def synchronize(self):
        pass

    def state_dict(self):
        return {"opt": self.optimizer.state_dict(), "ema": tuple(t.clone() for t in self.ema_params),
                "current_step": self.current_step, "gamma": self.gamma, "every_n_steps": self.every_n_steps}

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd["opt"])
        for dst, src in zip(self.ema_params, sd["ema"]):
            dst.copy_(src)
        self.current_step = sd["current_step"]
        self.gamma = sd["gamma"]
        self.every_n_steps = sd["every_n_steps"]


class EMA:
    """Callback with the reference's hooks (ema.py:35-123): wraps the trainer's optimizers at fit start,
    swaps EMA weights in for validation/test unless ``validate_original_weights``."""

    def __init__(self, ema_length: float, validate_original_weights: bool = False, every_n_steps: int = 1,
                 cpu_offload: bool = False):
        if not (0 <= ema_length <= 0.2886):
            raise MisconfigurationException("EMA length value must be between 0 and 0.2886")
        if cpu_offload:
            raise MisconfigurationException("cpu_offload is not supported: the EMA lives in HBM next to the weights")
        self.ema_length = ema_length
        self.gamma = sigma_rel_to_gamma(ema_length)
        self.validate_original_weights = validate_original_weights
        self.every_n_steps = every_n_steps
        self.cpu_offload = cpu_offload

    def on_fit_start(self, trainer, pl_module) -> None:
        trainer.optimizers = [
            o if isinstance(o, EMAOptimizer) else EMAOptimizer(o, device=pl_module.device, gamma=self.gamma,
                                                               every_n_steps=self.every_n_steps,
                                                               current_step=trainer.global_step)
            for o in trainer.optimizers]

    def _should_validate_ema_weights(self, trainer) -> bool:
        return not self.validate_original_weights and any(isinstance(o, EMAOptimizer) for o in trainer.optimizers)

    def swap_model_weights(self, trainer):
        for o in trainer.optimizers:
            assert isinstance(o, EMAOptimizer)
            o.switch_main_parameter_weights()

    def on_validation_start(self, trainer, pl_module) -> None:
        if self._should_validate_ema_weights(trainer):
            self.swap_model_weights(trainer)

    def on_validation_end(self, trainer, pl_module) -> None:
        if self._should_validate_ema_weights(trainer):
            self.swap_model_weights(trainer)

    on_test_start = on_validation_start
    on_test_end = on_validation_end

"""The whole training step as ONE hipGraph.

An eager step enqueues ~400 kernels from Python (autograd.Function bodies + ctypes): ~10-14 ms of host time per
step, i.e. the host, not the GPU, sets the step time once the kernels are fast.  Capturing
`training_step -> backward -> (gradient all-reduce) -> fused Adam+EMA` into a hipGraph and replaying it removes that
cost: a step is one input copy, one 48-byte parameter upload and one graph launch.  The step is captured as one chain (the
weight-gradient side stream of the eager step is switched off while capturing).

What changes between replays cannot be a by-value kernel argument (those are frozen at capture time), so the
per-step scalars -- Philox step/seed of dropout and the Diffuser, learning rate, EMA beta, Adam bias corrections,
1/world -- live in a device `edm_step_params` record (include/tinyedm_hip.h) that `StepParams.upload` rewrites from a
pinned host ring before every replay; the kernels read it through their `dyn` argument.

Reference call sites this replaces: the Lightning automatic-optimisation loop around `EDM.training_step`
(edm.py:205-236), `optim.Adam(fused=True).step` (edm.py:251) and `EMAOptimizer.step` (ema.py:229-291).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _runtime_env, networks, ops
from .ema import EMAOptimizer, FusedAdam


def _backward(model, loss):
    """the module's Lightning-style backward hook (trainer.LightningModule.backward: a cached, marked unit gradient instead of
    autograd's per-step ones_like) when it has one"""
    hook = getattr(model, "backward", None)
    if callable(hook):
        hook(loss)
    else:
        loss.backward()


class StepParams:
    """Device `edm_step_params` record + a ring of pinned host copies (so the host never rewrites a slot whose
    asynchronous upload has not executed yet)."""

    WORDS = 12      # 48 bytes

    def __init__(self, device, slots: int = 16):
        self.dev = torch.zeros(self.WORDS, dtype=torch.int32, device=device)
        self.host = torch.zeros(slots, self.WORDS, dtype=torch.int32).pin_memory()
        self._u = self.host.numpy().view(np.uint32)
        self._f = self.host.numpy().view(np.float32)
        self._events = [None] * slots
        self._i = 0

    def upload(self, *, step: int, seed: int, lr: float, ema_beta: float, grad_scale: float, bc1: float, bc2sqrt: float):
        k = self._i % len(self._events)
        self._i += 1
        if self._events[k] is not None:
            self._events[k].synchronize()
        u, f = self._u[k], self._f[k]
        u[0] = step & 0xFFFFFFFF
        u[1] = seed & 0xFFFFFFFF
        u[2] = (seed >> 32) & 0xFFFFFFFF
        f[4], f[5], f[6], f[7], f[8] = lr, ema_beta, grad_scale, bc1, bc2sqrt
        self.dev.copy_(self.host[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[k] = ev


class CapturedTrainStep:
    """`loss = step(batch)`: one optimisation step of `model` (an EDM) under `optimizer` (FusedAdam, optionally inside
    an EMAOptimizer), replayed from a hipGraph captured on the first call with a given batch shape.

    `reducer` (GradReducer, optional): the data-parallel step.  Its bucket all-reduces (RCCL) are issued by the same
    autograd hooks as in the eager step, on the comm stream forked from the capture stream, and joined before the
    optimizer kernel -- so they become nodes of the graph and an N-rank step costs the host one graph launch too (the
    eager step needs ~360 ctypes calls + the hook / collective launches per step from Python).  Every rank captures the
    same sequence of collectives, in the same order.  Semantics equal the eager sequence
        loss = model.training_step(batch, i); model.backward(loss); grad_scale = reducer.finish(); optimizer.step();
        optimizer.zero_grad()
    including the host-side counters (Philox step, Adam step, EMA step, weight epoch)."""

    WARMUP = 2
    MAX_GRAPHS = 4      # captured batch shapes kept (least recently used beyond that are released)

    def __init__(self, model, optimizer, grad_scale: float = 1.0, reducer=None, wgrad_fork=None):
        # wgrad_fork: keep the weight-gradient side stream (networks.WGRAD_STREAM) as a BRANCH of the captured graph.  None:
        # EDM_GRAPH_FORK (default off: the CIFAR-10 step replays 2.6 % slower with the branch; the 272 M ImageNet net is the
        # other way round -- round 6 measurements in DESIGN 6).
        import os
        self.wgrad_fork = (os.environ.get("EDM_GRAPH_FORK", "0") == "1") if wgrad_fork is None else bool(wgrad_fork)
        self.model = model
        self.opt = optimizer
        self.base = optimizer.optimizer if isinstance(optimizer, EMAOptimizer) else optimizer
        if not isinstance(self.base, FusedAdam):
            raise TypeError("CapturedTrainStep needs the flat-arena FusedAdam")
        _runtime_env.require_graph_replay_safe("CapturedTrainStep")
        self.ema = optimizer if isinstance(optimizer, EMAOptimizer) else None
        self.reducer = reducer if (reducer is not None and reducer.active) else None
        if self.reducer is not None:
            if not self.reducer.capturable():
                raise ValueError("CapturedTrainStep: the captured data-parallel step needs the RCCL ('nccl') backend and "
                                 "the in-place fp32 transport")
            grad_scale = grad_scale / self.reducer.world
        self.grad_scale = grad_scale
        self.params = StepParams(self.base.arena.theta.device)
        # warm-up steps and the capture run on ONE dedicated stream: autograd's AccumulateGrad nodes remember the
        # stream they were created on, and a node left over from a default-stream step would pull the default
        # stream into the capture (unjoined at capture end)
        self.stream = torch.cuda.Stream(self.base.arena.theta.device)
        self._graphs = {}
        self._seen = {}
        self._calls = 0
        # True once a capture failed on ANY rank of a multi-rank job: every rank then runs the eager step from here on
        # (ranks replaying captured collectives beside a rank that issues none would hang)
        self.fallback = False

    # ---- host-side bookkeeping identical to the eager path
    def _upload(self):
        g = self.base.param_groups[0]
        t = self.base.step_count + 1
        b1, b2 = g["betas"]
        beta = 1.0          # ema = 1*ema + 0*theta: the step leaves the EMA untouched
        if self.ema is not None and self.ema._should_update_at_step():
            beta = (1 - 1 / (self.ema.current_step + 1)) ** (self.ema.gamma + 1)
        self.params.upload(step=networks.rng.step, seed=networks.rng.seed, lr=float(g["lr"]), ema_beta=float(beta),
                           grad_scale=float(self.grad_scale), bc1=1.0 - b1 ** t, bc2sqrt=math.sqrt(1.0 - b2 ** t))

    def _advance(self):
        self.base.step_count += 1
        if self.ema is not None:
            self.ema.current_step += 1
        networks.rng.step += 1
        networks.bump_weight_epoch()

    def _snapshot(self):
        return (self.base.step_count, self.ema.current_step if self.ema is not None else 0, networks.rng.step)

    def _restore(self, snap):
        self.base.step_count, cs, networks.rng.step = snap
        if self.ema is not None:
            self.ema.current_step = cs

    def _eager(self, batch):
        """one step through the ordinary Python path, driven by the same device parameter record"""
        snap = self._snapshot()
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        side, networks.WGRAD_STREAM = networks.WGRAD_STREAM, self.wgrad_fork   # the launch structure of the capture that follows
        try:
            with torch.cuda.stream(self.stream):
                loss = self.model.training_step(batch, 0)
                _backward(self.model, loss)
                if self.reducer is not None:
                    self.reducer.finish()   # joins the comm stream; 1/world rides in the device record (grad_scale)
                self.base.step_dyn(self.params.dev, ema=self.ema.ema_arena if self.ema is not None else None)
                loss = loss.detach()        # drop the autograd graph (and with it the AccumulateGrad nodes) now
        except BaseException:
            if self.reducer is not None:
                self.reducer.reset()        # a half-consumed pass must not leak pending counts into the next step
            raise
        finally:
            networks.WGRAD_STREAM = side
        cur.wait_stream(self.stream)
        self._restore(snap)
        return loss

    def _capture_agreed(self, batch):
        """Capture, then -- with more than one rank -- agree on the outcome: a rank-local failure (out of memory, a runtime
        that refuses a node) must not leave the other ranks replaying collectives nobody answers.  Returns the graph entry,
        or None when the job falls back to the eager step (single rank: the exception propagates)."""
        ent, err = None, None
        try:
            ent = self._capture(batch)
        except Exception as e:      # noqa: BLE001
            err = e
        multi = self.reducer is not None and self.reducer.world > 1
        if multi:
            import torch.distributed as dist
            flag = torch.tensor([0 if err is not None else 1], device=self.base.arena.theta.device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.reducer.group)
            if int(flag.item()) == 0:
                import warnings
                warnings.warn("tinyedm_amd: capturing the data-parallel step failed on at least one rank"
                              + (f" (here: {type(err).__name__}: {str(err)[:200]})" if err is not None else "")
                              + "; every rank runs the eager step")
                self.fallback = True
                return None
        elif err is not None:
            raise err
        return ent

    def __call__(self, batch):
        x, y = batch
        key = (tuple(x.shape), x.dtype, None if y is None else (tuple(y.shape), y.dtype))
        self._upload()
        networks.rng.dyn = self.params.dev
        try:
            ent = self._graphs.get(key)
            seen = self._seen.get(key, 0)
            self._seen[key] = seen + 1
            # (a batch shape is captured on its SECOND visit: its first step runs eagerly so that the plans / scratch of that
            # shape are allocated outside the capture -- e.g. the ragged last batch of an epoch)
            if ent is None and not self.fallback and self._calls >= self.WARMUP and seen >= 1:
                ent = self._capture_agreed(batch)
                if ent is not None:
                    while len(self._graphs) >= self.MAX_GRAPHS:      # (dict order = least recently used first)
                        self._drop(next(iter(self._graphs)))
                    self._graphs[key] = ent
            elif ent is not None:
                self._graphs[key] = self._graphs.pop(key)            # most recently used last
            if ent is None:
                loss = self._eager(batch)          # allocator / plan / stream warm-up before capturing; the fallback
            else:
                graph, sx, sy, loss, _tok = ent
                sx.copy_(x, non_blocking=True)
                if sy is not None:
                    sy.copy_(y, non_blocking=True)
                graph.replay()
        finally:
            networks.rng.dyn = None
        self._calls += 1
        self._advance()
        return loss

    def _capture(self, batch):
        x, y = batch
        sx = x.clone()
        sy = None if y is None else y.clone()
        snap = self._snapshot()
        torch.cuda.synchronize()
        mode = "global"
        if self.reducer is not None:
            # torch's process-group watchdog thread polls the events of earlier (eager) collectives with hipEventQuery;
            # under a GLOBAL-mode capture that call from another thread is an error that invalidates the capture.  The
            # device is idle now, so give the watchdog a moment to retire what it still holds, and capture in
            # thread-local mode (only this thread's calls are policed).
            import time
            time.sleep(0.5)
            mode = "thread_local"
        ops.capture_begin()
        failed = False
        graph = torch.cuda.CUDAGraph()
        # the step is captured as ONE chain: a weight-gradient side branch replays slower than the chain (CIFAR-10:
        # 15.5 vs 15.1 ms) -- hipGraph schedules the branch less favourably than the host's enqueue order does
        side, networks.WGRAD_STREAM = networks.WGRAD_STREAM, self.wgrad_fork
        try:
            with torch.cuda.graph(graph, stream=self.stream, capture_error_mode=mode):
                loss = self.model.training_step((sx, sy), 0)
                _backward(self.model, loss)     # the reducer's hooks fork the comm stream off the capture stream ...
                if self.reducer is not None:
                    self.reducer.finish()       # ... and this joins it: the all-reduces are nodes of the graph
                self.base.step_dyn(self.params.dev, ema=self.ema.ema_arena if self.ema is not None else None)
                loss = loss.detach()
        except BaseException:
            if self.reducer is not None:
                self.reducer.reset()        # pending counts / fired set of the aborted pass
            failed = True
            raise
        finally:
            networks.WGRAD_STREAM = side
            token = ops.capture_end()
            if failed:
                ops.release_capture(token)  # no graph came of it: its launch-table slots are free again
            self._restore(snap)
        return graph, sx, sy, loss, token

    def _drop(self, key):
        ent = self._graphs.pop(key, None)
        if ent is not None:
            torch.cuda.synchronize()            # no replay of it may still be running when its slots are reused
            ops.release_capture(ent[4])

    def release(self):
        """drop every captured graph (their launch-table slots go back to the pool): call when the step object is retired"""
        for key in list(self._graphs):
            self._drop(key)

    def __del__(self):
        try:
            if self._graphs:
                torch.cuda.synchronize()        # (a replay of a dying graph may still be running: see _drop)
            for ent in self._graphs.values():
                ops.release_capture(ent[4])
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass

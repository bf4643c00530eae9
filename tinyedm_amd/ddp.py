"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce of the FLAT gradient
arena over RCCL (backend "nccl" on ROCm; "gloo" on CPU for tests), launched from autograd hooks as
each bucket's gradients become final so the collective overlaps the rest of the backward pass.

This replaces the implicit Lightning ``DDPStrategy`` -> torch DDP reducer -> NCCL of the reference
(experiments/conf/*.yaml ``devices: -1, strategy: auto``; train.py:26).  Differences by design:
gradients already live contiguously in one arena, so buckets are zero-copy slices of it (no
bucket flatten/unflatten), buckets are sized for xGMI rings (default 32 MiB), and the mean over
ranks is folded into the fused optimizer kernel's ``grad_scale`` instead of a divide pass.
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.distributed as dist


def _aux_streams():
    """side streams that may hold unfinished weight-gradient kernels -- none while the side stream is switched off (a
    captured step runs as one chain: waiting on a stream outside the capture would break its isolation)"""
    from . import networks, ops
    return ops.side_streams() if networks.WGRAD_STREAM else []


class GradReducer:
    """`transport`: "fp32" (default) all-reduces the arena slices in place; "bf16" sends a bf16 copy of each bucket
    (half the xGMI bytes; the sum of `world` bf16 values is then rounded to bf16 once more: ~2^-8 relative per
    element) and writes the fp32 result back.  `force=True` (or EDM_FORCE_REDUCE=1) registers the hooks and runs the
    collectives even in a one-rank group, so a single GPU exercises the whole comm-stream path through RCCL."""

    # buckets over the part of the arena whose gradients become final LAST (the lowest offsets: the first layers of the
    # network) are smaller: what is still to be all-reduced when the backward pass ends is what the step cannot hide
    TAIL_REGION_BYTES = 16 << 20
    TAIL_BUCKET_BYTES = 8 << 20
    W3_TAIL = 4         # 3x3 layers in the last grouped weight-gradient launch of a pass (networks.W3_TAIL)

    def __init__(self, arena, bucket_bytes: int = 32 << 20, process_group=None, transport: str = None, force=None):
        self.arena = arena
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if force is None:
            force = os.environ.get("EDM_FORCE_REDUCE") == "1"
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.transport = transport or os.environ.get("EDM_GRAD_TRANSPORT", "fp32")
        if self.transport not in ("fp32", "bf16"):
            raise ValueError("GradReducer: transport must be 'fp32' or 'bf16'")
        self.enabled = True                      # False during non-final gradient-accumulation micro-batches
        self.is_cuda = arena.grad.is_cuda
        # EDM_COMM_INLINE=1: issue the collectives on the compute stream (no comm stream, no overlap): a diagnostic / the
        # fallback if forked capture misbehaves
        self.inline = os.environ.get("EDM_COMM_INLINE") == "1"
        self.comm_stream = torch.cuda.Stream() if (self.is_cuda and not self.inline) else None
        # buckets = contiguous arena ranges, built in REVERSE parameter order (backward order) over the arena's body; the
        # arena's LATE region (FlatArena layout 3: the blocks' embed Linear weights, the Embedding module, then the 0-dim
        # block gains and gain_out) is ONE more bucket, the last -- those gradients are final only when the backward pass
        # ends (networks._EmbedAllFn.backward)
        per_bucket = max(1, bucket_bytes // 4)
        per_tail = max(1, min(bucket_bytes, self.TAIL_BUCKET_BYTES) // 4)
        tail_region = self.TAIL_REGION_BYTES // 4
        self.buckets: List[dict] = []
        cur = None
        scalar_lo = getattr(arena, "scalar_lo", arena.numel)
        order = sorted(range(len(arena.params)), key=lambda i: arena.offsets[i])      # by arena position
        body = [i for i in order if arena.offsets[i] < scalar_lo]
        tail = [i for i in order if arena.offsets[i] >= scalar_lo]
        for k in reversed(range(len(body))):
            idx = body[k]
            off = arena.offsets[idx]
            end = arena.offsets[body[k + 1]] if k + 1 < len(body) else scalar_lo
            in_tail = cur is not None and cur["hi"] <= tail_region
            crossing = cur is not None and not in_tail and end <= tail_region       # (a new bucket starts AT the region's edge)
            if cur is None or crossing or cur["hi"] - off > (per_tail if in_tail else per_bucket):
                cur = {"lo": off, "hi": end, "params": [], "pending": 0, "work": None}
                self.buckets.append(cur)
            cur["lo"] = off
            cur["params"].append(idx)
        if tail:
            self.buckets.append({"lo": scalar_lo, "hi": arena.numel, "params": tail, "pending": 0, "work": None})
        self._bucket_of = {}
        for b in self.buckets:
            for idx in b["params"]:
                self._bucket_of[idx] = b
        self._hooks = []
        self.measure = False                     # record events around finish()'s wait for the comm stream (bench.py)
        self._exposed = []
        self._w3_tail_prev = None
        if self.world > 1 and "EDM_W3_TAIL" not in os.environ:
            # a small final group (networks.W3_TAIL).  Process-global: it shapes every backward pass of this process while
            # the reducer lives -- build the reducer BEFORE capturing a step, and close() it when the run is over
            from . import networks
            self._w3_tail_prev = networks.W3_TAIL
            networks.W3_TAIL = self.W3_TAIL
        if self.active:
            for idx, p in enumerate(arena.params):
                hook = self._make_hook(idx)
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))
                if hasattr(p, "_edm_hooks"):      # gradients written directly by the HIP finish kernel
                    p._edm_hooks.append(hook)
        self.reset()

    def close(self):
        """the run is over: take the hooks off the parameters and give networks.W3_TAIL back (the setting used to
        outlive the reducer and regroup the weight-gradient launches of every later model in the process)"""
        for h in self._hooks:
            h.remove()
        for idx, p in enumerate(self.arena.params):
            hooks = getattr(p, "_edm_hooks", None)
            if hooks:
                hooks.clear()
        self._hooks = []
        self.active = False
        if self._w3_tail_prev is not None:
            from . import networks
            networks.W3_TAIL, self._w3_tail_prev = self._w3_tail_prev, None

    def capturable(self) -> bool:
        """True when the collectives of a step can be captured into a hipGraph: RCCL on GPU tensors, all-reduced in place"""
        return bool(self.is_cuda and self.transport == "fp32" and dist.is_initialized()
                    and dist.get_backend(self.group) == "nccl")

    def reset(self):
        for b in self.buckets:
            b["pending"] = len(b["params"])
            b["work"] = None
        self._fired = set()

    def _make_hook(self, idx):
        def hook(_param):
            if not self.enabled or idx in self._fired or getattr(_param, "_edm_deferred", False):
                return              # deferred: the grouped weight-gradient launch has not been enqueued yet
            # A parameter is counted once per backward pass.  Weights whose gradient the HIP finish kernel writes
            # straight into the arena announce themselves through `_edm_hooks`, and autograd may ALSO run the
            # post-accumulate hook for the same parameter (it does so even though the Function returned no gradient
            # for it): without this guard every bucket was launched before half of its gradients existed.
            self._fired.add(idx)
            b = self._bucket_of[idx]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch(b)
        return hook

    def _reduce(self, b, view):
        # GPU: the collective is issued stream-ordered on the comm stream (async_op=False does not block the host with
        # RCCL; it only means "ordered on the current stream") -- this form is also what a hipGraph capture of the step
        # accepts (round 3 probe, tools/rccl_capture_probe.py: a captured async_op=True work object crashes the runtime
        # at capture end).  CPU (gloo): asynchronous work objects, waited for in finish().
        buf = view
        if self.transport == "bf16":
            buf = b["wire"] = view.to(torch.bfloat16)
        if self.is_cuda:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            b["work"] = True
        else:
            b["work"] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _launch(self, b):
        view = self.arena.grad[b["lo"]:b["hi"]]
        if self.is_cuda and self.inline:
            for s in _aux_streams():
                torch.cuda.current_stream().wait_stream(s)
            self._reduce(b, view)
        elif self.is_cuda:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for s in _aux_streams():          # gradients finished on the side stream (networks._wgrad)
                self.comm_stream.wait_stream(s)
            with torch.cuda.stream(self.comm_stream):
                self._reduce(b, view)
        else:
            self._reduce(b, view)

    def finish(self) -> float:
        """Wait for every bucket (launching any whose hooks never fired, e.g. unused parameters) and
        return the scale (1/world) the optimizer must apply to the summed gradients."""
        if self.active:
            for b in self.buckets:
                if b["work"] is None:
                    self._launch(b)
            for b in self.buckets:
                if self.is_cuda:
                    if self.transport == "bf16":
                        with torch.cuda.stream(self.comm_stream or torch.cuda.current_stream()):   # behind the collective
                            self.arena.grad[b["lo"]:b["hi"]].copy_(b["wire"])
                else:
                    b["work"].wait()
                    if self.transport == "bf16":
                        self.arena.grad[b["lo"]:b["hi"]].copy_(b["wire"])
                b["wire"] = None
            if self.is_cuda and self.comm_stream is not None:
                if self.measure and not torch.cuda.is_current_stream_capturing():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()                  # the main stream gets here when the backward pass has ended ...
                    torch.cuda.current_stream().wait_stream(self.comm_stream)
                    e1.record()                  # ... and here when the last all-reduce has: the difference is exposed
                    self._exposed.append((e0, e1))
                else:
                    torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()
        return 1.0 / self.world

    def exposed_comm_ms(self):
        """mean time finish() kept the main stream waiting for the comm stream over the measured steps (call after a
        synchronisation; measure=True must have been set), or None"""
        if not self._exposed:
            return None
        ms = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return sum(ms) / len(ms)

    def broadcast_parameters(self, src: int = 0):
        """DDP's initial parameter broadcast: one collective over the parameter arena."""
        if self.active:
            dist.broadcast(self.arena.theta, src=src, group=self.group)

    def broadcast_buffers(self, model, src: int = 0):
        """DDP's initial buffer broadcast (the random `freqs` / `phases` of the Fourier embedding, networks.py:127-131):
        ranks must not depend on equal seeds to agree on them."""
        if self.active:
            for name, buf in model.named_buffers():
                if buf.numel() > 0 and buf.is_floating_point() and not name.endswith(("weighted_sum_squared_error",)):
                    dist.broadcast(buf.data, src=src, group=self.group)

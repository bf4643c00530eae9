"""EDM2-style magnitude-preserving U-Net on hand-written HIP kernels.

Drop-in for the reference's ``tinyedm/networks.py``: same class names, constructor arguments,
attribute names, ``state_dict`` keys and fp32 NCHW/OIHW parameter layouts -- so Hydra ``_target_``
YAML and reference checkpoints resolve -- but every forward/backward runs through the C-ABI
library (``tinyedm_amd/csrc``) on NHWC bf16 activations with fp32 accumulation and fp32 master
weights (the reference's ``bf16-mixed`` policy).  There is no CPU path: CPU tensors raise.

Each block is ONE ``torch.autograd.Function`` whose forward and backward are explicit kernel
sequences (so residual adds, mp_add scales and the weight-gradient projection are folded into
conv epilogues instead of being separate autograd nodes).

Reference citations are ``file:line`` of ``/root/reference/src/tinyedm/networks.py``.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import ops

bf16, f32 = torch.bfloat16, torch.float32


# --------------------------------------------------------------------------------------
# dropout RNG state (counter based: the backward regenerates the forward's mask)
# --------------------------------------------------------------------------------------
class _Rng:
    seed: int = 42
    step: int = 0
    dyn = None      # device edm_step_params record: set by a captured training step (graph.py), overrides seed/step


rng = _Rng()


def manual_seed(seed: int) -> None:
    """Seed of the Philox streams used by dropout and the Diffuser."""
    rng.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    rng.step = 0


# --------------------------------------------------------------------------------------
# weight-normalised layers (networks.py:22-64) with packed-weight caches
# --------------------------------------------------------------------------------------
_WEIGHT_EPOCH = 0


_IN_DENOISER = [False]  # a Denoiser forward is running (its plan chose the pack layouts for ITS input shape)
FORWARD_EPOCH = 0       # grad-enabled Denoiser forwards so far (FusedAdam.zero_grad: "was a gradient written since step()?")


def bump_weight_epoch() -> None:
    """Called by anything that rewrites parameters through raw pointers (the fused optimizer)."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


def _mp_coeffs(t: float):
    c = math.sqrt((1 - t) ** 2 + t ** 2)
    return (1 - t) / c, t / c


class _WNBase(nn.Module):
    """Holds the fp32 master weight and the kernel-layout copies derived from it."""

    weight: nn.Parameter

    def _init_cache(self):
        self._cache = None
        self._cache_key = None
        self._perm: Optional[Tensor] = None       # packed row -> master row (qkv conv only)
        self._ipad: Optional[int] = None          # zero-padded input channels (conv_in only)
        self._want = ("fwd", "dgrad")
        self._split_pack = None                   # (taps, O, 3*Ip) bf16 [w_hi | w_lo | w_hi]: the "f32x3" evaluation's operand
        self._split_key = None
        self._plain = None                        # plain packs for standalone module calls (see packs())

    def _taps(self) -> int:
        return self.weight[0, 0].numel() if self.weight.dim() == 4 else 1

    def packs(self):
        """(wp_fwd, wp_dgrad, w_hat) for the current master weight.  Training mode applies the
        forced in-place normalisation first (networks.py:32-34 / 55-57), exactly once per call."""
        w = self.weight
        if not w.is_cuda:
            raise RuntimeError("tinyedm_amd: parameters must live on the GPU (there is no CPU path)")
        if self.training and getattr(self, "_fresh", False):  # prepared by this forward's multi-tensor launch
            self._fresh = False
            return self._cache
        key = (w.data_ptr(), w._version, _WEIGHT_EPOCH)
        if not self.training and not _IN_DENOISER[0] and self._cache is not None and self._cache_key == key and any(
                getattr(t, "_edm_frag", False) for t in self._cache[:2] if t is not None):
            # a Denoiser forward at the benchmarked shape left FRAGMENT-MAJOR packs here (the layout of k_conv3x3_s only); a
            # standalone module call on another shape needs the plain layout: keep a private plain copy beside the plan's
            if self._plain is None or self._plain[0] != key:
                with torch.no_grad():
                    self._plain = (key, ops.weight_prep(w.data, self._taps(), Ipad=self._ipad, want_fwd="fwd" in self._want,
                                                        want_dgrad="dgrad" in self._want, want_hat="hat" in self._want,
                                                        perm=self._perm, normalize_inplace=False))
            return self._plain[1]
        if self.training or self._cache is None or self._cache_key != key:
            perm = self._perm
            if perm is not None and perm.device != w.device:
                perm = self._perm = perm.to(w.device)
            with torch.no_grad():
                self._cache = ops.weight_prep(
                    w.data, self._taps(), Ipad=self._ipad, want_fwd="fwd" in self._want,
                    want_dgrad="dgrad" in self._want, want_hat="hat" in self._want, perm=perm,
                    normalize_inplace=self.training)
            self._cache_key = (w.data_ptr(), w._version, _WEIGHT_EPOCH)
        return self._cache

    def finish_grad(self, slabs: Tensor, scale: float = 1.0) -> Tensor:
        """slabs (S, taps, O, Ipad) fp32 in packed order -> gradient w.r.t. the master weight."""
        w = self.weight
        I = w.shape[1]
        if w.grad is not None and getattr(w, "_edm_direct", False):
            # flat-arena mode: the reduction + projection accumulates straight into the gradient arena (no autograd
            # AccumulateGrad pass, no temporary).  It is DEFERRED: ~70 such tensors per step share a few
            # multi-tensor launches (_flush_fin), after which the data-parallel reducer is told they are final.
            key = w.device.index
            pend = _fin_pending.setdefault(key, [])
            pend.append((slabs, w, self._perm, self._taps(), I, scale))
            w._edm_deferred = True
            if len(pend) >= FIN_GROUP:
                _flush_fin(key)
            _queue_backward_end(w.device)
            return None
        g = ops.wgrad_finish(slabs, w.data, self._taps(), I, perm=self._perm, scale=scale)
        return g.view_as(w)


# Weight-gradient kernels on an auxiliary stream, off the dgrad -> elementwise -> dgrad chain of the backward pass
# (EDM_WGRAD_STREAM=0 disables).  Measured round 2, eager step (tools/bench_config.py, one gpurun call): CIFAR-10 15.2 ms
# either way (the grouped launches fill the chip), MNIST 18.2 vs 18.4, ImageNet-64 latent config 145.9 vs 153.4 ms (its
# long elementwise kernels overlap with MFMA-bound weight gradients).  Inside a captured hipGraph the side branch is
# scheduled less favourably (CIFAR-10 15.5 vs 15.1 ms): graph.CapturedTrainStep switches it off while it captures.
WGRAD_STREAM = os.environ.get("EDM_WGRAD_STREAM", "1") != "0"
# 3x3 weight gradients: layers per grouped stream-K launch (csrc/conv_wgrad3.hip; 0 = per-layer kernels only) and the
# largest layer (pixels) that joins a group.  A layer with more pixels fills the chip alone and its per-layer launch
# overlaps the backward pass at a finer grain (ImageNet-64 config, 64x64 layers at batch 176: 144 vs 147 ms).  Only with
# the side stream: on one chain (captured step) everything is grouped (154.6 vs 161 ms).
W3_GROUP = max(0, min(ops.W3_MAX_LAYERS, int(os.environ.get("EDM_W3_GROUP", "16"))))
# Data-parallel runs: the LAST grouped launch of a backward pass finishes at the very end of it, so the all-reduce of its
# layers' gradients has nothing left to hide under except the optimizer.  W3_TAIL > 0 cuts the groups so that the final one
# holds at most that many layers (CIFAR-10: 43 layers as 16 + 16 + 7 + 4 instead of 16 + 16 + 11: 9.4 MB = 6.6 % of the
# gradient bytes left for the exposed tail instead of 26 MB) at the price of one more launch.  The number of 3x3 layers of
# a pass is learned from the previous pass.  Set by ddp.GradReducer when more than one rank takes part (EDM_W3_TAIL overrides;
# 0 = off: one GPU keeps the three launches).
W3_TAIL = max(0, int(os.environ.get("EDM_W3_TAIL", "0")))
_w3_seen = {}               # device index -> 3x3 layers queued so far in the running backward pass
_w3_total = {}              # device index -> 3x3 layers queued by the previous (complete) backward pass
W3_MAXPIX = int(os.environ.get("EDM_W3_MAXPIX", str(1 << 18)))
FIN_GROUP = 40              # small weight gradients per multi-tensor finish launch (csrc/weights.hip)
# 1x1 weight gradients: layers per grouped launch (csrc/conv_wgrad1x1.hip k_wgrad1x1_group; 0 = one launch per layer)
W1_GROUP = max(0, min(16, int(os.environ.get("EDM_W1_GROUP", "16"))))
_w1_pending = {}            # device index -> [(mod, x, dy, scale)] 1x1 layers waiting for their grouped launch
_bwd_end_queued = set()     # devices whose end-of-backward callback is queued for the running backward pass
# ScaleLong gates: the batch sums that form the gate MLPs' weight gradients (the second launch of ops.skip_gate_bwd), for ALL
# the gates of a backward pass in ONE launch behind it (round 6: nine launches less on the backward's chain); their finish
# rides in the last multi-tensor finish.  Arena mode only; EDM_SG_DEFER=0: per gate, as round 5.
MOD_DEFER_UNFUSED = os.environ.get("EDM_MOD_DEFER_UNFUSED", "1") != "0"   # unfused modulation backward: finish with the others (A/B)
SG_MULTI = os.environ.get("EDM_SG_MULTI", "1") != "0"     # every decoder gate of a forward pass in one launch (round 6)
SG_HALVES = os.environ.get("EDM_SG_HALVES", "1") != "0"       # ... which also writes the gated-skip halves of cat / mp_silu(cat)
SG_BWD_MULTI = os.environ.get("EDM_SG_BWD_MULTI", "1") != "0"   # ... and their backward, deferred to the last of them
SG_DEFER = os.environ.get("EDM_SG_DEFER", "1") != "0"
_sg_pending = {}            # device index -> [(ScaleLong module, ws, mean, R)]
_w3_pending = {}            # device index -> [(mod, x, dy, scale)] 3x3 layers waiting for their grouped launch
_fin_pending = {}           # device index -> [(slabs, w, perm, taps, I, scale)] small weight gradients to finish


def _run_on_side(device, fn, tensors=()):
    """Run fn() on the auxiliary stream, ordered after everything already enqueued on the current stream."""
    if WGRAD_STREAM:
        side = ops.side_stream(device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        for t in tensors:
            t.record_stream(side)
    else:
        fn()


def _flush_w3(key):
    """Launch the pending 3x3 weight gradients of device `key` as one group (arena mode: results accumulate
    straight into the gradient arena), then tell the data-parallel reducer that those gradients are final."""
    items = _w3_pending.pop(key, None)
    if not items:
        return
    dev = items[0][0].weight.device
    args = [(x, dy, m.weight.data, m.weight.grad, m._perm, scale, True) for m, x, dy, scale in items]
    keep = [t for _, x, dy, _ in items for t in (x, dy)]
    _run_on_side(dev, lambda: ops.wgrad3_group(args), keep)
    _flush_fin(key)             # the small gradients of the same stretch of the backward pass ride along
    for m, _, _, _ in items:
        m.weight._edm_deferred = False
        for hook in getattr(m.weight, "_edm_hooks", ()):
            hook(m.weight)


def _flush_w1(key):
    """The pending 1x1 weight gradients of device `key` as ONE grouped slab launch; their finish (reduction over the
    splits + projection through the weight normalisation) joins the multi-tensor finish queue."""
    items = _w1_pending.pop(key, None)
    if not items:
        return
    dev = items[0][0].weight.device
    box = []
    _run_on_side(dev, lambda: box.extend(ops.conv_wgrad_1x1_group([(x, dy) for _, x, dy, _ in items])),
                 [t for _, x, dy, _ in items for t in (x, dy)])
    pend = _fin_pending.setdefault(key, [])
    for (m, _, _, scale), slabs in zip(items, box):
        w = m.weight
        pend.append((slabs, w, m._perm, 1, w.shape[1], scale))


def _flush_fin(key):
    """One multi-tensor launch for the pending small weight gradients of device `key` (on the auxiliary stream,
    ordered after both streams' producers of the slabs)."""
    _flush_w1(key)              # queued 1x1 layers first: their slabs are finished by this very launch
    items = _fin_pending.pop(key, None)
    if not items:
        return
    dev = items[0][1].device
    args = [(slabs, w.data, w.grad, perm, taps, I, scale, True) for slabs, w, perm, taps, I, scale in items]
    _run_on_side(dev, lambda: ops.wgrad_finish_multi(args), [a[0] for a in args])
    for _, w, *_rest in items:
        w._edm_deferred = False
        for hook in getattr(w, "_edm_hooks", ()):
            hook(w)


def _flush_sg(key):
    """the pending ScaleLong gates of device `key`: one launch for their weight-gradient batch sums; the projection through
    the weight normalisation joins the multi-tensor finish queue"""
    items = _sg_pending.pop(key, None)
    if not items:
        return
    outs = ops.skip_gate_wgrad_multi([(ws, mean, R) for _, ws, mean, R in items])
    for (sl, _, _, _), (gw1h, gw2h) in zip(items, outs):
        sl.layer1.finish_grad(gw1h.view(1, 1, *gw1h.shape))
        sl.layer2.finish_grad(gw2h.view(1, 1, *gw2h.shape))


def _backward_end(key):
    """End of a backward pass on device `key`: flush the last partial group, then make the stream that ran the
    backward wait for the auxiliary stream (the optimizer reads what the weight-gradient kernels wrote)."""
    _bwd_end_queued.discard(key)
    if key in _w3_seen:
        _w3_total[key] = _w3_seen.pop(key)
    _flush_sg(key)
    _flush_w3(key)
    _flush_fin(key)
    if WGRAD_STREAM:
        torch.cuda.current_stream(key).wait_stream(ops.side_stream(key))


def _queue_backward_end(device):
    key = torch.device(device).index
    if key in _bwd_end_queued:
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _backward_end(key))
        _bwd_end_queued.add(key)
    except RuntimeError:      # not inside a backward pass: finish right away
        _backward_end(key)


def reset_backward_state():
    """Forget deferred work of a backward pass that did not complete (an exception inside autograd leaves its
    end-of-backward callback unrun).  Called at the start of every Denoiser forward."""
    if _bwd_end_queued or _w3_pending or _fin_pending or _w1_pending or _sg_pending:
        _bwd_end_queued.clear()
        for items in _sg_pending.values():
            for sl, *_rest in items:
                sl.layer1.weight._edm_deferred = sl.layer2.weight._edm_deferred = False
        _sg_pending.clear()
        for items in _w1_pending.values():
            for m, _, _, _ in items:
                m.weight._edm_deferred = False
        _w1_pending.clear()
        for items in _fin_pending.values():
            for _, w, *_rest in items:
                w._edm_deferred = False
        _fin_pending.clear()
        for items in _w3_pending.values():
            for m, _, _, _ in items:
                m.weight._edm_deferred = False
        _w3_pending.clear()
        _w3_seen.clear()
        if WGRAD_STREAM:
            for s in ops.side_streams():
                torch.cuda.current_stream(s.device).wait_stream(s)


def _wgrad(mod, x, dy, taps, scale=1.0):
    """Weight gradient of a conv layer.  In flat-arena mode the result is only needed by the optimizer, so the
    kernels run on the side stream (EDM_WGRAD_STREAM=0 disables), off the critical dgrad -> elementwise -> dgrad
    chain of the backward pass, and 3x3 layers are collected into groups of W3_GROUP per launch."""
    w = mod.weight
    direct = w.grad is not None and getattr(w, "_edm_direct", False)
    if (taps == 9 and W3_GROUP and w.dim() == 4 and ops.wgrad3_supported(x, dy, w.shape[1])
            and (not WGRAD_STREAM or x.shape[0] * x.shape[1] * x.shape[2] <= W3_MAXPIX)):
        if direct:
            key = w.device.index
            pend = _w3_pending.setdefault(key, [])
            if pend and (pend[0][1].shape[2] + 2 > 64) != (x.shape[2] + 2 > 64):
                _flush_w3(key)                        # a group shares one halo class
                pend = _w3_pending.setdefault(key, [])
            pend.append((mod, x, dy, scale))
            w._edm_deferred = True                    # autograd may run the parameter's hooks before the group is launched
            seen = _w3_seen[key] = _w3_seen.get(key, 0) + 1
            if len(pend) >= W3_GROUP or (W3_TAIL and seen == _w3_total.get(key, 0) - W3_TAIL):
                _flush_w3(key)
            _queue_backward_end(w.device)
            return None
        g = torch.empty_like(w.data)
        ops.wgrad3_group([(x, dy, w.data, g, mod._perm, scale, False)])
        return g
    if direct and taps == 1 and W1_GROUP and w.dim() == 4 and ops.wgrad1x1_group_supported(x, dy):
        # flat-arena mode: the layer joins a group of 1x1 layers whose slabs come from ONE launch (and whose finish rides in
        # the multi-tensor finish launch that follows it)
        key = w.device.index
        pend = _w1_pending.setdefault(key, [])
        pend.append((mod, x, dy, scale))
        w._edm_deferred = True
        if len(pend) >= W1_GROUP:
            _flush_fin(key)
        _queue_backward_end(w.device)
        return None
    if direct:
        _run_on_side(w.device, lambda: mod.finish_grad(ops.conv_wgrad(x, dy, taps), scale=scale), (x, dy))
        _queue_backward_end(w.device)
        return None
    return mod.finish_grad(ops.conv_wgrad(x, dy, taps), scale=scale)


# bf16 tile of a weight-prep workgroup (rows x fan_in): 96 KB = 16 rows of a 256-channel 3x3 layer, two workgroups per CU
PREP_TILE_BYTES = int(os.environ.get("EDM_PREP_TILE_KB", "96")) * 1024


class _PrepPlan:
    """Multi-tensor weight preparation: ONE launch normalises and packs every weight of a network
    (edm_weight_prep_multi) into persistent kernel-layout buffers, instead of one launch per layer."""

    def __init__(self, mods, frag=None, wants=None, cat=None):
        """cat: Linear modules (same in_features) whose fp32 effective weights are laid out back to back in ONE buffer
        (`self.wcat`, rows in the order given): the batched embed Linear of a Denoiser reads it as one GEMM operand -- no
        torch.cat of 21 packs per forward (an ATen copy kernel in every captured step and sampler evaluation until round 6).
        frag: {module: (fwd, dgrad)} -- the 3x3 convs whose forward / dgrad pack is written FRAGMENT-MAJOR (csrc/weights.hip)
        because k_conv3x3_s, the 8x8 layers' kernel, runs them at the current input shape (Denoiser._frag_flags).
        wants: per module, the packs this plan keeps ("fwd", "dgrad", "hat", "split"); default = the module's own.  "split" =
        the split-bf16 operand of the "f32x3" evaluation, a PERSISTENT buffer rewritten in place by run() (a captured solve
        holds its address: ADVICE r4)."""
        self.mods = mods
        frag = frag or {}
        wants = [tuple(m._want) for m in mods] if wants is None else list(wants)
        self.wants = wants
        self.pins = 0               # captured graphs that read this plan's buffers (ops.note_capture_pin / release_capture)
        self._pinned_forever = False  # ... or a capture made without ops.capture_begin(): nobody can tell when it dies
        self.splits = []
        dev = mods[0].weight.device
        self.ptr_key = tuple(m.weight.data_ptr() for m in mods)
        desc = np.zeros(len(mods), dtype=np.dtype([
            ("w", "<u8"), ("fwd", "<u8"), ("dgrad", "<u8"), ("hat", "<u8"), ("perm", "<u8"),
            ("O", "<i4"), ("I", "<i4"), ("taps", "<i4"), ("Ipad", "<i4"), ("row0", "<i4"), ("pad", "<i4")]))
        assert desc.dtype.itemsize == 64
        groups, row0, lds = [], 0, 0
        self.caches = []
        self.wcat, cat_rows = None, {}
        if cat:
            kin = cat[0].weight.shape[1]
            if all(c.weight.dim() == 2 and c.weight.shape[1] == kin for c in cat):
                self.wcat = torch.empty(sum(c.weight.shape[0] for c in cat), kin, device=dev, dtype=f32)
                r = 0
                for c in cat:
                    cat_rows[c] = r
                    r += c.weight.shape[0]
        for k, m in enumerate(mods):
            w = m.weight
            O, I, taps = w.shape[0], w.shape[1], m._taps()
            ipad = I if m._ipad is None else m._ipad
            want = wants[k]
            wf = torch.empty(taps, O, ipad, device=dev, dtype=bf16) if "fwd" in want else None
            wd = torch.empty(taps, I, O, device=dev, dtype=bf16) if "dgrad" in want else None
            if m in cat_rows:
                wh = self.wcat[cat_rows[m]:cat_rows[m] + O]
            else:
                wh = torch.empty(O, I * taps, device=dev, dtype=f32) if ("hat" in want or "split" in want) else None
            self.splits.append(torch.empty(taps, O, 3 * ((I + 31) // 32 * 32), device=dev, dtype=bf16)
                               if "split" in want else None)
            if m._perm is not None and m._perm.device != dev:
                m._perm = m._perm.to(dev)
            # rows per workgroup: the bf16 tile rb x (I*taps) must fit LDS; power of two <= 32
            rb = 32
            while rb > 1 and (rb * I * taps * 2 > PREP_TILE_BYTES or rb > O):
                rb //= 2
            lds = max(lds, rb * I * taps * 2)
            ff, fd = frag.get(m, (False, False))
            if not (taps == 9 and I % 32 == 0 and O % 32 == 0 and ipad == I and rb % 8 == 0 and m._perm is None):
                ff = fd = False
            if ff and wf is not None:
                wf._edm_frag = True
            if fd and wd is not None:
                wd._edm_frag = True
            desc[k] = (w.data_ptr(), wf.data_ptr() if wf is not None else 0, wd.data_ptr() if wd is not None else 0,
                       wh.data_ptr() if wh is not None else 0, m._perm.data_ptr() if m._perm is not None else 0,
                       O, I, taps | (0x100 if ff and wf is not None else 0) | (0x200 if fd and wd is not None else 0), ipad,
                       row0, rb)
            groups += [(k, r) for r in range(0, O, rb)]
            row0 += O
            self.caches.append((wf, wd, wh))
        self.total_rows = row0
        self.desc = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
        self.groups = torch.tensor(groups, dtype=torch.int32).to(dev)
        self.lds_bytes = lds
        self.eval_key = None

    @property
    def pinned(self) -> bool:
        """a live captured graph reads this plan's buffers: never evicted"""
        return self.pins > 0 or self._pinned_forever

    def valid_for(self, mods):
        return tuple(m.weight.data_ptr() for m in mods) == self.ptr_key

    def run(self, training: bool):
        key = (tuple(m.weight._version for m in self.mods), _WEIGHT_EPOCH)
        if training or key != self.eval_key:
            ops.weight_prep_multi(self.desc, self.groups, self.lds_bytes, training)
            for m, c, sp in zip(self.mods, self.caches, self.splits):
                if sp is not None:          # in place: the address a captured f32x3 solve replays with stays valid
                    ops.split_pack(c[2], m._taps(), out=sp)
            # the in-place normalisation does not go through torch: remember what the packs correspond to
            self.eval_key = None if training else key
        if torch.cuda.is_current_stream_capturing() and not ops.note_capture_pin(self):
            self._pinned_forever = True
        for m, c, sp in zip(self.mods, self.caches, self.splits):
            m._cache = c
            m._cache_key = (m.weight.data_ptr(), m.weight._version, _WEIGHT_EPOCH)
            m._fresh = training
            m._split_pack = sp
            m._split_key = m._cache_key if sp is not None else None


class Conv2d(_WNBase):
    """networks.py:22-43.  Standalone call takes/returns NCHW like the reference."""

    def __init__(self, in_channels, out_channels, kernel_size):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        self._init_cache()

    def forward(self, x: Tensor) -> Tensor:
        xh = ops.nchw_to_nhwc_bf16(x.float().contiguous())
        y = _ConvFn.apply(xh, self.weight, self)
        return ops.nhwc_bf16_to_nchw(y).to(x.dtype)

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}"


class Linear(_WNBase):
    """networks.py:46-64 (always fp32, like the reference's autocast-disabled islands)."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.randn(out_features, in_features))
        self._init_cache()
        self._want = ("hat",)

    def forward(self, x: Tensor) -> Tensor:
        return _LinearFn.apply(x.float().contiguous(), self.weight, self)

    def extra_repr(self) -> str:
        return f"{self.in_features}, {self.out_features}"


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, mod: Conv2d):
        wf, wd, _ = mod.packs()
        ctx.mod, ctx.wd = mod, wd
        ctx.save_for_backward(x)
        return ops.conv_igemm(x, wf, mod._taps())

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        taps = ctx.mod._taps()
        gx = ops.conv_igemm(gy, ctx.wd, taps) if ctx.needs_input_grad[0] else None
        gw = _wgrad(ctx.mod, x, gy, taps) if ctx.needs_input_grad[1] else None
        return gx, gw, None


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, mod: Linear):
        _, _, wh = mod.packs()
        ctx.mod, ctx.wh = mod, wh
        ctx.save_for_backward(x)
        return ops.linear_fwd(x, wh)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = ops.linear_dgrad(gy, ctx.wh) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            dwh = ops.linear_wgrad(gy, x)
            gw = ctx.mod.finish_grad(dwh.view(1, 1, *dwh.shape))
        return gx, gw, None


# --------------------------------------------------------------------------------------
# L0 functions kept for API parity (NCHW in / out)
# --------------------------------------------------------------------------------------
def pixel_norm(x: Tensor, eps: float = 1e-4, dim=1) -> Tensor:
    """networks.py:9-14, channel-dim case on the GPU kernel."""
    if dim != 1 or x.dim() != 4 or eps != 1e-4:
        raise NotImplementedError("tinyedm_amd.pixel_norm implements the (B,C,H,W), dim=1, eps=1e-4 hot-path case")
    xn, _, _ = ops.pixelnorm_silu_fwd(ops.nchw_to_nhwc_bf16(x.float().contiguous()))
    return ops.nhwc_bf16_to_nchw(xn).to(x.dtype)


def mp_silu(x: Tensor) -> Tensor:
    """networks.py:83-84 for fp32 side tensors (the U-Net uses the fused bf16 kernels)."""
    return torch.nn.functional.silu(x) / 0.596


def mp_add(a: Tensor, b: Tensor, t: float = 0.5) -> Tensor:
    """networks.py:87-88."""
    return a.lerp(b, t) / np.sqrt((1 - t) ** 2 + t ** 2)


class UpSample(nn.Module):
    """networks.py:67-72 (nearest-exact x2), NCHW API."""

    def forward(self, x):
        return ops.nhwc_bf16_to_nchw(ops.up2(ops.nchw_to_nhwc_bf16(x.float().contiguous()))).to(x.dtype)


class DownSample(nn.Module):
    """networks.py:75-80 (2x2 average pool), NCHW API."""

    def forward(self, x):
        return ops.nhwc_bf16_to_nchw(ops.pool2(ops.nchw_to_nhwc_bf16(x.float().contiguous()))).to(x.dtype)


class _ResampleFn(torch.autograd.Function):
    """alias=True (downsample only): the input is handed back as a second output for the U-Net skip, so that the skip's
    gradient arrives HERE and is added inside the backward kernel instead of by an autograd `add` (see _ResBlockFn).
    want_silu=True (upsample only, round 6): the second output is mp_silu of the upsampled tensor -- the operand of the DecU
    block's first conv (networks.py:313-316) -- written by the same pass (ops.up2_silu); not differentiable: the block's
    backward applies mp_silu' itself."""

    @staticmethod
    def forward(ctx, x, up: bool, alias: bool = False, want_silu: bool = False):
        ctx.up = up
        ctx.set_materialize_grads(False)
        if up and want_silu:
            y, s = ops.up2_silu(x)
            ctx.mark_non_differentiable(s)
            return y, s
        y = ops.up2(x) if up else ops.pool2(x, 0.25)
        return (y, x) if alias else y

    @staticmethod
    def backward(ctx, g, g_alias=None):
        if g is None:
            return g_alias, None, None, None
        g = g.contiguous()
        if ctx.up:
            gx = ops.pool2(g, 1.0)
            if g_alias is not None:
                gx = ops.axpby(gx, 1.0, g_alias.contiguous(), 1.0)
            return gx, None, None, None
        return ops.up2(g, 0.25, add=None if g_alias is None else g_alias.contiguous()), None, None, None


class UncertaintyNet(nn.Module):
    """networks.py:91-103 (optional, off in every shipped config)."""

    def __init__(self, in_features: int, hidden_features: int):
        super().__init__()
        self.linear1 = Linear(in_features + 1, hidden_features)
        self.linear2 = Linear(hidden_features, 1)
        self.gain = nn.Parameter(torch.zeros(()))

    def forward(self, x: Tensor):
        x = torch.cat((x, torch.ones_like(x[:, 0:1])), dim=1)
        x = mp_silu(self.linear1(x))
        return self.gain * self.linear2(x)


# --------------------------------------------------------------------------------------
# ScaleLong skip gate + concat (networks.py:106-118, 309-311)
# --------------------------------------------------------------------------------------
class ScaleLong(nn.Module):
    def __init__(self, dim, r=16):
        super().__init__()
        self.layer1 = Conv2d(dim + 1, int(dim // r), 1)
        self.layer2 = Conv2d(int(dim // r), dim, 1)
        self.layer1._want = ("hat",)
        self.layer2._want = ("hat",)
        if SG_DEFER:    # their gradients are written by ONE launch behind the backward pass (_flush_sg): late parameters
            self.layer1.weight._edm_late = True         # of the flat arena (ema.FlatArena layout 3)
            self.layer2.weight._edm_late = True

    def forward(self, inp: Tensor) -> Tensor:
        """NCHW in -> gate (B,C,1,1), as the reference module."""
        xh = ops.nchw_to_nhwc_bf16(inp.float().contiguous())
        B, H, W, C = xh.shape
        w1h, w2h = self.layer1.packs()[2], self.layer2.packs()[2]
        _, gate, _ = ops.skip_gate_fwd(xh, w1h, w2h)
        return gate.view(B, C, 1, 1)


# ScaleLong gate: mean over H*W + MLP (and their backward) in one launch per direction (csrc/elementwise.hip
# k_skip_gate_*); EDM_SKIP_GATE_FUSED=0 keeps the two-launch form (A/B runs)
SKIP_GATE_FUSED = os.environ.get("EDM_SKIP_GATE_FUSED", "1") != "0"
# The saved pre-activation of a block carries its dropout mask (dropped <=> NaN; ops.conv3x3_mod(mark_dropped=True)), so
# the backward epilogue regenerates no Philox stream; EDM_U_MARKS=0: the mask is recomputed (A/B runs)
U_MARKS = os.environ.get("EDM_U_MARKS", "1") != "0"


# Copy-free torch.cat((input, skip * gate)) (networks.py:311; round 4): the kernel that produces a decoder block's output
# writes it -- and mp_silu of it -- straight into the left half of the NEXT block's concatenated operands
# (ops.conv_igemm(out=, silu_out=)), the skip half is filled by one small kernel, and in the backward the 1x1 dgrad that
# produces d loss / d cat writes its two halves to the tensors their consumers read (ops.conv_igemm(split=)).
# EDM_FUSE_CAT=0 keeps the standalone concat kernels (A/B runs).
FUSE_CAT = os.environ.get("EDM_FUSE_CAT", "1") != "0"
# round 6: launches that only existed because of where a tensor was materialised -- the 2x2 average pool of an EncD block
# without a 1x1 conv rides in the pixel-norm kernels (forward: ops.pool_pixelnorm_silu_fwd, the pooled tensor is never
# written; backward: the pixel-norm backward writes the gradient at the resolution before the pool), and a DecU block's
# upsample also emits mp_silu of its result (ops.up2_silu).  Bit-identical; EDM_FUSE_RESAMPLE=0 keeps the separate kernels.
FUSE_RESAMPLE = os.environ.get("EDM_FUSE_RESAMPLE", "1") != "0"
# fragment-major weight packs for the layers k_conv3x3_s runs (Denoiser._frag_flags); EDM_FRAG_PACKS=0: plain packs (A/B runs)
FRAG_PACKS = os.environ.get("EDM_FRAG_PACKS", "1") != "0"


def _col_block(buf: Tensor, C: int) -> Tensor:
    """(B, H, W, C) alias of the left C columns of the NHWC buffer `buf` (no autograd view relation: the rows are written
    by kernels through raw pointers, never by torch in-place ops)"""
    B, H, W, Ct = buf.shape
    return torch.empty(0, device=buf.device, dtype=buf.dtype).set_(buf.untyped_storage(), buf.storage_offset(),
                                                                   (B, H, W, C), (H * W * Ct, W * Ct, Ct, 1))


def _dest_ok(dest, x: Tensor, taps: int, Cin: int, Cout: int) -> bool:
    """the producer's final conv can write through an output descriptor (its kernel generation has that form)"""
    if dest is None:
        return False
    B, H, W, _ = x.shape
    return ops._igemm_entry(B * H * W, W, Cout, taps, Cin) in ops._KERNEL_ID and dest[0].shape[:3] == (B, H, W)


class _ConcatGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, skip, w1, w2, sl: ScaleLong, want_silu: bool = False):
        ctx.set_materialize_grads(False)          # no zero-filled gradient tensor for the non-differentiable `sil`
        B, H, W, Cs = skip.shape
        w1h, w2h = sl.layer1.packs()[2], sl.layer2.packs()[2]
        if SKIP_GATE_FUSED:
            mean, gate, z1 = ops.skip_gate_fwd(skip, w1h, w2h)     # mean over H*W + gate MLP: one launch
        else:
            mean = ops.reduce_hw(skip, scale=1.0 / (H * W))
            gate, z1 = ops.scalelong_fwd(mean, w1h, w2h)
        # want_silu: also emit mp_silu(cat), the input of the block's first conv, from the same pass
        cat, sil = ops.concat_gate_fwd(inp, skip, gate, want_silu)
        ctx.sl, ctx.Ci = sl, inp.shape[-1]
        ctx.save_for_backward(skip, mean, gate, z1, w1h, w2h)
        if want_silu:
            ctx.mark_non_differentiable(sil)
            return cat, sil
        return cat

    @staticmethod
    def backward(ctx, gcat, _gsil=None):
        skip, mean, gate, z1, w1h, w2h = ctx.saved_tensors
        gcat = gcat.contiguous()
        Ci, Cs = ctx.Ci, skip.shape[-1]
        if SKIP_GATE_FUSED:
            gmean, gw1h, gw2h = ops.skip_gate_bwd(gcat, Ci, skip, mean, w1h, w2h, gate, z1)
        else:
            ggate = ops.reduce_hw(gcat, C=Cs, c_off=Ci, y=skip)
            gmean, gw1h, gw2h = ops.scalelong_bwd(mean, w1h, w2h, gate, z1, ggate)
        ginp, gskip = ops.concat_gate_bwd(gcat, gate, gmean, Ci)
        gw1 = ctx.sl.layer1.finish_grad(gw1h.view(1, 1, *gw1h.shape))
        gw2 = ctx.sl.layer2.finish_grad(gw2h.view(1, 1, *gw2h.shape))
        return ginp, gskip, gw1, gw2, None, None


# --------------------------------------------------------------------------------------
# embeddings (networks.py:121-178), fp32
# --------------------------------------------------------------------------------------
class ClassEmbedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.linear = Linear(num_embeddings, embedding_dim)

    def forward(self, class_labels: Tensor):
        onehot = torch.nn.functional.one_hot(class_labels.flatten(), self.num_embeddings).float()
        return self.linear(onehot * np.sqrt(self.num_embeddings, dtype=np.float32))


class FourierEmbedding(nn.Module):
    def __init__(self, embedding_dim: int):
        super().__init__()
        self.register_buffer("freqs", 2 * np.pi * torch.randn(embedding_dim))
        self.register_buffer("phases", 2 * np.pi * torch.rand(embedding_dim))

    def forward(self, x):
        """x = c_noise = ln(sigma)/4 (networks.py:138-141)."""
        s = (x.float() * 4).exp().flatten().contiguous()
        return ops.fourier_fwd(s, self.freqs, self.phases, s.numel())


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigma, labels, w_sigma, w_cls, mod: "Embedding"):
        ctx.set_materialize_grads(False)
        B = max(sigma.numel(), labels.numel() if labels is not None else 1)
        four = ops.fourier_fwd(sigma, mod.fourier_embed.freqs, mod.fourier_embed.phases, B)
        wsh = mod.sigma_embed.packs()[2]
        wch = mod.class_embed.linear.packs()[2] if labels is not None else None
        es = ops.linear_fwd(four, wsh)
        pre, out = ops.embed_combine_fwd(es, wch, labels, mod.add_factor)
        ctx.mod, ctx.labels = mod, labels
        ctx.save_for_backward(four, pre)
        ctx.mark_non_differentiable(four)
        return four, out

    @staticmethod
    def backward(ctx, _gfour, gout):
        four, pre = ctx.saved_tensors
        mod = ctx.mod
        wshape = tuple(mod.class_embed.linear.weight.shape) if ctx.labels is not None else None
        ges, gwch = ops.embed_combine_bwd(gout.contiguous(), pre, ctx.labels, mod.add_factor, wshape)
        dws = ops.linear_wgrad(ges, four)
        gws = mod.sigma_embed.finish_grad(dws.view(1, 1, *dws.shape))
        gwc = mod.class_embed.linear.finish_grad(gwch.view(1, 1, *gwch.shape)) if gwch is not None else None
        return None, None, gws, gwc, None


class Embedding(nn.Module):
    """networks.py:144-178: (sigma, labels|None) -> (fourier (B,F), emb (B,E)), fp32."""

    def __init__(self, fourier_dim: int, embedding_dim: int, num_classes: int | None = None, add_factor: float = 0.5):
        super().__init__()
        self.fourier_dim = fourier_dim
        self.add_factor = add_factor
        self.embedding_dim = embedding_dim
        self.num_classes = num_classes
        self.fourier_embed = FourierEmbedding(fourier_dim)
        self.sigma_embed = Linear(fourier_dim, embedding_dim)
        self.sigma_embed.weight._edm_late = True      # (gradients written at the very end of a backward pass: ema.FlatArena)
        self.class_embed = None
        if num_classes is not None and num_classes != -1:
            self.class_embed = ClassEmbedding(num_classes, embedding_dim)
            self.class_embed.linear.weight._edm_late = True

    def forward(self, sigmas: Tensor, class_labels: Tensor | None = None):
        if class_labels is not None and self.class_embed is None:
            raise ValueError("class_labels is not None, but num_classes is None. ")
        if not sigmas.is_cuda:
            raise RuntimeError("tinyedm_amd.Embedding: sigma must be a GPU tensor (there is no CPU path)")
        sigma = sigmas.detach().float().flatten().contiguous()
        labels = None
        if class_labels is not None:
            labels = class_labels.detach().to(sigma.device).flatten().to(torch.int64).contiguous()
            if sigma.numel() not in (1, labels.numel()):
                raise ValueError("sigma and class_labels batch sizes differ")
        w_cls = self.class_embed.linear.weight if labels is not None else None
        four, out = _EmbeddingFn.apply(sigma, labels, self.sigma_embed.weight, w_cls, self)
        if sigmas.numel() == 1:
            four = four[:1]
        return four, out


# --------------------------------------------------------------------------------------
# attention (networks.py:181-207)
# --------------------------------------------------------------------------------------
def _qkv_perm(C: int, heads: int) -> Tensor:
    """packed output row (head, which, dd) of the qkv conv -> reference row head*3d + dd*3 + which
    (the interleaving implied by qkv.view(b, heads, -1, 3, hw), networks.py:194)."""
    d = C // heads
    h = torch.arange(heads).view(heads, 1, 1)
    w = torch.arange(3).view(1, 3, 1)
    dd = torch.arange(d).view(1, 1, d)
    return (h * 3 * d + dd * 3 + w).reshape(-1).to(torch.int32)


class CosineAttention(nn.Module):
    def __init__(self, embedding_dim: int, num_heads):
        super().__init__()
        assert embedding_dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = embedding_dim // num_heads
        self.embedding_dim = embedding_dim
        self.qkv_conv = Conv2d(embedding_dim, 3 * embedding_dim, 1)
        self.out_conv = Conv2d(embedding_dim, embedding_dim, 1)
        self.qkv_conv._perm = _qkv_perm(embedding_dim, num_heads)

    def forward_nhwc(self, x: Tensor, dest=None) -> Tensor:
        """dest = (cat, sil) buffers of the next decoder block: see FUSE_CAT"""
        # (grad mode is always off INSIDE Function.forward: whether a backward can follow is decided here)
        out = _AttnFn.apply(x, self.qkv_conv.weight, self.out_conv.weight, self, dest, torch.is_grad_enabled())
        if dest is not None and out.data_ptr() == dest[0].data_ptr():
            out._edm_cat = dest
        return out

    def forward(self, x: Tensor) -> Tensor:
        """NCHW API of the reference module."""
        y = self.forward_nhwc(ops.nchw_to_nhwc_bf16(x.float().contiguous()))
        return ops.nhwc_bf16_to_nchw(y).to(x.dtype)

    def forward_f32(self, x: Tensor, xp: Tensor | None = None, want=None):
        """reference-precision evaluation (NHWC fp32 in / out): qkv conv in the master row order, exact-fp32 attention.
        xp: x as split-bf16 pairs when the producer of x wrote them in the same launch; want: see _res_f32"""
        qkv = _conv_f32(self.qkv_conv, xp if (xp is not None and _split_ok(self.qkv_conv)) else x, 1)
        B, H, W, _ = qkv.shape
        if _split_ok(self.out_conv) and ops.split_attention_ok(self.embedding_dim, self.num_heads, H * W):
            # split back end: the attention runs on the bf16 matrix cores too (three passes over hi/lo pairs) and hands its
            # output to the out conv as pairs
            y = ops.split_attention(qkv, self.num_heads, pairs=True)
        else:
            y = ops.f32_attention(qkv, self.num_heads)
        a, b = _mp_coeffs(0.5)
        return _conv_f32(self.out_conv, y, 1, residual=x, alpha=b, beta=a, want=want)


class _AttnFn(torch.autograd.Function):
    """networks.py:191-207.  Where csrc/attention_fused.hip covers the shape (C = 256, 4 heads, <= 256 tokens: every attention
    block of the CIFAR-10 nets) the qkv projection runs INSIDE the attention kernel -- forward: fused kernel + out conv (with
    the mp_add epilogue), 2 launches instead of 3; backward: fused kernel (dO = b * gout . W_out, recomputed q / k / v) + the
    qkv conv's dgrad, 2 instead of 3 -- and the qkv tensor never exists in HBM; the 1x1 weight gradients join the grouped
    launches as before."""

    @staticmethod
    def forward(ctx, x, w_qkv, w_out, mod: CosineAttention, dest=None, grad_mode: bool = True):
        wf_qkv, wd_qkv, _ = mod.qkv_conv.packs()
        wf_out, wd_out, _ = mod.out_conv.packs()
        fused = ops.attention_qkv_supported(x, mod.num_heads)
        stat = qkv = None
        if fused:
            y, stat = ops.attention_qkv_fwd(x, wf_qkv, mod.num_heads, want_stat=grad_mode and any(ctx.needs_input_grad[:3]))
        else:
            qkv = ops.conv_igemm(x, wf_qkv, 1)
            y = ops.attention_fwd(qkv, mod.num_heads)
        a, b = _mp_coeffs(0.5)
        C = wf_out.shape[1]
        if _dest_ok(dest, y, 1, y.shape[-1], C):     # the block's output goes straight into the next block's cat / mp_silu(cat)
            out = ops.conv_igemm(y, wf_out, 1, residual=x, alpha=b, beta=a, out=_col_block(dest[0], C),
                                 silu_out=_col_block(dest[1], C))
            out._edm_cat = dest
        else:
            out = ops.conv_igemm(y, wf_out, 1, residual=x, alpha=b, beta=a)
        ctx.mod, ctx.fused = mod, fused
        if fused:
            ctx.save_for_backward(x, stat, y, wd_qkv, wd_out, wf_qkv)
        else:
            ctx.save_for_backward(x, qkv, y, wd_qkv, wd_out)
        return out

    @staticmethod
    def backward(ctx, gout):
        mod = ctx.mod
        gout = gout.contiguous()
        a, b = _mp_coeffs(0.5)
        if ctx.fused:
            x, stat, y, wd_qkv, wd_out, wf_qkv = ctx.saved_tensors
            if stat is None:
                raise RuntimeError("tinyedm_amd: attention backward without the forward's softmax statistics (the forward ran "
                                   "under torch.no_grad())")
            gqkv = ops.attention_qkv_bwd(x, y, gout, stat, wf_qkv, wd_out, mod.num_heads, alpha=b)
        else:
            x, qkv, y, wd_qkv, wd_out = ctx.saved_tensors
            gy = ops.conv_igemm(gout, wd_out, 1, alpha=b)
            gqkv = ops.attention_bwd(qkv, y, gy, mod.num_heads)
        gw_out = _wgrad(mod.out_conv, y, gout, 1, b)
        gx = ops.conv_igemm(gqkv, wd_qkv, 1, residual=gout, alpha=1.0, beta=a)
        gw_qkv = _wgrad(mod.qkv_conv, x, gqkv, 1)
        return gx, gw_qkv, gw_out, None, None, None


# --------------------------------------------------------------------------------------
# residual blocks (networks.py:210-329)
# --------------------------------------------------------------------------------------
class _ResBlockFn(torch.autograd.Function):
    """u -> [conv_1x1] -> [pixel_norm] -> mp_silu -> conv3x3 -> x(1+gain*embed) -> mp_silu -> dropout
    -> conv3x3 -> mp_add with the skip path (networks.py:246-263 encoder / 312-327 decoder)."""

    @staticmethod
    def forward(ctx, u, emb, w1x1, w1, w2, wemb, gain, blk, lin_view, glin_view, token, s_pre=None, alias=False,
                gm_view=None, skip=None, w_sl1=None, w_sl2=None, pre=None, dest=None, pool=False, gate_pre=None):
        # skip (decoder blocks with a U-Net skip and no upsample, FUSE_CAT): the concatenation of networks.py:311 happens
        # HERE.  pre = (cat, sil) whose left halves the producer of u already wrote (u is that half of cat): only the gated
        # skip half is filled in; otherwise the standalone concat kernel builds both.  dest = the (cat, sil) buffers of the
        # NEXT block: this block's last kernel writes its output (and mp_silu of it) into their left halves.
        # pool=True (encoder blocks with a downsample and no 1x1 conv, FUSE_RESAMPLE): u is the tensor BEFORE the 2x2 average
        # pool; the pool happens inside the pixel-norm kernels of both directions.
        # alias=True: the block input u is handed back as a second output.  The Denoiser takes the U-Net skip from that
        # output, so the skip's gradient arrives in THIS backward (g_alias) and is added by the kernel that writes the
        # input gradient -- not by an autograd `add` launch per skip (9-16 ATen kernels, 0.8 GB per step, round 1).
        enc = blk.is_encoder
        ctx.set_materialize_grads(False)
        has1 = w1x1 is not None
        taps = 9
        wf1, wd1, _ = blk.conv_3x3_1.packs()
        wf2, wd2, _ = blk.conv_3x3_2.packs()
        batched = lin_view is not None          # embed Linear evaluated for all blocks at once (_EmbedAllFn)
        weh = None if batched else blk.embed.packs()[2]
        wd11 = None
        ctx.has_skip = skip is not None
        if skip is not None:
            sl = blk.cat_factor
            ctx.sgb = None
            if gate_pre is not None:    # (SG_MULTI: the Denoiser computed every decoder gate in one launch behind the encoder)
                mean, gate, z1, w1h, w2h, ctx.sgb, halves = gate_pre
            else:
                w1h, w2h = sl.layer1.packs()[2], sl.layer2.packs()[2]
                mean, gate, z1 = ops.skip_gate_fwd(skip, w1h, w2h)      # mean over H*W + gate MLP: one launch
            ctx.Ci = u.shape[-1]
            if pre is not None and pre[0].shape[-1] == u.shape[-1] + skip.shape[-1] and pre[0].data_ptr() == u.data_ptr():
                cat, s_pre = pre
                if not (gate_pre is not None and halves == cat.data_ptr()):    # (else: the gate launch wrote them already)
                    ops.skip_half_fwd(skip, gate, cat, s_pre)
            else:
                cat, s_pre = ops.concat_gate_fwd(u.contiguous(), skip, gate, True)
            u = cat
            ctx.skip_saved = (skip, mean, gate, z1, w1h, w2h)
        ctx.pool = bool(pool)
        if enc and pool:
            assert not has1
            xres, s, dsave = ops.pool_pixelnorm_silu_fwd(u)
        elif enc:
            x = u
            if has1:
                wf11, wd11, _ = blk.conv_1x1.packs()
                x = ops.conv_igemm(u, wf11, 1)
            xres, s, dsave = ops.pixelnorm_silu_fwd(x)
        else:
            xres = u
            fold = False
            if has1:
                wf11, wd11, _ = blk.conv_1x1.packs()
                # round 6: the projection conv_1x1(cat) (networks.py:313) rides in the second 3x3 conv as a second reduction
                # behind its nine taps (ops.conv3x3_fold) where that conv runs on the static-schedule kernel; elsewhere it
                # stays a launch of its own whose result is the 3x3 conv's residual
                fold = u.is_contiguous() and ops.conv3x3_fold_supported((*u.shape[:3], wf1.shape[1]), wf2.shape[1], u.shape[-1])
                if not fold:
                    xres = ops.conv_igemm(u, wf11, 1)
            s = s_pre if s_pre is not None else ops.silu_fwd(u)     # s_pre: emitted by the concat kernel
            dsave = None
        lin = lin_view if batched else ops.linear_fwd(emb, weh)
        pdrop = blk.dropout_rate if blk.training else 0.0
        seed, sub, step = rng.seed, blk.rng_sub, rng.step
        if ops.FUSE_MOD and ops.IGEMM_VERSION == 0:
            # modulation + mp_silu + dropout ride in the conv epilogue; the pre-activation r1 is only written when
            # a backward pass will need it
            r1, a2 = ops.conv3x3_mod(s, wf1, lin, gain, pdrop, seed, sub, step, want_u=any(ctx.needs_input_grad),
                                     dyn=rng.dyn, mark_dropped=U_MARKS)
        else:
            r1 = ops.conv_igemm(s, wf1, taps)
            a2 = ops.mod_silu_drop_fwd(r1, lin, gain, pdrop, seed, sub, step, dyn=rng.dyn)
        a, b = _mp_coeffs(blk.add_factor)
        Co = wf2.shape[1]
        if (not enc) and fold:
            if _dest_ok(dest, a2, taps, a2.shape[-1], Co):
                out = ops.conv3x3_fold(a2, wf2, u, wf11, b, a, out=_col_block(dest[0], Co), silu_out=_col_block(dest[1], Co))
                out._edm_cat = dest
            else:
                out = ops.conv3x3_fold(a2, wf2, u, wf11, b, a)
        elif _dest_ok(dest, a2, taps, a2.shape[-1], Co):
            out = ops.conv_igemm(a2, wf2, taps, residual=xres, alpha=b, beta=a, out=_col_block(dest[0], Co),
                                 silu_out=_col_block(dest[1], Co))
            out._edm_cat = dest
        else:
            out = ops.conv_igemm(a2, wf2, taps, residual=xres, alpha=b, beta=a)
        ctx.blk, ctx.enc, ctx.has1 = blk, enc, has1
        ctx.drop = (pdrop, seed, sub, step, rng.dyn)
        ctx.u_marked = U_MARKS and ops.FUSE_MOD and ops.IGEMM_VERSION == 0
        ctx.batched, ctx.glin_view, ctx.gm_view = batched, glin_view, gm_view
        ctx.has_token = token is not None
        # (an encoder block without a 1x1 conv never reads u again: not kept -- with pool=True it is the 4x larger tensor)
        ctx.save_for_backward(u if (not enc or has1) else None, xres if enc else None, dsave, s, r1, lin, a2,
                              None if batched else emb, gain, wd1, wd2, wd11, weh)
        return (out, u) if alias else out

    @staticmethod
    def backward(ctx, gout, g_alias=None):
        u, xn, dsave, s, r1, lin, a2, emb, gain, wd1, wd2, wd11, weh = ctx.saved_tensors
        if g_alias is not None:
            g_alias = g_alias.contiguous()
        blk, enc, has1 = ctx.blk, ctx.enc, ctx.has1
        pdrop, seed, sub, step, dyn = ctx.drop
        gout = gout.contiguous()
        a, b = _mp_coeffs(blk.add_factor)
        # d loss / d lin goes (batched mode) into this block's column slice of the shared buffer; _EmbedAllFn.backward
        # turns the whole buffer into the embed-weight and embedding gradients with two GEMMs
        glin_out = ctx.glin_view if ctx.batched else None
        # flat-arena mode: d loss / d gain is accumulated straight into the gradient arena (no AccumulateGrad add)
        gp = blk.gain
        gdirect = gp.grad is not None and getattr(gp, "_edm_direct", False) and gp.grad.is_contiguous()
        ggain_out = gp.grad if gdirect else None
        deferred = False
        if ops.FUSE_MOD and ops.IGEMM_VERSION == 0 and (gout.shape[1] * gout.shape[2]) % 32 == 0:
            # conv2's dgrad with the modulation backward in its epilogue: ga2 never touches HBM.  With a shared gm buffer
            # (batched mode, arena gradients) the finish (glin = gm * gain, d loss / d gain) is NOT launched per block:
            # _EmbedAllFn.backward runs ONE edm_mod_finish_multi for all blocks; until then the gain's gradient is not
            # final (`_edm_deferred`: the data-parallel reducer must not count it yet)
            deferred = ctx.gm_view is not None and gdirect
            gr1, glin, ggain = ops.conv3x3_modbwd(gout, wd2, b, r1, lin, gain, pdrop, seed, sub, step, glin_out=glin_out,
                                                     ggain_out=ggain_out, dyn=dyn, gm_out=ctx.gm_view if deferred else None,
                                                     u_marked=ctx.u_marked)
            if deferred:
                gp._edm_deferred = True
        else:
            # (maps whose H*W is no multiple of 32 -- MNIST's 28x28 / 14x14 / 7x7 levels: the fused epilogue's 32-pixel MFMA row
            # blocks would straddle samples.)  The raw modulation gradient still goes to the shared buffer when there is one:
            # no finish launch per block (round 6: 27 launches of 6 us per MNIST step)
            deferred = MOD_DEFER_UNFUSED and ctx.gm_view is not None and gdirect
            ga2 = ops.conv_igemm(gout, wd2, 9, alpha=b)
            gr1, glin, ggain = ops.mod_silu_drop_bwd(r1, lin, gain, ga2, pdrop, seed, sub, step, glin_out=glin_out,
                                                        ggain_out=ggain_out, dyn=dyn, gm_out=ctx.gm_view if deferred else None)
            if deferred:
                gp._edm_deferred = True
        if gdirect:
            ggain = None
            if not deferred:
                for hook in getattr(gp, "_edm_hooks", ()):
                    hook(gp)
        gw2 = _wgrad(blk.conv_3x3_2, a2, gout, 9, b)
        gwemb = gemb = gtoken = None
        if ctx.batched:
            gtoken = ops.zeros_f32((1,), gout.device) if ctx.has_token else None
        else:
            dweh = ops.linear_wgrad(glin, emb)
            gwemb = blk.embed.finish_grad(dweh.view(1, 1, *dweh.shape))
            gemb = ops.linear_dgrad(glin, weh) if ctx.needs_input_grad[1] else None
        fuse = ops.FUSE_MOD and ops.IGEMM_VERSION == 0 and not enc
        gs = None if fuse else ops.conv_igemm(gr1, wd1, 9)      # decoder: mp_silu backward rides in the dgrad epilogue
        gw1 = _wgrad(blk.conv_3x3_1, s, gr1, 9)
        gw11 = None
        if enc and ctx.pool:
            # pixel-norm backward + the pool's backward (4 pixels per pooled pixel, 1/4 each) + the skip gradient: one pass
            gu = ops.pool_pixelnorm_silu_bwd(xn, dsave, gout, a, gs, gadd=g_alias)
            g_alias = None
        elif enc:
            gx = ops.pixelnorm_silu_bwd(xn, dsave, gout, a, gs, gadd=None if has1 else g_alias)
            if has1:
                gw11 = _wgrad(blk.conv_1x1, u, gx, 1)
                gu = ops.conv_igemm(gx, wd11, 1, residual=g_alias, alpha=1.0, beta=1.0 if g_alias is not None else 0.0)
            else:
                gu = gx
            g_alias = None
        else:
            if has1:
                t = ops.conv3x3_silubwd(gr1, wd1, u) if fuse else ops.silu_bwd(u, gs)
                if ctx.has_skip and ops._igemm_entry(gout.numel() // gout.shape[-1], gout.shape[2], wd11.shape[1], 1,
                                                     gout.shape[-1]) in ops._KERNEL_ID:
                    # d loss / d cat leaves the 1x1 dgrad as its two halves: d loss / d input (final) and the raw gradient
                    # of the gated skip -- no gcat tensor, no concat backward pass over it
                    Ci = ctx.Ci
                    B_, H_, W_, Ct = u.shape
                    gu = torch.empty(B_, H_, W_, Ci, device=u.device, dtype=bf16)
                    gcs = torch.empty(B_, H_, W_, Ct - Ci, device=u.device, dtype=bf16)
                    ops.conv_igemm(gout, wd11, 1, residual=t, alpha=a, beta=1.0, split=(Ci, gu, gcs))
                else:
                    gu = ops.conv_igemm(gout, wd11, 1, residual=t, alpha=a, beta=1.0)
                    gcs = None
                gw11 = _wgrad(blk.conv_1x1, u, gout, 1, a)
            else:
                gu = ops.conv3x3_silubwd(gr1, wd1, u, gout, a) if fuse else ops.silu_bwd(u, gs, gout, a)
                gcs = None
        gskip = gwsl1 = gwsl2 = None
        if ctx.has_skip:
            skip, mean, gate, z1, w1h, w2h = ctx.skip_saved
            Ci = ctx.Ci
            sl = blk.cat_factor
            wsl = (sl.layer1.weight, sl.layer2.weight)
            defer = SG_DEFER and all(w.grad is not None and getattr(w, "_edm_direct", False) for w in wsl)
            sgb = getattr(ctx, "sgb", None)
            queued = False
            if gcs is None:             # (no split form for this shape: the skip half is read out of gcat in place)
                gmean, *gws = ops.skip_gate_bwd(gu, Ci, skip, mean, w1h, w2h, gate, z1, defer_wgrad=defer)
                gu, gskip = ops.concat_gate_bwd(gu, gate, gmean, Ci)
            elif defer and sgb is not None and SG_BWD_MULTI:
                # nothing on the backward's chain needs this gate's backward: only the gradient of the U-Net skip does, and
                # the ENCODER consumes that.  gskip goes out as a placeholder; the last decoder gate of the pass to get here
                # runs the gate backward of all of them in one launch and fills every placeholder in a second (_SgbPass)
                gskip = torch.empty_like(gcs)
                sgb.pending.append((gcs, skip, w1h, w2h, gate, z1, gskip, sl, mean))
                queued = True
            else:
                gmean, *gws = ops.skip_gate_bwd(gcs, 0, skip, mean, w1h, w2h, gate, z1, defer_wgrad=defer)
                gskip = ops.skip_half_bwd(gcs, gate, gmean)
            if defer:       # the batch sums of every gate's weight gradients: one launch at the end of the pass (_flush_sg)
                if not queued:
                    _sg_pending.setdefault(wsl[0].device.index, []).append((sl, gws[0], mean, w1h.shape[0]))
                wsl[0]._edm_deferred = wsl[1]._edm_deferred = True
                _queue_backward_end(wsl[0].device)
            else:
                gw1h, gw2h = gws
                gwsl1 = sl.layer1.finish_grad(gw1h.view(1, 1, *gw1h.shape))
                gwsl2 = sl.layer2.finish_grad(gw2h.view(1, 1, *gw2h.shape))
            if sgb is not None:
                sgb.arrived += 1
                if sgb.arrived == sgb.expected:
                    sgb.flush()
        if g_alias is not None:         # decoder blocks are never asked for an alias; kept for completeness
            gu = ops.axpby(gu, 1.0, g_alias, 1.0)
        return (gu, gemb, gw11, gw1, gw2, gwemb, ggain, None, None, None, gtoken, None, None, None, gskip, gwsl1, gwsl2,
                None, None, None, None)


_rng_sub_counter = [0]


class _BlockBase(nn.Module):
    is_encoder = True

    def _common(self, out_channels, embedding_dim, attention, num_heads, dropout_rate):
        self.conv_3x3_2 = Conv2d(out_channels, out_channels, 3)
        self.dropout = nn.Dropout(dropout_rate)
        self.attention = CosineAttention(out_channels, num_heads) if attention else nn.Identity()
        self.embed = Linear(embedding_dim, out_channels)
        self.embed.weight._edm_late = True        # its gradient is written by _EmbedAllFn.backward, behind every block (ema.FlatArena)
        self.gain = nn.Parameter(torch.ones(()))
        _rng_sub_counter[0] += 1
        self.rng_sub = _rng_sub_counter[0]

    def _res(self, u: Tensor, embedding: Tensor, lin=None, s_pre=None, alias=False, skip=None, dest=None, pool=False,
             gate=None):
        """alias=True: returns (out, alias of u) -- see _ResBlockFn.forward.  skip / dest: FUSE_CAT (decoder blocks);
        pool: u is the block input BEFORE its 2x2 average pool (FUSE_RESAMPLE, encoder blocks)"""
        w11 = self.conv_1x1.weight if isinstance(self.conv_1x1, Conv2d) else None
        has_attn = isinstance(self.attention, CosineAttention)
        sk = ()
        if skip is not None or dest is not None or pool:
            cf = self.cat_factor if skip is not None else None
            sk = (skip, cf.layer1.weight if cf is not None else None, cf.layer2.weight if cf is not None else None,
                  getattr(u, "_edm_cat", None) if skip is not None else None, None if has_attn else dest, bool(pool),
                  gate if skip is not None else None)
        if lin is None:
            out = _ResBlockFn.apply(u, embedding, w11, self.conv_3x3_1.weight, self.conv_3x3_2.weight,
                                    self.embed.weight, self.gain, self, None, None, None, s_pre, alias, None, *sk)
        else:
            lin_view, glin_view, token, gm_view = lin
            out = _ResBlockFn.apply(u, None, w11, self.conv_3x3_1.weight, self.conv_3x3_2.weight, None, self.gain,
                                    self, lin_view, glin_view, token, s_pre, alias, gm_view, *sk)
        ualias = None
        if alias:
            out, ualias = out
        if has_attn:
            out = self.attention.forward_nhwc(out, dest)
        elif dest is not None and out.data_ptr() == dest[0].data_ptr():
            out._edm_cat = dest
        return (out, ualias) if alias else out


# Reference-precision evaluation, two conv back ends (Denoiser.set_eval_dtype): "f32" = exact fp32 products on the
# f32-input MFMA (csrc/eval_f32.hip k_conv_f32, 1/16 of the bf16 rate); "f32x3" (round 4) = split-bf16: activations and
# weights as (hi, lo) bf16 pairs, three bf16 MFMA passes hi.w_hi + hi.w_lo + lo.w_hi accumulated in fp32 on the tuned bf16
# kernels (ops.split_conv) -- 2^-17 per operand instead of exact, a third of the bf16 rate instead of a sixteenth.  Everything
# between the convs (fp32 activations, elementwise kernels, attention) is shared.
_SPLIT_EVAL = [False]
# round 6: in the split evaluation a block's last conv writes what its CONSUMER reads (the input halves of the next block's
# concatenated operands, mp_silu pairs, pairs for the attention's qkv conv) instead of fp32 + a copy / elementwise kernel per
# consumer; EDM_F32_FUSE_OUT=0 keeps the round-5 sequence (A/B runs)
F32_FUSE_OUT = os.environ.get("EDM_F32_FUSE_OUT", "1") != "0"


def _split_ok(mod) -> bool:
    """this conv runs on the split-bf16 back end in the current evaluation (so its input may arrive as pairs)"""
    return bool(_SPLIT_EVAL[0]) and isinstance(mod, _WNBase) and mod.weight.shape[1] % 32 == 0 and mod.weight.shape[0] % 8 == 0


class _CatPre:
    """what a decoder block of the split evaluation hands to a successor that concatenates its output with a U-Net skip:
    the pairs buffers of torch.cat((input, skip * gate)) and of mp_silu of it (networks.py:311, 316) with their LEFT column
    blocks already written by the producer's last kernel (ops.split_conv(dest=)); the successor fills the skip halves"""
    __slots__ = ("cat", "sil")

    def __init__(self, cat, sil):
        self.cat, self.sil = cat, sil


def _conv_f32(mod: "_WNBase", x: Tensor, taps: int, pairs_out: bool = False, want=None, **kw):
    """x: fp32 NHWC, or -- split back end only -- bf16 (hi, lo) pairs written by the producer (ops.f32_pixelnorm_silu /
    f32_silu / f32_concat_gate / split_conv with pairs=True).  pairs_out: hand the result on as pairs (split back end only).
    want (split back end only; round 6: what the CONSUMER of this conv's output reads, written by this launch instead of by
    a copy / elementwise kernel of its own): "pairs" -> (fp32, pairs); "silu" -> (fp32, mp_silu as pairs);
    ("dest", cat, sil) -> _CatPre(cat, sil), the left column blocks of the next block's concatenated operands"""
    w_hat = mod.packs()[2]
    if _split_ok(mod):
        key = (mod.weight.data_ptr(), mod.weight._version, _WEIGHT_EPOCH)
        if mod._split_pack is None or mod._split_key != key:
            # not under a Denoiser plan (the plan's run() writes its persistent buffer and sets the key): build it here
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("tinyedm_amd: split-bf16 weight pack is stale under stream capture (the plan refreshes "
                                   "it before the capture: Denoiser._prep_all)")
            mod._split_pack, mod._split_key = ops.split_pack(w_hat, taps), key
        xp = x if x.dtype == bf16 else ops.f32_to_pairs(x)
        if want == "pairs":
            return ops.split_conv(xp, mod._split_pack, taps, also_pairs=True, **kw)
        if want == "silu":
            return ops.split_conv(xp, mod._split_pack, taps, silu_pairs=True, **kw)
        if want is not None:
            _, cat, sil = want
            ops.split_conv(xp, mod._split_pack, taps, dest=(cat, sil), **kw)
            return _CatPre(cat, sil)
        return ops.split_conv(xp, mod._split_pack, taps, pairs_out=pairs_out, **kw)
    if x.dtype == bf16 or pairs_out or want is not None:
        raise RuntimeError("tinyedm_amd: split-bf16 pairs reached a conv that runs on the exact-fp32 kernel")
    return ops.f32_conv(x, w_hat, taps, **kw)


def _res_f32(blk, xres, s: Tensor, lin: Tensor, want=None):
    """the residual branch of a block in the reference-precision evaluation path: conv3x3 -> modulation + mp_silu (fused
    epilogue; eval: no dropout) -> conv3x3 + mp_add with the skip path (networks.py:253-263 / 317-327).
    want: what the block's output must come as (see _conv_f32; decided by Denoiser._forward_f32 from the NEXT block)"""
    # (split back end: a2 only feeds conv2 -- it travels as pairs, written by conv1's epilogue)
    a2 = _conv_f32(blk.conv_3x3_1, s, 9, lin=lin, gain=blk.gain.detach(),
                   pairs_out=_split_ok(blk.conv_3x3_1) and _split_ok(blk.conv_3x3_2))
    a, b = _mp_coeffs(blk.add_factor)
    # xres = ("fold", x pairs): the block's skip projection conv_1x1(x) rides in the second conv (ops.split_conv(fold=))
    kw = dict(residual=xres)
    if isinstance(xres, tuple):
        m1 = blk.conv_1x1
        kw = dict(fold=(xres[1], m1._split_pack))
    if isinstance(blk.attention, CosineAttention):
        # (the attention reads the block's output twice: as the qkv conv's operand -- pairs, from the same launch -- and as
        # the fp32 residual of its out conv)
        both = _split_ok(blk.conv_3x3_2) and _split_ok(blk.attention.qkv_conv)
        out = _conv_f32(blk.conv_3x3_2, a2, 9, alpha=b, beta=a, want="pairs" if both else None, **kw)
        out, outp = out if both else (out, None)
        return blk.attention.forward_f32(out, outp, want)
    return _conv_f32(blk.conv_3x3_2, a2, 9, alpha=b, beta=a, want=want, **kw)


def _out_conv_split(blk) -> bool:
    """the conv that writes this block's output runs on the split back end (so it can write what the consumer reads)"""
    return _split_ok(blk.attention.out_conv if isinstance(blk.attention, CosineAttention) else blk.conv_3x3_2)


def _as_nhwc(x: Tensor):
    """Accept the reference's NCHW float tensors at module boundaries; internal tensors are NHWC bf16."""
    if x.dtype == bf16 and getattr(x, "_edm_nhwc", False):
        return x, False
    return ops.nchw_to_nhwc_bf16(x.float().contiguous()), True


def _tag(x: Tensor) -> Tensor:
    x._edm_nhwc = True
    return x


class EncoderBlock(_BlockBase):
    """networks.py:210-265."""
    is_encoder = True

    def __init__(self, in_channels: int, out_channels: int, embedding_dim: int, down: bool, attention: bool,
                 num_heads: int = 4, dropout_rate: float = 0.0, add_factor: float = 0.3):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.add_factor = add_factor
        self.resample = DownSample() if down else nn.Identity()
        self.conv_1x1 = Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()
        self.conv_3x3_1 = Conv2d(out_channels, out_channels, 3)
        self._common(out_channels, embedding_dim, attention, num_heads, dropout_rate)

    def forward(self, input: Tensor, embedding: Tensor, _lin=None, _alias: bool = False, _dest=None):
        """_alias=True (Denoiser only, NHWC input): returns (out, alias of the input): the tensor the U-Net skip should
        be taken from, so that the skip gradient is summed inside this block's backward kernels.  _dest: see Denoiser._silu_dest"""
        x, conv = _as_nhwc(input)
        ualias = None
        emb = None if _lin is not None else _emb32(embedding, x.shape[0])
        if (isinstance(self.resample, DownSample) and FUSE_RESAMPLE and not isinstance(self.conv_1x1, Conv2d)
                and x.shape[-1] <= 1024 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0):
            # the pool rides in the block's pixel-norm kernels (forward and backward); the alias is the tensor before it
            out = self._res(x, emb, _lin, alias=_alias, pool=True, dest=_dest)
            if _alias:
                out, ualias = out
        else:
            if isinstance(self.resample, DownSample):
                if _alias:
                    x, ualias = _ResampleFn.apply(x, False, True)
                else:
                    x = _ResampleFn.apply(x, False)
            if _alias and ualias is None:
                out, ualias = self._res(x, emb, _lin, alias=True, dest=_dest)
            else:
                out = self._res(x, emb, _lin, dest=_dest)
        if conv:
            out = ops.nhwc_bf16_to_nchw(out).to(input.dtype)
        else:
            cat_pre = getattr(out, "_edm_cat", None)
            out = _tag(out)
            if cat_pre is not None:
                out._edm_cat = cat_pre
        return (out, _tag(ualias)) if _alias else out

    def forward_f32(self, x: Tensor, lin: Tensor, want=None):
        """networks.py:246-265 on NHWC fp32 activations (evaluation only); lin = this block's embed Linear output (B, C)"""
        pairs = _split_ok(self.conv_3x3_1)
        if (F32_FUSE_OUT and isinstance(self.resample, DownSample) and not isinstance(self.conv_1x1, Conv2d)
                and x.shape[-1] <= 1024):
            xn, s = ops.f32_pool_pixelnorm_silu(x, pairs=pairs)     # the pooled tensor is never written
        else:
            if isinstance(self.resample, DownSample):
                x = ops.f32_pool2(x)
            if isinstance(self.conv_1x1, Conv2d):
                x = _conv_f32(self.conv_1x1, x, 1)
            xn, s = ops.f32_pixelnorm_silu(x, pairs=pairs)
        return _res_f32(self, xn, s, lin, want)


class DecoderBlock(_BlockBase):
    """networks.py:268-329."""
    is_encoder = False

    def __init__(self, in_channels: int, out_channels: int, embedding_dim: int, up: bool, attention: bool,
                 num_heads: int = 4, skip_channels: int = 0, dropout_rate: float = 0.0, add_factor: float = 0.3):
        super().__init__()
        self.add_factor = add_factor
        self.dropout_rate = dropout_rate
        self.cat_factor = ScaleLong(skip_channels) if skip_channels > 0 else None
        self.resample = UpSample() if up else nn.Identity()
        total = in_channels + skip_channels
        self.conv_1x1 = Conv2d(total, out_channels, 1) if total != out_channels else nn.Identity()
        self.conv_3x3_1 = Conv2d(total, out_channels, 3)
        self._common(out_channels, embedding_dim, attention, num_heads, dropout_rate)

    def _gate_in_block(self) -> bool:
        """the concatenation, the ScaleLong gate and their backward live inside this block's autograd node"""
        return (FUSE_CAT and SKIP_GATE_FUSED and not isinstance(self.resample, UpSample)
                and isinstance(self.conv_1x1, Conv2d) and self.cat_factor is not None)

    def forward(self, input: Tensor, embedding: Tensor, skip: Tensor | None = None, _lin=None, _dest=None, _gate=None) -> Tensor:
        x, conv = _as_nhwc(input)
        s_pre = None
        if skip is not None and FUSE_CAT and SKIP_GATE_FUSED and not isinstance(self.resample, UpSample) \
                and isinstance(self.conv_1x1, Conv2d):
            # the concatenation, the ScaleLong gate and their backward live inside the block's autograd node
            assert self.cat_factor is not None
            sk, _ = _as_nhwc(skip)
            out = self._res(x, None if _lin is not None else _emb32(embedding, x.shape[0]), _lin, skip=sk, dest=_dest,
                            gate=_gate)
            return ops.nhwc_bf16_to_nchw(out).to(input.dtype) if conv else _tag(out)
        if skip is not None:
            assert self.cat_factor is not None
            sk, _ = _as_nhwc(skip)
            fuse_silu = not isinstance(self.resample, UpSample)    # mp_silu(cat) is conv1's input unless an upsample follows
            x = _ConcatGateFn.apply(x, sk, self.cat_factor.layer1.weight, self.cat_factor.layer2.weight, self.cat_factor,
                                    fuse_silu)
            if fuse_silu:
                x, s_pre = x
        if skip is None and not isinstance(self.resample, UpSample):
            pre = getattr(x, "_edm_cat", None)       # (out, mp_silu(out)) written by the producer of x: Denoiser._silu_dest
            if pre is not None and pre[0].data_ptr() == x.data_ptr() and pre[0].shape == x.shape:
                s_pre = pre[1]
        if isinstance(self.resample, UpSample):
            if FUSE_RESAMPLE:
                x, s_pre = _ResampleFn.apply(x, True, False, True)      # mp_silu of the upsampled tensor from the same pass
            else:
                x = _ResampleFn.apply(x, True)
                s_pre = None
        out = self._res(x, None if _lin is not None else _emb32(embedding, x.shape[0]), _lin, s_pre, dest=_dest)
        return ops.nhwc_bf16_to_nchw(out).to(input.dtype) if conv else _tag(out)

    @staticmethod
    def _split_pack_ready(mod) -> bool:
        """the module's persistent split-bf16 pack is current (written by the Denoiser's plan before this forward)"""
        key = (mod.weight.data_ptr(), mod.weight._version, _WEIGHT_EPOCH)
        return mod._split_pack is not None and mod._split_key == key

    def forward_f32(self, x, lin: Tensor, skip: Tensor | None = None, want=None):
        """networks.py:306-329 on NHWC fp32 activations (evaluation only).  x: fp32, or what the producer wrote for THIS
        block (split back end, _conv_f32(want=)): (fp32, mp_silu pairs), or a _CatPre whose left halves hold x"""
        s = silp = pre = None
        if isinstance(x, _CatPre):
            pre = x
        elif isinstance(x, tuple):
            x, silp = x
        up = isinstance(self.resample, UpSample)
        if skip is not None:
            cf = self.cat_factor
            gate = ops.f32_skip_gate(skip, cf.layer1.packs()[2], cf.layer2.packs()[2])
            if pre is not None:
                ops.f32_skip_half(skip, gate, pre.cat, pre.sil)    # the input halves are there already: no concat copy
                x, s = pre.cat, pre.sil
            else:
                # (split back end, no upsample behind the concat: cat and mp_silu(cat) only feed convs -- pairs)
                pairs = (not up) and isinstance(self.conv_1x1, Conv2d) and _split_ok(self.conv_1x1) and _split_ok(self.conv_3x3_1)
                x, s = ops.f32_concat_gate(x, skip, gate, not up, pairs=pairs)
        elif pre is not None:
            raise RuntimeError("tinyedm_amd: a pre-concatenated input reached a decoder block without a skip")
        if up and F32_FUSE_OUT:
            x, s = ops.f32_up2_silu(x, pairs=_split_ok(self.conv_3x3_1))   # upsample + mp_silu of it: one pass
        elif up:
            x, s = ops.f32_up2(x), None
        if (isinstance(self.conv_1x1, Conv2d) and F32_FUSE_OUT and x.dtype == bf16 and _split_ok(self.conv_1x1)
                and _split_ok(self.conv_3x3_1) and _split_ok(self.conv_3x3_2) and self._split_pack_ready(self.conv_1x1)
                and ops.split_conv_fold_supported((*x.shape[:3], 2 * self.conv_3x3_2.weight.shape[1]),
                                                  self.conv_3x3_2.weight.shape[0], x.shape[-1] // 2)):
            xres = ("fold", x)          # the projection rides in the block's second 3x3 conv (ops.split_conv(fold=))
        else:
            xres = _conv_f32(self.conv_1x1, x, 1) if isinstance(self.conv_1x1, Conv2d) else x
        if s is None:
            s = silp if (silp is not None and _split_ok(self.conv_3x3_1)) else ops.f32_silu(x, pairs=_split_ok(self.conv_3x3_1))
        return _res_f32(self, xres, s, lin, want)


def _emb32(embedding: Tensor, B: int) -> Tensor:
    e = embedding.float()
    if e.dim() != 2:
        raise ValueError("embedding must be (B, E)")
    if e.shape[0] != B:
        if e.shape[0] != 1:
            raise ValueError(f"embedding batch {e.shape[0]} does not match input batch {B}")
        e = e.expand(B, -1)
    return e.contiguous()


# --------------------------------------------------------------------------------------
# architecture tables + builders (networks.py:332-487)
# --------------------------------------------------------------------------------------
def get_encoder_blocks_types() -> tuple[str, ...]:
    return ("Enc",) * 3 + ("EncD",) + ("Enc",) * 3 + ("EncD",) + ("EncA",) * 3 + ("EncD",) + ("EncA",) * 3


def get_decoder_blocks_types() -> tuple[str, ...]:
    return (("DecA", "Dec") + ("DecA",) * 4 + ("DecU",) + ("DecA",) * 4 + ("DecU",) + ("Dec",) * 4 + ("DecU",)
            + ("Dec",) * 4)


def get_encoder_out_channels() -> tuple[int, ...]:
    return (192,) * 4 + (384,) * 4 + (576,) * 4 + (768,) * 3


def get_decoder_out_channels() -> tuple[int, ...]:
    return (768,) * 6 + (576,) * 5 + (384,) * 6 + (192,) * 4


def get_skip_connections() -> tuple[bool, ...]:
    """The indices of decoder blocks that have skip connections."""
    return (False, False) + (True,) * 4 + ((False,) + (True,) * 4) * 3


def get_skip_channels(encoder_out_channels, decoder_out_channels, skip_connections) -> tuple[int, ...]:
    pool = iter(list(encoder_out_channels[::-1]) + [encoder_out_channels[0]])  # + the input block
    return tuple(int(next(pool)) if flag else 0 for flag in skip_connections)


def build_encoder_blocks(block_types, out_channels, **kwargs):
    blocks = nn.ModuleList()
    cin = out_channels[0]
    for t, cout in zip(block_types, out_channels):
        blocks.append(EncoderBlock(in_channels=cin, out_channels=cout, down=t.endswith("D"), attention=t.endswith("A"),
                                   **kwargs))
        cin = cout
    return blocks


def build_decoder_blocks(block_types, out_channels, skip_channels, **kwargs):
    blocks = nn.ModuleList()
    cin = out_channels[0]
    for t, cout, sc in zip(block_types, out_channels, skip_channels):
        blocks.append(DecoderBlock(in_channels=cin, out_channels=cout, skip_channels=sc, up=t.endswith("U"),
                                   attention=t.endswith("A"), **kwargs))
        cin = cout
    return blocks


# --------------------------------------------------------------------------------------
# Denoiser (networks.py:490-605)
# --------------------------------------------------------------------------------------
class _EmbedAllFn(torch.autograd.Function):
    """The per-block embed Linears (networks.py:256, 320) of ALL blocks as one fp32 GEMM: emb (B,E) x Wcat^T
    (E, sum C) -> lin_all (B, sum C).  Blocks read/write column slices; their d loss / d lin lands in a shared
    buffer, and this node's backward (which autograd runs after every block, via the scalar `token` edge) turns
    it into all embed-weight gradients and d loss / d emb with two more GEMMs."""

    @staticmethod
    def forward(ctx, emb, den, *weights):
        ctx.set_materialize_grads(False)
        blocks = den._res_blocks()
        wcat = den._wcat(blocks)
        lin_all = ops.linear_fwd(emb, wcat)
        glin_all = ops.zeros_f32(lin_all.shape, lin_all.device)
        # shared raw modulation-gradient buffer + the per-block table of ONE finish launch (only when every block gain's
        # gradient lives in the flat arena; otherwise the blocks finish one by one)
        table = den._modfin_table(blocks) if any(ctx.needs_input_grad) else None
        gm_all = ops.zeros_f32(lin_all.shape, lin_all.device) if table is not None else None
        ctx.den, ctx.glin_all, ctx.gm_all, ctx.lin_all, ctx.table = den, glin_all, gm_all, lin_all, table
        ctx.save_for_backward(emb, wcat)
        token = ops.zeros_f32((1,), emb.device)
        if gm_all is not None:
            ctx.mark_non_differentiable(lin_all, glin_all, gm_all)
        else:
            ctx.mark_non_differentiable(lin_all, glin_all)
        return lin_all, glin_all, token, gm_all

    @staticmethod
    def backward(ctx, _g1, _g2, _gtoken, _g4=None):
        emb, wcat = ctx.saved_tensors
        glin_all = ctx.glin_all
        if ctx.gm_all is not None:
            # every block's modulation finish in one launch; the block gains' gradients are final from here on
            ops.mod_finish_multi(ctx.gm_all, ctx.lin_all, glin_all, ctx.table, len(ctx.den._res_blocks()))
            for b in ctx.den._res_blocks():
                gp = b.gain
                if getattr(gp, "_edm_deferred", False):
                    gp._edm_deferred = False
                    for hook in getattr(gp, "_edm_hooks", ()):
                        hook(gp)
        dw = ops.linear_wgrad(glin_all, emb)                    # (sum C, E)
        gemb = ops.linear_dgrad(glin_all, wcat) if ctx.needs_input_grad[0] else None
        gws, off = [], 0
        for b in ctx.den._res_blocks():
            C = b.embed.weight.shape[0]
            gws.append(b.embed.finish_grad(dw[off:off + C].view(1, 1, C, -1)))
            off += C
        return (gemb, None, *gws)


class _ConvInFn(torch.autograd.Function):
    """c_in * noisy, ones channel, conv_in 3x3 (networks.py:584-587); input padded to 32 channels."""

    @staticmethod
    def forward(ctx, noisy, sigma, w, den: "Denoiser"):
        wf, _, _ = den.conv_in.packs()
        xin = ops.precond_in(noisy, sigma, den.sigma_data, den.conv_in._ipad)
        ctx.den = den
        ctx.save_for_backward(xin)
        return ops.conv_igemm(xin, wf, 9)

    @staticmethod
    def backward(ctx, gy):
        (xin,) = ctx.saved_tensors
        gw = _wgrad(ctx.den.conv_in, xin, gy.contiguous(), 9)
        return None, None, gw, None


class _ConvOutFn(torch.autograd.Function):
    """conv_out * gain_out * c_out + noisy * c_skip (networks.py:602-603), fp32 NCHW result."""

    @staticmethod
    def forward(ctx, x, w, gain_out, noisy, sigma, den: "Denoiser"):
        wh = den.conv_out.packs()[2]
        D, Fraw = ops.conv_out_fwd(x, wh, gain_out, noisy, sigma, den.sigma_data, want_fraw=True)
        ctx.den = den
        ctx.save_for_backward(x, wh, gain_out, Fraw, sigma)
        return D

    @staticmethod
    def backward(ctx, gD):
        x, wh, gain_out, Fraw, sigma = ctx.saved_tensors
        # flat-arena mode: d loss / d gain_out is accumulated straight into the gradient arena (like the block gains):
        # no AccumulateGrad node ever runs for a parameter, which is also what keeps a stream capture of the step
        # independent of the stream earlier (eager) steps ran on
        gp = ctx.den.gain_out
        gdirect = gp.grad is not None and getattr(gp, "_edm_direct", False) and gp.grad.is_contiguous()
        gx, gwh, gg = ops.conv_out_bwd(x, wh, gain_out, Fraw, gD.contiguous().float(), sigma, ctx.den.sigma_data,
                                       gg_out=gp.grad if gdirect else None)
        if gdirect:
            gg = None
            for hook in getattr(gp, "_edm_hooks", ()):
                hook(gp)
        gw = ctx.den.conv_out.finish_grad(gwh.view(1, 1, *gwh.shape))
        return gx, gw, gg, None, None, None


class Denoiser(nn.Module):
    def __init__(
        self,
        in_channels: int = 3,
        out_channels: int = 3,
        encoder_block_types: tuple[str, ...] = get_encoder_blocks_types(),
        decoder_block_types: tuple[str, ...] = get_decoder_blocks_types(),
        encoder_out_channels: tuple[int, ...] = get_encoder_out_channels(),
        decoder_out_channels: tuple[int, ...] = get_decoder_out_channels(),
        skip_connections: tuple[bool, ...] = get_skip_connections(),
        dropout_rate: float = 0.0,
        sigma_data: float = 0.5,
        encoder_add_factor: float = 0.3,
        decoder_add_factor: float = 0.3,
        embedding_dim: int = 768,
        num_heads: int = 4,
    ):
        super().__init__()
        assert len(encoder_block_types) == len(encoder_out_channels), (
            f"encoder_block_types and encoder_out_channels must have the same length, got "
            f"{len(encoder_block_types)} and {len(encoder_out_channels)}")
        assert len(decoder_block_types) == len(decoder_out_channels), (
            f"decoder_block_types and decoder_out_channels must have the same length, got "
            f"{len(decoder_block_types)} and {len(decoder_out_channels)}")
        assert len(skip_connections) == len(decoder_out_channels), (
            f"skip_connections must have the same length as decoder_out_channels, got "
            f"{len(skip_connections)} and {len(decoder_out_channels)}")
        encoder_block_types, decoder_block_types = tuple(encoder_block_types), tuple(decoder_block_types)
        encoder_out_channels, decoder_out_channels = tuple(encoder_out_channels), tuple(decoder_out_channels)
        skip_connections = tuple(bool(s) for s in skip_connections)

        self.conv_in = Conv2d(in_channels + 1, encoder_out_channels[0], 3)
        self.conv_in._ipad = 32 * ((in_channels + 1 + 31) // 32)
        self.conv_in._want = ("fwd",)
        self.conv_out = Conv2d(decoder_out_channels[-1], out_channels, 1)
        self.conv_out._want = ("hat",)
        self.gain_out = nn.Parameter(torch.zeros(()))

        self.encoder_blocks = build_encoder_blocks(
            encoder_block_types, encoder_out_channels, embedding_dim=embedding_dim, dropout_rate=dropout_rate,
            add_factor=encoder_add_factor, num_heads=num_heads)
        skip_channels = get_skip_channels(encoder_out_channels, decoder_out_channels, skip_connections)
        self.decoder_blocks = build_decoder_blocks(
            decoder_block_types, decoder_out_channels, skip_channels, embedding_dim=embedding_dim,
            dropout_rate=dropout_rate, add_factor=decoder_add_factor, num_heads=num_heads)

        self.in_channels = in_channels
        self.out_channels = out_channels
        self.encoder_block_types = encoder_block_types
        self.decoder_block_types = decoder_block_types
        self.encoder_out_channels = encoder_out_channels
        self.decoder_out_channels = decoder_out_channels
        self.skip_connections = skip_connections
        self.dropout_rate = dropout_rate
        self.sigma_data = sigma_data
        self.encoder_add_factor = encoder_add_factor
        self.decoder_add_factor = decoder_add_factor
        self.embedding_dim = embedding_dim
        self.num_heads = num_heads

    def _res_blocks(self):
        return list(self.encoder_blocks) + list(self.decoder_blocks)

    def _modfin_table(self, blocks):
        """Device table of edm_mod_finish_multi ({gain ptr, d loss / d gain ptr, first column, channels} per block), or
        None unless every block gain's gradient is a view into the flat gradient arena.  Cached on the pointers."""
        gains = [b.gain for b in blocks]
        if not all(g.grad is not None and getattr(g, "_edm_direct", False) and g.grad.is_contiguous() for g in gains):
            return None
        key = tuple((g.data_ptr(), g.grad.data_ptr()) for g in gains)
        cached = getattr(self, "_modfin", None)
        if cached is None or cached[0] != key:
            rec = np.zeros(len(blocks), dtype=np.dtype([("gain", "<u8"), ("ggain", "<u8"), ("col0", "<i4"), ("C", "<i4")]))
            assert rec.dtype.itemsize == 24
            off = 0
            for k, (b, g) in enumerate(zip(blocks, gains)):
                C = b.embed.weight.shape[0]
                rec[k] = (g.data_ptr(), g.grad.data_ptr(), off, C)
                off += C
            cached = self._modfin = (key, torch.from_numpy(rec.view(np.uint8).copy()).to(gains[0].device))
        return cached[1]

    @staticmethod
    def _decoder_gates(dec, skips, dests=None):
        return _decoder_gates_impl(dec, skips, dests)

    @staticmethod
    def _silu_dest(block, nxt, x, up, down=False):
        """(out, sil) buffers for `block`'s output when its consumer `nxt` is a decoder block WITHOUT a skip, an upsample or
        a 1x1 conv in front of its first 3x3 conv: that conv reads mp_silu(input) (networks.py:313-316), which the producer's
        last kernel writes beside the output (mode 3 of the conv epilogue) instead of a k_silu_fwd launch (round 6)"""
        if not FUSE_CAT or isinstance(nxt.resample, UpSample) or nxt.cat_factor is not None:
            return None
        Bx, Hx, Wx, _ = x.shape
        if down:
            Hx, Wx = Hx // 2, Wx // 2
        Co = block.conv_3x3_2.weight.shape[0]
        return (torch.empty(Bx, Hx * up, Wx * up, Co, device=x.device, dtype=bf16),
                torch.empty(Bx, Hx * up, Wx * up, Co, device=x.device, dtype=bf16))

    def _frag_flags(self, B: int, H: int, W: int):
        """{conv module: (forward pack, dgrad pack) fragment-major?} for an input of this shape: the 3x3 convs that the
        default dispatch runs on k_conv3x3_s (the 8x8 layers at batch 128) get packs in the layout that kernel loads with
        coalesced 1-KiB instructions.  Depends on the shape only (the resolution of every block follows from the
        architecture), cached per shape."""
        cache = self.__dict__.setdefault("_fragcache", {})
        key = (B, H, W, ops.IGEMM_VERSION)      # (a forced kernel generation changes which layers the 8x8 kernel runs)
        if key not in cache:
            flags = {}
            h, w = H, W

            def mark(conv, hh, ww):
                O, I = conv.weight.shape[:2]
                npix = B * hh * ww
                f = (ops.uses_s_kernel(npix, ww, I, O), ops.uses_s_kernel(npix, ww, O, I))
                if f[0] or f[1]:
                    flags[conv] = f
            for blk in self.encoder_blocks:
                if isinstance(blk.resample, DownSample):
                    h, w = h // 2, w // 2
                mark(blk.conv_3x3_1, h, w)
                mark(blk.conv_3x3_2, h, w)
            for blk in self.decoder_blocks:
                if isinstance(blk.resample, UpSample):
                    h, w = h * 2, w * 2
                mark(blk.conv_3x3_1, h, w)
                mark(blk.conv_3x3_2, h, w)
            cache[key] = flags
        return cache[key]

    MAX_PLANS = 6       # unpinned plans of the live weights kept per network (least recently used beyond that are dropped)

    def _eff_wants(self, mods):
        """per module, the packs the current mode needs.  Training and bf16 evaluation: the module's own (_want); the
        reference-precision evaluation paths add the fp32 effective weight ("hat") and -- "f32x3", the convs the split back
        end runs -- the persistent split-bf16 pack.  The TRAINING plan's key therefore never changes when a sampling callback
        switches the evaluation precision for a while (set_eval_dtype used to grow every module's _want for good)."""
        hat = (not self.training) and self.eval_dtype != "bf16"
        split = (not self.training) and self.eval_dtype == "f32x3"
        out = []
        for m in mods:
            w = tuple(m._want)
            if hat and "hat" not in w:
                w += ("hat",)
            if split and m.weight.dim() == 4 and m.weight.shape[1] % 32 == 0 and m.weight.shape[0] % 8 == 0:
                w += ("split",)
            out.append(w)
        return tuple(out)

    def _prep_all(self, shape=None):
        """One multi-tensor launch for every weight of the U-Net (forced normalisation in training + packs).
        Plans (the persistent pack buffers) are kept per (weights, wanted packs, fragment-major set).  A plan that a captured
        step or solve was recorded with is PINNED (the graph holds its addresses) and stays; of the others, those of weights
        that no longer exist (re-allocated parameters) are dropped, and at most MAX_PLANS of the live weights are kept."""
        mods = [m for m in self.modules() if isinstance(m, _WNBase)]
        frag = self._frag_flags(*shape) if (shape is not None and FRAG_PACKS) else {}
        ptrs = tuple(m.weight.data_ptr() for m in mods)
        base = (ptrs, self._eff_wants(mods))
        key = base + (tuple((k, v) for k, v in enumerate(frag.get(m) for m in mods) if v),)
        plans = self.__dict__.setdefault("_plans", {})
        if shape is None and not self.training:
            # no shape given (the sampler's refresh before a graph replay, the fp32 evaluation route): every plan of these
            # weights and this precision is brought up to date -- a captured solve reads the packs of the plan of ITS input
            # shape (each run() is a no-op unless the master weights changed since that plan's last run)
            for k, plan in plans.items():
                if k[:2] == base and k != key:
                    plan.run(False)
        plan = plans.pop(key, None)
        if plan is None:
            plan = _PrepPlan(mods, frag, base[1], cat=[b.embed for b in self._res_blocks()])
            for k in [k for k, p in plans.items() if k[0] != ptrs and not p.pinned]:
                del plans[k]                    # packs of parameters that have been re-allocated since
            live = [k for k, p in plans.items() if not p.pinned]
            for k in live[:max(0, len(live) + 1 - self.MAX_PLANS)]:
                del plans[k]                    # (dict order = least recently used first)
        plans[key] = plan                       # most recently used last
        plan.run(self.training)
        self.__dict__["_embed_wcat"] = plan.wcat

    def _wcat(self, blocks) -> Tensor:
        """(sum C, E) fp32: the blocks' effective embed weights as one GEMM operand -- the current plan's contiguous buffer
        when the modules' packs are its row blocks (always, inside a Denoiser forward), else a concatenation"""
        whs = [b.embed.packs()[2] for b in blocks]
        w = self.__dict__.get("_embed_wcat")
        if w is not None and all(wh._base is w for wh in whs) and sum(wh.shape[0] for wh in whs) == w.shape[0]:
            return w
        return torch.cat(whs, 0)

    # ---- reference-precision evaluation (round 3): the reference samples / validates in fp32 (generate.py:39-44,
    # callbacks.py:41-49).  eval_dtype "f32" routes EVAL-mode forwards (no grad) through the exact-fp32 kernels of
    # csrc/eval_f32.hip (fp32 activations, fp32 effective weights, v_mfma_f32_32x32x2_f32); training is unaffected.
    eval_dtype = "bf16"

    def set_eval_dtype(self, dtype: str) -> "Denoiser":
        """"bf16" (default: the training path's kernels), "f32" (reference precision: exact fp32 products, ~9x slower) or
        "f32x3" (reference precision to 2^-17 per operand: split-bf16 on the bf16 kernels, see _conv_f32)"""
        dtype = {"float32": "f32", "fp32": "f32", "bfloat16": "bf16", "split": "f32x3"}.get(str(dtype).replace("torch.", ""), str(dtype))
        if dtype not in ("bf16", "f32", "f32x3"):
            raise ValueError("Denoiser.set_eval_dtype: 'bf16', 'f32' (exact fp32 products) or 'f32x3' (split-bf16, fp32-accurate)")
        # (the evaluation plans keep the fp32 effective weights / split packs: _eff_wants -- the modules' own _want, and with it
        # the training plan, is left alone)
        self.eval_dtype = dtype
        return self

    def _forward_f32(self, noisy: Tensor, sig: Tensor, emb: Tensor) -> Tensor:
        blocks = self._res_blocks()
        lin_all = ops.linear_fwd(emb, self._wcat(blocks))
        lins, off = {}, 0
        for b in blocks:
            C = b.embed.weight.shape[0]
            lins[b] = lin_all[:, off:off + C]
            off += C
        cp = 8 * ((self.in_channels + 1 + 7) // 8)
        x = ops.f32_conv(ops.f32_precond_in(noisy, sig, self.sigma_data, cp), self.conv_in.packs()[2], 9)
        skips = [x]
        dec = list(zip(self.decoder_blocks, self.skip_connections))
        split = bool(_SPLIT_EVAL[0]) and F32_FUSE_OUT

        def want_of(block, nxt, n_skip, hw, skip_c):
            """what `block`'s last conv writes for its consumer `nxt` (round 6: the eval-shaped forward of the split back
            end): the input halves of nxt's concatenated operands / fp32 + mp_silu pairs / plain fp32"""
            if not split or nxt is None or isinstance(nxt.resample, UpSample) or not _out_conv_split(block):
                return None
            if not _split_ok(nxt.conv_3x3_1):
                return None
            if n_skip:
                if not (isinstance(nxt.conv_1x1, Conv2d) and _split_ok(nxt.conv_1x1)):
                    return None
                Ct = block.conv_3x3_2.weight.shape[0] + skip_c
                shape = (noisy.shape[0], hw[0], hw[1], 2 * Ct)
                return ("dest", torch.empty(shape, device=noisy.device, dtype=bf16),
                        torch.empty(shape, device=noisy.device, dtype=bf16))
            return None if isinstance(nxt.conv_1x1, Conv2d) else "silu"

        enc = list(self.encoder_blocks)
        h, w = x.shape[1], x.shape[2]
        for i, block in enumerate(enc):
            if isinstance(block.resample, DownSample):
                h, w = h // 2, w // 2
            # (only the LAST encoder output has a consumer besides the skip stack: the first decoder block, which has no skip)
            want = want_of(block, dec[0][0], dec[0][1], (h, w), 0) if (i + 1 == len(enc) and dec and not dec[0][1]) else None
            x = block.forward_f32(x, lins[block], want)
            skips.append(x[0] if isinstance(x, tuple) else x)
        for i, (block, has_skip) in enumerate(dec):
            skip = skips.pop() if has_skip else None
            if isinstance(block.resample, UpSample):
                h, w = 2 * h, 2 * w
            nxt, n_skip = dec[i + 1] if i + 1 < len(dec) else (None, False)
            want = want_of(block, nxt, n_skip, (h, w), skips[-1].shape[-1] if (n_skip and skips) else 0)
            x = block.forward_f32(x, lins[block], skip, want)
        return ops.f32_conv_out(x, self.conv_out.packs()[2], self.gain_out.detach(), noisy, sig, self.sigma_data)

    def forward(self, noisy_image: Tensor, sigma: Tensor, embedding: Tensor):
        prev, _IN_DENOISER[0] = _IN_DENOISER[0], True
        try:
            return self._forward(noisy_image, sigma, embedding)
        finally:
            _IN_DENOISER[0] = prev

    def _forward(self, noisy_image: Tensor, sigma: Tensor, embedding: Tensor):
        if not noisy_image.is_cuda:
            raise RuntimeError("tinyedm_amd.Denoiser: inputs must be GPU tensors (there is no CPU path)")
        if self.eval_dtype != "bf16" and not self.training:
            if torch.is_grad_enabled() and (noisy_image.requires_grad or embedding.requires_grad):
                raise RuntimeError("tinyedm_amd.Denoiser: the fp32 evaluation path is forward-only (use torch.no_grad())")
            with torch.no_grad():
                self._prep_all()
                noisy = noisy_image.float().contiguous()
                _SPLIT_EVAL[0] = self.eval_dtype == "f32x3"
                try:
                    D = self._forward_f32(noisy, sigma.detach().float().flatten().contiguous(),
                                          _emb32(embedding.detach(), noisy.shape[0]))
                finally:
                    _SPLIT_EVAL[0] = False
            return D.to(noisy_image.dtype)
        reset_backward_state()
        if torch.is_grad_enabled():      # eval-mode forwards with autograd on (fine-tuning, parity runs) write gradients too
            global FORWARD_EPOCH
            FORWARD_EPOCH += 1
        with torch.no_grad():
            self._prep_all((noisy_image.shape[0], noisy_image.shape[2], noisy_image.shape[3]))
        noisy = noisy_image.float().contiguous()
        B = noisy.shape[0]
        sig = sigma.detach().float().flatten().contiguous()
        emb = _emb32(embedding, B)

        blocks = self._res_blocks()
        lin_all, glin_all, token, gm_all = _EmbedAllFn.apply(emb, self, *[b.embed.weight for b in blocks])
        lins, off = {}, 0
        for k, b in enumerate(blocks):
            C = b.embed.weight.shape[0]
            # the ordering edge goes to the FIRST block only: every other block is downstream of its output, so its
            # backward -- and with it _EmbedAllFn.backward -- runs after all of them (one edge instead of 21 that
            # autograd would sum with 20 tiny adds)
            lins[b] = (lin_all[:, off:off + C], glin_all[:, off:off + C], token if k == 0 else None,
                       None if gm_all is None else gm_all[:, off:off + C])
            off += C

        x = _ConvInFn.apply(noisy, sig, self.conv_in.weight, self)
        skips = []
        dec = list(zip(self.decoder_blocks, self.skip_connections))
        enc = list(self.encoder_blocks)
        for k, block in enumerate(enc):
            # the skip is taken from the block's alias of its own input: the decoder's skip gradient then lands in the
            # block's backward and is summed by a kernel that runs anyway, not by an autograd add
            dest = None
            if k + 1 == len(enc) and dec and not dec[0][1]:     # the last encoder output feeds a decoder block without a skip
                dest = self._silu_dest(block, dec[0][0], x, 1, down=isinstance(block.resample, DownSample))
            x, x_in = block(_tag(x), None, _lin=lins[block], _alias=True, _dest=dest)
            skips.append(x_in)
        skips.append(x)
        # the next block concatenates a block's output with a skip: that block's last kernel writes its output and mp_silu
        # of it into the left halves of those operands (no concat copy); the buffers exist before the decoder runs
        dests = _decoder_dests_impl(dec, skips, x) if FUSE_CAT and SKIP_GATE_FUSED else {}
        gates = self._decoder_gates(dec, skips, dests) if SG_MULTI else {}
        for i, (block, has_skip) in enumerate(dec):
            skip = skips.pop() if has_skip else None
            dest = dests.get(i)
            if dest is not None:
                pass
            elif i + 1 < len(dec):
                dest = self._silu_dest(block, dec[i + 1][0], x, 2 if isinstance(block.resample, UpSample) else 1)
            xin = _tag(x)
            cat_pre = getattr(x, "_edm_cat", None)
            if cat_pre is not None:
                xin._edm_cat = cat_pre
            x = block(xin, None, _tag(skip) if has_skip else None, _lin=lins[block], _dest=dest, _gate=gates.get(block))
        D = _ConvOutFn.apply(x, self.conv_out.weight, self.gain_out, noisy, sig, self)
        if self.training:
            rng.step += 1
        return D.to(noisy_image.dtype)


class _SgbPass:
    """The decoder gates of ONE forward pass whose backward is deferred (SG_BWD_MULTI): each block's backward queues
    (gcs, skip, weights, gate, z1, its gskip placeholder) and counts itself in; the last one flushes -- one
    skip_gate_bwd_multi per channel count, one skip_half_bwd_multi over every placeholder -- before any consumer of a
    placeholder (the encoder's backward; autograd's sum for the last encoder output) can run."""

    def __init__(self, expected: int):
        self.expected, self.arrived, self.pending = expected, 0, []

    def flush(self):
        items, self.pending = self.pending, []
        self.arrived = 0            # (a second backward over a retained graph counts its gates in again)
        if not items:
            return
        by_c = {}
        for it in items:
            by_c.setdefault(it[1].shape[-1], []).append(it)
        halves = []
        for group in by_c.values():
            for k0 in range(0, len(group), 32):
                part = group[k0:k0 + 32]
                res = ops.skip_gate_bwd_multi([(gcs, 0, skip, w1h, w2h, gate, z1) + ((gskip,) if SG_HALVES else ())
                                               for gcs, skip, w1h, w2h, gate, z1, gskip, _, _ in part])
                for (gcs, skip, w1h, w2h, gate, z1, gskip, sl, mean), (gmean, ws) in zip(part, res):
                    if not SG_HALVES:
                        halves.append((gcs, gate, gmean, gskip))
                    _sg_pending.setdefault(sl.layer1.weight.device.index, []).append((sl, ws, mean, w1h.shape[0]))
        for k0 in range(0, len(halves), 32):
            ops.skip_half_bwd_multi(halves[k0:k0 + 32])


def _decoder_dests_impl(dec, skips, x):
    """{decoder index i: (cat, sil) buffers of block i + 1} where block i + 1 concatenates a U-Net skip in its autograd node
    (FUSE_CAT): block i's last kernel writes its output and mp_silu of it into their LEFT halves (no concat copy).  Allocated
    before the decoder runs (round 6) so that the gate launch behind the encoder can fill the RIGHT halves too."""
    out = {}
    st = list(skips)
    h, w = x.shape[1], x.shape[2]
    for i, (block, has_skip) in enumerate(dec):
        if has_skip:
            st.pop()
        if isinstance(block.resample, UpSample):
            h, w = 2 * h, 2 * w
        if i + 1 < len(dec) and dec[i + 1][1]:
            nxt = dec[i + 1][0]
            if not isinstance(nxt.resample, UpSample) and isinstance(nxt.conv_1x1, Conv2d):
                Ct = block.conv_3x3_2.weight.shape[0] + st[-1].shape[-1]
                out[i] = (torch.empty(x.shape[0], h, w, Ct, device=x.device, dtype=bf16),
                          torch.empty(x.shape[0], h, w, Ct, device=x.device, dtype=bf16))
    return out


def _decoder_gates_impl(dec, skips, dests=None):
    """{decoder block: (mean, gate, z1, w1h, w2h, backward token, halves written)} for every block whose ScaleLong gate lives
    in its autograd node: the gates depend on the skip tensors and two small weights only, so all of them are computed HERE,
    behind the encoder, by one launch per channel count (ops.skip_gate_fwd_multi) instead of one half-empty launch per
    decoder block -- and the workgroup that reduced a sample also writes its gated skip (and mp_silu of it) into the right
    halves of the block's (cat, sil) buffers (`dests[i - 1]`), the work of the block's k_skip_half_fwd launch."""
    st = list(skips)
    todo = {}
    for i, (block, has_skip) in enumerate(dec):
        if not has_skip:
            continue
        skip = st.pop()
        if not (isinstance(block, DecoderBlock) and block._gate_in_block() and skip.dim() == 4 and skip.dtype == bf16):
            continue
        sl = block.cat_factor
        pre = dests.get(i - 1) if dests and SG_HALVES else None
        if pre is not None and not (pre[0].shape[:3] == skip.shape[:3] and pre[0].shape[-1] > skip.shape[-1]):
            pre = None
        todo.setdefault(skip.shape[-1], []).append((block, skip, sl.layer1.packs()[2], sl.layer2.packs()[2], pre))
    out = {}
    tok = _SgbPass(sum(len(v) for v in todo.values())) if torch.is_grad_enabled() else None
    for items in todo.values():
        for k0 in range(0, len(items), 32):
            part = items[k0:k0 + 32]
            res = ops.skip_gate_fwd_multi([(skip, w1h, w2h) + (pre if pre is not None else ()) for _, skip, w1h, w2h, pre in part])
            for (block, _, w1h, w2h, pre), (mean, gate, z1) in zip(part, res):
                out[block] = (mean, gate, z1, w1h, w2h, tok, None if pre is None else pre[0].data_ptr())
    return out


class DenoiserWrapper(nn.Module):
    """networks.py:608-646: generic EDM preconditioning around an arbitrary net (unused by the
    shipped configs; elementwise torch math, kept for API completeness)."""

    def __init__(self, net: nn.Module, sigma_data: float):
        super().__init__()
        self.net = net
        self._sigma_data = sigma_data

    @property
    def sigma_data(self) -> float:
        return self._sigma_data

    def forward(self, noisy_image: Tensor, sigma: Tensor, embedding: Tensor | None = None) -> Tensor:
        sigma = sigma.view(-1, 1, 1, 1)
        sd = self.sigma_data
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
        c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
        c_noise = sigma.log() / 4
        F_ = self.net(c_in * noisy_image, c_noise.flatten(), embedding)
        return c_skip * noisy_image + c_out * F_

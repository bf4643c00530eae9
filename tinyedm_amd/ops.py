"""Tensor-level wrappers around the C-ABI kernels: validate shapes/dtypes on the host (a kernel
that faults can take the whole GPU node down), then pass raw pointers + the current stream.

Layout conventions: activations are NHWC bf16 tensors of shape (B, H, W, C) (contiguous);
"rows" means B*H*W.  Per-(sample,channel) quantities are fp32 (B, C).
"""
import ctypes
import os
from typing import Optional

import torch

from . import _lib

bf16, f32 = torch.bfloat16, torch.float32

# When set to a dict, the MFMA kernels below record (start_event, end_event, algorithmic FLOPs, algorithmic
# HBM bytes) per launch on the launch stream (bench.py's roofline leg).  None = no instrumentation.
PROFILE = None


class _prof:
    def __init__(self, name, flops, nbytes):
        self.name, self.flops, self.nbytes = name, flops, nbytes

    def __enter__(self):
        if PROFILE is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if PROFILE is not None:
            self.e.record()
            PROFILE.setdefault(self.name, []).append((self.s, self.e, self.flops, self.nbytes))
        return False


def _stream():
    s = torch.cuda.current_stream()
    if s.device_index not in _lib._inited_devices:
        _lib.init_device(s.device_index)
    return ctypes.c_void_p(s.cuda_stream)


_SIDE_STREAMS = {}


def side_stream(device):
    """Per-device auxiliary HIP stream: weight-gradient kernels (off the backward's critical path) run here, under
    the HBM-bound elementwise kernels of the main stream.  Created on first use."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    s = _SIDE_STREAMS.get(idx)
    if s is None:
        s = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return s


def side_streams():
    return list(_SIDE_STREAMS.values())


class _ZeroPool:
    """Zero-initialised fp32 scratch for the atomically-accumulated outputs (reductions, small weight grads).

    One training step needs ~110 such tensors; a `torch.zeros` each is a fill launch each.  The pool zeroes a
    16 MiB chunk with ONE fill and hands out disjoint 256-byte-aligned pieces; a chunk is never reused (pieces keep
    its storage alive), so lifetime rules are those of ordinary tensors.  Pieces are fresh tensors on the shared
    storage, not views, so they do not share autograd version counters.  A captured graph must re-zero its
    accumulators on every replay: `capture_begin()` / `capture_end()` drop the current chunk, so every chunk used
    inside a capture is allocated -- and its ONE fill recorded -- inside that capture."""
    CHUNK = 1 << 22  # floats

    def __init__(self):
        self.buf, self.off = None, 0

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) // 64 * 64
        if n_al > self.CHUNK // 4:
            return torch.zeros(shape, device=device, dtype=f32)
        if self.buf is None or self.off + n_al > self.CHUNK or self.buf.device != device:
            self.buf, self.off = torch.zeros(self.CHUNK, device=device, dtype=f32), 0
        out = torch.empty(0, device=device, dtype=f32).set_(self.buf.untyped_storage(), self.off, tuple(shape))
        self.off += n_al
        return out


_zero_pool = _ZeroPool()


def zeros_f32(shape, device):
    return _zero_pool.take(tuple(shape), device)


def capture_begin():
    """call right before a stream capture starts (and capture_end() right after it ends)"""
    _zero_pool.buf = None
    _tables.begin()


def note_capture_pin(obj) -> bool:
    """`obj` (a weight-prep plan) is read by the graph being captured: it gets one pin (`obj.pins += 1`) that
    release_capture() of this capture's token takes back.  False when the capture runs without capture_begin() /
    capture_end() (nobody could release the pin: the caller pins for good)."""
    if not _tables.deferring:
        return False
    if not any(o is obj for o in _tables.pins):
        obj.pins += 1
        _tables.pins.append(obj)
    return True


def capture_end():
    """-> token of the launch-table slots the capture took: hand it to release_capture() when the graph is destroyed"""
    _zero_pool.buf = None
    return _tables.end()


def release_capture(token):
    """the graph captured between the capture_begin() / capture_end() that returned `token` no longer exists: its
    launch-table slots may be reused by later captures (a long-lived process that re-captures does not grow)"""
    _tables.release(token)


class _LaunchTables:
    """Staging for the launch tables of the grouped kernels (edm_wgrad3_group, edm_conv_wgrad_1x1_group,
    edm_wgrad_finish_multi; include/tinyedm_hip.h "LAUNCH TABLES"): the C side writes a table into host memory and it is
    copied into device memory that the kernels read -- nothing larger than a few hundred bytes travels as a by-value
    kernel argument.
    * Eager steps: (pinned host, device) slot pairs from a ring, one stream-ordered copy per table issued by the C side; a
      slot is reused only after the event recorded behind its last use has completed.
    * Between ops.capture_begin() and ops.capture_end() (graph.CapturedTrainStep): the table of a captured launch never
      changes between replays, so it is uploaded ONCE -- the C side only fills a host buffer (defer_upload), the device
      slot comes from a pool allocated up front, and capture_end() copies all of them before the first replay can run
      (no memcpy nodes in the graph: nine 4-us copies per replayed step otherwise).  capture_end() returns the slots as a
      token; release_capture(token) puts them on a free list once the graph is gone.
    * A capture made WITHOUT those hooks still works: the copy becomes a memcpy node that every replay executes again from
      the same host address, so those slots come from a bump-allocated pinned pool with ITS OWN index (`bare_i`: deferred
      captures do not eat into it), never reused or freed (allocated up front: pinning memory is not a capturable call).
    Per-device state is created by the first EAGER launch (or capture_begin()): creating it under capture would pin memory
    and hipMalloc inside the capture, so that raises instead."""
    RING, POOL = 64, 512

    def __init__(self):
        self.dev = {}
        self.deferring = False
        self.pending = []           # (device tensor, host tensor) pairs to upload at capture_end()
        self.taken = []             # (device index, slot) of the running deferred capture
        self.pins = []              # objects pinned by the running deferred capture (note_capture_pin)

    def _state(self, device):
        key = torch.device(device).index
        key = torch.cuda.current_device() if key is None else key
        st = self.dev.get(key)
        if st is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("tinyedm_amd: the launch-table pools of this device do not exist yet and cannot be "
                                   "created under stream capture (pinned + device allocations): run one eager step first, "
                                   "or bracket the capture with ops.capture_begin() / capture_end()")
            nb = max(int(_lib.call("edm_wgrad3_table_bytes")), int(_lib.call("edm_conv_wgrad_1x1_group_table_bytes")),
                     int(_lib.call("edm_wgrad_finish_multi_table_bytes")), int(_lib.call("edm_skip_gate_wgrad_multi_table_bytes")),
                     int(_lib.call("edm_skip_gate_fwd_multi_table_bytes")), int(_lib.call("edm_skip_gate_bwd_multi_table_bytes")))
            nb = (nb + 255) // 256 * 256
            d = torch.device("cuda", key)
            st = self.dev[key] = {
                "key": key, "nb": nb, "i": 0, "events": [None] * self.RING,
                "host": torch.empty(self.RING, nb, dtype=torch.uint8).pin_memory(),
                "devb": torch.empty(self.RING, nb, dtype=torch.uint8, device=d),
                "pool": torch.empty(self.POOL, nb, dtype=torch.uint8).pin_memory(), "bare_i": 0,
                "bare_devb": torch.empty(self.POOL, nb, dtype=torch.uint8, device=d),
                "pool_i": 0, "free": [],
                # device slots for captured launches, allocated HERE (eagerly, ordinary memory): a buffer allocated inside a
                # capture belongs to the graph's pool and a write to it from outside the graph (the deferred upload) did
                # not reach what the replayed kernels read -- the first replay hung on a garbage table
                "pool_devb": [torch.empty(self.POOL, nb, dtype=torch.uint8, device=d)]}
        return st

    def _reserve(self, st, want):
        """make sure `want` more captured launches find a device slot -- called OUTSIDE capture (capture_begin): a process
        that holds many live graphs grows the pool instead of running out"""
        while len(st["free"]) + len(st["pool_devb"]) * self.POOL - st["pool_i"] < want:
            st["pool_devb"].append(torch.empty(self.POOL, st["nb"], dtype=torch.uint8, device=st["devb"].device))

    def take(self, device):
        """-> (host pointer, device pointer, defer_upload, release()): call release() right after the launch"""
        st = self._state(device)
        if torch.cuda.is_current_stream_capturing():
            if self.deferring:
                if st["free"]:
                    k = st["free"].pop()
                else:
                    k = st["pool_i"]
                    if k >= len(st["pool_devb"]) * self.POOL:
                        raise RuntimeError("tinyedm_amd: out of launch-table slots for this capture (more than the 64 "
                                           "reserved by ops.capture_begin(): ops._LaunchTables._reserve)")
                    st["pool_i"] = k + 1
                self.taken.append((st["key"], k))
                dev = st["pool_devb"][k // self.POOL][k % self.POOL]
                host = torch.empty(st["nb"], dtype=torch.uint8)
                self.pending.append((dev, host))
                return host.data_ptr(), dev.data_ptr(), 1, (lambda: None)
            k = st["bare_i"]
            if k >= self.POOL:
                raise RuntimeError("tinyedm_amd: out of launch-table slots for bare captures (captures made without "
                                   f"ops.capture_begin() / capture_end() share {self.POOL} pinned slots per process)")
            st["bare_i"] = k + 1                    # never reused: the graph's memcpy node keeps reading the pinned slot
            return st["pool"][k].data_ptr(), st["bare_devb"][k].data_ptr(), 0, (lambda: None)
        k = st["i"] % self.RING
        st["i"] += 1
        if st["events"][k] is not None:
            st["events"][k].synchronize()

        def release(st=st, k=k):
            ev = torch.cuda.Event()
            ev.record()
            st["events"][k] = ev
        return st["host"][k].data_ptr(), st["devb"][k].data_ptr(), 0, release

    def begin(self):
        self.deferring, self.pending, self.taken, self.pins = True, [], [], []
        if torch.cuda.is_available():
            self._reserve(self._state(torch.cuda.current_device()), 64)     # (a captured training step takes 12)

    def end(self):
        """upload the tables of the capture that just ended (their launches have only been recorded so far)"""
        pend, self.pending, self.deferring = self.pending, [], False
        taken, self.taken = tuple(self.taken), []
        pins, self.pins = tuple(self.pins), []
        for dev, host in pend:
            dev.copy_(host)
        if pend:
            torch.cuda.synchronize(pend[0][0].device)
        return _CaptureToken(taken, pins)

    def release(self, token):
        if token is None or token.released:
            return
        token.released = True
        for key, k in token.slots:
            st = self.dev.get(key)
            if st is not None and k not in st["free"]:
                st["free"].append(k)
        for obj in token.pins:          # the graph that read these plans' buffers is gone (they used to stay
            obj.pins = max(0, obj.pins - 1)   # pinned -- and their packs alive -- for the life of the Denoiser)


class _CaptureToken:
    """what a captured graph holds on to: launch-table slots and pinned weight-prep plans (ops.release_capture frees both)"""
    __slots__ = ("slots", "pins", "released")

    def __init__(self, slots, pins):
        self.slots, self.pins, self.released = slots, pins, False

    def __iter__(self):             # (round-4/5 callers iterated the token as its slot list)
        return iter(self.slots)

    def __len__(self):
        return len(self.slots)


_tables = _LaunchTables()


class GraphCorruptionError(RuntimeError):
    """raised by check_health(): a training step or a sampler solve produced non-finite values"""


_HEALTH = {}


def health(device):
    """Per-device sticky 32-bit health word.  The optimizer kernel ORs bit 0 into it when a non-finite gradient or
    weight passes through a step, the Heun update kernels bit 1 when the sampler state goes non-finite.  It is the
    in-graph sentinel of the hipGraph paths: a replay that ran with corrupted kernel arguments (tinyedm_amd/_runtime_env.py)
    leaves the bit behind, and check_health() turns it into an exception at the next host read point."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    h = _HEALTH.get(idx)
    if h is None:
        h = _HEALTH[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    return h


def check_health(device, what: str):
    """Read (one host sync) and clear the health word of `device`; raise GraphCorruptionError if a bit is set."""
    h = health(device)
    v = int(h.item())
    if v:
        h.zero_()
        bits = [n for b, n in ((1, "non-finite gradient/weight in an optimizer step"), (2, "non-finite sampler state")) if v & b]
        raise GraphCorruptionError(f"{what}: {' and '.join(bits)} (health word {v:#x}).  The values computed on the GPU are "
                                   "garbage: a diverged run, NaN inputs, or a corrupted hipGraph replay "
                                   "(see tinyedm_amd/_runtime_env.py); nothing after this point can be trusted")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t: torch.Tensor, dtype, name: str, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/HIP tensor -- tinyedm_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _nhwc(t, name):
    _chk(t, bf16, name)
    if t.dim() != 4:
        raise ValueError(f"{name}: expected (B,H,W,C)")
    return t.shape


# ------------------------------------------------------------------ elementwise
def pixelnorm_silu_fwd(x):
    B, H, W, C = _nhwc(x, "x")
    xn, a = torch.empty_like(x), torch.empty_like(x)
    d = torch.empty(B * H * W, device=x.device, dtype=f32)
    _lib.call("edm_pixelnorm_silu_fwd", _p(x), _p(xn), _p(a), _p(d), B * H * W, C, _stream())
    return xn, a, d


def pixelnorm_silu_bwd(xn, d, gxn, gxn_scale, ga, gadd=None):
    """gadd (optional): a gradient of the same tensor from another consumer (the U-Net skip), added in the same pass"""
    B, H, W, C = _nhwc(xn, "xn")
    _chk(d, f32, "d", (B * H * W,))
    if gxn is not None:
        _chk(gxn, bf16, "gxn", xn.shape)
    if ga is not None:
        _chk(ga, bf16, "ga", xn.shape)
    if gadd is not None:
        _chk(gadd, bf16, "gadd", xn.shape)
    gx = torch.empty_like(xn)
    _lib.call("edm_pixelnorm_silu_bwd", _p(xn), _p(d), _p(gxn), float(gxn_scale), _p(ga), _p(gadd), _p(gx), B * H * W, C,
              _stream())
    return gx


def pool_pixelnorm_silu_fwd(x):
    """(xn, a, d) of the 2x2-average-pooled x, the pooled tensor never written (== pixelnorm_silu_fwd(pool2(x)), bit for bit)"""
    B, H, W, C = _nhwc(x, "x")
    if H % 2 or W % 2 or C > 1024:
        raise ValueError("pool_pixelnorm_silu_fwd: H, W even and C <= 1024 required")
    xn = torch.empty(B, H // 2, W // 2, C, device=x.device, dtype=bf16)
    a = torch.empty_like(xn)
    d = torch.empty(B * (H // 2) * (W // 2), device=x.device, dtype=f32)
    _lib.call("edm_pool_pixelnorm_silu_fwd", _p(x), _p(xn), _p(a), _p(d), B, H // 2, W // 2, C, _stream())
    return xn, a, d


def pool_pixelnorm_silu_bwd(xn, d, gxn, gxn_scale, ga, gadd=None):
    """backward of pool_pixelnorm_silu_fwd: the gradient w.r.t. the tensor BEFORE the pool, (B, 2H, 2W, C); gadd (optional, that
    shape): a gradient of the same tensor from another consumer (the U-Net skip) (== up2(pixelnorm_silu_bwd(...), 0.25, add=gadd))"""
    B, H, W, C = _nhwc(xn, "xn")
    _chk(d, f32, "d", (B * H * W,))
    if gxn is not None:
        _chk(gxn, bf16, "gxn", xn.shape)
    if ga is not None:
        _chk(ga, bf16, "ga", xn.shape)
    if gadd is not None:
        _chk(gadd, bf16, "gadd", (B, 2 * H, 2 * W, C))
    gx = torch.empty(B, 2 * H, 2 * W, C, device=xn.device, dtype=bf16)
    _lib.call("edm_pool_pixelnorm_silu_bwd", _p(xn), _p(d), _p(gxn), float(gxn_scale), _p(ga), _p(gadd), _p(gx), B, H, W, C,
              _stream())
    return gx


def up2_silu(x):
    """(y, a): y = nearest-exact x2 upsample of x, a = mp_silu(y), one pass"""
    B, H, W, C = _nhwc(x, "x")
    y = torch.empty(B, 2 * H, 2 * W, C, device=x.device, dtype=bf16)
    a = torch.empty_like(y)
    _lib.call("edm_up2_silu", _p(x), _p(y), _p(a), B, 2 * H, 2 * W, C, _stream())
    return y, a


def silu_fwd(x):
    _chk(x, bf16, "x")
    a = torch.empty_like(x)
    _lib.call("edm_silu_fwd", _p(x), _p(a), x.numel(), _stream())
    return a


def silu_bwd(x, ga, gextra=None, extra_scale=1.0):
    _chk(x, bf16, "x")
    _chk(ga, bf16, "ga", x.shape)
    if gextra is not None:
        _chk(gextra, bf16, "gextra", x.shape)
    gx = torch.empty_like(x)
    _lib.call("edm_silu_bwd", _p(x), _p(ga), _p(gextra), float(extra_scale), _p(gx), x.numel(), _stream())
    return gx


def axpby(a, alpha, b=None, beta=0.0):
    _chk(a, bf16, "a")
    if b is not None:
        _chk(b, bf16, "b", a.shape)
    out = torch.empty_like(a)
    _lib.call("edm_axpby", _p(a), float(alpha), _p(b), float(beta), _p(out), a.numel(), _stream())
    return out


def _lin_view(lin, B, C, name):
    """(B, C) fp32 view whose rows may be strided (a column slice of the batched embed-linear output)."""
    if not lin.is_cuda or lin.dtype != f32 or lin.dim() != 2 or tuple(lin.shape) != (B, C) or lin.stride(1) != 1:
        raise ValueError(f"{name}: expected a (B={B}, C={C}) fp32 GPU tensor with unit column stride")
    if lin.stride(0) < C:
        raise ValueError(f"{name}: row stride {lin.stride(0)} < C={C}")
    return lin.stride(0)


def _dyn(dyn):
    """device edm_step_params record (48 bytes) or None"""
    if dyn is None:
        return None
    if not dyn.is_cuda or dyn.numel() * dyn.element_size() < 48 or not dyn.is_contiguous():
        raise ValueError("dyn: expected a contiguous device buffer of >= 48 bytes (edm_step_params)")
    return ctypes.c_void_p(dyn.data_ptr())


def mod_silu_drop_fwd(r, lin, gain, pdrop, seed, sub, step, dyn=None):
    B, H, W, C = _nhwc(r, "r")
    ls = _lin_view(lin, B, C, "lin")
    _chk(gain, f32, "gain")
    a = torch.empty_like(r)
    _lib.call("edm_mod_silu_drop_fwd", _p(r), _p(lin), ls, _p(gain), _p(a), B, H * W, C, float(pdrop), int(seed), int(sub),
              int(step), _dyn(dyn), _stream())
    return a


def mod_silu_drop_bwd(r, lin, gain, ga, pdrop, seed, sub, step, glin_out=None, ggain_out=None, dyn=None, gm_out=None):
    """glin_out: optional (B, C) strided fp32 view to receive d loss / d lin (else a fresh tensor);
    ggain_out: optional 0-dim fp32 tensor that d loss / d gain is ACCUMULATED into (else a fresh zero scalar).
    gm_out: a zero-filled (B, C) fp32 view with unit column stride (a column slice of the buffer all blocks share): the raw
    modulation gradient is accumulated there and NOT finished -- returns (gr, None, None); one mod_finish_multi over the
    shared buffer finishes every block (as conv3x3_modbwd(gm_out=...))."""
    B, H, W, C = _nhwc(r, "r")
    ls = _lin_view(lin, B, C, "lin")
    _chk(ga, bf16, "ga", r.shape)
    gr = torch.empty_like(r)
    if gm_out is not None:
        gms = _lin_view(gm_out, B, C, "gm_out")
        _chk(gain, f32, "gain")
        _lib.call("edm_mod_silu_drop_bwd_raw", _p(r), _p(lin), ls, _p(gain), _p(ga), _p(gr), _p(gm_out), gms, B, H * W, C,
                  float(pdrop), int(seed), int(sub), int(step), _dyn(dyn), _stream())
        return gr, None, None
    gm = zeros_f32((B, C), r.device)
    glin = torch.empty(B, C, device=r.device, dtype=f32) if glin_out is None else glin_out
    gs = _lin_view(glin, B, C, "glin")
    ggain = zeros_f32((), r.device) if ggain_out is None else _chk(ggain_out, f32, "ggain_out", ())
    _lib.call("edm_mod_silu_drop_bwd", _p(r), _p(lin), ls, _p(gain), _p(ga), _p(gr), _p(gm), _p(glin), gs, _p(ggain), B,
              H * W, C, float(pdrop), int(seed), int(sub), int(step), _dyn(dyn), _stream())
    return gr, glin, ggain


def dropout_mask(n, pdrop, seed, sub, step, device):
    m = torch.empty(n, device=device, dtype=torch.uint8)
    _lib.call("edm_dropout_mask", _p(m), n, float(pdrop), int(seed), int(sub), int(step), _stream())
    return m


def pool2(x, scale=0.25):
    B, H, W, C = _nhwc(x, "x")
    if H % 2 or W % 2:
        raise ValueError("pool2: H and W must be even")
    y = torch.empty(B, H // 2, W // 2, C, device=x.device, dtype=bf16)
    _lib.call("edm_pool2", _p(x), _p(y), B, H // 2, W // 2, C, float(scale), _stream())
    return y


def up2(x, scale=1.0, add=None):
    """y = scale * nearest-exact x2 upsample of x (+ add, a bf16 tensor of the output shape)"""
    B, H, W, C = _nhwc(x, "x")
    if add is not None:
        _chk(add, bf16, "add", (B, 2 * H, 2 * W, C))
    y = torch.empty(B, 2 * H, 2 * W, C, device=x.device, dtype=bf16)
    _lib.call("edm_up2", _p(x), _p(add), _p(y), B, 2 * H, 2 * W, C, float(scale), _stream())
    return y


def reduce_hw(x, C=None, c_off=0, y=None, scale=1.0):
    """out[b,c] = scale * sum_hw x[b,hw,c_off+c] (* y[b,hw,c])."""
    B, H, W, Cx = _nhwc(x, "x")
    C = Cx - c_off if C is None else C
    if c_off % 8 or c_off + C > Cx:
        raise ValueError("reduce_hw: bad channel slice")
    out = torch.empty(B, C, device=x.device, dtype=f32)      # written (not accumulated) in a fixed summation order
    ys = 0
    if y is not None:
        By, Hy, Wy, Cy = _nhwc(y, "y")
        if (By, Hy, Wy) != (B, H, W) or Cy < C:
            raise ValueError("reduce_hw: y shape mismatch")
        ys = Cy
    xp = ctypes.c_void_p(x.data_ptr() + 2 * c_off)
    _lib.call("edm_reduce_hw", xp, Cx, _p(y), ys, _p(out), B, H * W, C, float(scale), _stream())
    return out


def scalelong_fwd(mean, w1h, w2h):
    B, C = mean.shape
    R = w1h.shape[0]
    _chk(mean, f32, "mean")
    _chk(w1h, f32, "w1h", (R, C + 1))
    _chk(w2h, f32, "w2h", (C, R))
    gate = torch.empty(B, C, device=mean.device, dtype=f32)
    z1 = torch.empty(B, R, device=mean.device, dtype=f32)
    _lib.call("edm_scalelong_fwd", _p(mean), _p(w1h), _p(w2h), _p(gate), _p(z1), B, C, R, _stream())
    return gate, z1


def scalelong_bwd(mean, w1h, w2h, gate, z1, ggate):
    B, C = mean.shape
    R = w1h.shape[0]
    _chk(ggate, f32, "ggate", (B, C))
    gmean = torch.empty(B, C, device=mean.device, dtype=f32)
    gw1 = zeros_f32(w1h.shape, w1h.device)
    gw2 = zeros_f32(w2h.shape, w2h.device)
    _lib.call("edm_scalelong_bwd", _p(mean), _p(w1h), _p(w2h), _p(gate), _p(z1), _p(ggate), _p(gmean), _p(gw1), _p(gw2),
              B, C, R, _stream())
    return gmean, gw1, gw2


def skip_gate_fwd(skip, w1h, w2h):
    """ScaleLong gate of a skip tensor in one launch: (mean (B,C), gate (B,C), z1 (B,R)) == reduce_hw(skip)/HW followed by
    scalelong_fwd."""
    B, H, W, C = _nhwc(skip, "skip")
    R = w1h.shape[0]
    _chk(w1h, f32, "w1h", (R, C + 1))
    _chk(w2h, f32, "w2h", (C, R))
    mean = torch.empty(B, C, device=skip.device, dtype=f32)
    gate = torch.empty(B, C, device=skip.device, dtype=f32)
    z1 = torch.empty(B, R, device=skip.device, dtype=f32)
    _lib.call("edm_skip_gate_fwd", _p(skip), _p(w1h), _p(w2h), _p(mean), _p(gate), _p(z1), B, H * W, C, R, _stream())
    return mean, gate, z1


def skip_gate_fwd_multi(items):
    """skip_gate_fwd for several skip tensors of ONE channel count in one launch: items = sequence (<= 32) of
    (skip, w1h, w2h) or (skip, w1h, w2h, cat, sil) -> list of (mean, gate, z1), the values of the per-tensor call.
    cat / sil: (B, H, W, Ci + C) contiguous bf16 buffers whose right C columns receive skip * gate and mp_silu of it
    (== skip_half_fwd(skip, gate, cat, sil), by the workgroup that just reduced the sample)."""
    n = len(items)
    if not 0 < n <= 32:
        raise ValueError("skip_gate_fwd_multi: 1..32 gates per launch")
    arr = (_lib.SkipGateFwdItem * n)()
    out = []
    C0 = items[0][0].shape[-1]
    for k, item in enumerate(items):
        skip, w1h, w2h = item[:3]
        cat, sil = item[3:5] if len(item) > 3 else (None, None)
        B, H, W, C = _nhwc(skip, "skip")
        Ci = 0
        if cat is not None:
            Ci = cat.shape[-1] - C
            _chk(cat, bf16, "cat", (B, H, W, Ci + C))
            if sil is not None:
                _chk(sil, bf16, "sil", (B, H, W, Ci + C))
            if Ci <= 0 or Ci % 8:
                raise ValueError("skip_gate_fwd_multi: cat must have Ci + C channels with Ci a positive multiple of 8")
        elif sil is not None:
            raise ValueError("skip_gate_fwd_multi: sil needs cat")
        if C != C0:
            raise ValueError("skip_gate_fwd_multi: the gates of one launch share a channel count")
        R = w1h.shape[0]
        _chk(w1h, f32, "w1h", (R, C + 1))
        _chk(w2h, f32, "w2h", (C, R))
        mean = torch.empty(B, C, device=skip.device, dtype=f32)
        gate = torch.empty(B, C, device=skip.device, dtype=f32)
        z1 = torch.empty(B, R, device=skip.device, dtype=f32)
        arr[k] = _lib.SkipGateFwdItem(skip.data_ptr(), w1h.data_ptr(), w2h.data_ptr(), mean.data_ptr(), gate.data_ptr(),
                                      z1.data_ptr(), _p(cat), _p(sil), B, H * W, C, R, Ci, 0)
        out.append((mean, gate, z1))
    th, td, defer, release = _tables.take(items[0][0].device)
    _lib.call("edm_skip_gate_fwd_multi", ctypes.byref(arr), n, ctypes.c_void_p(th), ctypes.c_void_p(td), defer, _stream())
    release()
    return out


def skip_gate_bwd_multi(items):
    """the per-sample pass of skip_gate_bwd(defer_wgrad=True) for several gates of ONE channel count in one launch:
    items = sequence (<= 32) of (gcat, Ci, skip, w1h, w2h, gate, z1[, gskip]) -> list of (gmean (B, C), ws (B, C + 2R)).
    gskip: an uninitialised (B, H, W, C) bf16 tensor that receives gcat[..., Ci:] * gate + gmean / HW (== skip_half_bwd), written
    by the workgroup that just reduced the sample."""
    n = len(items)
    if not 0 < n <= 32:
        raise ValueError("skip_gate_bwd_multi: 1..32 gates per launch")
    arr = (_lib.SkipGateBwdItem * n)()
    out = []
    C0 = items[0][2].shape[-1]
    for k, item in enumerate(items):
        gcat, Ci, skip, w1h, w2h, gate, z1 = item[:7]
        gskip = item[7] if len(item) > 7 else None
        B, H, W, Ct = _nhwc(gcat, "gcat")
        Bs, Hs, Ws, C = _nhwc(skip, "skip")
        if (Bs, Hs, Ws) != (B, H, W) or Ct != Ci + C or C != C0:
            raise ValueError("skip_gate_bwd_multi: gcat/skip shape mismatch, or mixed channel counts in one launch")
        R = w1h.shape[0]
        _chk(w1h, f32, "w1h", (R, C + 1))
        _chk(w2h, f32, "w2h", (C, R))
        _chk(gate, f32, "gate", (B, C))
        _chk(z1, f32, "z1", (B, R))
        gmean = torch.empty(B, C, device=skip.device, dtype=f32)
        ws = torch.empty(B, C + 2 * R, device=skip.device, dtype=f32)
        if gskip is not None:
            _chk(gskip, bf16, "gskip", (B, H, W, C))
        arr[k] = _lib.SkipGateBwdItem(gcat.data_ptr(), Ct, skip.data_ptr(), w1h.data_ptr(), w2h.data_ptr(), gate.data_ptr(),
                                      z1.data_ptr(), gmean.data_ptr(), ws.data_ptr(), _p(gskip), Ci, B, H * W, C, R, 0)
        out.append((gmean, ws))
    th, td, defer, release = _tables.take(items[0][2].device)
    _lib.call("edm_skip_gate_bwd_multi", ctypes.byref(arr), n, ctypes.c_void_p(th), ctypes.c_void_p(td), defer, _stream())
    release()
    return out


def skip_half_bwd_multi(items):
    """skip_half_bwd for several tensors in one launch: items = sequence (<= 32) of (gcs, gate, gmean, gskip) with gskip an
    uninitialised tensor like gcs that is WRITTEN (the caller handed it out earlier as a placeholder)."""
    n = len(items)
    if not 0 < n <= 32:
        raise ValueError("skip_half_bwd_multi: 1..32 tensors per launch")
    arr = (_lib.SkipHalfBwdItem * n)()
    for k, (gcs, gate, gmean, gskip) in enumerate(items):
        B, H, W, Cs = _nhwc(gcs, "gcs")
        _chk(gate, f32, "gate", (B, Cs))
        _chk(gmean, f32, "gmean", (B, Cs))
        _chk(gskip, bf16, "gskip", (B, H, W, Cs))
        arr[k] = _lib.SkipHalfBwdItem(gcs.data_ptr(), gate.data_ptr(), gmean.data_ptr(), gskip.data_ptr(), B, H * W, Cs, 0)
    th, td, defer, release = _tables.take(items[0][0].device)
    _lib.call("edm_skip_half_bwd_multi", ctypes.byref(arr), n, ctypes.c_void_p(th), ctypes.c_void_p(td), defer, _stream())
    release()


def skip_gate_bwd(gcat, Ci, skip, mean, w1h, w2h, gate, z1, defer_wgrad=False):
    """backward of skip_gate_fwd given the gradient of cat = [inp, skip*gate]: (gmean, gw1h, gw2h) == reduce_hw(gcat[...,
    Ci:] * skip) followed by scalelong_bwd.  defer_wgrad=True: -> (gmean, ws): the batch sums that form gw1h / gw2h are left
    to ONE skip_gate_wgrad_multi over every gate of the backward pass (ws = this gate's per-sample vectors)."""
    B, H, W, Ct = _nhwc(gcat, "gcat")
    Bs, Hs, Ws, C = _nhwc(skip, "skip")
    if (Bs, Hs, Ws) != (B, H, W) or Ci + C != Ct or Ci % 8:
        raise ValueError("skip_gate_bwd: gcat / skip shape mismatch")
    R = w1h.shape[0]
    _chk(mean, f32, "mean", (B, C))
    _chk(gate, f32, "gate", (B, C))
    _chk(z1, f32, "z1", (B, R))
    _chk(w1h, f32, "w1h", (R, C + 1))
    _chk(w2h, f32, "w2h", (C, R))
    gmean = torch.empty(B, C, device=skip.device, dtype=f32)
    gw1 = None if defer_wgrad else torch.empty(w1h.shape, device=w1h.device, dtype=f32)
    gw2 = None if defer_wgrad else torch.empty(w2h.shape, device=w2h.device, dtype=f32)
    ws = torch.empty(B, C + 2 * R, device=skip.device, dtype=f32)
    _lib.call("edm_skip_gate_bwd", _p(gcat), Ct, Ci, _p(skip), _p(mean), _p(w1h), _p(w2h), _p(gate), _p(z1), _p(gmean),
              _p(gw1), _p(gw2), _p(ws), B, H * W, C, R, _stream())
    return (gmean, ws) if defer_wgrad else (gmean, gw1, gw2)


def skip_gate_wgrad_multi(items):
    """items: sequence (<= 32) of (ws, mean, R) from skip_gate_bwd(defer_wgrad=True) and the gate's forward -> list of
    (gw1h (R, C+1), gw2h (C, R)), all from ONE launch"""
    n = len(items)
    if not 0 < n <= 32:
        raise ValueError("skip_gate_wgrad_multi: 1..32 gates per launch")
    arr = (_lib.SkipGateWgradItem * n)()
    out = []
    for k, (ws, mean, R) in enumerate(items):
        _chk(mean, f32, "mean")
        B, C = mean.shape
        _chk(ws, f32, "ws", (B, C + 2 * R))
        gw1 = torch.empty(R, C + 1, device=ws.device, dtype=f32)
        gw2 = torch.empty(C, R, device=ws.device, dtype=f32)
        arr[k] = _lib.SkipGateWgradItem(ws.data_ptr(), mean.data_ptr(), gw1.data_ptr(), gw2.data_ptr(), B, C, R, 0)
        out.append((gw1, gw2))
    th, td, defer, release = _tables.take(items[0][0].device)
    _lib.call("edm_skip_gate_wgrad_multi", ctypes.byref(arr), n, ctypes.c_void_p(th), ctypes.c_void_p(td), defer, _stream())
    release()
    return out


def concat_gate_fwd(inp, skip, gate, want_silu):
    B, H, W, Ci = _nhwc(inp, "inp")
    Bs, Hs, Ws, Cs = _nhwc(skip, "skip")
    if (Bs, Hs, Ws) != (B, H, W):
        raise ValueError("concat_gate_fwd: inp/skip spatial mismatch")
    _chk(gate, f32, "gate", (B, Cs))
    cat = torch.empty(B, H, W, Ci + Cs, device=inp.device, dtype=bf16)
    sil = torch.empty_like(cat) if want_silu else None
    _lib.call("edm_concat_gate_fwd", _p(inp), _p(skip), _p(gate), _p(cat), _p(sil), B, H * W, Ci, Cs, _stream())
    return cat, sil


def concat_gate_bwd(gcat, gate, gmean, Ci):
    B, H, W, Ct = _nhwc(gcat, "gcat")
    Cs = Ct - Ci
    _chk(gate, f32, "gate", (B, Cs))
    _chk(gmean, f32, "gmean", (B, Cs))
    ginp = torch.empty(B, H, W, Ci, device=gcat.device, dtype=bf16)
    gskip = torch.empty(B, H, W, Cs, device=gcat.device, dtype=bf16)
    _lib.call("edm_concat_gate_bwd", _p(gcat), _p(gate), _p(gmean), _p(ginp), _p(gskip), B, H * W, Ci, Cs, _stream())
    return ginp, gskip


def skip_half_fwd(skip, gate, cat, sil=None):
    """cat[..., Ci:] = skip*gate (and sil[..., Ci:] = mp_silu of it) in place: the skip half of the decoder's concatenated
    operands; the input half was written by the producer of `input` (conv_igemm(out=, silu_out=))."""
    B, H, W, Cs = _nhwc(skip, "skip")
    _chk(gate, f32, "gate", (B, Cs))
    Ct = cat.shape[-1]
    _chk(cat, bf16, "cat", (B, H, W, Ct))
    if sil is not None:
        _chk(sil, bf16, "sil", (B, H, W, Ct))
    if Ct <= Cs:
        raise ValueError("skip_half_fwd: cat must be wider than the skip")
    _lib.call("edm_skip_half_fwd", _p(skip), _p(gate), _p(cat), _p(sil), B, H * W, Ct - Cs, Cs, _stream())


def skip_half_bwd(gcs, gate, gmean):
    """gskip = gcs*gate + gmean/HW from the skip half gcs of d loss / d cat (conv_igemm(split=) wrote it)"""
    B, H, W, Cs = _nhwc(gcs, "gcs")
    _chk(gate, f32, "gate", (B, Cs))
    _chk(gmean, f32, "gmean", (B, Cs))
    gskip = torch.empty_like(gcs)
    _lib.call("edm_skip_half_bwd", _p(gcs), _p(gate), _p(gmean), _p(gskip), B, H * W, Cs, _stream())
    return gskip


def _sigma_arg(sigma, B):
    _chk(sigma, f32, "sigma")
    if sigma.numel() == 1:
        return 0
    if sigma.numel() != B:
        raise ValueError(f"sigma must have 1 or {B} elements, got {sigma.numel()}")
    return 1


def precond_in(noisy, sigma, sigma_data, CP):
    _chk(noisy, f32, "noisy")
    B, Cimg, H, W = noisy.shape
    ss = _sigma_arg(sigma, B)
    out = torch.empty(B, H, W, CP, device=noisy.device, dtype=bf16)
    _lib.call("edm_precond_in", _p(noisy), _p(sigma), ss, float(sigma_data), _p(out), B, Cimg, H * W, CP, _stream())
    return out


def conv_out_fwd(x, w_hat, gain_out, noisy, sigma, sigma_data, want_fraw=True):
    B, H, W, C = _nhwc(x, "x")
    Co = w_hat.shape[0]
    _chk(w_hat, f32, "w_hat", (Co, C))
    _chk(noisy, f32, "noisy", (B, Co, H, W))
    _chk(gain_out, f32, "gain_out")
    ss = _sigma_arg(sigma, B)
    D = torch.empty(B, Co, H, W, device=x.device, dtype=f32)
    Fraw = torch.empty(B, Co, H, W, device=x.device, dtype=f32) if want_fraw else None
    _lib.call("edm_conv_out_fwd", _p(x), _p(w_hat), _p(gain_out), _p(noisy), _p(sigma), ss, float(sigma_data), _p(D),
              _p(Fraw), B, H * W, C, Co, _stream())
    return D, Fraw


def conv_out_bwd(x, w_hat, gain_out, Fraw, dD, sigma, sigma_data, gg_out=None):
    """gg_out: optional 0-dim fp32 tensor that d loss / d gain_out is ACCUMULATED into (else a fresh zero scalar)"""
    B, H, W, C = _nhwc(x, "x")
    Co = w_hat.shape[0]
    _chk(dD, f32, "dD", (B, Co, H, W))
    _chk(Fraw, f32, "Fraw", (B, Co, H, W))
    ss = _sigma_arg(sigma, B)
    gx = torch.empty_like(x)
    gw = zeros_f32(w_hat.shape, w_hat.device)
    gg = zeros_f32((), x.device) if gg_out is None else _chk(gg_out, f32, "gg_out", ())
    _lib.call("edm_conv_out_bwd", _p(x), _p(w_hat), _p(gain_out), _p(Fraw), _p(dD), _p(sigma), ss, float(sigma_data),
              _p(gx), _p(gw), _p(gg), B, H * W, C, Co, _stream())
    return gx, gw, gg


def nchw_to_nhwc_bf16(x):
    _chk(x, f32, "x")
    B, C, H, W = x.shape
    y = torch.empty(B, H, W, C, device=x.device, dtype=bf16)
    _lib.call("edm_nchw_to_nhwc_bf16", _p(x), _p(y), B, C, H * W, _stream())
    return y


def nhwc_bf16_to_nchw(x):
    B, H, W, C = _nhwc(x, "x")
    y = torch.empty(B, C, H, W, device=x.device, dtype=f32)
    _lib.call("edm_nhwc_bf16_to_nchw", _p(x), _p(y), B, C, H * W, _stream())
    return y


# ------------------------------------------------------------------ convolution
# 0 = pick per shape (default), 1 = register-staged 128x128 kernel, 2 = LDS-DMA 256x128 kernel,
# 5 = small-map split-K kernel, 6 = static-schedule 3x3 kernel (tests force each generation through this variable)
IGEMM_VERSION = int(os.environ.get("EDM_IGEMM", "0"))


def _igemm_entry(npix, W, Cout, taps, Cin=0):
    if taps == 9 and W > 64:
        return "edm_conv_igemm"
    if IGEMM_VERSION == 1:
        return "edm_conv_igemm"
    if IGEMM_VERSION == 2:
        return "edm_conv_igemm_v2"
    if IGEMM_VERSION == 6:
        ok = taps == 9 and Cin % 64 == 0 and Cin <= 2016 and W <= 64
        return "edm_conv_igemm_v6" if ok else "edm_conv_igemm"
    if IGEMM_VERSION == 5:
        return "edm_conv_igemm_s" if (taps == 9 and Cin % 256 == 0 and Cin <= 2016 and W <= 16) else "edm_conv_igemm"
    # per-shape choice from the microbenchmarks (tools/microbench_conv.py): the LDS-DMA tall-tile kernels only pay
    # off when they still give every CU >= 2 tiles; small feature maps keep the 128x128 register-staged kernel.
    if taps == 9:
        tm = (npix + 511) // 512
        tiles3 = tm * ((Cout + 127) // 128)
        v6_ok = Cin % 64 == 0 and Cin <= 2016 and W <= 64
        if tiles3 >= 512:
            return "edm_conv_igemm_v6" if v6_ok else "edm_conv_igemm"    # (conv_in, Cin = 32: 37 us there; the tall-tile
                                                                             #  k_conv_igemm3 it used to take needed 47 and was retired)
        if v6_ok and tm * ((Cout + 63) // 64) >= 256:     # 512x64 tiles of the same kernel (16x16 layers at batch 128)
            return "edm_conv_igemm_v6"
        ts = ((npix + 127) // 128) * ((Cout + 63) // 64)
        if W <= 16 and Cin % 256 == 0 and Cin <= 2016 and 128 <= ts <= 1024:   # small maps: K split over the waves
            return "edm_conv_igemm_s"
        return "edm_conv_igemm"
    tiles2 = ((npix + 255) // 256) * ((Cout + 127) // 128)
    return "edm_conv_igemm_v2" if tiles2 >= 1024 else "edm_conv_igemm"


def uses_s_kernel(npix, W, Cin, Cout):
    """the default dispatch runs a 3x3 conv of this shape on k_conv3x3_s (the 8x8 layers' kernel) AND that kernel can take
    a fragment-major weight pack for it (networks._PrepPlan writes one for such layers)"""
    return (IGEMM_VERSION == 0 and Cout % 64 == 0 and Cin % 32 == 0
            and _igemm_entry(npix, W, Cout, 9, Cin) == "edm_conv_igemm_s")


def _wfrag(wp, entry, who):
    """1 when wp is a fragment-major pack (tagged by the plan that wrote it); such a pack is only valid on k_conv3x3_s"""
    if not getattr(wp, "_edm_frag", False):
        return 0
    if entry != "edm_conv_igemm_s":
        raise ValueError(f"{who}: a fragment-major weight pack reached a shape that does not run on k_conv3x3_s ({entry})")
    return 1


V46 = "_v6"     # profile-key suffix of the static-schedule 3x3 kernel (its 32x32x16 predecessor, "_v4", was retired in round 4)
# kernel ids of edm_conv_igemm_o (include/tinyedm_hip.h)
_KERNEL_ID = {"edm_conv_igemm": 1, "edm_conv_igemm_v2": 2, "edm_conv_igemm_s": 5, "edm_conv_igemm_v6": 6}


def _v4_suffix(entry, npix, Cout):
    """profile-key suffix naming the kernel that runs: _v6 = k_conv3x3_v6 with 512x128 tiles, _v6s = the same kernel with
    512x64 tiles (16x16 layers), _s = k_conv3x3_s (8x8 layers)"""
    if entry == "edm_conv_igemm_s":
        return "_s"
    if entry != "edm_conv_igemm_v6":
        return entry[len("edm_conv_igemm"):]
    pad4, pad2 = (Cout + 127) // 128 * 128, (Cout + 63) // 64 * 64     # (conv_igemm6.hip: narrow tiles when 128 would mostly pad)
    wide = ((npix + 511) // 512) * ((Cout + 127) // 128) >= 512 and not pad2 * 80 < pad4 * 69
    return V46 if wide else V46 + "s"


def conv_igemm(x, wp, taps, residual=None, alpha=1.0, beta=0.0, out=None, silu_out=None, split=None):
    """Y = alpha*conv(x, wp) + beta*residual.  wp: bf16 (taps, Cout, Cin).
    Output descriptor (edm_conv_igemm_o; how the decoder's concat stops being a copy):
      out       a (B, H, W, Cout) VIEW with unit channel stride and a row stride >= Cout (the left column block of a wider
                NHWC buffer): the result is written there and `out` is returned;
      silu_out  a view with the same strides as `out`: also receives mp_silu(Y);
      split     (c, ya, yb): output channels < c go to ya (B, H, W, c), the others to yb (B, H, W, Cout - c), both
                contiguous; returns (ya, yb)."""
    B, H, W, Cin = _nhwc(x, "x")
    _chk(wp, bf16, "wp")
    if wp.dim() != 3 or wp.shape[0] != taps or wp.shape[2] != Cin:
        raise ValueError(f"conv_igemm: packed weight {tuple(wp.shape)} does not match taps={taps}, Cin={Cin}")
    Cout = wp.shape[1]
    if residual is not None:
        _chk(residual, bf16, "residual", (B, H, W, Cout))
    npix = B * H * W
    entry = _igemm_entry(npix, W, Cout, taps, Cin)
    # profile key names the kernel generation that runs ("conv3x3_igemm_v6", "conv1x1_igemm", ...)
    pname = ("conv3x3_igemm" if taps == 9 else "conv1x1_igemm") + _v4_suffix(entry, npix, Cout)
    nbytes = 2.0 * (npix * (Cin + Cout * ((2 if residual is not None else 1) + (1 if silu_out is not None else 0))) + wp.numel())
    wfrag = _wfrag(wp, entry, "conv_igemm")
    if out is None and silu_out is None and split is None and not wfrag:
        y = torch.empty(B, H, W, Cout, device=x.device, dtype=bf16)
        with _prof(pname, 2.0 * npix * Cin * Cout * taps, nbytes):
            _lib.call(entry, _p(x), _p(wp), _p(y), _p(residual), float(alpha), float(beta), B, H, W, Cin, Cout, taps, _stream())
        return y
    kid = _KERNEL_ID.get(entry)
    if kid is None:
        raise ValueError(f"conv_igemm: {entry} has no output-descriptor form (shape B={B} H={H} W={W} Cin={Cin} Cout={Cout})")
    ld, ya, yb, ldb, c = 0, None, None, 0, 0
    if split is not None:
        if out is not None or silu_out is not None:
            raise ValueError("conv_igemm: split excludes out / silu_out")
        c, ya, yb = split
        _chk(ya, bf16, "split[1]", (B, H, W, c))
        _chk(yb, bf16, "split[2]", (B, H, W, Cout - c))
        if c % 8 or not (0 < c < Cout):
            raise ValueError("conv_igemm: split channel must be a multiple of 8 inside (0, Cout)")
        ld, ldb = c, Cout - c
    else:
        if out is None:
            out = torch.empty(B, H, W, Cout, device=x.device, dtype=bf16)
        ld = _row_view(out, B, H, W, Cout, "out")
        if silu_out is not None and _row_view(silu_out, B, H, W, Cout, "silu_out") != ld:
            raise ValueError("conv_igemm: out and silu_out must have the same row stride")
        ya = out
    with _prof(pname, 2.0 * npix * Cin * Cout * taps, nbytes):
        _lib.call("edm_conv_igemm_o", _p(x), _p(wp), _p(ya), ld, _p(silu_out), _p(yb), ldb, c, _p(residual), float(alpha),
                  float(beta), B, H, W, Cin, Cout, taps, kid, wfrag, _stream())
    return (ya, yb) if split is not None else out


FOLD_PROJ = os.environ.get("EDM_FOLD_PROJ", "1") != "0"


def conv3x3_fold_supported(xshape, Cout, C2):
    """the shape (B, H, W, Cin) -> Cout of a decoder block's second 3x3 conv lets its skip projection (C2 -> Cout) ride along
    (conv3x3_fold)"""
    B, H, W, Cin = xshape
    return bool(FOLD_PROJ and IGEMM_VERSION == 0 and _lib.call("edm_conv3x3_fold_supported", B, H, W, Cin, Cout, C2))


def conv3x3_fold(x, wp, x2, w2p, alpha3, alpha1, out=None, silu_out=None):
    """Y = alpha3 * conv3x3(x, wp) + alpha1 * conv1x1(x2, w2p) in one launch (edm_conv3x3_fold): wp (9, Cout, Cin), x2
    (B, H, W, C2) contiguous, w2p (1, Cout, C2).  out / silu_out: the output descriptor of conv_igemm."""
    B, H, W, Cin = _nhwc(x, "x")
    _chk(wp, bf16, "wp")
    _chk(w2p, bf16, "w2p")
    B2, H2, W2, C2 = _nhwc(x2, "x2")
    Cout = wp.shape[1]
    if wp.dim() != 3 or wp.shape[0] != 9 or wp.shape[2] != Cin or tuple(w2p.shape) != (1, Cout, C2) or (B2, H2, W2) != (B, H, W):
        raise ValueError(f"conv3x3_fold: operands do not match (x {tuple(x.shape)}, wp {tuple(wp.shape)}, x2 {tuple(x2.shape)}, "
                         f"w2p {tuple(w2p.shape)})")
    if getattr(wp, "_edm_frag", False):
        raise ValueError("conv3x3_fold: a fragment-major weight pack belongs to k_conv3x3_s")
    if out is None:
        out = torch.empty(B, H, W, Cout, device=x.device, dtype=bf16)
    ld = _row_view(out, B, H, W, Cout, "out")
    if silu_out is not None and _row_view(silu_out, B, H, W, Cout, "silu_out") != ld:
        raise ValueError("conv3x3_fold: out and silu_out must have the same row stride")
    npix = B * H * W
    pname = "conv3x3_igemm" + _v4_suffix("edm_conv_igemm_v6", npix, Cout) + "_fold"
    with _prof(pname, 2.0 * npix * Cout * (9 * Cin + C2),
               2.0 * (npix * (Cin + C2 + Cout * (2 if silu_out is not None else 1)) + wp.numel() + w2p.numel())):
        _lib.call("edm_conv3x3_fold", _p(x), _p(wp), _p(x2), C2, _p(w2p), C2, _p(out), ld, _p(silu_out), float(alpha3),
                  float(alpha1), B, H, W, Cin, Cout, _stream())
    return out


def _row_view(t, B, H, W, C, name):
    """row stride (elements) of a (B, H, W, C) bf16 view whose pixels are rows of one flat [B*H*W][ld] buffer"""
    if t.dtype != bf16 or not t.is_cuda or tuple(t.shape) != (B, H, W, C):
        raise ValueError(f"{name}: expected a cuda bf16 tensor of shape {(B, H, W, C)}, got {t.dtype} {tuple(t.shape)}")
    sb, sh, sw, sc = t.stride()
    if sc != 1 or sw < C or sw % 8 or sh != W * sw or sb != H * sh or (t.data_ptr() % 16):
        raise ValueError(f"{name}: strides {t.stride()} are not those of a column block of an NHWC buffer")
    return sw


FUSE_MOD = os.environ.get("EDM_FUSE_MOD", "1") != "0"


def conv3x3_mod(x, wp, lin, gain, pdrop, seed, sub, step, want_u=True, dyn=None, mark_dropped=False):
    """First 3x3 conv of a block with the modulation epilogue fused: returns (u, a2) with u = conv(x) (None when
    want_u is False) and a2 = dropout(mp_silu(u*(lin*gain+1))) -- same values as conv_igemm + mod_silu_drop_fwd.
    mark_dropped: the elements of u that the dropout removed come back as NaN (their value is never needed again);
    conv3x3_modbwd(u_marked=True) and mod_silu_drop_bwd read the mask from there."""
    B, H, W, Cin = _nhwc(x, "x")
    _chk(wp, bf16, "wp")
    if wp.dim() != 3 or wp.shape[0] != 9 or wp.shape[2] != Cin:
        raise ValueError(f"conv3x3_mod: pack shape {tuple(wp.shape)} does not match taps=9, Cin={Cin}")
    Cout = wp.shape[1]
    ls = _lin_view(lin, B, Cout, "lin")
    _chk(gain, f32, "gain")
    u = torch.empty(B, H, W, Cout, device=x.device, dtype=bf16) if want_u else None
    a2 = torch.empty(B, H, W, Cout, device=x.device, dtype=bf16)
    npix = B * H * W
    entry = _igemm_entry(npix, W, Cout, 9, Cin)
    pname = "conv3x3_igemm" + (_v4_suffix(entry, npix, Cout) if entry in ("edm_conv_igemm_v6", "edm_conv_igemm_s") else "")
    with _prof(pname, 2.0 * npix * Cin * Cout * 9, 2.0 * (npix * (Cin + Cout * (2 if want_u else 1)) + wp.numel())):
        _lib.call("edm_conv3x3_mod", _p(x), _p(wp), _p(u), _p(a2), _p(lin), ls, _p(gain), float(pdrop), int(seed),
                  int(sub), int(step), int(bool(mark_dropped)), B, H, W, Cin, Cout, _dyn(dyn), _wfrag(wp, entry, "conv3x3_mod"),
                  _stream())
    return u, a2


def conv3x3_modbwd(gout, wd, alpha, r1, lin, gain, pdrop, seed, sub, step, glin_out=None, ggain_out=None, dyn=None,
                   gm_out=None, u_marked=False):
    """dgrad of a block's second 3x3 conv with the modulation backward in its epilogue: returns (gr1, glin, ggain),
    the values conv_igemm(gout, wd, 9, alpha=alpha) followed by mod_silu_drop_bwd would give (H*W % 32 == 0).
    gm_out: a zero-filled (B, Cout) fp32 view with unit column stride (a column slice of a buffer shared by all blocks):
    the raw modulation gradient is accumulated there and NOT finished -- returns (gr1, None, None); one
    mod_finish_multi over the shared buffer turns it into glin / ggain for every block.
    u_marked: r1 comes from conv3x3_mod(mark_dropped=True) with the same pdrop: dropped <=> NaN, no Philox stream is
    regenerated."""
    B, H, W, Cin = _nhwc(gout, "gout")
    _chk(wd, bf16, "wd")
    if wd.dim() != 3 or wd.shape[0] != 9 or wd.shape[2] != Cin:
        raise ValueError(f"conv3x3_modbwd: pack shape {tuple(wd.shape)} does not match taps=9, Cin={Cin}")
    Cout = wd.shape[1]
    _chk(r1, bf16, "r1", (B, H, W, Cout))
    if (H * W) % 32:
        raise ValueError("conv3x3_modbwd: H*W must be a multiple of 32")
    ls = _lin_view(lin, B, Cout, "lin")
    _chk(gain, f32, "gain")
    gr = torch.empty_like(r1)
    if gm_out is not None:
        gms = _lin_view(gm_out, B, Cout, "gm_out")
        gm = gm_out
    else:
        gms = 0
        gm = zeros_f32((B, Cout), r1.device)
        glin = torch.empty(B, Cout, device=r1.device, dtype=f32) if glin_out is None else glin_out
        gs = _lin_view(glin, B, Cout, "glin")
        ggain = zeros_f32((), r1.device) if ggain_out is None else _chk(ggain_out, f32, "ggain_out", ())
    npix = B * H * W
    entry = _igemm_entry(npix, W, Cout, 9, Cin)
    pname = "conv3x3_igemm" + (_v4_suffix(entry, npix, Cout) if entry in ("edm_conv_igemm_v6", "edm_conv_igemm_s") else "") + "_modbwd"
    with _prof(pname, 2.0 * npix * Cin * Cout * 9, 2.0 * (npix * (Cin + 2 * Cout) + wd.numel())):
        _lib.call("edm_conv3x3_modbwd", _p(gout), _p(wd), float(alpha), _p(r1), _p(lin), ls, _p(gain), _p(gr), _p(gm), gms,
                  float(pdrop), int(seed), int(sub), int(step), int(bool(u_marked)), B, H, W, Cin, Cout, _dyn(dyn),
                  _wfrag(wd, entry, "conv3x3_modbwd"), _stream())
    if gm_out is not None:
        return gr, None, None
    _lib.call("edm_mod_finish", _p(gm), _p(lin), ls, _p(gain), _p(glin), gs, _p(ggain), B, Cout, _stream())
    return gr, glin, ggain


def mod_finish_multi(gm_all, lin_all, glin_all, items, n_items):
    """items: uint8 device tensor of n_items 24-byte records {gain ptr, ggain ptr, col0, C} (csrc/elementwise.hip
    ModFinItem): glin_all[:, col0:col0+C] += gm_all * gain, ggain += sum gm_all * lin_all, for every block at once."""
    _chk(gm_all, f32, "gm_all")
    _chk(lin_all, f32, "lin_all", gm_all.shape)
    _chk(glin_all, f32, "glin_all", gm_all.shape)
    _chk(items, torch.uint8, "items")
    if gm_all.dim() != 2 or items.numel() != 24 * n_items:
        raise ValueError("mod_finish_multi: bad buffers")
    _lib.call("edm_mod_finish_multi", _p(gm_all), _p(lin_all), _p(glin_all), gm_all.shape[1], _p(items), int(n_items),
              gm_all.shape[0], _stream())


def conv3x3_silubwd(g, wd, xpre, gextra=None, extra_scale=1.0):
    """dgrad of a block's first 3x3 conv with the mp_silu backward in its epilogue:
    mp_silu'(xpre) * conv_igemm(g, wd, 9) + extra_scale * gextra  (== conv_igemm followed by silu_bwd)."""
    B, H, W, Cin = _nhwc(g, "g")
    _chk(wd, bf16, "wd")
    if wd.dim() != 3 or wd.shape[0] != 9 or wd.shape[2] != Cin:
        raise ValueError(f"conv3x3_silubwd: pack shape {tuple(wd.shape)} does not match taps=9, Cin={Cin}")
    Cout = wd.shape[1]
    _chk(xpre, bf16, "xpre", (B, H, W, Cout))
    if gextra is not None:
        _chk(gextra, bf16, "gextra", xpre.shape)
    gx = torch.empty_like(xpre)
    npix = B * H * W
    entry = _igemm_entry(npix, W, Cout, 9, Cin)
    pname = "conv3x3_igemm" + (_v4_suffix(entry, npix, Cout) if entry in ("edm_conv_igemm_v6", "edm_conv_igemm_s") else "") + "_silubwd"
    with _prof(pname, 2.0 * npix * Cin * Cout * 9,
               2.0 * (npix * (Cin + Cout * (3 if gextra is not None else 2)) + wd.numel())):
        _lib.call("edm_conv3x3_silubwd", _p(g), _p(wd), _p(xpre), _p(gextra), float(extra_scale), _p(gx), B, H, W, Cin,
                  Cout, _wfrag(wd, entry, "conv3x3_silubwd"), _stream())
    return gx


WGRAD_1X1 = os.environ.get("EDM_WGRAD_1X1", "1") != "0"
WGRAD_VERSION = 2       # (generation 1, the register-staged kernel, was retired in round 6; the name stays for the tests' asserts)


def conv_wgrad(x, dy, taps):
    """fp32 split-K slabs (S, taps, Cout, Cin) of dW in packed order."""
    B, H, W, Cin = _nhwc(x, "x")
    Bd, Hd, Wd, Cout = _nhwc(dy, "dy")
    if (Bd, Hd, Wd) != (B, H, W):
        raise ValueError("conv_wgrad: x/dy spatial mismatch")
    npix = B * H * W
    if taps == 1 and WGRAD_1X1 and Cin % 32 == 0 and Cout % 32 == 0:
        # dedicated 1x1 kernel: 256x128-output tiles (the 64x64 tiles below are L2->LDS bound on 1x1 layers)
        S = _lib.call("edm_conv_wgrad_1x1_nsplit", npix, Cin, Cout)
        slabs = torch.empty(S, 1, Cout, Cin, device=x.device, dtype=f32)
        with _prof("conv1x1_wgrad", 2.0 * npix * Cin * Cout, 2.0 * npix * (Cin + Cout) + 4.0 * slabs.numel()):
            _lib.call("edm_conv_wgrad_1x1", _p(x), _p(dy), _p(slabs), npix, Cin, Cout, S, _stream())
        return slabs
    S = _lib.call("edm_conv_wgrad_nsplit", B, H, W, Cin, Cout, taps)
    slabs = torch.empty(S, taps, Cout, Cin, device=x.device, dtype=f32)
    with _prof("conv3x3_wgrad" if taps == 9 else "conv1x1_wgrad", 2.0 * npix * Cin * Cout * taps,
               2.0 * npix * (Cin + Cout) + 4.0 * slabs.numel()):
        _lib.call("edm_conv_wgrad_v2", _p(x), _p(dy), _p(slabs), B, H, W, Cin, Cout, taps, S, _stream())
    return slabs


def wgrad1x1_group_supported(x, dy):
    return WGRAD_1X1 and x.shape[-1] % 32 == 0 and dy.shape[-1] % 32 == 0


def conv_wgrad_1x1_group(pairs):
    """pairs: sequence (<= 16) of (x, dy) NHWC bf16 tensors of 1x1 layers -> list of their fp32 split-K slabs
    (S, 1, Cout, Cin), all produced by ONE launch (csrc/conv_wgrad1x1.hip k_wgrad1x1_group)."""
    n = len(pairs)
    if not 0 < n <= 16:
        raise ValueError("conv_wgrad_1x1_group: 1..16 layers per group")
    arr = (_lib.WGrad1Item * n)()
    out = []
    flops = nbytes = 0.0
    for k, (x, dy) in enumerate(pairs):
        B, H, W, Cin = _nhwc(x, "x")
        Bd, Hd, Wd, Cout = _nhwc(dy, "dy")
        if (Bd, Hd, Wd) != (B, H, W) or Cin % 32 or Cout % 32:
            raise ValueError("conv_wgrad_1x1_group: x/dy mismatch or channels not multiples of 32")
        npix = B * H * W
        S = _lib.call("edm_conv_wgrad_1x1_nsplit_grouped", npix, Cin, Cout)
        slabs = torch.empty(S, 1, Cout, Cin, device=x.device, dtype=f32)
        arr[k] = _lib.WGrad1Item(x.data_ptr(), dy.data_ptr(), slabs.data_ptr(), npix, Cin, Cout, S, 0)
        out.append(slabs)
        flops += 2.0 * npix * Cin * Cout
        nbytes += 2.0 * npix * (Cin + Cout) + 4.0 * slabs.numel()
    th, td, defer, release = _tables.take(pairs[0][0].device)
    with _prof("conv1x1_wgrad", flops, nbytes):
        _lib.call("edm_conv_wgrad_1x1_group", ctypes.byref(arr), n, ctypes.c_void_p(th), ctypes.c_void_p(td), defer, _stream())
    release()
    return out


W3_MAX_LAYERS = 48      # csrc/conv_wgrad3.hip MAXL (edm_wgrad3_max_layers; checked by tests/test_host_cpu.py)


def wgrad3_supported(x, dy, I):
    """shapes the grouped 3x3 weight-gradient path (csrc/conv_wgrad3.hip) covers"""
    B, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    return Cin % 32 == 0 and Cout % 32 == 0 and W <= 126 and I * 9 * 4 <= 64 * 1024 and B * (H + 1) * (W + 1) < (1 << 30)


def wgrad3_plan_ksplit(shapes):
    """K shares per tile that edm_wgrad3_group's plan gives each layer of a group (diagnostics, host logic only: no
    launch, no device memory).  shapes: sequence of (B, H, W, Cin, Cout)."""
    n = len(shapes)
    arr = (_lib.WGrad3Item * n)()
    for k, (B, H, W, Cin, Cout) in enumerate(shapes):
        arr[k] = _lib.WGrad3Item(1, 1, 1, 1, None, B, H, W, Cin, Cout, Cin, 1.0, 0)     # (pointers only checked for NULL)
    out = (ctypes.c_int * n)()
    _lib.call("edm_wgrad3_plan_ksplit", ctypes.byref(arr), n, out)
    return list(out)


def wgrad3_group(items):
    """Weight gradients of up to W3_MAX_LAYERS (48) 3x3 conv layers in ONE stream-K launch + ONE finish launch.
    items: sequence of (x, dy, w, grad, perm, scale, accumulate) with x (B,H,W,Cin) / dy (B,H,W,Cout) NHWC bf16,
    w the fp32 master weight (Cout, I, 3, 3) with I <= Cin, grad an fp32 tensor like w that receives (accumulate=0)
    or accumulates (1) the projected gradient, perm the optional packed-row permutation (int32)."""
    n = len(items)
    if not 0 < n <= W3_MAX_LAYERS:
        raise ValueError(f"wgrad3_group: 1..{W3_MAX_LAYERS} layers per group")
    arr = (_lib.WGrad3Item * n)()
    flops = nbytes = 0.0
    halo = None
    for k, (x, dy, w, grad, perm, scale, accumulate) in enumerate(items):
        B, H, W, Cin = _nhwc(x, "x")
        Bd, Hd, Wd, Cout = _nhwc(dy, "dy")
        if (Bd, Hd, Wd) != (B, H, W):
            raise ValueError("wgrad3_group: x/dy spatial mismatch")
        _chk(w, f32, "w")
        if w.dim() != 4 or w.shape[0] != Cout or tuple(w.shape[2:]) != (3, 3) or w.shape[1] > Cin:
            raise ValueError(f"wgrad3_group: weight {tuple(w.shape)} does not match Cout={Cout}, Cin={Cin}")
        _chk(grad, f32, "grad", w.shape)
        if not wgrad3_supported(x, dy, w.shape[1]):
            raise ValueError(f"wgrad3_group: unsupported shape x={tuple(x.shape)} dy={tuple(dy.shape)}")
        if perm is not None:
            _chk(perm, torch.int32, "perm", (Cout,))
        h = W + 2 > 64
        if halo is not None and h != halo:
            raise ValueError("wgrad3_group: layers of one group must be all W <= 62 or all W > 62")
        halo = h
        arr[k] = _lib.WGrad3Item(x.data_ptr(), dy.data_ptr(), w.data_ptr(), grad.data_ptr(),
                                 None if perm is None else perm.data_ptr(), B, H, W, Cin, Cout, w.shape[1], float(scale),
                                 int(bool(accumulate)))
        flops += 2.0 * B * H * W * Cin * Cout * 9
        nbytes += 2.0 * B * H * W * (Cin + Cout) + 4.0 * w.numel()
    nb = _lib.call("edm_wgrad3_workspace", ctypes.byref(arr), n)
    if nb <= 0:
        raise _lib.HipKernelError(f"edm_wgrad3_workspace failed: {_lib.lib().edm_last_error().decode()}")
    work = torch.empty(nb // 4, device=items[0][0].device, dtype=f32)
    th, td, defer, release = _tables.take(items[0][0].device)
    probe = None
    if PROFILE is not None:
        # the entry point uploads its launch table, runs k_wgrad3 AND k_wgrad3_finish: "conv3x3_wgrad" times the call,
        # "conv3x3_wgrad_kernel" the k_wgrad3 launch alone (events recorded inside the call: edm_wgrad3_probe) -- the figure
        # the rocprofv3 kernel traces show (888 us per call vs 813-831 us for the kernel)
        probe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for ev in probe:
            ev.record()             # (torch creates the hipEvent_t on the first record)
        _lib.call("edm_wgrad3_probe", ctypes.c_void_p(probe[0].cuda_event), ctypes.c_void_p(probe[1].cuda_event))
    with _prof("conv3x3_wgrad", flops, nbytes):
        _lib.call("edm_wgrad3_group", ctypes.byref(arr), n, _p(work), nb, ctypes.c_void_p(th), ctypes.c_void_p(td), defer,
                  _stream())
    if probe is not None:
        PROFILE.setdefault("conv3x3_wgrad_kernel", []).append((probe[0], probe[1], flops, nbytes))
    release()
    return work


# ------------------------------------------------------------------ weights
def weight_prep(w, taps, Ipad=None, want_fwd=True, want_dgrad=True, want_hat=False, perm=None, normalize_inplace=False):
    """w: fp32 master (O, I, k, k) or (O, I).  Returns (wp_fwd, wp_dgrad, w_hat)."""
    _chk(w, f32, "w")
    O = w.shape[0]
    I = w.shape[1]
    if w.numel() != O * I * taps:
        raise ValueError("weight_prep: taps does not match the weight shape")
    Ipad = I if Ipad is None else Ipad
    dev = w.device
    wf = torch.empty(taps, O, Ipad, device=dev, dtype=bf16) if want_fwd else None
    wd = torch.empty(taps, I, O, device=dev, dtype=bf16) if want_dgrad else None
    wh = torch.empty(O, I * taps, device=dev, dtype=f32) if want_hat else None
    if perm is not None:
        _chk(perm, torch.int32, "perm", (O,))
    _lib.call("edm_weight_prep", _p(w), O, I, taps, Ipad, _p(wf), _p(wd), _p(wh), _p(perm), int(normalize_inplace),
              _stream())
    return wf, wd, wh


def weight_prep_multi(desc, groups, lds_bytes, normalize_inplace):
    """desc: uint8 tensor of 64-byte PrepDesc records (see csrc/weights.hip), groups: int32 [n_groups, 2] of
    (record, first row); lds_bytes: the largest rb*I*taps*2 over the records."""
    _chk(desc, torch.uint8, "desc")
    _chk(groups, torch.int32, "groups")
    if desc.numel() % 64 or groups.dim() != 2 or groups.shape[1] != 2:
        raise ValueError("weight_prep_multi: bad descriptor / group tables")
    _lib.call("edm_weight_prep_multi", _p(desc), _p(groups), int(groups.shape[0]), int(lds_bytes),
              int(bool(normalize_inplace)), _stream())


def wgrad_finish(slabs, w, taps, I, perm=None, scale=1.0, out=None):
    """Reduce slabs (S, taps, O, Ipad) and project through the weight normalisation -> grad like w."""
    _chk(slabs, f32, "slabs")
    _chk(w, f32, "w")
    S, t_, O, Ipad = slabs.shape
    if t_ != taps or w.shape[0] != O or w.numel() != O * I * taps:
        raise ValueError("wgrad_finish: shape mismatch")
    accumulate = out is not None
    if out is None:
        out = torch.empty_like(w)
    else:
        _chk(out, f32, "out", w.shape)
    _lib.call("edm_wgrad_finish", _p(slabs), S, _p(w), _p(out), _p(perm), O, I, Ipad, taps, float(scale), int(accumulate),
              _stream())
    return out


def wgrad_finish_multi(items):
    """items: sequence of (slabs, w, grad, perm, taps, I, scale, accumulate): wgrad_finish for all of them, 40 per launch"""
    for c0 in range(0, len(items), 40):
        chunk = items[c0:c0 + 40]
        arr = (_lib.FinishItem * len(chunk))()
        for k, (slabs, w, grad, perm, taps, I, scale, accumulate) in enumerate(chunk):
            _chk(slabs, f32, "slabs")
            _chk(w, f32, "w")
            S, t_, O, Ipad = slabs.shape
            if t_ != taps or w.shape[0] != O or w.numel() != O * I * taps:
                raise ValueError("wgrad_finish_multi: shape mismatch")
            _chk(grad, f32, "grad", w.shape)
            if perm is not None:
                _chk(perm, torch.int32, "perm", (O,))
            arr[k] = _lib.FinishItem(slabs.data_ptr(), w.data_ptr(), grad.data_ptr(), None if perm is None else perm.data_ptr(),
                                     S, O, I, Ipad, taps, float(scale), int(bool(accumulate)))
        th, td, defer, release = _tables.take(chunk[0][0].device)
        _lib.call("edm_wgrad_finish_multi", ctypes.byref(arr), len(chunk), ctypes.c_void_p(th), ctypes.c_void_p(td), defer,
                  _stream())
        release()


# ------------------------------------------------------------------ attention
def attention_fwd(qkv, heads):
    """qkv (B,H,W,3C) bf16 with per-token channel order [head][q|k|v][d] -> y (B,H,W,C)."""
    B, H, W, C3 = _nhwc(qkv, "qkv")
    C = C3 // 3
    y = torch.empty(B, H, W, C, device=qkv.device, dtype=bf16)
    N = H * W
    with _prof("attention_fwd", 4.0 * B * N * N * C, 2.0 * B * N * 4 * C):
        _lib.call("edm_attention_fwd", _p(qkv), _p(y), B, N, C, heads, _stream())
    return y


def attention_bwd(qkv, y, gy, heads):
    B, H, W, C3 = _nhwc(qkv, "qkv")
    C = C3 // 3
    _chk(y, bf16, "y", (B, H, W, C))
    _chk(gy, bf16, "gy", (B, H, W, C))
    gqkv = torch.empty_like(qkv)
    N = H * W
    with _prof("attention_bwd", 14.0 * B * N * N * C, 2.0 * B * N * 8 * C):
        _lib.call("edm_attention_bwd", _p(qkv), _p(y), _p(gy), _p(gqkv), B, N, C, heads, _stream())
    return gqkv


ATTN_FUSED = os.environ.get("EDM_ATTN_FUSED", "1") != "0"      # qkv projection inside the attention kernels (attention_fused.hip)
ATTN_HP = int(os.environ.get("EDM_ATTN_HP", "0"))               # heads per workgroup of the FORWARD (0 = the kernel's default;
                                                                # the backward always runs one head per workgroup)


def attention_qkv_supported(x, heads):
    """the fused kernels cover this shape (C = 256, 4 heads of 64, 33..256 tokens)"""
    B, H, W, C = x.shape
    return bool(ATTN_FUSED and _lib.call("edm_attention_qkv_supported", H * W, C, heads))


def attention_qkv_fwd(x, wf_qkv, heads, want_stat=True):
    """x (B,H,W,C) bf16, wf_qkv (1, 3C, C) bf16 forward pack of the qkv conv (rows [head][q|k|v][d]) ->
    y (B,H,W,C) bf16 = cosine attention of qkv_conv(x), stat (B, heads, H*W) fp32 or None (networks.py:193-202)"""
    B, H, W, C = _nhwc(x, "x")
    _chk(wf_qkv, bf16, "wf_qkv", (1, 3 * C, C))
    N = H * W
    y = torch.empty_like(x)
    stat = torch.empty(B, heads, N, device=x.device, dtype=f32) if want_stat else None
    with _prof("attention_qkv_fwd", 2.0 * B * N * C * 3 * C + 4.0 * B * N * N * C, 2.0 * B * N * 2 * C):
        _lib.call("edm_attention_qkv_fwd", _p(x), _p(wf_qkv), _p(y), _p(stat), B, N, C, heads, ATTN_HP, _stream())
    return y, stat


def attention_qkv_bwd(x, y, gout, stat, wf_qkv, wd_out, heads, alpha=1.0):
    """-> gqkv (B,H,W,3C) bf16 (packed channel order) = d loss / d qkv_conv(x); gout = d loss / d out_conv(y),
    wd_out (1, C, C) = the out conv's dgrad pack; dO = alpha * gout . W_out is formed inside the kernel"""
    B, H, W, C = _nhwc(x, "x")
    _chk(y, bf16, "y", x.shape)
    _chk(gout, bf16, "gout", x.shape)
    _chk(stat, f32, "stat", (B, heads, H * W))
    _chk(wf_qkv, bf16, "wf_qkv", (1, 3 * C, C))
    _chk(wd_out, bf16, "wd_out", (1, C, C))
    N = H * W
    gqkv = torch.empty(B, H, W, 3 * C, device=x.device, dtype=bf16)
    with _prof("attention_qkv_bwd", 2.0 * B * N * C * 4 * C + 10.0 * B * N * N * C, 2.0 * B * N * 6 * C):
        _lib.call("edm_attention_qkv_bwd", _p(x), _p(y), _p(gout), _p(stat), _p(wf_qkv), _p(wd_out), _p(gqkv), float(alpha),
                  B, N, C, heads, 0, _stream())
    return gqkv


# ------------------------------------------------------------------ fp32 linears / embedding
def linear_fwd(x, w):
    _chk(x, f32, "x")
    _chk(w, f32, "w")
    M, K = x.shape
    N = w.shape[0]
    if w.shape != (N, K):
        raise ValueError("linear_fwd: shape mismatch")
    y = torch.empty(M, N, device=x.device, dtype=f32)
    _lib.call("edm_linear_fwd", _p(x), _p(w), _p(y), M, N, K, _stream())
    return y


def linear_dgrad(dy, w):
    _chk(dy, f32, "dy")
    _chk(w, f32, "w")
    M, N = dy.shape
    K = w.shape[1]
    if w.shape[0] != N:
        raise ValueError("linear_dgrad: shape mismatch")
    # (a pre-zeroed piece of the scratch pool + accumulate: the split-K form of the GEMM adds its partial sums with atomics and
    # would otherwise clear dx with a memset node of its own in every captured step)
    dx = zeros_f32((M, K), dy.device)
    _lib.call("edm_linear_dgrad", _p(dy), _p(w), _p(dx), M, N, K, 1, _stream())
    return dx


def linear_wgrad(dy, x):
    _chk(dy, f32, "dy")
    _chk(x, f32, "x")
    M, N = dy.shape
    K = x.shape[1]
    if x.shape[0] != M:
        raise ValueError("linear_wgrad: shape mismatch")
    dw = torch.empty(N, K, device=dy.device, dtype=f32)
    _lib.call("edm_linear_wgrad", _p(dy), _p(x), _p(dw), M, N, K, 0, _stream())
    return dw


def fourier_fwd(sigma, freqs, phases, B):
    ss = _sigma_arg(sigma, B)
    _chk(freqs, f32, "freqs")
    _chk(phases, f32, "phases", freqs.shape)
    out = torch.empty(B, freqs.numel(), device=freqs.device, dtype=f32)
    _lib.call("edm_fourier_fwd", _p(sigma), ss, _p(freqs), _p(phases), _p(out), B, freqs.numel(), _stream())
    return out


def _labels_arg(labels, B):
    if labels is None:
        return None
    if not labels.is_cuda:
        raise RuntimeError("labels must be a CUDA/HIP tensor")
    labels = labels.flatten().to(torch.int64).contiguous()
    if labels.numel() != B:
        raise ValueError(f"labels must have {B} elements")
    return labels


def embed_combine_fwd(emb_sigma, wcls_hat, labels, add_factor):
    _chk(emb_sigma, f32, "emb_sigma")
    B, E = emb_sigma.shape
    K = 0
    if labels is not None:
        _chk(wcls_hat, f32, "wcls_hat")
        K = wcls_hat.shape[1]
    pre, out = torch.empty_like(emb_sigma), torch.empty_like(emb_sigma)
    _lib.call("edm_embed_combine_fwd", _p(emb_sigma), _p(wcls_hat), _p(labels), float(add_factor), K, _p(pre), _p(out), B,
              E, _stream())
    return pre, out


def embed_combine_bwd(gout, pre, labels, add_factor, wcls_shape):
    _chk(gout, f32, "gout", pre.shape)
    B, E = pre.shape
    ges = torch.empty_like(pre)
    gw, K = None, 0
    if labels is not None:
        gw = zeros_f32(wcls_shape, pre.device)
        K = wcls_shape[1]
    _lib.call("edm_embed_combine_bwd", _p(gout), _p(pre), _p(labels), float(add_factor), K, _p(ges), _p(gw), B, E, _stream())
    return ges, gw


# ------------------------------------------------------------------ step-level kernels
def diffuse(clean, P_mean, P_std, seed, step, dyn=None):
    _chk(clean, f32, "clean")
    B = clean.shape[0]
    noisy = torch.empty_like(clean)
    sigma = torch.empty(B, device=clean.device, dtype=f32)
    _lib.call("edm_diffuse", _p(clean), _p(noisy), _p(sigma), float(P_mean), float(P_std), B, clean.numel() // B, int(seed),
              int(step), _dyn(dyn), _stream())
    return noisy, sigma


def diffuse_given(clean, eps, noise, P_mean, P_std):
    _chk(clean, f32, "clean")
    B = clean.shape[0]
    _chk(eps, f32, "eps", (B,))
    _chk(noise, f32, "noise", clean.shape)
    noisy = torch.empty_like(clean)
    sigma = torch.empty(B, device=clean.device, dtype=f32)
    _lib.call("edm_diffuse_given", _p(clean), _p(eps), _p(noise), _p(noisy), _p(sigma), float(P_mean), float(P_std), B,
              clean.numel() // B, _stream())
    return noisy, sigma


def weighted_mse(D, clean, sigma, sigma_data, weight=None, want_grad=True, acc_sum=None, acc_total=None):
    """acc_sum (fp32, 1 element) / acc_total (int64, 1 element): the metric's epoch state, accumulated in the same pass"""
    _chk(D, f32, "D")
    _chk(clean, f32, "clean", D.shape)
    B = D.shape[0]
    if weight is not None:
        _chk(weight, f32, "weight", (B,))
    else:
        _chk(sigma, f32, "sigma", (B,))
    loss = zeros_f32((), D.device)
    dD = torch.empty_like(D) if want_grad else None
    if acc_sum is not None:
        _chk(acc_sum, f32, "acc_sum")
        _chk(acc_total, torch.int64, "acc_total")
    _lib.call("edm_weighted_mse", _p(D), _p(clean), _p(sigma), _p(weight), float(sigma_data), _p(loss), _p(dD), B,
              D.numel() // B, _p(acc_sum), _p(acc_total), _stream())
    return loss, dD


def adam_ema(theta, grad, m, v, ema, lr, b1, b2, eps, step, ema_beta, grad_scale=1.0, dyn=None, zero_grad=False):
    for t, nme in ((theta, "theta"), (grad, "grad"), (m, "m"), (v, "v")):
        _chk(t, f32, nme)
        if t.numel() != theta.numel():
            raise ValueError("adam_ema: arena size mismatch")
    if ema is not None:
        _chk(ema, f32, "ema")
    _lib.call("edm_adam_ema", _p(theta), _p(grad), _p(m), _p(v), _p(ema), theta.numel(), float(lr), float(b1), float(b2),
              float(eps), int(step), float(ema_beta), float(grad_scale), _dyn(dyn), int(bool(zero_grad)),
              _p(health(theta.device)), _stream())


def heun_euler(x, D, t0, t1):
    _chk(x, f32, "x")
    _chk(D, f32, "D", x.shape)
    dx, x1 = torch.empty_like(x), torch.empty_like(x)
    _lib.call("edm_heun_euler", _p(x), _p(D), float(t0), float(t1), _p(dx), _p(x1), x.numel(), _p(health(x.device)),
              _stream())
    return dx, x1


def heun_correct(x, dx, x1, D1, t0, t1):
    _chk(x, f32, "x")
    out = torch.empty_like(x)
    _lib.call("edm_heun_correct", _p(x), _p(dx), _p(x1), _p(D1), float(t0), float(t1), _p(out), x.numel(),
              _p(health(x.device)), _stream())
    return out


def scale_f32(x, s):
    _chk(x, f32, "x")
    y = torch.empty_like(x)
    _lib.call("edm_scale_f32", _p(x), float(s), _p(y), x.numel(), _stream())
    return y


# ------------------------------------------------------------------ reference-precision evaluation (csrc/eval_f32.hip)
def _nhwc32(t, name):
    _chk(t, f32, name)
    if t.dim() != 4:
        raise ValueError(f"{name}: expected (B,H,W,C)")
    return t.shape


def f32_conv(x, w_hat, taps, residual=None, alpha=1.0, beta=0.0, lin=None, gain=None):
    """x (B,H,W,Cin) fp32 NHWC, w_hat (Cout, I*taps) fp32 (master OIHW order, I <= Cin) -> (B,H,W,Cout) fp32:
    alpha*conv + beta*residual, or with lin/gain the block's modulation epilogue mp_silu(alpha*conv*(lin*gain+1))."""
    B, H, W, Cin = _nhwc32(x, "x")
    _chk(w_hat, f32, "w_hat")
    if w_hat.dim() != 2 or w_hat.shape[1] % taps:
        raise ValueError(f"f32_conv: w_hat {tuple(w_hat.shape)} does not match taps={taps}")
    Cout, I = w_hat.shape[0], w_hat.shape[1] // taps
    if I > Cin:
        raise ValueError(f"f32_conv: weight has {I} input channels, x only {Cin}")
    if residual is not None:
        _chk(residual, f32, "residual", (B, H, W, Cout))
    ls = 0
    if lin is not None:
        ls = _lin_view(lin, B, Cout, "lin")
        _chk(gain, f32, "gain")
    y = torch.empty(B, H, W, Cout, device=x.device, dtype=f32)
    with _prof("f32_conv3x3" if taps == 9 else "f32_conv1x1", 2.0 * B * H * W * I * Cout * taps,
               4.0 * (B * H * W * (Cin + Cout * (2 if residual is not None else 1)) + w_hat.numel())):
        _lib.call("edm_f32_conv", _p(x), _p(w_hat), _p(y), _p(residual), float(alpha), float(beta), _p(lin), ls, _p(gain),
                  B, H, W, Cin, I, Cout, taps, _stream())
    return y


def f32_to_pairs(x):
    """(B,H,W,C) fp32 -> (B,H,W,2C) bf16 = [hi | lo], hi = bf16(x), lo = bf16(x - hi): the operand format of split_conv"""
    B, H, W, C = _nhwc32(x, "x")
    if C % 8:
        raise ValueError("f32_to_pairs: C % 8 required")
    p = torch.empty(B, H, W, 2 * C, device=x.device, dtype=bf16)
    _lib.call("edm_f32_to_pairs", _p(x), _p(p), B * H * W, C, _stream())
    return p


def split_pack(w_hat, taps, out=None):
    """w_hat (Cout, I*taps) fp32 -> (taps, Cout, 3*Ip) bf16 = [w_hi | w_lo | w_hi], Ip = I rounded up to 32.
    out: rewrite this buffer in place (the persistent pack of a weight-prep plan: a captured solve keeps its address)"""
    _chk(w_hat, f32, "w_hat")
    Cout, I = w_hat.shape[0], w_hat.shape[1] // taps
    Ip = (I + 31) // 32 * 32
    pk = torch.empty(taps, Cout, 3 * Ip, device=w_hat.device, dtype=bf16) if out is None else \
        _chk(out, bf16, "out", (taps, Cout, 3 * Ip))
    _lib.call("edm_split_pack", _p(w_hat), _p(pk), Cout, I, taps, Ip, _stream())
    return pk


def split_conv_fold_supported(xp_shape, Cout, C2):
    """split_conv(fold=) covers this 3x3 shape ((B, H, W, 2C) pairs -> Cout) with a C2-channel projection riding along"""
    B, H, W, C2x = xp_shape
    C = C2x // 2
    return bool(FOLD_PROJ and C % 64 == 0 and C2 % 32 == 0 and
                _lib.call("edm_conv3x3_fold_supported", B, H, W, 3 * C, Cout, 3 * C2))


def split_conv(xp, pack3, taps, residual=None, alpha=1.0, beta=0.0, lin=None, gain=None, pairs_out=False, also_pairs=False,
               dest=None, silu_pairs=False, fold=None):
    """fp32-accurate conv in three bf16 MFMA passes: xp (B,H,W,2C) bf16 pairs (f32_to_pairs), pack3 (taps,Cout,3C) bf16
    (split_pack) -> (B,H,W,Cout) fp32 = alpha*conv + beta*residual, or with lin/gain mp_silu(alpha*conv*(lin*gain+1));
    pairs_out=True: the result comes back as (B,H,W,2Cout) bf16 pairs instead (it only feeds another split_conv);
    also_pairs=True: (fp32 result, the same as pairs) from the one launch;
    silu_pairs=True: (fp32 result, mp_silu of it as pairs): what a decoder block without a skip reads (networks.py:313-316);
    dest=(cat, sil): (B,H,W,2 Ct) bf16 pairs buffers of the NEXT decoder block's concatenated operands, Ct > Cout: the result
    and mp_silu of it are written into their left column blocks (hi at columns [0, Cout), lo at [Ct, Ct + Cout)) and nothing
    else; returns cat (f32_skip_half fills the right blocks)."""
    B, H, W, C2 = _nhwc(xp, "xp")
    C = C2 // 2
    _chk(pack3, bf16, "pack3")
    if pack3.dim() != 3 or pack3.shape[0] != taps or pack3.shape[2] != 3 * C or C % 32:
        raise ValueError(f"split_conv: pack {tuple(pack3.shape)} does not match taps={taps}, C={C} (C % 32 required)")
    Cout = pack3.shape[1]
    if residual is not None:
        _chk(residual, f32, "residual", (B, H, W, Cout))
    ls = 0
    if lin is not None:
        ls = _lin_view(lin, B, Cout, "lin")
        _chk(gain, f32, "gain")
    if sum(map(bool, (pairs_out, also_pairs, dest is not None, silu_pairs))) > 1:
        raise ValueError("split_conv: pairs_out / also_pairs / silu_pairs / dest exclude one another")
    if fold is not None:
        # fold = (x2p, pack3_1x1): alpha * conv3x3(xp) + beta * conv1x1(x2p) in one launch (the decoder's skip projection,
        # edm_split_conv_fold); `beta` is the projection's coefficient, there is no residual
        x2p, pk1 = fold
        if residual is not None or lin is not None or taps != 9:
            raise ValueError("split_conv: fold excludes residual / modulation and needs taps == 9")
        B2, H2, W2, C22 = _nhwc(x2p, "fold[0]")
        C2 = C22 // 2
        _chk(pk1, bf16, "fold[1]", (1, Cout, 3 * C2))
        if (B2, H2, W2) != (B, H, W):
            raise ValueError("split_conv: fold operand shape mismatch")
        y = yp = ys = None
        ld = lo = 0
        ret = None
        if dest is not None:
            cat, sil = dest
            Ct = cat.shape[-1] // 2
            _chk(cat, bf16, "dest[0]", (B, H, W, 2 * Ct))
            if sil is not None:
                _chk(sil, bf16, "dest[1]", (B, H, W, 2 * Ct))
            if Ct <= Cout or Ct % 4:
                raise ValueError("split_conv: dest must be wider than the result")
            yp, ys, ld, lo, ret = cat, sil, 2 * Ct, Ct, cat
        elif pairs_out:
            ret = yp = torch.empty(B, H, W, 2 * Cout, device=xp.device, dtype=bf16)
        else:
            y = torch.empty(B, H, W, Cout, device=xp.device, dtype=f32)
            if also_pairs or silu_pairs:
                p2 = torch.empty(B, H, W, 2 * Cout, device=xp.device, dtype=bf16)
                yp, ys = (p2, None) if also_pairs else (None, p2)
                ret = (y, p2)
            else:
                ret = y
        with _prof("split_conv3x3_fold", 2.0 * B * H * W * Cout * (9 * C + C2), 4.0 * B * H * W * (C + C2 + Cout) + 2.0 * pack3.numel()):
            _lib.call("edm_split_conv_fold", _p(xp), _p(pack3), _p(x2p), _p(pk1), C2, _p(y), _p(yp), ld, lo, _p(ys), float(alpha),
                      float(beta), B, H, W, C, Cout, _stream())
        return ret
    nb = 4.0 * B * H * W * (C + Cout * (2 if residual is not None else 1)) + 2.0 * pack3.numel()
    pname = "split_conv3x3" if taps == 9 else "split_conv1x1"
    if dest is not None or also_pairs or silu_pairs:
        y = yp = ys = None
        ld = lo = 0
        if dest is not None:
            cat, sil = dest
            Ct = cat.shape[-1] // 2
            _chk(cat, bf16, "dest[0]", (B, H, W, 2 * Ct))
            if sil is not None:
                _chk(sil, bf16, "dest[1]", (B, H, W, 2 * Ct))
            if Ct <= Cout or Ct % 4:
                raise ValueError("split_conv: dest must be wider than the result")
            yp, ys, ld, lo = cat, sil, 2 * Ct, Ct
        else:
            y = torch.empty(B, H, W, Cout, device=xp.device, dtype=f32)
            p2 = torch.empty(B, H, W, 2 * Cout, device=xp.device, dtype=bf16)
            yp, ys = (p2, None) if also_pairs else (None, p2)
        with _prof(pname, 2.0 * B * H * W * C * Cout * taps, nb):
            _lib.call("edm_split_conv_o", _p(xp), _p(pack3), _p(y), _p(yp), ld, lo, _p(ys), _p(residual), float(alpha),
                      float(beta), _p(lin), ls, _p(gain), B, H, W, C, Cout, taps, _stream())
        return cat if dest is not None else (y, p2)
    y = None if pairs_out else torch.empty(B, H, W, Cout, device=xp.device, dtype=f32)
    yp = torch.empty(B, H, W, 2 * Cout, device=xp.device, dtype=bf16) if pairs_out else None
    with _prof(pname, 2.0 * B * H * W * C * Cout * taps, nb):
        _lib.call("edm_split_conv", _p(xp), _p(pack3), _p(y), _p(yp), _p(residual), float(alpha), float(beta), _p(lin), ls,
                  _p(gain), B, H, W, C, Cout, taps, _stream())
    return yp if pairs_out else y


def f32_skip_half(skip, gate, cat, sil=None):
    """cat[..., Ci:Ct] (hi) / [Ct+Ci:2Ct] (lo) = pairs of skip*gate, sil likewise of mp_silu(skip*gate): the skip half of the
    split evaluation's concatenated operands (the input half: split_conv(dest=))"""
    B, H, W, Cs = _nhwc32(skip, "skip")
    _chk(gate, f32, "gate", (B, Cs))
    Ct = cat.shape[-1] // 2
    _chk(cat, bf16, "cat", (B, H, W, 2 * Ct))
    if sil is not None:
        _chk(sil, bf16, "sil", (B, H, W, 2 * Ct))
    if Ct <= Cs:
        raise ValueError("f32_skip_half: cat must be wider than the skip")
    _lib.call("edm_f32_skip_half", _p(skip), _p(gate), _p(cat), _p(sil), B, H * W, Ct - Cs, Cs, _stream())


def f32_pool_pixelnorm_silu(x, pairs=False):
    """(xn, s) of the 2x2-average-pooled x in one pass (== f32_pixelnorm_silu(f32_pool2(x)), bit for bit)"""
    B, H, W, C = _nhwc32(x, "x")
    if H % 2 or W % 2 or C > 1024:
        raise ValueError("f32_pool_pixelnorm_silu: H, W even and C <= 1024 required")
    xn = torch.empty(B, H // 2, W // 2, C, device=x.device, dtype=f32)
    s = torch.empty(B, H // 2, W // 2, 2 * C, device=x.device, dtype=bf16) if pairs else torch.empty_like(xn)
    _lib.call("edm_f32_pool_pixelnorm_silu", _p(x), _p(xn), _p(s), B, H // 2, W // 2, C, int(bool(pairs)), _stream())
    return xn, s


def f32_up2_silu(x, pairs=False):
    """(y, s): y = nearest-exact x2 of x, s = mp_silu(y) as fp32 or pairs, one pass"""
    B, H, W, C = _nhwc32(x, "x")
    y = torch.empty(B, 2 * H, 2 * W, C, device=x.device, dtype=f32)
    s = torch.empty(B, 2 * H, 2 * W, 2 * C, device=x.device, dtype=bf16) if pairs else torch.empty_like(y)
    _lib.call("edm_f32_up2_silu", _p(x), _p(y), _p(s), B, 2 * H, 2 * W, C, int(bool(pairs)), _stream())
    return y, s


def f32_attention(qkv, heads):
    B, H, W, C3 = _nhwc32(qkv, "qkv")
    C = C3 // 3
    y = torch.empty(B, H, W, C, device=qkv.device, dtype=f32)
    N = H * W
    with _prof("f32_attention", 4.0 * B * N * N * C, 4.0 * B * N * 4 * C):
        _lib.call("edm_f32_attention", _p(qkv), _p(y), B, N, C, heads, _stream())
    return y


def split_attention_ok(C, heads, N):
    """shapes edm_split_attention covers (attention_split.hip): head_dim 64, at most 256 tokens"""
    return C % heads == 0 and C // heads == 64 and N <= 256


def split_attention(qkv, heads, pairs=False):
    """fp32-accurate attention on the bf16 matrix cores (three MFMA passes over hi/lo pairs): qkv (B,H,W,3C) fp32 in the qkv
    conv's own channel order -> (B,H,W,C) fp32, or -- pairs=True -- (B,H,W,2C) bf16 [hi | lo] pairs for the out conv"""
    B, H, W, C3 = _nhwc32(qkv, "qkv")
    C = C3 // 3
    N = H * W
    if not split_attention_ok(C, heads, N):
        raise ValueError(f"split_attention: head_dim {C // heads}, {N} tokens not covered (head_dim 64, <= 256 tokens)")
    y = torch.empty(B, H, W, 2 * C, device=qkv.device, dtype=bf16) if pairs else torch.empty(B, H, W, C, device=qkv.device, dtype=f32)
    with _prof("split_attention", 3 * 4.0 * B * N * N * C, 4.0 * B * N * 4 * C):
        _lib.call("edm_split_attention", _p(qkv), None if pairs else _p(y), _p(y) if pairs else None, B, N, C, heads, _stream())
    return y


def f32_pixelnorm_silu(x, pairs=False):
    """-> (xn fp32, s): s = mp_silu(xn) as fp32, or -- pairs=True -- as (B,H,W,2C) bf16 split pairs (split_conv's operand)"""
    B, H, W, C = _nhwc32(x, "x")
    xn = torch.empty_like(x)
    s = torch.empty(B, H, W, 2 * C, device=x.device, dtype=bf16) if pairs else torch.empty_like(x)
    _lib.call("edm_f32_pixelnorm_silu", _p(x), _p(xn), _p(s), B * H * W, C, int(bool(pairs)), _stream())
    return xn, s


def f32_silu(x, pairs=False):
    _chk(x, f32, "x")
    C = x.shape[-1]
    s = torch.empty(*x.shape[:-1], 2 * C, device=x.device, dtype=bf16) if pairs else torch.empty_like(x)
    _lib.call("edm_f32_silu", _p(x), _p(s), x.numel(), C if pairs else 0, _stream())
    return s


def f32_pool2(x):
    B, H, W, C = _nhwc32(x, "x")
    if H % 2 or W % 2:
        raise ValueError("f32_pool2: H and W must be even")
    y = torch.empty(B, H // 2, W // 2, C, device=x.device, dtype=f32)
    _lib.call("edm_f32_pool2", _p(x), _p(y), B, H // 2, W // 2, C, _stream())
    return y


def f32_up2(x):
    B, H, W, C = _nhwc32(x, "x")
    y = torch.empty(B, 2 * H, 2 * W, C, device=x.device, dtype=f32)
    _lib.call("edm_f32_up2", _p(x), _p(y), B, 2 * H, 2 * W, C, _stream())
    return y


def f32_skip_gate(skip, w1h, w2h):
    B, H, W, C = _nhwc32(skip, "skip")
    R = w1h.shape[0]
    _chk(w1h, f32, "w1h", (R, C + 1))
    _chk(w2h, f32, "w2h", (C, R))
    gate = torch.empty(B, C, device=skip.device, dtype=f32)
    _lib.call("edm_f32_skip_gate", _p(skip), _p(w1h), _p(w2h), _p(gate), B, H * W, C, R, _stream())
    return gate


def f32_concat_gate(inp, skip, gate, want_silu, pairs=False):
    """pairs=True: cat and sil come back as (B,H,W,2(Ci+Cs)) bf16 split pairs (they only feed convolutions)"""
    B, H, W, Ci = _nhwc32(inp, "inp")
    Bs, Hs, Ws, Cs = _nhwc32(skip, "skip")
    if (Bs, Hs, Ws) != (B, H, W):
        raise ValueError("f32_concat_gate: inp/skip spatial mismatch")
    _chk(gate, f32, "gate", (B, Cs))
    if pairs:
        cat = torch.empty(B, H, W, 2 * (Ci + Cs), device=inp.device, dtype=bf16)
    else:
        cat = torch.empty(B, H, W, Ci + Cs, device=inp.device, dtype=f32)
    sil = torch.empty_like(cat) if want_silu else None
    _lib.call("edm_f32_concat_gate", _p(inp), _p(skip), _p(gate), _p(cat), _p(sil), B, H * W, Ci, Cs, int(bool(pairs)),
              _stream())
    return cat, sil


def f32_precond_in(noisy, sigma, sigma_data, CP):
    _chk(noisy, f32, "noisy")
    B, Cimg, H, W = noisy.shape
    ss = _sigma_arg(sigma, B)
    out = torch.empty(B, H, W, CP, device=noisy.device, dtype=f32)
    _lib.call("edm_f32_precond_in", _p(noisy), _p(sigma), ss, float(sigma_data), _p(out), B, Cimg, H * W, CP, _stream())
    return out


def f32_conv_out(x, w_hat, gain_out, noisy, sigma, sigma_data):
    B, H, W, C = _nhwc32(x, "x")
    Co = w_hat.shape[0]
    _chk(w_hat, f32, "w_hat", (Co, C))
    _chk(noisy, f32, "noisy", (B, Co, H, W))
    _chk(gain_out, f32, "gain_out")
    ss = _sigma_arg(sigma, B)
    D = torch.empty(B, Co, H, W, device=x.device, dtype=f32)
    _lib.call("edm_f32_conv_out", _p(x), _p(w_hat), _p(gain_out), _p(noisy), _p(sigma), ss, float(sigma_data), _p(D), B,
              H * W, C, Co, _stream())
    return D


def f32_nchw_to_nhwc(x):
    _chk(x, f32, "x")
    B, C, H, W = x.shape
    y = torch.empty(B, H, W, C, device=x.device, dtype=f32)
    _lib.call("edm_f32_nchw_to_nhwc", _p(x), _p(y), B, C, H * W, _stream())
    return y


def f32_nhwc_to_nchw(x):
    B, H, W, C = _nhwc32(x, "x")
    y = torch.empty(B, C, H, W, device=x.device, dtype=f32)
    _lib.call("edm_f32_nhwc_to_nchw", _p(x), _p(y), B, C, H * W, _stream())
    return y


# ------------------------------------------------------------------ data formats (csrc/data.hip)
def u8_gather_normalize(data, index, mean=0.5, std=0.5, flip=False, seed=0, epoch=0):
    """data uint8 (N,C,H,W) resident on the device, index int64 (B,) -> fp32 (B,C,H,W) = (x/255-mean)/std,
    optionally flipped left-right per sample (Philox(seed; epoch, b))."""
    _chk(data, torch.uint8, "data")
    _chk(index, torch.int64, "index")
    if data.dim() != 4 or index.dim() != 1:
        raise ValueError("u8_gather_normalize: data must be (N,C,H,W), index (B,)")
    N, C, H, W = data.shape
    B = index.shape[0]
    out = torch.empty(B, C, H, W, device=data.device, dtype=f32)
    _lib.call("edm_u8_gather_normalize", _p(data), _p(index), _p(out), B, C, H, W, N, float(mean), float(std),
              int(bool(flip)), int(seed) & 0xFFFFFFFFFFFFFFFF, int(epoch) & 0xFFFFFFFF, _stream())
    return out


def denormalize_u8(x, scale=127.5, offset=128.0):
    """fp32 tensor -> uint8, same shape: (x*scale+offset).clip(0,255) truncated."""
    _chk(x, f32, "x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    _lib.call("edm_denormalize_u8", _p(x), _p(out), x.numel(), float(scale), float(offset), _stream())
    return out


def prediction_to_u8_nhwc(pred, mean, std):
    """fp32 (B,C,H,W) -> uint8 (B,H,W,C): clamp(pred*std*2+mean, 0, 1)*255 truncated (per-channel mean/std tensors)."""
    _chk(pred, f32, "pred")
    B, C, H, W = pred.shape
    _chk(mean, f32, "mean", (C,))
    _chk(std, f32, "std", (C,))
    out = torch.empty(B, H, W, C, device=pred.device, dtype=torch.uint8)
    _lib.call("edm_prediction_to_u8_nhwc", _p(pred), _p(out), B, C, H, W, _p(mean), _p(std), _stream())
    return out

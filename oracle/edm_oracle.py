"""CPU oracle for the EDM/EDM2 hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-torch, functional restatement of the reference algorithm
(YichengDWu/tinyedm).  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The product path
(``tinyedm_amd``) never routes through it and has no CPU fallback.

Pinning: the fp32 mode of every function below is checked in
``tests/test_oracle_golden.py`` against golden vectors produced by importing
the reference's own ``networks.py`` / ``solvers.py`` (script:
``oracle/make_golden.py``; fixtures: ``tests/golden/*.npz``) and against the
closed-form known-answer values of SURVEY.md section 8(c).

Parameters live in a flat ``dict[str, Tensor]`` whose keys and layouts are the
reference's ``state_dict`` keys (OIHW fp32 conv weights, (out,in) linears).

``bf16=True`` inserts a round-to-bf16 (``q``) at exactly the points where the
HIP path materialises a bf16 tensor in HBM (conv operands/outputs, activation
tensors); accumulation stays fp32.  ``bf16=False`` is the fp32 reference
semantics.

Reference citations are ``file:line`` relative to ``/root/reference/src/tinyedm``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
EPS = 1e-4
SILU_DIV = 0.596


# --------------------------------------------------------------------------
# configuration records (mirror the ctor kwargs of networks.py:145-161, 491-506)
# --------------------------------------------------------------------------
@dataclass
class EmbeddingCfg:
    fourier_dim: int
    embedding_dim: int
    num_classes: Optional[int] = None
    add_factor: float = 0.5


@dataclass
class DenoiserCfg:
    in_channels: int = 3
    out_channels: int = 3
    encoder_block_types: Sequence[str] = ()
    decoder_block_types: Sequence[str] = ()
    encoder_out_channels: Sequence[int] = ()
    decoder_out_channels: Sequence[int] = ()
    skip_connections: Sequence[bool] = ()
    dropout_rate: float = 0.0
    sigma_data: float = 0.5
    encoder_add_factor: float = 0.3
    decoder_add_factor: float = 0.3
    embedding_dim: int = 768
    num_heads: int = 4

    def skip_channels(self) -> List[int]:
        """networks.py:435-444 -- reversed encoder widths + the input block,
        scattered onto the True skip flags."""
        pool = list(self.encoder_out_channels)[::-1] + [self.encoder_out_channels[0]]
        out, it = [], iter(pool)
        for flag in self.skip_connections:
            out.append(next(it) if flag else 0)
        return out


def cifar10_cfg(num_classes: Optional[int] = None) -> Tuple[EmbeddingCfg, DenoiserCfg]:
    """experiments/conf/cifar10.yaml:26-44."""
    e = EmbeddingCfg(64, 256, num_classes)
    d = DenoiserCfg(
        3, 3,
        ["Enc", "Enc", "EncD", "EncA", "EncA", "EncD", "EncA", "EncA"],
        ["DecA", "Dec", "DecA", "DecA", "DecA", "DecU", "DecA", "DecA", "DecA", "DecU", "Dec", "Dec", "Dec"],
        [256] * 8, [256] * 13,
        [False, False, True, True, True, False, True, True, True, False, True, True, True],
        dropout_rate=0.13, sigma_data=0.5, embedding_dim=256, num_heads=4)
    return e, d


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def q_bf16(x: Tensor) -> Tensor:
    """Round-to-nearest-even to bf16, kept in fp32 storage (straight-through grad)."""
    return x + (x.detach().to(torch.bfloat16).to(x.dtype) - x.detach())


def _ident(x: Tensor) -> Tensor:
    return x


def rms_div(x: Tensor, dims: Sequence[int], eps: float = EPS) -> Tensor:
    """networks.py:9-14: x / (eps + ||x||_2 * sqrt(1/len(dims-slice)))."""
    n = torch.linalg.vector_norm(x.float(), dim=list(dims), keepdim=True)
    cnt = 1
    for d in dims:
        cnt *= x.shape[d]
    return x / (eps + n * np.float32(1.0 / math.sqrt(cnt)))


def weight_normalize(w: Tensor, eps: float = EPS) -> Tensor:
    """networks.py:17-19: per-output-row normalisation over all other dims."""
    return rms_div(w, list(range(1, w.ndim)), eps)


def effective_weight(w: Tensor) -> Tensor:
    """networks.py:35-36 / 58-59: normalize(w) / sqrt(fan_in)."""
    fan_in = w[0].numel()
    return weight_normalize(w) / np.float32(math.sqrt(fan_in))


def mp_silu(x: Tensor) -> Tensor:
    """networks.py:83-84."""
    return F.silu(x) / SILU_DIV


def mp_add(a: Tensor, b: Tensor, t: float = 0.5) -> Tensor:
    """networks.py:87-88."""
    return (a + (b - a) * t) / math.sqrt((1 - t) ** 2 + t ** 2)


def wn_conv(x: Tensor, w: Tensor, q=_ident) -> Tensor:
    """networks.py:35-37 (bias-free, padding=same); q rounds operands and output."""
    k = w.shape[-1]
    return q(F.conv2d(q(x), q(effective_weight(w)), padding=k // 2))


def wn_linear(x: Tensor, w: Tensor) -> Tensor:
    """networks.py:58-60 (always fp32 on the hot path: networks.py:164,255,319)."""
    return F.linear(x, effective_weight(w))


def force_normalize_(P: Dict[str, Tensor]) -> None:
    """Training-forward side effect networks.py:32-34 / 55-57 applied to every
    Conv2d/Linear weight (each weight is used by exactly one layer per forward,
    so doing them all up front is equivalent)."""
    with torch.no_grad():
        for k, v in P.items():
            if k.endswith(".weight"):
                v.copy_(weight_normalize(v))


# --------------------------------------------------------------------------
# parameter construction (shapes / names = reference state_dict)
# --------------------------------------------------------------------------
def _block_flags(t: str) -> Tuple[bool, bool, bool]:
    return t.endswith("D"), t.endswith("U"), t.endswith("A")


def init_params(ecfg: EmbeddingCfg, dcfg: DenoiserCfg, gen: torch.Generator,
                gains_nonzero: bool = True) -> Dict[str, Tensor]:
    """Same tensors as instantiating tinyedm.Embedding + tinyedm.Denoiser
    (networks.py:28,49,135-136,244,304,538); values drawn from ``gen``.
    ``gains_nonzero`` draws gain_out / block gains away from their 0 / 1 init so
    that fixtures exercise the whole network."""
    P: Dict[str, Tensor] = {}

    def rn(*shape):
        return torch.randn(*shape, generator=gen)

    P["embedding.fourier_embed.freqs"] = 2 * math.pi * rn(ecfg.fourier_dim)
    P["embedding.fourier_embed.phases"] = 2 * math.pi * torch.rand(ecfg.fourier_dim, generator=gen)
    P["embedding.sigma_embed.weight"] = rn(ecfg.embedding_dim, ecfg.fourier_dim)
    if ecfg.num_classes is not None and ecfg.num_classes != -1:
        P["embedding.class_embed.linear.weight"] = rn(ecfg.embedding_dim, ecfg.num_classes)

    def gain(init):
        if gains_nonzero:
            return torch.tensor(init) + 0.5 + 0.25 * rn(())
        return torch.tensor(float(init))

    enc_c, dec_c = list(dcfg.encoder_out_channels), list(dcfg.decoder_out_channels)
    P["denoiser.gain_out"] = gain(0.0)
    P["denoiser.conv_in.weight"] = rn(enc_c[0], dcfg.in_channels + 1, 3, 3)
    P["denoiser.conv_out.weight"] = rn(dcfg.out_channels, dec_c[-1], 1, 1)

    cin = enc_c[0]
    for i, (t, cout) in enumerate(zip(dcfg.encoder_block_types, enc_c)):
        p = f"denoiser.encoder_blocks.{i}."
        _, _, attn = _block_flags(t)
        if cin != cout:
            P[p + "conv_1x1.weight"] = rn(cout, cin, 1, 1)
        P[p + "conv_3x3_1.weight"] = rn(cout, cout, 3, 3)
        P[p + "conv_3x3_2.weight"] = rn(cout, cout, 3, 3)
        if attn:
            P[p + "attention.qkv_conv.weight"] = rn(3 * cout, cout, 1, 1)
            P[p + "attention.out_conv.weight"] = rn(cout, cout, 1, 1)
        P[p + "embed.weight"] = rn(cout, dcfg.embedding_dim)
        P[p + "gain"] = gain(1.0) if gains_nonzero else torch.tensor(1.0)
        cin = cout

    cin = dec_c[0]
    for i, (t, cout, sc) in enumerate(zip(dcfg.decoder_block_types, dec_c, dcfg.skip_channels())):
        p = f"denoiser.decoder_blocks.{i}."
        _, _, attn = _block_flags(t)
        tot = cin + sc
        if sc > 0:
            P[p + "cat_factor.layer1.weight"] = rn(sc // 16, sc + 1, 1, 1)
            P[p + "cat_factor.layer2.weight"] = rn(sc, sc // 16, 1, 1)
        if tot != cout:
            P[p + "conv_1x1.weight"] = rn(cout, tot, 1, 1)
        P[p + "conv_3x3_1.weight"] = rn(cout, tot, 3, 3)
        P[p + "conv_3x3_2.weight"] = rn(cout, cout, 3, 3)
        if attn:
            P[p + "attention.qkv_conv.weight"] = rn(3 * cout, cout, 1, 1)
            P[p + "attention.out_conv.weight"] = rn(cout, cout, 1, 1)
        P[p + "embed.weight"] = rn(cout, dcfg.embedding_dim)
        P[p + "gain"] = gain(1.0) if gains_nonzero else torch.tensor(1.0)
        cin = cout
    return P


def trainable_keys(P: Dict[str, Tensor]) -> List[str]:
    return [k for k in P if not (k.endswith("freqs") or k.endswith("phases"))]


# --------------------------------------------------------------------------
# Embedding (networks.py:121-178) -- always fp32
# --------------------------------------------------------------------------
def fourier_features(P, sigma_log4: Tensor) -> Tensor:
    """networks.py:138-141."""
    y = torch.outer(sigma_log4.flatten(), P["embedding.fourier_embed.freqs"]) + P["embedding.fourier_embed.phases"]
    return y.cos() * np.float32(math.sqrt(2.0))


def embedding_forward(P, ecfg: EmbeddingCfg, sigma: Tensor, labels: Optional[Tensor] = None):
    """networks.py:163-178 -> (fourier, emb)."""
    c_noise = sigma.float().log() / 4
    four = fourier_features(P, c_noise)
    emb = wn_linear(four, P["embedding.sigma_embed.weight"])
    if labels is not None:
        if "embedding.class_embed.linear.weight" not in P:
            raise ValueError("class_labels is not None, but num_classes is None. ")
        K = P["embedding.class_embed.linear.weight"].shape[1]
        onehot = F.one_hot(labels.flatten(), K).float() * np.float32(math.sqrt(K))
        emb = mp_add(emb, wn_linear(onehot, P["embedding.class_embed.linear.weight"]), ecfg.add_factor)
    return four, mp_silu(emb)


# --------------------------------------------------------------------------
# U-Net pieces
# --------------------------------------------------------------------------
def scale_long_gate(P, prefix: str, skip: Tensor) -> Tensor:
    """networks.py:112-118.  (B,C,H,W) -> (B,C,1,1).  The HIP path keeps this
    tiny per-sample MLP in fp32, so no q() here in either mode."""
    m = skip.float().mean(dim=(2, 3), keepdim=True)
    m = torch.cat((m, torch.ones_like(m[:, :1])), dim=1)
    h = mp_silu(F.conv2d(m, effective_weight(P[prefix + "layer1.weight"])))
    return torch.sigmoid(F.conv2d(h, effective_weight(P[prefix + "layer2.weight"])))


def cosine_attention(P, prefix: str, x: Tensor, heads: int, q=_ident) -> Tensor:
    """networks.py:191-207."""
    b, c, h, w = x.shape
    d = c // heads
    qkv = wn_conv(x, P[prefix + "qkv_conv.weight"], q)
    qkv = qkv.view(b, heads, d, 3, h * w)
    qkv = q(rms_div(qkv, [2]))
    qq, kk, vv = qkv.unbind(3)                      # (b, heads, d, hw)
    s = torch.einsum("bhdi,bhdj->bhij", qq, kk) / math.sqrt(d)
    p = q(torch.softmax(s, dim=-1))
    y = q(torch.einsum("bhij,bhdj->bhdi", p, vv)).reshape(b, c, h, w)
    y = wn_conv(y, P[prefix + "out_conv.weight"], q)
    return q(mp_add(x, y, 0.5))


def _modulate(P, p: str, res: Tensor, emb: Tensor) -> Tensor:
    """networks.py:255-258 / 319-322 (fp32)."""
    m = wn_linear(emb, P[p + "embed.weight"]) * P[p + "gain"] + 1
    return res * m[:, :, None, None]


def _dropout(x: Tensor, rate: float, training: bool, mask: Optional[Tensor]) -> Tensor:
    if not training or rate == 0.0:
        return x
    if mask is None:
        return F.dropout(x, rate, True)
    return x * mask / (1.0 - rate)


def encoder_block(P, p: str, x: Tensor, emb: Tensor, down: bool, attn: bool, heads: int,
                  t: float, rate: float, training: bool, q=_ident, mask=None) -> Tensor:
    """networks.py:246-265."""
    if down:
        x = q(F.avg_pool2d(x, 2, 2))
    if p + "conv_1x1.weight" in P:
        x = wn_conv(x, P[p + "conv_1x1.weight"], q)
    x = q(rms_div(x, [1]))
    r = wn_conv(q(mp_silu(x)), P[p + "conv_3x3_1.weight"], q)
    r = q(_dropout(mp_silu(_modulate(P, p, r, emb)), rate, training, mask))
    r = wn_conv(r, P[p + "conv_3x3_2.weight"], q)
    out = q(mp_add(x, r, t))
    if attn:
        out = cosine_attention(P, p + "attention.", out, heads, q)
    return out


def decoder_block(P, p: str, x: Tensor, emb: Tensor, skip: Optional[Tensor], up: bool, attn: bool,
                  heads: int, t: float, rate: float, training: bool, q=_ident, mask=None) -> Tensor:
    """networks.py:306-329."""
    if skip is not None:
        x = torch.cat((x, q(skip * scale_long_gate(P, p + "cat_factor.", skip))), dim=1)
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest-exact")
    r = x
    if p + "conv_1x1.weight" in P:
        x = wn_conv(x, P[p + "conv_1x1.weight"], q)
    r = wn_conv(q(mp_silu(r)), P[p + "conv_3x3_1.weight"], q)
    r = q(_dropout(mp_silu(_modulate(P, p, r, emb)), rate, training, mask))
    r = wn_conv(r, P[p + "conv_3x3_2.weight"], q)
    out = q(mp_add(x, r, t))
    if attn:
        out = cosine_attention(P, p + "attention.", out, heads, q)
    return out


def precond_scalars(sigma: Tensor, sd: float):
    """networks.py:578-581 -> c_skip, c_out, c_in each (B,1,1,1)."""
    s = sigma.reshape(-1, 1, 1, 1).float()
    return sd ** 2 / (s ** 2 + sd ** 2), s * sd / (s ** 2 + sd ** 2).sqrt(), 1 / (sd ** 2 + s ** 2).sqrt()


def denoiser_forward(P, dcfg: DenoiserCfg, noisy: Tensor, sigma: Tensor, emb: Tensor,
                     training: bool = False, bf16: bool = False,
                     dropout_masks: Optional[Dict[str, Tensor]] = None,
                     record: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """networks.py:577-605.  ``dropout_masks`` maps block prefix -> {0,1} keep
    mask (B,C,H,W) so that tests can inject the HIP path's Philox masks.  ``record`` (a dict) receives every block's
    output under its parameter prefix (tools/block_parity.py: per-block HIP-vs-oracle error table)."""
    q = q_bf16 if bf16 else _ident
    masks = dropout_masks or {}
    c_skip, c_out, c_in = precond_scalars(sigma, dcfg.sigma_data)
    x = c_in * noisy
    x = torch.cat((x, torch.ones_like(x[:, :1])), dim=1)
    x = wn_conv(x, P["denoiser.conv_in.weight"], q)
    skips = [x]
    for i, t in enumerate(dcfg.encoder_block_types):
        p = f"denoiser.encoder_blocks.{i}."
        down, _, attn = _block_flags(t)
        x = encoder_block(P, p, x, emb, down, attn, dcfg.num_heads, dcfg.encoder_add_factor,
                          dcfg.dropout_rate, training, q, masks.get(p))
        if record is not None:
            record[p] = x
        skips.append(x)
    for i, (t, has_skip) in enumerate(zip(dcfg.decoder_block_types, dcfg.skip_connections)):
        p = f"denoiser.decoder_blocks.{i}."
        _, up, attn = _block_flags(t)
        x = decoder_block(P, p, x, emb, skips.pop() if has_skip else None, up, attn, dcfg.num_heads,
                          dcfg.decoder_add_factor, dcfg.dropout_rate, training, q, masks.get(p))
        if record is not None:
            record[p] = x
    # conv_out result is consumed in fp32 by the fused epilogue (no q on the output)
    f = F.conv2d(q(x), q(effective_weight(P["denoiser.conv_out.weight"]))) * P["denoiser.gain_out"]
    return f * c_out + noisy * c_skip


def edm_forward(P, ecfg, dcfg, noisy, sigma, labels=None, bf16=False):
    """edm.py:280-286."""
    _, emb = embedding_forward(P, ecfg, sigma, labels)
    return denoiser_forward(P, dcfg, noisy, sigma, emb, False, bf16)


# --------------------------------------------------------------------------
# UncertaintyNet (networks.py:91-103)
# --------------------------------------------------------------------------
def uncertainty_forward(U: Dict[str, Tensor], fourier: Tensor) -> Tensor:
    x = torch.cat((fourier, torch.ones_like(fourier[:, :1])), dim=1)
    x = mp_silu(wn_linear(x, U["u.linear1.weight"]))
    return U["u.gain"] * wn_linear(x, U["u.linear2.weight"])


# --------------------------------------------------------------------------
# Diffuser, loss, training step (edm.py:84-93, 205-236; metric.py:8-18)
# --------------------------------------------------------------------------
def diffuse(clean: Tensor, eps: Tensor, noise: Tensor, P_mean: float, P_std: float):
    """edm.py:86-93 with the two normal draws injected."""
    sigma = (P_mean + eps * P_std).exp()
    return clean + noise * sigma.view(-1, 1, 1, 1), sigma


def loss_weight(sigma: Tensor, sd: float) -> Tensor:
    """edm.py:212."""
    return (sigma ** 2 + sd ** 2) / (sigma * sd) ** 2


def weighted_mse(weight: Tensor, pred: Tensor, target: Tensor) -> Tensor:
    """metric.py:8-18 + 38-49: sum_i mean_j(w_i d_ij^2) / N."""
    n = target.shape[0]
    d = pred.reshape(n, -1) - target.reshape(n, -1)
    return torch.mean(weight.view(n, 1) * d * d, dim=1).sum() / n


def training_loss(P, ecfg, dcfg, clean, eps, noise, P_mean, P_std, labels=None, bf16=False,
                  dropout_masks=None, normalize_weights=True):
    """edm.py:205-236 (use_uncertainty=False branch), training-mode forward."""
    if normalize_weights:
        force_normalize_(P)
    noisy, sigma = diffuse(clean, eps, noise, P_mean, P_std)
    _, emb = embedding_forward(P, ecfg, sigma, labels)
    den = denoiser_forward(P, dcfg, noisy, sigma, emb, True, bf16, dropout_masks)
    return weighted_mse(loss_weight(sigma, dcfg.sigma_data), den, clean)


# --------------------------------------------------------------------------
# optimizer side: Adam (edm.py:251-253), LR (edm.py:305-320), EMA (ema.py)
# --------------------------------------------------------------------------
def lr_lambda(step: int, rampup: int, steady: int) -> float:
    """edm.py:307-317."""
    if step < rampup:
        return 1e-8 + (1.0 - 1e-8) * step / rampup
    if step < rampup + steady:
        return 1.0
    return 1.0 / math.sqrt(1 + (step - rampup - steady) / steady)


def adam_step(theta, grad, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam semantics (no weight decay, no amsgrad); step is 1-based."""
    m.mul_(b1).add_(grad, alpha=1 - b1)
    v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    theta.addcdiv_(m, denom, value=-lr / bc1)


def sigma_rel_to_gamma(sigma_rel: float) -> float:
    """ema.py:29-32."""
    t = sigma_rel ** -2
    return float(np.roots([1, 7, 16 - t, 12 - t]).real.max())


def ema_beta(current_step: int, gamma: float) -> float:
    """ema.py:273 (current_step is 0-based, incremented after the update)."""
    return (1 - 1 / (current_step + 1)) ** (gamma + 1)


def ema_step(ema: Tensor, theta: Tensor, beta: float) -> None:
    """ema.py:137-140."""
    ema.mul_(beta).add_(theta, alpha=1.0 - beta)


# --------------------------------------------------------------------------
# sampler (solvers.py:13-59)
# --------------------------------------------------------------------------
def karras_schedule(num_steps=18, sigma_min=0.002, sigma_max=80.0, rho=7.0) -> Tensor:
    """solvers.py:33-41, fp32, N+1 entries, last one 0."""
    i = torch.arange(num_steps, dtype=torch.float32)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros(1)])


def heun_solve(model, x0: Tensor, t_steps: Tensor, labels=None) -> Tensor:
    """solvers.py:43-59.  ``model(x, sigma0d, labels) -> D``."""
    n = t_steps.numel() - 1
    x1 = x0.float() * t_steps[0]
    for i in range(n):
        t0, t1 = t_steps[i], t_steps[i + 1]
        x = x1
        dx = (x - model(x, t0, labels).float()) / t0
        x1 = x + (t1 - t0) * dx
        if i < n - 1:
            dxp = (x1 - model(x1, t1, labels).float()) / t1
            x1 = x + (t1 - t0) * (0.5 * dx + 0.5 * dxp)
    return x1

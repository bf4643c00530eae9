"""Whole-network gradient parity on the BENCHMARKED network (BASELINE configs[1] and [2]: the CIFAR-10 EDM2 U-Net,
35.6 M parameters, unconditional and class-conditional), training mode with dropout 0.13.

One training step's backward pass through the HIP path -- every per-parameter gradient (all 136+ tensors: 3x3 / 1x1
conv weights through the grouped and per-layer weight-gradient kernels, embed Linears, ScaleLong MLPs, block gains,
gain_out) -- is compared with autograd through the CPU oracle (oracle/edm_oracle.py, pinned to the reference by
tests/test_oracle_golden.py) on the SAME inputs, the SAME noise draws and the kernel's OWN Philox dropout masks
(ops.dropout_mask regenerates each block's mask from (seed, block stream, step), injected through the oracle's
`dropout_masks=`), twice: with the bf16 rounding points of the HIP path (limit 3e-2 relative L2 per tensor) and in plain
fp32 (what the reference's fp32 autograd would give; recorded, limit 6e-2).  Reference: networks.py:32-37, 246-329,
edm.py:205-236.  Every per-tensor figure goes to gpurun_out/grad_parity_r06.json (copied to profiles/)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import edm_oracle as O
from parity_log import record

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _block_shapes(den, B, H):
    """(prefix, block, (B, h, w, C)) of every residual block's dropout site (the output of its first 3x3 conv)"""
    out = []
    h = H
    for i, (blk, t) in enumerate(zip(den.encoder_blocks, den.encoder_block_types)):
        if t.endswith("D"):
            h //= 2
        out.append((f"denoiser.encoder_blocks.{i}.", blk, (B, h, h, blk.conv_3x3_2.weight.shape[1])))
    for i, (blk, t) in enumerate(zip(den.decoder_blocks, den.decoder_block_types)):
        if t.endswith("U"):
            h *= 2
        out.append((f"denoiser.decoder_blocks.{i}.", blk, (B, h, h, blk.conv_3x3_2.weight.shape[1])))
    return out


def _grad_parity(tag, ecfg, dcfg, shape, P_mean, P_std, seed_params, seed_data, lim_bf16=3e-2, lim_fp32=6e-2,
                 fp32_leg=True):
    """One training step's backward through the HIP path against autograd through the CPU oracle on the same inputs,
    noise draws and the kernel's OWN Philox dropout masks; every per-parameter gradient.  -> {leg: summary}"""
    import tinyedm_amd as T
    from tinyedm_amd import metric, networks as N, ops
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    B, Cimg, H, W = shape
    assert H == W
    conditional = ecfg.num_classes is not None and ecfg.num_classes > 0
    P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(seed_params), gains_nonzero=True)
    N._rng_sub_counter[0] = 0
    T.manual_seed(1234)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                     tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                     tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), dcfg.dropout_rate,
                     dcfg.sigma_data, dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim, dcfg.num_heads)
    emb.load_state_dict({k[len("embedding."):]: v for k, v in P.items() if k.startswith("embedding.")})
    den.load_state_dict({k[len("denoiser."):]: v for k, v in P.items() if k.startswith("denoiser.")})
    emb, den = emb.to(DEV).train(), den.to(DEV).train()
    named = {("embedding." + k): v for k, v in emb.named_parameters()}
    named.update({("denoiser." + k): v for k, v in den.named_parameters()})
    # the benchmarked configuration keeps gradients in the flat arena (FusedAdam): direct accumulation by the kernels,
    # grouped 3x3 weight gradients, one modulation finish per step
    opt = T.FusedAdam(list(named.values()), lr=1e-3)
    opt.zero_grad()

    g = torch.Generator().manual_seed(seed_data)
    clean = 0.5 * torch.randn(B, Cimg, H, W, generator=g)
    eps, noise = torch.randn(B, generator=g), torch.randn(B, Cimg, H, W, generator=g)
    labels = torch.randint(0, ecfg.num_classes, (B,), generator=g) if conditional else None
    noisy, sigma = O.diffuse(clean, eps, noise, P_mean, P_std)

    seed, step0 = N.rng.seed, N.rng.step
    _, e = emb(sigma.to(DEV), None if labels is None else labels.to(DEV))
    D = den(noisy.to(DEV), sigma.to(DEV), e)
    sd = dcfg.sigma_data
    w = (sigma ** 2 + sd ** 2) / (sigma * sd) ** 2
    loss = metric.weighted_mse_loss(w.to(DEV), D, clean.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert N.rng.step == step0 + 1

    # the kernel's own dropout masks, block by block (NHWC element order -> NCHW for the oracle)
    masks = None
    if dcfg.dropout_rate > 0:
        masks, kept = {}, []
        for prefix, blk, (b, h, w_, c) in _block_shapes(den, B, H):
            m = ops.dropout_mask(b * h * w_ * c, blk.dropout_rate, seed, blk.rng_sub, step0, DEV)
            masks[prefix] = m.view(b, h, w_, c).permute(0, 3, 1, 2).float().cpu().contiguous()
            kept.append(masks[prefix].mean().item())
        assert abs(float(np.mean(kept)) - (1 - dcfg.dropout_rate)) < 1e-2, np.mean(kept)

    out = {}
    legs = (("bf16_oracle", True, lim_bf16),) + ((("fp32_autograd", False, lim_fp32),) if fp32_leg else ())
    for leg, bf16, lim in legs:
        Pb = {k: v.clone() for k, v in P.items()}
        keys = O.trainable_keys(Pb)
        for k in keys:
            Pb[k].requires_grad_(True)
        lo = O.training_loss(Pb, ecfg, dcfg, clean, eps, noise, P_mean, P_std, labels, bf16=bf16, dropout_masks=masks)
        lo.backward()
        assert sorted(keys) == sorted(named), set(keys) ^ set(named)
        lrel = abs(loss.item() - lo.item()) / abs(lo.item())
        record(f"gradparity/{tag}/loss_vs_{leg}", lrel, 2e-2)
        assert lrel <= 2e-2, (loss.item(), lo.item())
        scal = [Pb[k].grad.abs().item() for k in keys if Pb[k].numel() == 1]
        scal_rms = float(np.sqrt(np.mean(np.square(scal))))
        per = {}
        for k in keys:
            gr, go = named[k].grad, Pb[k].grad
            assert gr is not None and torch.isfinite(gr).all(), k
            if gr.numel() == 1:     # scalar gains: sums with heavy cancellation -> judged against the scale of the scalar grads
                per[k] = abs(gr.item() - go.item()) / max(abs(go.item()), scal_rms)
            else:
                per[k] = rel(gr, go)
        worst = max(per, key=per.get)
        out[leg] = {"worst_tensor": worst, "worst": per[worst], "limit": lim, "median": float(np.median(list(per.values()))),
                    "n_tensors": len(per), "per_tensor": {k: round(v, 6) for k, v in sorted(per.items())}}
        record(f"gradparity/{tag}/worst_tensor_vs_{leg}[{worst}]", per[worst], lim)
        assert per[worst] <= lim, f"{leg}: {worst} rel {per[worst]:.3e} (limit {lim})"
        # the in-place weight normalisation of the training forward agrees too (networks.py:32-34)
        for k in keys:
            if Pb[k].dim() >= 2:
                assert rel(named[k].detach(), Pb[k].detach()) <= 1e-5, k
    _dump(tag, {"shape": list(shape), "batch": B, "dropout": dcfg.dropout_rate,
                "n_params": int(sum(v.numel() for v in named.values())), **out})
    return out


def _dump(tag, entry):
    path = os.path.join(ROOT, "gpurun_out", "grad_parity_r06.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old[tag] = entry
        with open(path, "w") as f:
            json.dump(old, f, indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("conditional", [False, True], ids=["cifar10", "cifar10_cond"])
def test_whole_network_gradients_vs_oracle(conditional):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ecfg, dcfg = O.cifar10_cfg(10 if conditional else None)
    assert dcfg.dropout_rate == 0.13
    _grad_parity("cifar10_cond" if conditional else "cifar10", ecfg, dcfg, (4, 3, 32, 32), -1.2, 1.2, 21, 77)


def _mnist_cfg():
    from test_configs_gpu import mnist_cfg
    return mnist_cfg()


def _imagenet_cfg(channels):
    from test_configs_gpu import imagenet_cfg
    return imagenet_cfg(channels)


def test_whole_network_gradients_mnist_config():
    """BASELINE configs[0] (experiments/conf/mnist.yaml: 1 channel, 28x28 -> 14x14 -> 7x7, widths 128/256/512, head dims 32 /
    64 / 128, dropout 0.1; ragged 49-token attention maps): every per-parameter gradient of a training step vs the bf16-rounding
    oracle with the kernel's own Philox masks (round-4 review: was 'loss within 3e-2 + every gradient finite')."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ecfg, dcfg = _mnist_cfg()
    _grad_parity("mnist", ecfg, dcfg, (2, 1, 28, 28), -1.2, 1.2, 31, 79)


@pytest.mark.parametrize("case", ["imagenet64_pixel", "latent32"])
def test_whole_network_gradients_default_net(case):
    """BASELINE configs[3] / [4]: the default 272 M-parameter net (networks.py:332-432 tables: widths 192/384/576/768, head
    dims 48 / 96 / 144 / 192, class-conditional) on (1, 3, 64, 64) pixels and on (2, 4, 32, 32) latents
    (experiments/conf/imagenet.yaml: P_mean -0.4, P_std 1.0, dropout 0): the wgrad / attention-backward / skip-gate-backward
    kernels at those widths as a WHOLE-NETWORK backward, every per-parameter gradient vs the bf16-rounding oracle."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if case == "imagenet64_pixel":
        ecfg, dcfg = _imagenet_cfg(3)
        shape = (1, 3, 64, 64)
    else:
        ecfg, dcfg = _imagenet_cfg(4)
        shape = (2, 4, 32, 32)
    _grad_parity(case, ecfg, dcfg, shape, -0.4, 1.0, 41, 81, fp32_leg=False)


def test_whole_network_gradients_at_batch_128():
    """The same comparison at the BENCHMARKED batch (B = 128: the shapes bench.py dispatches -- 512-pixel tiles that span
    images, the grouped weight-gradient launches over 131 072 pixels, k_conv3x3_s with fragment-major packs on the 8x8
    layers, the copy-free concat) without a B = 128 CPU backward pass: samples are independent in this network, so the loss
    weights of all but eight images (first / last of the batch and both sides of the tile boundaries at 32 and 64) are set to
    zero, and the oracle runs those eight images with the kernel's own Philox masks.  Every per-parameter gradient of the
    B = 128 HIP step must then equal 8/128 of the oracle's B = 8 gradient (bf16 rounding points; limit 3e-2 per tensor)."""
    import tinyedm_amd as T
    from tinyedm_amd import metric, networks as N, ops
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ecfg, dcfg = O.cifar10_cfg(None)
    P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(21), gains_nonzero=True)
    N._rng_sub_counter[0] = 0
    T.manual_seed(4321)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                     tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                     tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), dcfg.dropout_rate,
                     dcfg.sigma_data, dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim, dcfg.num_heads)
    emb.load_state_dict({k[len("embedding."):]: v for k, v in P.items() if k.startswith("embedding.")})
    den.load_state_dict({k[len("denoiser."):]: v for k, v in P.items() if k.startswith("denoiser.")})
    emb, den = emb.to(DEV).train(), den.to(DEV).train()
    named = {("embedding." + k): v for k, v in emb.named_parameters()}
    named.update({("denoiser." + k): v for k, v in den.named_parameters()})
    opt = T.FusedAdam(list(named.values()), lr=1e-3)
    opt.zero_grad()

    g = torch.Generator().manual_seed(78)
    B, sel = 128, [0, 1, 31, 32, 63, 64, 126, 127]
    clean = 0.5 * torch.randn(B, 3, 32, 32, generator=g)
    eps, noise = torch.randn(B, generator=g), torch.randn(B, 3, 32, 32, generator=g)
    noisy, sigma = O.diffuse(clean, eps, noise, -1.2, 1.2)
    seed, step0 = N.rng.seed, N.rng.step
    _, e = emb(sigma.to(DEV), None)
    D = den(noisy.to(DEV), sigma.to(DEV), e)
    w = (sigma ** 2 + 0.25) / (sigma * 0.5) ** 2
    wsel = torch.zeros_like(w)
    wsel[sel] = w[sel]
    loss = metric.weighted_mse_loss(wsel.to(DEV), D, clean.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    # the plan of this forward really is the benchmarked one: fragment-major packs on the 8x8 layers
    assert any(getattr(t, "_edm_frag", False) for plan in den._plans.values() for c in plan.caches for t in c[:2] if t is not None)

    masks = {}
    for prefix, blk, (b, h, w_, c) in _block_shapes(den, B, 32):
        m = ops.dropout_mask(b * h * w_ * c, blk.dropout_rate, seed, blk.rng_sub, step0, DEV)
        masks[prefix] = m.view(b, h, w_, c)[sel].permute(0, 3, 1, 2).float().cpu().contiguous()
    Pb = {k: v.clone() for k, v in P.items()}
    keys = O.trainable_keys(Pb)
    for k in keys:
        Pb[k].requires_grad_(True)
    lo = O.training_loss(Pb, ecfg, dcfg, clean[sel], eps[sel], noise[sel], -1.2, 1.2, None, bf16=True, dropout_masks=masks)
    lo.backward()
    f = len(sel) / B
    lrel = abs(loss.item() - f * lo.item()) / abs(f * lo.item())
    record("gradparity/b128/loss_vs_bf16_oracle", lrel, 2e-2)
    assert lrel <= 2e-2, (loss.item(), f * lo.item())
    scal = [f * Pb[k].grad.abs().item() for k in keys if Pb[k].numel() == 1]
    scal_rms = float(np.sqrt(np.mean(np.square(scal))))
    per = {}
    for k in keys:
        gr, go = named[k].grad, f * Pb[k].grad
        assert gr is not None and torch.isfinite(gr).all(), k
        per[k] = abs(gr.item() - go.item()) / max(abs(go.item()), scal_rms) if gr.numel() == 1 else rel(gr, go)
    worst = max(per, key=per.get)
    record(f"gradparity/b128/worst_tensor_vs_bf16_oracle[{worst}]", per[worst], 3e-2)
    _dump("cifar10_b128", {"batch": B, "images_with_loss_weight": sel, "dropout": dcfg.dropout_rate,
                           "bf16_oracle": {"worst_tensor": worst, "worst": per[worst], "limit": 3e-2,
                                           "median": float(np.median(list(per.values()))), "n_tensors": len(per),
                                           "per_tensor": {k: round(v, 6) for k, v in sorted(per.items())}}})
    assert per[worst] <= 3e-2, f"{worst} rel {per[worst]:.3e}"

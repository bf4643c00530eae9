"""Golden eval-forward outputs of the reference's FULL-SIZE configurations, produced by the reference itself
(runs only in the build container, where /root/reference exists).  TEST INFRASTRUCTURE ONLY.

    tests/golden/configs.npz : for MNIST (conf/mnist.yaml), CIFAR-10 class-conditional (conf/cifar10.yaml + labels)
    and the default `Denoiser` tables (conf/imagenet.yaml: 272.9 M parameters, head dims 144 / 192) -- inputs
    (noisy, sigma, labels), the reference's fp32 output D and its output under bf16 autocast.

Weights are not stored: they are regenerated from the seed through oracle.edm_oracle.init_params; a digest of the
generated parameters is stored so a silent RNG change is detected.

Usage: python oracle/make_golden_configs.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import edm_oracle as O  # noqa: E402
from oracle.make_golden import _load, build_ref, grad_digest  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def config_cases(net):
    """name -> (ecfg, dcfg, input shape, seed)"""
    mn_e = O.EmbeddingCfg(64, 256, 10)
    mn_d = O.DenoiserCfg(
        1, 1,
        ["Enc", "Enc", "Enc", "EncD", "EncA", "EncA", "EncA", "EncD", "EncA", "EncA", "EncA"],
        ["DecA", "Dec", "DecA", "DecA", "DecA", "DecA", "DecU", "DecA", "DecA", "DecA", "DecA", "DecU", "Dec", "Dec",
         "Dec", "Dec"],
        [128, 128, 128, 128, 256, 256, 256, 256, 512, 512, 512],
        [512, 512, 512, 512, 512, 512, 512, 256, 256, 256, 256, 256, 128, 128, 128, 128],
        [False, False, True, True, True, True, False, True, True, True, True, False, True, True, True, True],
        dropout_rate=0.1, sigma_data=0.5, embedding_dim=256, num_heads=4)
    ci_e, ci_d = O.cifar10_cfg(num_classes=10)
    im_e = O.EmbeddingCfg(192, 768, 1000)
    im_d = O.DenoiserCfg(4, 4, list(net.get_encoder_blocks_types()), list(net.get_decoder_blocks_types()),
                         list(net.get_encoder_out_channels()), list(net.get_decoder_out_channels()),
                         list(net.get_skip_connections()), dropout_rate=0.0, sigma_data=0.5, embedding_dim=768,
                         num_heads=4)
    return {"mnist": (mn_e, mn_d, (2, 1, 28, 28), 21), "cifar10_cond": (ci_e, ci_d, (2, 3, 32, 32), 22),
            "default32": (im_e, im_d, (1, 4, 32, 32), 23)}


def make_inputs(ecfg, shape, seed):
    g = torch.Generator().manual_seed(seed)
    noisy = torch.randn(*shape, generator=g) * 1.3
    sigma = torch.exp(torch.randn(shape[0], generator=g) * 1.2 - 1.2)
    labels = torch.randint(0, ecfg.num_classes, (shape[0],), generator=g)
    return noisy, sigma, labels


def main():
    torch.set_num_threads(8)
    net = _load("networks")
    out = {}
    for name, (ecfg, dcfg, shape, seed) in config_cases(net).items():
        P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(seed))
        emb_m, den_m = build_ref(net, ecfg, dcfg, P)
        emb_m.eval(); den_m.eval()
        noisy, sigma, labels = make_inputs(ecfg, shape, seed + 100)
        with torch.no_grad():
            _, e = emb_m(sigma, labels)
            D = den_m(noisy, sigma, e)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                _, eb = emb_m(sigma, labels)
                Db = den_m(noisy, sigma, eb).float()
        keys = sorted(P)
        dig = np.stack([grad_digest(P[k]) for k in keys[:: max(1, len(keys) // 16)]])
        out[name + "::noisy"], out[name + "::sigma"], out[name + "::labels"] = noisy.numpy(), sigma.numpy(), labels.numpy()
        out[name + "::D"], out[name + "::D_autocast_bf16"], out[name + "::param_digest"] = D.numpy(), Db.numpy(), dig
        nparams = sum(v.numel() for k, v in P.items() if "freqs" not in k and "phases" not in k)
        print(f"{name}: {nparams / 1e6:.1f} M params, |D| {D.norm():.4f}, ref bf16-vs-fp32 rel "
              f"{((Db - D).norm() / D.norm()).item():.3e}", flush=True)
    np.savez_compressed(os.path.join(OUT, "configs.npz"), **out)


if __name__ == "__main__":
    main()

"""How far are the two sampler precisions of this build from the reference's fp32 sampler?  

The reference samples in fp32 (generate.py:39-44, solvers.py:43-59).  This tool integrates the SAME 32-step Heun trajectory
(63 evaluations) of the CIFAR-10 U-Net (35.6 M parameters, seeded weights with non-zero gains) from the same x0
  (a) on the HIP path with the bf16 network (the training path's kernels; hipGraph),
  (a') on the HIP path with the fp32 network (the reference-precision evaluation, csrc/eval_f32.hip; hipGraph), and
  (b) on the CPU oracle in fp32 (the reference's arithmetic: oracle pinned to the reference's golden vectors), plus the
      same oracle with every convolution operand rounded to TF32,
and writes the relative L2 distances of the final images to gpurun_out/r03_sampler_parity.json (copied to profiles/).
    python tools/sampler_parity.py [--images 2]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import edm_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--bf16-oracle", action="store_true", help="also integrate the oracle with bf16 rounding points")
    ap.add_argument("--no-tf32-oracle", action="store_true",
                    help="skip the leg that rounds every convolution operand of the fp32 oracle to TF32 (10-bit mantissa): "
                         "what cuDNN does by default for fp32 convolutions on the GPUs the reference targets "
                         "(torch.backends.cudnn.allow_tf32 = True; generate.py does not change it)")
    a = ap.parse_args()
    import tinyedm_amd as T
    dev = torch.device("cuda", 0)
    ecfg, dcfg = O.cifar10_cfg()
    P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(1), gains_nonzero=True)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types), tuple(dcfg.decoder_block_types),
                     tuple(dcfg.encoder_out_channels), tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections),
                     dcfg.dropout_rate, dcfg.sigma_data, dcfg.encoder_add_factor, dcfg.decoder_add_factor,
                     dcfg.embedding_dim, dcfg.num_heads)
    emb.load_state_dict({k[len("embedding."):]: v for k, v in P.items() if k.startswith("embedding.")})
    den.load_state_dict({k[len("denoiser."):]: v for k, v in P.items() if k.startswith("denoiser.")})
    emb, den = emb.to(dev).eval(), den.to(dev).eval()

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb, self.den = emb, den

        def forward(self, x, t, lab):
            _, e = self.emb(t, lab)
            return self.den(x, t, e)
    x0 = torch.randn(a.images, 3, 32, 32, generator=torch.Generator().manual_seed(7))
    sol = T.DeterministicSolver(num_steps=a.steps)
    x_hip = sol.solve(Model().eval(), x0.to(dev), None, graph=True).cpu()
    den.set_eval_dtype("f32")
    x_hip32 = sol.solve(Model().eval(), x0.to(dev), None, graph=True).cpu()
    den.set_eval_dtype("bf16")

    import bench
    torch.set_num_threads(bench.host_cores())
    t0 = time.time()
    ts = O.karras_schedule(a.steps)
    calls = [0]

    def net(bf16):
        def f(x, t, l):
            calls[0] += 1
            if calls[0] % 8 == 0:
                print(f"[sampler_parity] oracle evaluation {calls[0]} ({time.time() - t0:.0f} s)", flush=True)
            return O.edm_forward(P, ecfg, dcfg, x, t, l, bf16=bf16)
        return f
    def tf32(t):
        """round-to-nearest-even to a 10-bit mantissa (the TF32 operand format)"""
        i = t.contiguous().view(torch.int32)
        i = (i + 0x0FFF + ((i >> 13) & 1)) & ~0x1FFF
        return i.view(torch.float32)

    with torch.no_grad():
        x_f32 = O.heun_solve(net(False), x0, ts, None)
        x_bf = O.heun_solve(net(True), x0, ts, None) if a.bf16_oracle else x_f32
        x_tf = None
        if not a.no_tf32_oracle:
            conv = O.F.conv2d
            O.F.conv2d = lambda x, w, *aa, **kk: conv(tf32(x.float()), tf32(w.float()), *aa, **kk)
            try:
                x_tf = O.heun_solve(net(False), x0, ts, None)
            finally:
                O.F.conv2d = conv
    cpu_s = time.time() - t0

    def rel(u, v):
        return ((u.double() - v.double()).norm() / v.double().norm()).item()
    out = {"config": "CIFAR-10 unconditional U-Net (35.6M params, seeded weights, gains non-zero)", "heun_steps": a.steps,
           "nfe": 2 * a.steps - 1, "images": a.images,
           "hip_bf16net_vs_fp32_oracle_rel_l2": rel(x_hip, x_f32),
           "hip_fp32net_vs_fp32_oracle_rel_l2": rel(x_hip32, x_f32),
           "hip_fp32net_max_abs_diff_vs_fp32": (x_hip32 - x_f32).abs().max().item(),
           "hip_bf16net_vs_bf16_oracle_rel_l2": rel(x_hip, x_bf) if a.bf16_oracle else None,
           "bf16_oracle_vs_fp32_oracle_rel_l2": rel(x_bf, x_f32) if a.bf16_oracle else None,
           "tf32conv_oracle_vs_fp32_oracle_rel_l2": rel(x_tf, x_f32) if x_tf is not None else None,
           "hip_bf16net_vs_tf32conv_oracle_rel_l2": rel(x_hip, x_tf) if x_tf is not None else None,
           "final_image_rms": x_f32.pow(2).mean().sqrt().item(),
           "max_abs_diff_vs_fp32": (x_hip - x_f32).abs().max().item(),
           "note": "state integrated in fp32 on both sides; only the network evaluation differs (bf16 operands, fp32 "
                   "accumulation on the HIP path). 1/255 of the [-1,1] image range is 7.8e-3.",
           "oracle_cpu_seconds": round(cpu_s, 1)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r03_sampler_parity.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""Library-GEMM reference rates on this GPU (torch.matmul -> hipBLASLt) for the shapes the conv kernels compute.
Used only to put the hand-written kernels' PFLOP/s in context; not part of the product path."""
import torch


def rate(M, N, K, iters=20):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"M={M:7d} N={N:5d} K={K:5d}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    rate(8192, 8192, 8192)
    rate(131072, 256, 2304)   # 32x32 3x3 256->256 fwd as a plain GEMM
    rate(131072, 256, 4608)   # 512->256
    rate(32768, 256, 2304)    # 16x16
    rate(8192, 256, 2304)     # 8x8
    rate(2304, 256, 131072)   # wgrad shape (K = pixels)
    rate(131072, 256, 512)    # 1x1 512->256
    rate(512, 256, 131072)    # 1x1 wgrad

// Diagnostic: cost of the 32x32 -> 64-bit multiplies of Philox4x32-10 on gfx950, written as v_mul_hi_u32 + v_mul_lo_u32
// (what hipcc emits for __umulhi(a, b) and a * b) or as ONE v_mad_u64_u32.  Same results, timed over a VALU-only loop.
//   hipcc --offload-arch=gfx950 -O3 -o philox_mul philox_mul.hip && ./philox_mul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__device__ __forceinline__ void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  if (MODE == 0) {
    hi = __umulhi(a, b);
    lo = a * b;
  } else {
    unsigned long long r;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b) : "vcc");
    hi = (uint32_t)(r >> 32);
    lo = (uint32_t)r;
  }
}
template <int MODE>
__global__ void k(uint32_t* out, int iters) {
  uint32_t c0 = threadIdx.x, c1 = blockIdx.x, c2 = 7, c3 = 9;
  for (int it = 0; it < iters; ++it) {
    uint32_t k0 = 0x1234u + it, k1 = 0x5678u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0, lo0, hi1, lo1;
      mulhilo<MODE>(0xD2511F53u, c0, hi0, lo0);
      mulhilo<MODE>(0xCD9E8D57u, c2, hi1, lo1);
      uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;
}

int main() {
  uint32_t *o0, *o1;
  const int n = 1024 * 256, iters = 2000;
  hipMalloc(&o0, n * 4); hipMalloc(&o1, n * 4);
  for (int rep = 0; rep < 2; ++rep) {
    for (int mode = 0; mode < 2; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, o0, iters);
      else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, o1, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.2f ms  (%.1f G Philox4x32-10 calls/s)\n", mode ? "v_mad_u64_u32        " : "v_mul_hi + v_mul_lo_u32", ms,
             (double)n * iters / ms / 1e6);
    }
  }
  uint32_t h0[64], h1[64];
  hipMemcpy(h0, o0, 256, hipMemcpyDeviceToHost); hipMemcpy(h1, o1, 256, hipMemcpyDeviceToHost);
  int same = 1;
  for (int i = 0; i < 64; ++i) same &= h0[i] == h1[i];
  printf("results identical: %s\n", same ? "yes" : "NO");
  return 0;
}

// Diagnostic: what does the f32-input matrix instruction sustain on this device?  A bare loop of
// v_mfma_f32_32x32x2_f32 on random data, every operand re-read from LDS (one dword per lane per MFMA, as k_conv_f32 does),
// 2 x 2 accumulator blocks per wave, NW waves per workgroup, one workgroup per CU; reports TFLOP/s and the shader clock
// (s_memtime / s_memrealtime).  The data-sheet peak is 157.3 TF at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32 mfma_f32.hip && ./mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ out, int iters,
                                         unsigned long long* __restrict__ clk) {
  __shared__ float sm[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = sm + wave * 256 + lane;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const float* p = base + (it & 15) * 256;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float a0 = p[ks * 64], a1 = p[ks * 64 + 2048], b0 = p[ks * 64 + 1024], b1 = p[ks * 64 + 3072];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 256) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int iters = 40000;
  std::vector<float> h(8192);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *d, *o; unsigned long long* c;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 512 * 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int nw : {4, 8}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(64 * nw), 0, 0, d, o, iters, c);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
      const double flop = 256.0 * nw * iters * 16.0 * 2 * 32 * 32 * 2;
      // s_memrealtime ticks at 100 MHz
      printf("f32 32x32x2, %d waves/CU: %.1f TFLOP/s  (%.2f ms, shader clock %.2f GHz)\n", nw, flop / ms / 1e9, ms,
             (double)hc[0] / ((double)hc[1] / 100e6) / 1e9);
    }
  }
  return 0;
}

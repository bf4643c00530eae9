// Diagnostic (MI355X_MICROARCH.md, DVFS give-back item 7 / cdna_hip_programming.md rule 28): does the chip hold a
// higher clock on v_mfma_f32_16x16x32_bf16 than on v_mfma_f32_32x32x16_bf16 at the same output tile per wave?
// Bare loops on random data, every operand re-read from LDS by ds_read_b128, 128(co) x 64(px) accumulators per wave
// (the tile of k_conv3x3_v4), NW waves per workgroup, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const unsigned short* __restrict__ src, float* __restrict__ out, int iters,
                                         unsigned long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // 48 KiB of random operand data in LDS
  for (int i = threadIdx.x; i < 48 * 1024 / 2; i += blockDim.x) ((unsigned short*)smem)[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = smem + wave * 4096 + lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (SHAPE == 32) {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const char* p = base + (it & 7) * 1024;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {            // two k16 steps = K 32
        bf16x8 a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(p + (ks * 6 + i) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(p + (ks * 6 + 4 + j) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const char* p = base + (it & 7) * 1024;
      bf16x8 a[8], b[4];                          // one k32 step: the same 12 ds_read_b128
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(p + (8 + j) * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x < 256) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int iters = 20000;
  std::vector<unsigned short> h(48 * 1024 / 2);
  srand(1);
  for (auto& v : h) {      // random bf16 in (-2, 2): sign, exponent 126..128, random mantissa
    v = (unsigned short)(((rand() & 1) << 15) | ((126 + rand() % 3) << 7) | (rand() & 0x7F));
  }
  unsigned short* d; float* o; unsigned long long* c;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 512 * 8);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int nw : {4, 8}) {
    for (int rep = 0; rep < 3; ++rep)
      for (int shape : {32, 16}) {
        auto kern = shape == 32 ? k<32> : k<16>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * nw), 96 * 1024, 0, d, o, 2000, c);   // warm
        hipEventRecord(e0);
        for (int q = 0; q < 20; ++q) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * nw), 96 * 1024, 0, d, o, iters, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[512]; hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
        double ghz = 0; for (int b = 0; b < 256; ++b) ghz += (double)hc[2 * b] / hc[2 * b + 1] * 0.1; ghz /= 256;
        const double flop = 20.0 * iters * 256.0 * nw * 2.0 * 128 * 64 * 32;
        printf("waves/WG %d  shape %2d: %8.2f ms  %7.1f TFLOP/s  in-kernel clock %.3f GHz\n", nw, shape, ms, flop / ms / 1e9, ghz);
      }
  }
  return 0;
}

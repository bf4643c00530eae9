// How long does it take 8 waves per CU to pull a 256-token x 256-channel bf16 tile (128 KB) as MFMA B fragments straight
// from global memory (lane = token row, 16 loads of 16 B at a 32-B stride: the load pattern of attention_fused.hip) --
// against the same bytes loaded row-contiguously (a lane takes 16 B of a 512-B row, 32 lanes per row) and against an
// LDS-DMA of the tile?   hipcc --offload-arch=gfx950 -O3 -o frag_load frag_load.hip && ./frag_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int C = 256;

template <int MODE, int NTENS>
__global__ __launch_bounds__(512) void k_load(const unsigned short* __restrict__ x, unsigned int* __restrict__ out, int B, int N,
                                              int groups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
  const int b = (k / groups) * 8 + xcd;
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < NTENS; ++t) {
    const unsigned short* base = x + (long)t * B * N * C + (long)b * N * C;
    if (MODE == 0) {            // fragment pattern: row = wave*32 + l31, 16 pieces of 16 B at 32-B stride
      const unsigned short* r = base + (long)(wave * 32 + l31) * C + lhi * 8;
      u32x4 v[16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) v[kk] = *reinterpret_cast<const u32x4*>(r + kk * 16);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc += v[kk];
    } else if (MODE == 1) {     // row-contiguous: instruction i covers rows wave*32 + 2 i, +1 (32 lanes x 16 B = one 512-B row each)
      u32x4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const u32x4*>(base + (long)(wave * 32 + 2 * i + lhi) * C + l31 * 8);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += v[i];
    } else {                    // LDS-DMA of the wave's 16 KB, then fragment reads from LDS
      char* dst = smem + wave * 16384;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (long)(wave * 32 + 2 * i + lhi) * C + l31 * 8), (lptr_t)(dst + i * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc += *reinterpret_cast<const u32x4*>(dst + l31 * 512 + kk * 32 + lhi * 16);
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) out[threadIdx.x] = acc[0];
}

template <int MODE, int NTENS>
float run(const unsigned short* x, unsigned int* out, int B, int N, int groups, int iters) {
  const int grid = ((B + 7) / 8) * 8 * groups;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const size_t lds = MODE == 2 ? 8 * 16384 : 0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_load<MODE, NTENS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_load<MODE, NTENS>), dim3(grid), dim3(512), lds, 0, x, out, B, N, groups);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_load<MODE, NTENS>), dim3(grid), dim3(512), lds, 0, x, out, B, N, groups);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const int B = 128, N = 256;
  unsigned short* x;
  unsigned int* out;
  hipMalloc(&x, (size_t)3 * B * N * C * 2);
  hipMemset(x, 1, (size_t)3 * B * N * C * 2);
  hipMalloc(&out, 4096);
  for (int groups : {1, 2, 4}) {
    printf("B=128, 256 tokens x 256 ch bf16 per sample, %d workgroup(s) per sample (each loads the whole sample):\n", groups);
    printf("  1 tensor : fragment pattern %6.1f us   row-contiguous %6.1f us   LDS-DMA + ds_read %6.1f us\n",
           run<0, 1>(x, out, B, N, groups, 50), run<1, 1>(x, out, B, N, groups, 50), run<2, 1>(x, out, B, N, groups, 50));
    printf("  2 tensors: fragment pattern %6.1f us   row-contiguous %6.1f us   LDS-DMA + ds_read %6.1f us\n",
           run<0, 2>(x, out, B, N, groups, 50), run<1, 2>(x, out, B, N, groups, 50), run<2, 2>(x, out, B, N, groups, 50));
  }
  return 0;
}

// Shared device helpers for the tinyedm_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define EDM_OK 0
#define EDM_ERR_ARG (-1)
#define EDM_ERR_LAUNCH (-2)
#define EDM_ERR_UNSUPPORTED (-3)

extern "C" void edm_set_error(const char* fmt, ...);

#define EDM_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      edm_set_error(__VA_ARGS__);         \
      return EDM_ERR_ARG;                 \
    }                                     \
  } while (0)

#define EDM_CHECK_LAUNCH(name)                                        \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      edm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return EDM_ERR_LAUNCH;                                          \
    }                                                                 \
  } while (0)

#define SILU_DIV 0.596f
#define NORM_EPS 1e-4f

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// magnitude-preserving SiLU  (reference networks.py:83-84)
__device__ __forceinline__ float mp_silu_f(float x) { return x * sigmoidf_(x) * (1.0f / SILU_DIV); }
// d/dx of mp_silu
__device__ __forceinline__ float mp_silu_grad_f(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s)) * (1.0f / SILU_DIV);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void load8(const bf16* p, float (&f)[8]) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ void store8(bf16* p, const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
  *reinterpret_cast<bf16x8*>(p) = v;
}

// ---------------- Philox4x32-10 (counter-based RNG; stateless, replayable) -----------
struct Philox4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u32_to_unit(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }  // [0,1)

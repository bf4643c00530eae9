// Cosine self-attention of the split-bf16 ("f32x3") evaluation path: an fp32-accurate attention on the bf16 matrix cores.
//
//   q, k, v rows pixel-normalised over head_dim (networks.py:195), softmax(q k^T / sqrt(d)) v (:201)   -- as attention.hip,
// but from the fp32 qkv tensor of the reference-precision path (the qkv conv's own channel order: head*3d + 3*dd +
// {q, k, v}, networks.py:194) and with every matrix product taken in THREE bf16 MFMA passes over (hi, lo) operand pairs,
// x = hi + lo, hi = bf16(x), lo = bf16(x - hi):   a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi   (the dropped lo.lo term is 2^-18 of
// a product; the scheme of edm_split_conv, conv_dispatch.hip).  Normalisation, softmax and the accumulations are fp32.
// It replaces the SIMT kernel k_attn_f32_g (eval_f32.hip, 30 TF/s: 11.5 % of a split-bf16 solve) on the shapes the bf16
// training kernel covers: head_dim 64, N <= 256 tokens.
//
// Structure (attention.hip's "query on the lane" formulation): one workgroup per (sample, head), one wave per 32-query
// block.  K and V live in LDS as four bf16 images (K_hi, K_lo, V_hi, V_lo: 147 KB at 256 tokens); Q never touches LDS -- a
// lane stages the rows of its OWN query block with the very (row, 8-channel group) assignment the MFMA B operand has
// (row = lane & 31, groups 2 s + (lane >> 5)), so the normalised q values it has just computed ARE its fragments, and the k /
// v values that came with the same 96-byte loads go to the images.  S^T = K Q^T (32 keys x 32 queries per tile, scores in
// registers), softmax in registers, O^T = V^T P^T with the probabilities split hi / lo in registers as well.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef short4v __attribute__((address_space(3))) * lds_s4p;

constexpr int D = 64;
constexpr int RS = 2 * D + 16;  // padded LDS row bytes (bank-conflict-free ds_read_b128 fragments), as attention.hip

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
  short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p0));
  short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p1));
  short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}
__device__ __forceinline__ const bf16x8& ld128(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// x[8] -> hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = (bf16)x[e];
    lo[e] = (bf16)(x[e] - (float)hi[e]);
  }
}
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// NT = 32-token blocks (2: up to 64 tokens, 8: up to 256); NT waves
template <int NT>
__global__ __launch_bounds__(NT * 64) void k_attn_split(const float* __restrict__ qkv, float* __restrict__ y,
                                                         bf16* __restrict__ ypairs, int N, int C, int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT * 32;
  char* const Kh = smem;
  char* const Kl = Kh + NP * RS;
  char* const Vh = Kl + NP * RS;
  char* const Vl = Vh + NP * RS;
  const int b = blockIdx.x / heads, head = blockIdx.x % heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const float* base = qkv + ((long)b * N) * 3 * C + (long)head * 3 * D;

  // ---- stage: this lane's token row, channel groups c8 = 2 s + lhi (s = 0..3): 24 contiguous floats = 8 (q, k, v) triples
  // each.  All 24 loads are issued before the first is consumed (one workgroup per CU: nothing else hides the latency).
  const int row = wave * 32 + l31;
  const bool valid = row < N;
  bf16x8 bqh[4], bql[4];
  {
    const float* rp = base + (long)(valid ? row : 0) * 3 * C + lhi * 24;
    f32x4 raw[4][6];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 6; ++j) raw[s][j] = ld4(rp + s * 48 + j * 4);
    float q[4][8], k[4][8], v[4][8];
    float ssq = 0.f, ssk = 0.f, ssv = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        q[s][e] = raw[s][(3 * e) >> 2][(3 * e) & 3];
        k[s][e] = raw[s][(3 * e + 1) >> 2][(3 * e + 1) & 3];
        v[s][e] = raw[s][(3 * e + 2) >> 2][(3 * e + 2) & 3];
        ssq += q[s][e] * q[s][e];
        ssk += k[s][e] * k[s][e];
        ssv += v[s][e] * v[s][e];
      }
    ssq += __shfl_xor(ssq, 32, 64);   // the other half of the row's 64 channels
    ssk += __shfl_xor(ssk, 32, 64);
    ssv += __shfl_xor(ssv, 32, 64);
    // pixel_norm: x / (eps + |x| / sqrt(D))   (networks.py:9-13)
    const float iq = 1.0f / (NORM_EPS + sqrtf(ssq) * 0.125f);
    const float ik = 1.0f / (NORM_EPS + sqrtf(ssk) * 0.125f);
    const float iv = 1.0f / (NORM_EPS + sqrtf(ssv) * 0.125f);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 kh, kl, vh, vl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        q[s][e] *= iq;
        k[s][e] = valid ? k[s][e] * ik : 0.f;   // rows >= N: zero images (their keys are masked, their values multiplied by 0)
        v[s][e] = valid ? v[s][e] * iv : 0.f;
      }
      split8(q[s], bqh[s], bql[s]);
      split8(k[s], kh, kl);
      split8(v[s], vh, vl);
      const int off = row * RS + (2 * s + lhi) * 16;
      *reinterpret_cast<bf16x8*>(Kh + off) = kh;
      *reinterpret_cast<bf16x8*>(Kl + off) = kl;
      *reinterpret_cast<bf16x8*>(Vh + off) = vh;
      *reinterpret_cast<bf16x8*>(Vl + off) = vl;
    }
  }
  __syncthreads();

  const float sl2 = 0.125f * 1.44269504088896341f;  // 1/sqrt(64) * log2(e): the scale rides in the fma in front of exp2
  const bool full = N == NP;
  const int tr_row = (lane & 15) >> 2;
  const int tr_col = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  // ---- S^T tiles (32 keys x 32 queries): keys in the accumulator registers, this lane's query on its column
  f32x16 St[NT];
  float m = -1e30f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int koff = (kt * 32 + l31) * RS + lhi * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 ah = ld128(Kh + koff + s * 32), al = ld128(Kl + koff + s * 32);
      acc = MFMA32(al, bqh[s], acc);     // small terms first
      acc = MFMA32(ah, bql[s], acc);
      acc = MFMA32(ah, bqh[s], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (!full) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        acc[r] = key < N ? acc[r] : -1e30f;
      }
      m = fmaxf(m, acc[r]);
    }
    St[kt] = acc;
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float mb = m * sl2;
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(St[kt][r], sl2, -mb));
      St[kt][r] = p;
      l += p;
    }
  l += __shfl_xor(l, 32, 64);
  const float linv = 1.0f / l;

  // ---- O^T = V^T P^T: the probability tile goes straight back as the B operand (its k-order is absorbed into the row
  // addresses of the transposing V reads, as attention.hip), split hi / lo on the way
  f32x16 acc[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = St[kt][8 * s2 + j];
      bf16x8 ph, pl;
      split8(pv, ph, pl);
      const int row0 = kt * 32 + 16 * s2 + 4 * lhi + tr_row;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt) {
        const int o = row0 * RS + dt * 64 + tr_col;
        const bf16x8 vh = tr_frag(Vh + o, Vh + o + 8 * RS), vl = tr_frag(Vl + o, Vl + o + 8 * RS);
        acc[dt] = MFMA32(vl, ph, acc[dt]);
        acc[dt] = MFMA32(vh, pl, acc[dt]);
        acc[dt] = MFMA32(vh, ph, acc[dt]);
      }
    }
  }
  if (valid) {
    const long tok = (long)b * N + row;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = head * D + dt * 32 + 8 * g + 4 * lhi;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = acc[dt][4 * g + r] * linv;
        if (y) *reinterpret_cast<f32x4*>(y + tok * C + c) = o;
        if (ypairs) {   // the same values as (hi, lo) bf16 pairs, rows [hi(C) | lo(C)]: the out conv's operand format
          bf16x4 hi, lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            hi[r] = (bf16)o[r];
            lo[r] = (bf16)(o[r] - (float)hi[r]);
          }
          *reinterpret_cast<bf16x4*>(ypairs + tok * 2 * C + c) = hi;
          *reinterpret_cast<bf16x4*>(ypairs + tok * 2 * C + C + c) = lo;
        }
      }
  }
}

template <int NT>
void launch_split(const float* qkv, float* y, void* yp, int B, int N, int C, int heads, hipStream_t st) {
  auto kern = k_attn_split<NT>;
  constexpr size_t lds = (size_t)4 * NT * 32 * RS;
  EDM_MAX_LDS(kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(NT * 64), lds, st, qkv, y, (bf16*)yp, N, C, heads);
}

}  // namespace

// qkv [B*N][3C] fp32 in the qkv conv's own channel order (head*3d + 3*dd + {q,k,v}) -> y [B*N][C] fp32 (head*d + dd) and /
// or ypairs [B*N][2C] bf16 = [hi | lo] of the same values (either may be NULL).  head_dim 64, N <= 256:
// EDM_ERR_UNSUPPORTED (-3) otherwise (the caller falls back to edm_f32_attention).
extern "C" int edm_split_attention(const float* qkv, float* y, void* ypairs, int B, int N, int C, int heads, hipStream_t st) {
  EDM_REQUIRE(qkv && (y || ypairs) && B > 0 && N > 0 && heads > 0 && C > 0 && C % heads == 0, "split_attention: bad args");
  if (C / heads != D || N > 256) return EDM_ERR_UNSUPPORTED;
  static_assert((size_t)4 * 256 * RS <= 160 * 1024, "LDS budget");
  if (N <= 64) launch_split<2>(qkv, y, ypairs, B, N, C, heads, st);
  else if (N <= 128) launch_split<4>(qkv, y, ypairs, B, N, C, heads, st);
  else launch_split<8>(qkv, y, ypairs, B, N, C, heads, st);
  EDM_CHECK_LAUNCH("split_attention");
  return EDM_OK;
}

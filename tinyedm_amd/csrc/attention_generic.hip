// Cosine self-attention for head dims other than 64 (reference networks.py:181-207 with the default num_heads=4:
// MNIST 512/4 = 128, ImageNet-64 / latent nets 576/4 = 144 and 768/4 = 192).
//
// Same math and MFMA formulation as attention.hip ("query on the lane", accumulator-as-operand), but with
// head_dim up to 192 the Q/K/V images of one (sample, head) no longer fit the 160 KiB LDS (256 tokens x 144 dims
// x 3 = 258 KiB), so the operands are STREAMED: the wave's own query (or key) block lives in registers as the MFMA
// B operand, the other side passes through LDS in 64-row tiles shared by the four waves.
//
//   forward :  sweep K tiles -> S^T in registers -> softmax -> sweep V tiles -> O^T
//   backward:  pass 1 (query-major)  sweep K -> softmax stats; sweep K+V -> dQ
//              pass 2 (key-major)    sweep Q+dO tiles -> dK, dV
// The q/k/v pixel-norm (networks.py:195) is applied while staging (row norms from a pre-pass) and its backward is
// fused into the stores.  head_dim must be a multiple of 16; the V / dO images are zero-padded to a multiple of 32.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef short4v __attribute__((address_space(3))) * lds_s4p;

constexpr int TK = 64;  // rows per streamed tile

template <int D>
struct Geo {
  static constexpr int DT = D / 16;         // k-steps of the score contraction
  static constexpr int DB = (D + 31) / 32;  // 32-row blocks of the O^T / dQ^T / dK^T / dV^T accumulators
  static constexpr int DP = DB * 32;        // padded dims of an LDS image row
  static constexpr int RS = DP * 2 + 16;    // padded LDS row bytes
};

__device__ __forceinline__ bf16x8 tr_frag_g(const char* p0, const char* p1) {
  short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p0));
  short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p1));
  short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}
__device__ __forceinline__ bf16x8 pack8_g(const f32x16& x, int s2) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)x[8 * s2 + j];
  return o;
}
__device__ __forceinline__ const bf16x8& ld128_g(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// rows [row0, row0+TK) of a [N][D] slice (row stride in elements) -> LDS tile, scaled by 1/dn[row] (dn == nullptr: raw)
template <int D>
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ src, long row_stride, const float* dn, int row0, int N,
                                           char* tile) {
  constexpr int C8 = Geo<D>::DP / 8, RS = Geo<D>::RS;
  for (int idx = threadIdx.x; idx < TK * C8; idx += 256) {
    const int r = idx / C8, c8 = idx - r * C8;
    const int row = row0 + r;
    float v[8];
    if (row < N && c8 * 8 < D) {
      load8(src + (long)row * row_stride + c8 * 8, v);
      if (dn) {
        const float inv = 1.0f / dn[row];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= inv;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
    store8(reinterpret_cast<bf16*>(tile + r * RS + c8 * 16), v);
  }
}

// the wave's own 32-row block as MFMA B operand: lane (l31, lhi) holds row l31, dims s*16 + lhi*8 .. +7
template <int D>
__device__ __forceinline__ void load_block_b(const bf16* __restrict__ src, long row_stride, const float* dn, int row,
                                             int N, int lhi, bf16x8 (&out)[Geo<D>::DT]) {
  const float inv = (dn && row < N) ? 1.0f / dn[row] : 1.0f;
#pragma unroll
  for (int s = 0; s < Geo<D>::DT; ++s) {
    float v[8];
    if (row < N) {
      load8(src + (long)row * row_stride + s * 16 + lhi * 8, v);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) out[s][i] = (bf16)(v[i] * inv);
  }
}

template <int D>
__device__ __forceinline__ f32x16 score_tile_g(const char* a_rows, const bf16x8 (&b)[Geo<D>::DT]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < Geo<D>::DT; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld128_g(a_rows + s * 32), b[s], acc, 0, 0, 0);
  return acc;
}

// row norms d = eps + ||x||/sqrt(D) of the q, k, v rows of this (sample, head): dsave[which*NP + row]
template <int D>
__device__ __forceinline__ void row_norms(const bf16* __restrict__ src, long row_stride, int N, int NP, float* dsave) {
  for (int idx = threadIdx.x; idx < 3 * NP; idx += 256) {
    const int which = idx / NP, row = idx - which * NP;
    float ss = 0.f;
    if (row < N) {
      const bf16* p = src + (long)row * row_stride + which * D;
#pragma unroll 2
      for (int c8 = 0; c8 < D / 8; ++c8) {
        float v[8];
        load8(p + c8 * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
      }
    }
    dsave[idx] = NORM_EPS + sqrtf(ss) * rsqrtf((float)D);
  }
}

// ------------------------------------------------------------------------------------------------ forward
template <int D, int NT>
__global__ __launch_bounds__(256) void k_attn_fwd_g(const bf16* __restrict__ qkv, bf16* __restrict__ y, int N, int C,
                                                      int heads) {
  using G = Geo<D>;
  constexpr int NP = NT * 32, RS = G::RS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                           // [TK][RS]
  float* dsave = reinterpret_cast<float*>(smem + TK * RS);     // [3][NP]
  const int b = blockIdx.x / heads, head = blockIdx.x % heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const long rs = 3L * C;
  const bf16* src = qkv + ((long)b * N) * rs + (long)head * 3 * D;
  row_norms<D>(src, rs, N, NP, dsave);
  __syncthreads();

  const float scale = rsqrtf((float)D);
  const int tr_row = (lane & 15) >> 2;
  const int tr_col = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  for (int round = 0; round < (NT + 3) / 4; ++round) {
    const int qb = round * 4 + wave;
    const bool active = qb < NT;
    const int qi = qb * 32 + l31;
    bf16x8 bq[G::DT];
    load_block_b<D>(src, rs, dsave, active ? qi : N, N, lhi, bq);
    f32x16 St[NT];
    float m = -1e30f;
    // ---- sweep 1: scores against every key tile
#pragma unroll
    for (int t = 0; t < (NT + 1) / 2; ++t) {
      __syncthreads();
      stage_tile<D>(src + D, rs, dsave + NP, t * TK, N, tile);
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int kt = 2 * t + kk;
        if (kt < NT) {
          St[kt] = score_tile_g<D>(tile + (kk * 32 + l31) * RS + lhi * 16, bq);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            float s = key < N ? St[kt][r] * scale : -1e30f;
            St[kt][r] = s;
            m = fmaxf(m, s);
          }
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __expf(St[kt][r] - m);
        St[kt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);
    const float linv = 1.0f / l;
    f32x16 acc[G::DB];
#pragma unroll
    for (int dt = 0; dt < G::DB; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    // ---- sweep 2: O^T += V^T P^T per value tile
#pragma unroll
    for (int t = 0; t < (NT + 1) / 2; ++t) {
      __syncthreads();
      stage_tile<D>(src + 2 * D, rs, dsave + 2 * NP, t * TK, N, tile);
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int kt = 2 * t + kk;
        if (kt < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) St[kt][r] *= linv;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pb = pack8_g(St[kt], s2);
            const int row0 = kk * 32 + 16 * s2 + 4 * lhi + tr_row;
#pragma unroll
            for (int dt = 0; dt < G::DB; ++dt) {
              const char* p0 = tile + row0 * RS + dt * 64 + tr_col;
              acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_g(p0, p0 + 8 * RS), pb, acc[dt], 0, 0, 0);
            }
          }
        }
      }
    }
    if (active && qi < N) {
      bf16* dst = y + ((long)b * N + qi) * C + (long)head * D;
#pragma unroll
      for (int dt = 0; dt < G::DB; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = dt * 32 + 8 * g + 4 * lhi;
          if (d0 < D) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[dt][4 * g + r];
            *reinterpret_cast<bf16x4*>(dst + d0) = o;
          }
        }
    }
  }
}

// dx = (g - xn*<g,xn>*d/(D*(d-eps)))/d for one token per lane; g in O^T-style accumulators, xn re-derived from the raw row
template <int D>
__device__ __forceinline__ void norm_bwd_store_g(f32x16 (&g)[Geo<D>::DB], const bf16* __restrict__ raw, float dn,
                                                 bf16* __restrict__ dst, int lhi, bool valid) {
  using G = Geo<D>;
  const float inv = 1.0f / dn;
  float dot = 0.f;
  if (valid) {
#pragma unroll
    for (int dt = 0; dt < G::DB; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d0 = dt * 32 + 8 * gq + 4 * lhi;
        if (d0 < D) {
          bf16x4 v = *reinterpret_cast<const bf16x4*>(raw + d0);
#pragma unroll
          for (int r = 0; r < 4; ++r) dot += g[dt][4 * gq + r] * (float)(bf16)((float)v[r] * inv);
        }
      }
  }
  dot += __shfl_xor(dot, 32, 64);
  const float s = dn - NORM_EPS;
  const float coef = s > 0.f ? dot * dn / ((float)D * s) : 0.f;
  if (valid) {
#pragma unroll
    for (int dt = 0; dt < G::DB; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d0 = dt * 32 + 8 * gq + 4 * lhi;
        if (d0 < D) {
          bf16x4 v = *reinterpret_cast<const bf16x4*>(raw + d0);
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xn = (float)(bf16)((float)v[r] * inv);
            o[r] = (bf16)((g[dt][4 * gq + r] - xn * coef) * inv);
          }
          *reinterpret_cast<bf16x4*>(dst + d0) = o;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ backward
template <int D, int NT>
__global__ __launch_bounds__(256) void k_attn_bwd_g(const bf16* __restrict__ qkv, const bf16* __restrict__ y,
                                                      const bf16* __restrict__ gy, bf16* __restrict__ gqkv, int N, int C,
                                                      int heads) {
  using G = Geo<D>;
  constexpr int NP = NT * 32, RS = G::RS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tileA = smem;                                               // [TK][RS]
  char* tileB = smem + TK * RS;                                     // [TK][RS]
  float* dsave = reinterpret_cast<float*>(smem + 2 * TK * RS);      // [3][NP]
  float* st_m = dsave + 3 * NP;                                     // [NP] row max (scaled)
  float* st_l = st_m + NP;                                          // [NP] 1/sum
  float* st_d = st_l + NP;                                          // [NP] delta = <dO, O>
  const int b = blockIdx.x / heads, head = blockIdx.x % heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const long rs = 3L * C;
  const bf16* src = qkv + ((long)b * N) * rs + (long)head * 3 * D;
  const bf16* gsrc = gy + ((long)b * N) * C + (long)head * D;
  const bf16* ysrc = y + ((long)b * N) * C + (long)head * D;
  bf16* gdst = gqkv + ((long)b * N) * rs + (long)head * 3 * D;
  row_norms<D>(src, rs, N, NP, dsave);
  for (int row = threadIdx.x; row < NP; row += 256) {
    float dl = 0.f;
    if (row < N) {
#pragma unroll 2
      for (int c8 = 0; c8 < D / 8; ++c8) {
        float g[8], o[8];
        load8(gsrc + (long)row * C + c8 * 8, g);
        load8(ysrc + (long)row * C + c8 * 8, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) dl += g[i] * o[i];
      }
    }
    st_d[row] = dl;
  }
  __syncthreads();

  const float scale = rsqrtf((float)D);
  const int tr_row = (lane & 15) >> 2;
  const int tr_col = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  // ================= pass 1: query-major -> softmax stats, dQ =================
  for (int round = 0; round < (NT + 3) / 4; ++round) {
    const int qb = round * 4 + wave;
    const bool active = qb < NT;
    const int qi = qb * 32 + l31;
    const int qrow = active ? qi : N;  // N = "no row": zeros
    f32x16 St[NT];
    float m = -1e30f;
    {
      bf16x8 bq[G::DT];
      load_block_b<D>(src, rs, dsave, qrow, N, lhi, bq);
#pragma unroll
      for (int t = 0; t < (NT + 1) / 2; ++t) {
        __syncthreads();
        stage_tile<D>(src + D, rs, dsave + NP, t * TK, N, tileA);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int kt = 2 * t + kk;
          if (kt < NT) {
            St[kt] = score_tile_g<D>(tileA + (kk * 32 + l31) * RS + lhi * 16, bq);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
              float s = key < N ? St[kt][r] * scale : -1e30f;
              St[kt][r] = s;
              m = fmaxf(m, s);
            }
          }
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __expf(St[kt][r] - m);
        St[kt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);
    const float linv = (active && qi < N) ? 1.0f / l : 0.f;
    if (active && lhi == 0) {
      st_m[qi] = m;
      st_l[qi] = linv;
    }
    const float delta = active ? st_d[qi] : 0.f;
    bf16x8 bdo[G::DT];
    load_block_b<D>(gsrc, (long)C, nullptr, qrow, N, lhi, bdo);
    f32x16 accq[G::DB];
#pragma unroll
    for (int dt = 0; dt < G::DB; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accq[dt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < (NT + 1) / 2; ++t) {
      __syncthreads();
      stage_tile<D>(src + D, rs, dsave + NP, t * TK, N, tileA);
      stage_tile<D>(src + 2 * D, rs, dsave + 2 * NP, t * TK, N, tileB);
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int kt = 2 * t + kk;
        if (kt < NT) {
          f32x16 dP = score_tile_g<D>(tileB + (kk * 32 + l31) * RS + lhi * 16, bdo);  // dP^T tile: keys x queries
#pragma unroll
          for (int r = 0; r < 16; ++r) dP[r] = St[kt][r] * linv * (dP[r] - delta) * scale;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 ds = pack8_g(dP, s2);
            const int row0 = kk * 32 + 16 * s2 + 4 * lhi + tr_row;
#pragma unroll
            for (int dt = 0; dt < G::DB; ++dt) {
              const char* p0 = tileA + row0 * RS + dt * 64 + tr_col;
              accq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_g(p0, p0 + 8 * RS), ds, accq[dt], 0, 0, 0);
            }
          }
        }
      }
    }
    const bool valid = active && qi < N;
    const int qs = valid ? qi : 0;
    norm_bwd_store_g<D>(accq, src + (long)qs * rs, dsave[0 * NP + qs], gdst + (long)qs * rs, lhi, valid);
  }
  __syncthreads();  // softmax stats of every query block are now in LDS

  // ================= pass 2: key-major -> dK, dV =================
  for (int round = 0; round < (NT + 3) / 4; ++round) {
    const int kb = round * 4 + wave;
    const bool active = kb < NT;
    const int ki = kb * 32 + l31;
    const int krow = active ? ki : N;
    bf16x8 bk[G::DT], bv[G::DT];
    load_block_b<D>(src + D, rs, dsave + NP, krow, N, lhi, bk);
    load_block_b<D>(src + 2 * D, rs, dsave + 2 * NP, krow, N, lhi, bv);
    f32x16 acck[G::DB], accv[G::DB];
#pragma unroll
    for (int dt = 0; dt < G::DB; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acck[dt][r] = 0.f;
        accv[dt][r] = 0.f;
      }
#pragma unroll 1
    for (int t = 0; t < (NT + 1) / 2; ++t) {
      __syncthreads();
      stage_tile<D>(src, rs, dsave, t * TK, N, tileA);               // normalised Q rows
      stage_tile<D>(gsrc, (long)C, nullptr, t * TK, N, tileB);       // dO rows
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int qt = 2 * t + kk;
        if (qt < NT) {
          // S tile: rows = queries (registers), cols = keys (lane)
          f32x16 S = score_tile_g<D>(tileA + (kk * 32 + l31) * RS + lhi * 16, bk);
          f32x16 dP = score_tile_g<D>(tileB + (kk * 32 + l31) * RS + lhi * 16, bv);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int q0 = qt * 32 + 8 * g + 4 * lhi;
            const f32x4 mm = *reinterpret_cast<const f32x4*>(st_m + q0);
            const f32x4 ll = *reinterpret_cast<const f32x4*>(st_l + q0);
            const f32x4 dd = *reinterpret_cast<const f32x4*>(st_d + q0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = (active && ki < N) ? __expf(S[4 * g + r] * scale - mm[r]) * ll[r] : 0.f;
              S[4 * g + r] = p;
              dP[4 * g + r] = p * (dP[4 * g + r] - dd[r]) * scale;
            }
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pb = pack8_g(S, s2), ds = pack8_g(dP, s2);
            const int row0 = kk * 32 + 16 * s2 + 4 * lhi + tr_row;
#pragma unroll
            for (int dt = 0; dt < G::DB; ++dt) {
              const char* p0 = tileB + row0 * RS + dt * 64 + tr_col;
              accv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_g(p0, p0 + 8 * RS), pb, accv[dt], 0, 0, 0);
              const char* p1 = tileA + row0 * RS + dt * 64 + tr_col;
              acck[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_g(p1, p1 + 8 * RS), ds, acck[dt], 0, 0, 0);
            }
          }
        }
      }
    }
    const bool valid = active && ki < N;
    const int ks = valid ? ki : 0;
    norm_bwd_store_g<D>(acck, src + (long)ks * rs + D, dsave[1 * NP + ks], gdst + (long)ks * rs + D, lhi, valid);
    norm_bwd_store_g<D>(accv, src + (long)ks * rs + 2 * D, dsave[2 * NP + ks], gdst + (long)ks * rs + 2 * D, lhi, valid);
  }
}

template <int D, int NT>
void launch_fwd_g(const void* qkv, void* y, int B, int N, int C, int heads, hipStream_t st) {
  auto kern = k_attn_fwd_g<D, NT>;
  EDM_MAX_LDS(kern, 160 * 1024);
  const size_t lds = (size_t)TK * Geo<D>::RS + (size_t)3 * NT * 32 * sizeof(float);
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(256), lds, st, (const bf16*)qkv, (bf16*)y, N, C, heads);
}
template <int D, int NT>
void launch_bwd_g(const void* qkv, const void* y, const void* gy, void* gqkv, int B, int N, int C, int heads,
                  hipStream_t st) {
  auto kern = k_attn_bwd_g<D, NT>;
  EDM_MAX_LDS(kern, 160 * 1024);
  const size_t lds = (size_t)2 * TK * Geo<D>::RS + (size_t)6 * NT * 32 * sizeof(float);
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(256), lds, st, (const bf16*)qkv, (const bf16*)y, (const bf16*)gy,
                     (bf16*)gqkv, N, C, heads);
}

template <int D>
void dispatch_fwd(const void* qkv, void* y, int B, int N, int C, int heads, hipStream_t st) {
  const int nt = (N + 31) / 32;
  if (nt <= 2) launch_fwd_g<D, 2>(qkv, y, B, N, C, heads, st);
  else if (nt <= 4) launch_fwd_g<D, 4>(qkv, y, B, N, C, heads, st);
  else launch_fwd_g<D, 8>(qkv, y, B, N, C, heads, st);
}
template <int D>
void dispatch_bwd(const void* qkv, const void* y, const void* gy, void* gqkv, int B, int N, int C, int heads,
                  hipStream_t st) {
  const int nt = (N + 31) / 32;
  if (nt <= 2) launch_bwd_g<D, 2>(qkv, y, gy, gqkv, B, N, C, heads, st);
  else if (nt <= 4) launch_bwd_g<D, 4>(qkv, y, gy, gqkv, B, N, C, heads, st);
  else launch_bwd_g<D, 8>(qkv, y, gy, gqkv, B, N, C, heads, st);
}

}  // namespace

// Called by edm_attention_fwd/bwd (attention.hip) for head_dim != 64.  Returns EDM_ERR_UNSUPPORTED (-3) for head dims
// that are not instantiated (built: 32, 128, 144, 192 -- every head_dim of the reference's configs besides 64).
int edm_attention_fwd_generic(const void* qkv, void* y, int B, int N, int C, int heads, int D, hipStream_t st) {
  switch (D) {
    case 32: dispatch_fwd<32>(qkv, y, B, N, C, heads, st); break;
    case 128: dispatch_fwd<128>(qkv, y, B, N, C, heads, st); break;
    case 144: dispatch_fwd<144>(qkv, y, B, N, C, heads, st); break;
    case 192: dispatch_fwd<192>(qkv, y, B, N, C, heads, st); break;
    default: return EDM_ERR_UNSUPPORTED;
  }
  return EDM_OK;
}
int edm_attention_bwd_generic(const void* qkv, const void* y, const void* gy, void* gqkv, int B, int N, int C, int heads,
                              int D, hipStream_t st) {
  switch (D) {
    case 32: dispatch_bwd<32>(qkv, y, gy, gqkv, B, N, C, heads, st); break;
    case 128: dispatch_bwd<128>(qkv, y, gy, gqkv, B, N, C, heads, st); break;
    case 144: dispatch_bwd<144>(qkv, y, gy, gqkv, B, N, C, heads, st); break;
    case 192: dispatch_bwd<192>(qkv, y, gy, gqkv, B, N, C, heads, st); break;
    default: return EDM_ERR_UNSUPPORTED;
  }
  return EDM_OK;
}

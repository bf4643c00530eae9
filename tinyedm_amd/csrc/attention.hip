// Cosine self-attention core of the reference's CosineAttention (networks.py:194-202):
//   q,k,v rows pixel-normalised over head_dim (:195), softmax(q k^T / sqrt(d)) v (:201).
// Low-resolution only (N = H*W <= 256 tokens, head_dim 64), so one workgroup owns one
// (sample, head): Q,K,V live in LDS for the whole kernel and the N x N score matrix never
// leaves registers.
//
// qkv   [B*N, 3C] bf16, channel order per token = [head][q|k|v][d]   (the qkv conv's output rows
//        are permuted into this order by edm_weight_prep)
// y     [B*N, C]  bf16, channel = head*d + dd
//
// MFMA formulation ("query on the lane"): S^T = K Q^T puts one query per lane column and its keys
// in the accumulator registers, so the softmax is in-register plus one lane^32 exchange, and the
// probability tile is fed straight back as the B operand of O^T = V^T P^T (accumulator-as-operand:
// its k-order permutation is absorbed into the row addresses of the transposing V reads).
// The backward runs the same trick twice: query-major for dQ, key-major for dK and dV.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef short4v __attribute__((address_space(3))) * lds_s4p;

constexpr int D = 64;
constexpr int RS = 2 * D + 16;  // padded LDS row bytes (bank-conflict-free ds_read_b128 fragments)

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
  short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p0));
  short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p1));
  short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s2) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)x[8 * s2 + j];
  return o;
}
__device__ __forceinline__ const bf16x8& ld128(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// load + pixel-normalise NIMG [N][D] slices (stride between slices = D channels) into LDS images.  The trip count is a
// compile-time constant (NIMG * NP * 8 / THREADS 16-byte pieces per thread: 12 for three 256-token images) and ALL of a
// thread's loads are issued before the first is consumed: with one workgroup per CU nothing else hides the HBM latency,
// and the rolled loop paid it once per piece (round 3: fwd 35 -> see DESIGN 3.3).
template <bool SAVE, int NIMG, int NP, int THREADS>
__device__ __forceinline__ void stage_normalised(const bf16* __restrict__ src, long row_stride, char* img0, int N,
                                                 float* dsave) {
  constexpr int TOTAL = NIMG * NP * 8;
  constexpr int IT = (TOTAL + THREADS - 1) / THREADS;
  u32x4 raw[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int idx = threadIdx.x + it * THREADS;
    const int which = idx / (NP * 8), rem = idx % (NP * 8);
    const int row = rem >> 3, c8 = rem & 7;
    raw[it] = u32x4{0u, 0u, 0u, 0u};
    if (idx < TOTAL && row < N)
      raw[it] = *reinterpret_cast<const u32x4*>(src + (long)row * row_stride + which * D + c8 * 8);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int idx = threadIdx.x + it * THREADS;
    const int which = idx / (NP * 8), rem = idx % (NP * 8);
    const int row = rem >> 3, c8 = rem & 7;
    const bf16x8 bv = __builtin_bit_cast(bf16x8, raw[it]);
    float v[8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = (float)bv[i];
      ss += v[i] * v[i];
    }
    ss = group_sum<8>(ss);
    const float dn = NORM_EPS + sqrtf(ss) * 0.125f;  // 1/sqrt(64)
    const float inv = 1.0f / dn;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= inv;
    if (idx < TOTAL) {
      store8(reinterpret_cast<bf16*>(img0 + (long)which * NP * RS + row * RS + c8 * 16), v);
      if (SAVE && c8 == 0) dsave[which * NP + row] = dn;
    }
  }
}

// S^T tile (32 keys x 32 queries) = K_tile Q_blk^T over D
__device__ __forceinline__ f32x16 score_tile(const char* a_rows, const bf16x8 (&bq)[D / 16]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < D / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld128(a_rows + s * 32), bq[s], acc, 0, 0, 0);
  return acc;
}

// One wave per 32-query block: NT = 8 (256 tokens) runs 8 waves -- two per SIMD, so one wave's softmax (VALU/exp)
// overlaps the other's MFMAs (fwd 56 -> 37 us, bwd 126 -> 77 us on the CIFAR-10 16x16 layers).  Smaller token counts
// keep 4 waves: the extra waves have no query block but speed up the staging pass (2 waves: 9.9 -> 12.8 us at 8x8).
template <int NT>
constexpr int attn_threads() { return NT >= 8 ? 512 : 256; }

template <int NT>
__global__ __launch_bounds__(attn_threads<NT>()) void k_attn_fwd(const bf16* __restrict__ qkv, bf16* __restrict__ y, int N, int C,
                                                    int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT * 32;
  char* Qn = smem;
  char* Kn = Qn + NP * RS;
  char* Vn = Kn + NP * RS;
  const int b = blockIdx.x / heads, head = blockIdx.x % heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const bf16* src = qkv + ((long)b * N) * 3 * C + head * 3 * D;
  stage_normalised<false, 3, NP, attn_threads<NT>()>(src, 3L * C, Qn, N, nullptr);
  __syncthreads();

  // The softmax is the vector-ALU bill of this kernel (128 scores per lane against 64 MFMAs per wave, two waves per SIMD):
  // round 4 cut it from ~9 to ~4 instructions per score -- the 1/sqrt(d) scale and log2(e) ride in ONE fma in front of
  // v_exp_f32 (the max is taken on the raw scores), the key mask is compiled out when every tile is full (N = 32 NT: the
  // 16x16 and 8x8 maps), and the 1/sum normalisation is applied to the 32 output values instead of the 128 probabilities
  // (the probabilities go to the second MFMA unnormalised, in (0, 1]: the same relative bf16 rounding).
  const float sl2 = 0.125f * 1.44269504088896341f;  // 1/sqrt(64) * log2(e)
  const bool full = N == NT * 32;
  // transposing-read geometry for V^T fragments
  const int tr_row = (lane & 15) >> 2;
  const int tr_col = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  for (int qb = wave; qb < NT; qb += attn_threads<NT>() / 64) {
    bf16x8 bq[D / 16];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) bq[s] = ld128(Qn + (qb * 32 + l31) * RS + s * 32 + lhi * 16);
    f32x16 St[NT];
    float m = -1e30f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      St[kt] = score_tile(Kn + (kt * 32 + l31) * RS + lhi * 16, bq);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!full) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          St[kt][r] = key < N ? St[kt][r] : -1e30f;
        }
        m = fmaxf(m, St[kt][r]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mb = m * sl2;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(fmaf(St[kt][r], sl2, -mb));
        St[kt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);
    const float linv = 1.0f / l;
    f32x16 acc[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack8(St[kt], s2);
        const int row0 = kt * 32 + 16 * s2 + 4 * lhi + tr_row;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          const char* p0 = Vn + row0 * RS + dt * 64 + tr_col;
          acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(p0, p0 + 8 * RS), pb, acc[dt], 0, 0, 0);
        }
      }
    }
    const int qi = qb * 32 + l31;
    if (qi < N) {
      bf16* dst = y + ((long)b * N + qi) * C + head * D;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[dt][4 * g + r] * linv);
          *reinterpret_cast<bf16x4*>(dst + dt * 32 + 8 * g + 4 * lhi) = o;
        }
    }
  }
}

// dx = (g - xn*<g,xn>*d/(D*(d-eps)))/d for one token per lane, g given as O^T-style accumulators
__device__ __forceinline__ void norm_bwd_store(f32x16 (&g)[D / 32], const char* xn_row, float dn, bf16* dst, int lhi,
                                               bool valid) {
  float xn[D / 32][16];
  float dot = 0.f;
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      bf16x4 v = *reinterpret_cast<const bf16x4*>(xn_row + (dt * 32 + 8 * gq + 4 * lhi) * 2);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xn[dt][4 * gq + r] = (float)v[r];
        dot += g[dt][4 * gq + r] * xn[dt][4 * gq + r];
      }
    }
  dot += __shfl_xor(dot, 32, 64);
  const float s = dn - NORM_EPS;
  const float coef = s > 0.f ? dot * dn / ((float)D * s) : 0.f;
  const float inv = 1.0f / dn;
  if (valid) {
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)((g[dt][4 * gq + r] - xn[dt][4 * gq + r] * coef) * inv);
        *reinterpret_cast<bf16x4*>(dst + dt * 32 + 8 * gq + 4 * lhi) = o;
      }
  }
}

template <int NT>
__global__ __launch_bounds__(attn_threads<NT>()) void k_attn_bwd(const bf16* __restrict__ qkv, const bf16* __restrict__ y,
                                                    const bf16* __restrict__ gy, bf16* __restrict__ gqkv, int N, int C,
                                                    int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT * 32;
  char* Qn = smem;
  char* Kn = Qn + NP * RS;
  char* Vn = Kn + NP * RS;
  char* dO = Vn + NP * RS;
  float* dsave = reinterpret_cast<float*>(dO + NP * RS);  // [3][NP]
  float* st_m = dsave + 3 * NP;                            // [NP] row max (scaled)
  float* st_l = st_m + NP;                                 // [NP] 1/sum
  float* st_d = st_l + NP;                                 // [NP] delta = <dO, O>
  const int b = blockIdx.x / heads, head = blockIdx.x % heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const bf16* src = qkv + ((long)b * N) * 3 * C + head * 3 * D;
  // dO image + delta: its loads are issued FIRST (before the q/k/v staging consumes anything), all at once
  constexpr int THR = attn_threads<NT>();
  constexpr int ITD = (NP * 8 + THR - 1) / THR;
  u32x4 graw[ITD], oraw[ITD];
#pragma unroll
  for (int it = 0; it < ITD; ++it) {
    const int idx = threadIdx.x + it * THR;
    const int row = idx >> 3, c8 = idx & 7;
    graw[it] = oraw[it] = u32x4{0u, 0u, 0u, 0u};
    if (idx < NP * 8 && row < N) {
      graw[it] = *reinterpret_cast<const u32x4*>(gy + ((long)b * N + row) * C + head * D + c8 * 8);
      oraw[it] = *reinterpret_cast<const u32x4*>(y + ((long)b * N + row) * C + head * D + c8 * 8);
    }
  }
  stage_normalised<true, 3, NP, THR>(src, 3L * C, Qn, N, dsave);
#pragma unroll
  for (int it = 0; it < ITD; ++it) {
    const int idx = threadIdx.x + it * THR;
    const int row = idx >> 3, c8 = idx & 7;
    const bf16x8 gv = __builtin_bit_cast(bf16x8, graw[it]), ov = __builtin_bit_cast(bf16x8, oraw[it]);
    float dl = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) dl += (float)gv[i] * (float)ov[i];
    dl = group_sum<8>(dl);
    if (idx < NP * 8) {
      *reinterpret_cast<u32x4*>(dO + row * RS + c8 * 16) = graw[it];
      if (c8 == 0) st_d[row] = dl;
    }
  }
  __syncthreads();

  const float scale = 0.125f;
  const float sl2 = 0.125f * 1.44269504088896341f;  // scale * log2(e): see k_attn_fwd
  const bool full = N == NT * 32;
  const int tr_row = (lane & 15) >> 2;
  const int tr_col = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  // ================= pass 1: query-major -> softmax stats, dQ =================
  for (int qb = wave; qb < NT; qb += attn_threads<NT>() / 64) {
    bf16x8 bq[D / 16], bdo[D / 16];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) {
      bq[s] = ld128(Qn + (qb * 32 + l31) * RS + s * 32 + lhi * 16);
      bdo[s] = ld128(dO + (qb * 32 + l31) * RS + s * 32 + lhi * 16);
    }
    const int qi = qb * 32 + l31;
    f32x16 St[NT];
    float m = -1e30f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      St[kt] = score_tile(Kn + (kt * 32 + l31) * RS + lhi * 16, bq);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!full) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          St[kt][r] = key < N ? St[kt][r] : -1e30f;
        }
        m = fmaxf(m, St[kt][r]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mb = m * sl2;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(fmaf(St[kt][r], sl2, -mb));
        St[kt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);
    const float linv = (qi < N) ? 1.0f / l : 0.f;
    if (lhi == 0) {
      st_m[qi] = mb;          // (the row maximum times scale * log2(e): pass 2 feeds it to the same fma)
      st_l[qi] = linv;
    }
    const float delta = st_d[qi];
    const float ls = linv * scale;
    f32x16 accq[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accq[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 dP = score_tile(Vn + (kt * 32 + l31) * RS + lhi * 16, bdo);  // dP^T tile: keys x queries
#pragma unroll
      for (int r = 0; r < 16; ++r) dP[r] = St[kt][r] * ls * (dP[r] - delta);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 ds = pack8(dP, s2);
        const int row0 = kt * 32 + 16 * s2 + 4 * lhi + tr_row;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          const char* p0 = Kn + row0 * RS + dt * 64 + tr_col;
          accq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(p0, p0 + 8 * RS), ds, accq[dt], 0, 0, 0);
        }
      }
    }
    norm_bwd_store(accq, Qn + qi * RS, dsave[0 * NP + qi], gqkv + ((long)b * N + qi) * 3 * C + head * 3 * D, lhi,
                   qi < N);
  }
  __syncthreads();  // softmax stats of every query block are now in LDS

  // ================= pass 2: key-major -> dK, dV =================
  for (int kb = wave; kb < NT; kb += attn_threads<NT>() / 64) {
    bf16x8 bk[D / 16], bv[D / 16];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) {
      bk[s] = ld128(Kn + (kb * 32 + l31) * RS + s * 32 + lhi * 16);
      bv[s] = ld128(Vn + (kb * 32 + l31) * RS + s * 32 + lhi * 16);
    }
    const int ki = kb * 32 + l31;
    f32x16 acck[D / 32], accv[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acck[dt][r] = 0.f;
        accv[dt][r] = 0.f;
      }
#pragma unroll 1
    for (int qt = 0; qt < NT; ++qt) {
      // S tile: rows = queries (registers), cols = keys (lane)
      f32x16 S = score_tile(Qn + (qt * 32 + l31) * RS + lhi * 16, bk);
      f32x16 dP = score_tile(dO + (qt * 32 + l31) * RS + lhi * 16, bv);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int q0 = qt * 32 + 8 * g + 4 * lhi;
        const f32x4 mm = *reinterpret_cast<const f32x4*>(st_m + q0);
        const f32x4 ll = *reinterpret_cast<const f32x4*>(st_l + q0);
        const f32x4 dd = *reinterpret_cast<const f32x4*>(st_d + q0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(S[4 * g + r], sl2, -mm[r])) * ll[r];
          if (!full) p = (ki < N) ? p : 0.f;
          S[4 * g + r] = p;
          dP[4 * g + r] = p * scale * (dP[4 * g + r] - dd[r]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack8(S, s2), ds = pack8(dP, s2);
        const int row0 = qt * 32 + 16 * s2 + 4 * lhi + tr_row;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          const char* p0 = dO + row0 * RS + dt * 64 + tr_col;
          accv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(p0, p0 + 8 * RS), pb, accv[dt], 0, 0, 0);
          const char* p1 = Qn + row0 * RS + dt * 64 + tr_col;
          acck[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(p1, p1 + 8 * RS), ds, acck[dt], 0, 0, 0);
        }
      }
    }
    bf16* dst = gqkv + ((long)b * N + ki) * 3 * C + head * 3 * D;
    norm_bwd_store(acck, Kn + ki * RS, dsave[1 * NP + ki], dst + D, lhi, ki < N);
    norm_bwd_store(accv, Vn + ki * RS, dsave[2 * NP + ki], dst + 2 * D, lhi, ki < N);
  }
}

template <int NT>
size_t lds_fwd() { return (size_t)3 * NT * 32 * RS; }
template <int NT>
size_t lds_bwd() { return (size_t)4 * NT * 32 * RS + (size_t)6 * NT * 32 * sizeof(float); }

template <int NT>
void launch_fwd(const void* qkv, void* y, int B, int N, int C, int heads, hipStream_t st) {
  auto kern = k_attn_fwd<NT>;
  EDM_MAX_LDS(kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(attn_threads<NT>()), lds_fwd<NT>(), st, (const bf16*)qkv, (bf16*)y, N, C, heads);
}
template <int NT>
void launch_bwd(const void* qkv, const void* y, const void* gy, void* gqkv, int B, int N, int C, int heads,
                hipStream_t st) {
  auto kern = k_attn_bwd<NT>;
  EDM_MAX_LDS(kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(B * heads), dim3(attn_threads<NT>()), lds_bwd<NT>(), st, (const bf16*)qkv, (const bf16*)y,
                     (const bf16*)gy, (bf16*)gqkv, N, C, heads);
}

int check(int B, int N, int C, int heads, const char* who) {
  EDM_REQUIRE(B > 0 && N > 0 && heads > 0 && C > 0 && C % heads == 0, "%s: bad args", who);
  EDM_REQUIRE(N <= 256, "%s: at most 256 tokens (got %d)", who, N);
  return EDM_OK;
}

}  // namespace

// attention_generic.hip: streamed-operand kernels for head_dim != 64
int edm_attention_fwd_generic(const void* qkv, void* y, int B, int N, int C, int heads, int D, hipStream_t st);
int edm_attention_bwd_generic(const void* qkv, const void* y, const void* gy, void* gqkv, int B, int N, int C, int heads,
                              int D, hipStream_t st);

extern "C" int edm_attention_fwd(const void* qkv, void* y, int B, int N, int C, int heads, hipStream_t st) {
  if (int rc = check(B, N, C, heads, "attention_fwd")) return rc;
  if (C != heads * D) {
    const int rc = edm_attention_fwd_generic(qkv, y, B, N, C, heads, C / heads, st);
    EDM_REQUIRE(rc == EDM_OK, "attention_fwd: head_dim %d is not built (64, 32, 128, 144, 192 are)", C / heads);
    EDM_CHECK_LAUNCH("attention_fwd (generic)");
    return EDM_OK;
  }
  const int nt = (N + 31) / 32;
  if (nt <= 1) launch_fwd<1>(qkv, y, B, N, C, heads, st);
  else if (nt <= 2) launch_fwd<2>(qkv, y, B, N, C, heads, st);
  else if (nt <= 4) launch_fwd<4>(qkv, y, B, N, C, heads, st);
  else launch_fwd<8>(qkv, y, B, N, C, heads, st);
  EDM_CHECK_LAUNCH("attention_fwd");
  return EDM_OK;
}

extern "C" int edm_attention_bwd(const void* qkv, const void* y, const void* gy, void* gqkv, int B, int N, int C,
                                 int heads, hipStream_t st) {
  if (int rc = check(B, N, C, heads, "attention_bwd")) return rc;
  if (C != heads * D) {
    const int rc = edm_attention_bwd_generic(qkv, y, gy, gqkv, B, N, C, heads, C / heads, st);
    EDM_REQUIRE(rc == EDM_OK, "attention_bwd: head_dim %d is not built (64, 32, 128, 144, 192 are)", C / heads);
    EDM_CHECK_LAUNCH("attention_bwd (generic)");
    return EDM_OK;
  }
  const int nt = (N + 31) / 32;
  if (nt <= 1) launch_bwd<1>(qkv, y, gy, gqkv, B, N, C, heads, st);
  else if (nt <= 2) launch_bwd<2>(qkv, y, gy, gqkv, B, N, C, heads, st);
  else if (nt <= 4) launch_bwd<4>(qkv, y, gy, gqkv, B, N, C, heads, st);
  else launch_bwd<8>(qkv, y, gy, gqkv, B, N, C, heads, st);
  EDM_CHECK_LAUNCH("attention_bwd");
  return EDM_OK;
}

// CosineAttention with the qkv projection INSIDE the attention kernel (round 5) -- reference networks.py:191-202:
//   qkv = qkv_conv(x) -> view(b, heads, d, 3, hw) -> pixel_norm over d -> softmax(q k^T / sqrt(d)) v
// for the shape the benchmarked nets run it at: C = 256 channels, 4 heads of head_dim 64, N = H*W <= 256 tokens.
// The qkv tensor (3C channels per token) never exists in HBM: a workgroup owns `HP` heads of one sample,
//   1. every wave keeps the 32 tokens x 256 channels of its token block as MFMA B fragments in 64 registers (loaded once),
//   2. the head's 192 weight rows stream through a 3-deep LDS ring by LDS-DMA (counted vmcnt, one barrier per 32-channel
//      chunk) and are the A operand: the product lands TOKEN-ON-THE-LANE, channels in the accumulator registers -- the
//      layout in which the pixel norm is a register sum + one lane^32 exchange,
//   3. normalised K / V go to LDS images (K in accumulator order: a row read of it matches the k order of the Q fragments
//      built from the accumulators, common `pack8`; V in channel order for the transposing reads), Q stays in registers,
//   4. the scores are STREAMED: cosine attention bounds every logit by |q.k| / sqrt(d) <= sqrt(d) = 8 (q and k are
//      pixel-normalised), so softmax needs no running maximum -- p = exp(s - 8) is in [e^-16, ~1], summed in fp32 -- and a
//      32-key tile goes QK^T -> exp -> P.V without any other tile being live (16 score registers instead of 128).
// The backward (k_attn_qkv_bwd) recomputes q, k, v the same way from x, forms dO = b * gout . W_out for its head in the same
// layout, and runs attention.hip's two passes from LDS images; it writes gqkv (the qkv weight gradient and the input
// gradient are GEMMs over all heads / samples: separate launches).
//
// x     [B*N, C]   bf16 NHWC tokens          Wqkv [3C, C] bf16 forward pack, rows in [head][q|k|v][d] order (edm_weight_prep)
// y     [B*N, C]   bf16, channel = head*64 + d
// stat  [B, heads, N] fp32: 1 / sum_j exp(s_ij - 8) (saved for the backward)
#include "common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef short4v __attribute__((address_space(3))) * lds_s4p;

constexpr int D = 64, C = 256;
constexpr int RS = 2 * D + 16;          // padded image row (bytes): conflict-free ds_read_b128 / ds_write_b128 at a row per lane
constexpr int KC = 32;                  // channels per streamed weight chunk
constexpr int NCH = C / KC;             // 8 chunks per head
constexpr int WROWB = KC * 2;           // 64-byte LDS rows of a chunk
constexpr int WRING = 3;
constexpr float LOG2E = 1.44269504088896341f;

__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ const bf16x8& ld128(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
  short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p0));
  short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p1));
  short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}
// registers 8*s2 .. 8*s2+7 of a 32x32 accumulator tile as the 8-element operand of k-step s2 (element j <-> tile row
// 16*s2 + 8*(j>>2) + 4*(lane>>5) + (j&3): the "accumulator order" of the K image and the Q fragments)
__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s2) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)x[8 * s2 + j];
  return o;
}
__device__ __forceinline__ float rbf(float v) { return (float)(bf16)v; }

// S^T tile (32 keys x 32 queries) = K_tile Q_blk^T over D
__device__ __forceinline__ f32x16 score_tile(const char* a_rows, const bf16x8 (&bq)[D / 16]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < D / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld128(a_rows + s * 32), bq[s], acc, 0, 0, 0);
  return acc;
}

// ---- streamed GEMM  acc[rb] (32 rows x 32 tokens) = Wrows[rb*32 .., 0..255] . X^T   for NRB row blocks
// The ROWS rows x 256 channels of the A operand go through the LDS ring in NCH chunks of 32 channels (64-byte rows,
// 16-byte pieces XOR-swizzled with (row >> 2) & 3 on the DMA source and on the fragment read: conv_igemm2.hip's layout).
// Chunks are numbered by a counter `t` that runs across calls (ring slot t % 3): a call may find its first two chunks
// already in flight (issued by the previous call's tail) and issues the first two of `next` (nullptr: none) in its own tail.
template <int NT, int NRB>
struct Streamed {
  static constexpr int ROWS = NRB * 32;
  static constexpr int SLOTS = ROWS / 16;            // 1-KiB DMA instructions per chunk
  static constexpr int SLOTB = ROWS * WROWB;         // bytes of a ring slot
  static constexpr int PER_WAVE = (SLOTS + NT - 1) / NT;

  // wave `wave` issues its share of chunk kc of the rows starting at `wrows` (row stride C elements) into ring slot `slot`
  static __device__ __forceinline__ void issue(const bf16* wrows, int kc, char* ring, int slot, int wave, int lane,
                                               const bf16* zeros) {
    const int drow = lane >> 2, dp = lane & 3;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int s = wave + NT * i;
      if (SLOTS % NT == 0 || s < SLOTS) {
        const int row = s * 16 + drow;
        const int c = dp ^ ((row >> 2) & 3);
        dma16(wrows + (long)row * C + kc * KC + c * 8, ring + slot * SLOTB + s * 1024);
      }
    }
    (void)zeros;
  }
  // leave the chunk issued AFTER the one being retired in flight (`more`), or drain
  static __device__ __forceinline__ void retire(bool more, int wave) {
    if (!more) {
      wait_vmcnt<0>();
    } else if (SLOTS % NT == 0) {
      wait_vmcnt<SLOTS / NT>();
    } else {                                        // uneven share (12 slots over 8 waves): the wave's own count
      if (wave < SLOTS % NT) wait_vmcnt<PER_WAVE>();
      else wait_vmcnt<PER_WAVE - 1>();
    }
  }
};

template <int NT>
__global__ __launch_bounds__(NT * 64) void k_attn_qkv_fwd(const bf16* __restrict__ x, const bf16* __restrict__ Wqkv,
                                                            bf16* __restrict__ y, float* __restrict__ stat,
                                                            const bf16* __restrict__ zeros, int B, int N, int heads,
                                                            int HP) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT * 32;
  using G = Streamed<NT, 6>;
  char* Kn = smem;
  char* Vn = Kn + NP * RS;
  char* ring = Vn + NP * RS;
  const int groups = heads / HP;
  const int id = blockIdx.x, xcd = id & 7, kk_ = id >> 3;
  const int b = (kk_ / groups) * 8 + xcd, hg = kk_ % groups;          // the head groups of a sample share an XCD (x in its L2)
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int tok = wave * 32 + l31;
  const bool tvalid = tok < N;

  // ---- the wave's token block as B fragments: bx[kk] = x[tok][16 kk + 8 lhi .. +8]
  bf16x8 bx[C / 16];
  {
    const bf16* xr = x + ((long)b * N + (tvalid ? tok : 0)) * C + lhi * 8;
#pragma unroll
    for (int kk = 0; kk < C / 16; ++kk) bx[kk] = *reinterpret_cast<const bf16x8*>(xr + kk * 16);
    if (!tvalid) {
#pragma unroll
      for (int kk = 0; kk < C / 16; ++kk) bx[kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  const int head0 = hg * HP;
  G::issue(Wqkv + (long)head0 * 3 * D * C, 0, ring, 0, wave, lane, zeros);
  G::issue(Wqkv + (long)head0 * 3 * D * C, 1, ring, 1, wave, lane, zeros);

  const int a_sw = (l31 >> 2) & 3;
  const int a_off0 = l31 * WROWB + (((0 + lhi) ^ a_sw) << 4);
  const int a_off1 = l31 * WROWB + (((2 + lhi) ^ a_sw) << 4);
  const float sl2 = 0.125f * LOG2E, c0 = 8.0f * LOG2E;       // p = exp2(s * sl2 - c0) = exp(s / sqrt(d) - 8)
  const bool full = N == NP;
  const int tr_row = (lane & 15) >> 2;
  const int tr_col = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  int base = 0;                                               // ring slot of this head's chunk 0
  for (int hi = 0; hi < HP; ++hi) {
    const int head = head0 + hi;
    const bf16* wrows = Wqkv + (long)head * 3 * D * C;
    f32x16 acc[6];
#pragma unroll
    for (int rb = 0; rb < 6; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      const bool next_head = hi + 1 < HP;
      const bool more = (kc + 1 < NCH) || next_head;
      if (kc == 0 && hi > 0) wait_vmcnt<0>();                 // (the previous head's y stores sit behind the prefetched chunks)
      else G::retire(more, wave);
      __builtin_amdgcn_s_barrier();
      {                                                        // chunk t+2: its slot was last read before this barrier
        int sl = base + (kc + 2) % WRING;
        sl -= sl >= WRING ? WRING : 0;
        if (kc + 2 < NCH) G::issue(wrows, kc + 2, ring, sl, wave, lane, zeros);
        else if (next_head) G::issue(wrows + (long)3 * D * C, kc + 2 - NCH, ring, sl, wave, lane, zeros);
      }
      int cur = base + kc % WRING;
      cur -= cur >= WRING ? WRING : 0;
      const char* wt = ring + cur * G::SLOTB;
      // all twelve fragment reads of the chunk are issued before its first MFMA (left to itself hipcc keeps ONE read in
      // flight: ds_read -> lgkmcnt(0) -> MFMA, twelve exposed LDS round trips per chunk)
      bf16x8 fa[6], fb[6];
#pragma unroll
      for (int rb = 0; rb < 6; ++rb) fa[rb] = ld128(wt + rb * 32 * WROWB + a_off0);
#pragma unroll
      for (int rb = 0; rb < 6; ++rb) fb[rb] = ld128(wt + rb * 32 * WROWB + a_off1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rb = 0; rb < 6; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rb], bx[2 * kc], acc[rb], 0, 0, 0);
#pragma unroll
      for (int rb = 0; rb < 6; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[rb], bx[2 * kc + 1], acc[rb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    base = (base + NCH) % WRING;

    // ---- pixel norm of q, k, v over the 64 channels of the token on this lane (networks.py:195), bf16 rounding points of
    // the unfused path: the conv output, then the normalised value
    float inv[3];
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      float ss = 0.f;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = rbf(acc[2 * w + rb][r]);
          acc[2 * w + rb][r] = v;
          ss += v * v;
        }
      ss += __shfl_xor(ss, 32, 64);
      inv[w] = 1.0f / (NORM_EPS + sqrtf(ss) * 0.125f);
    }
    bf16x8 bq[D / 16];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[rb][r] *= inv[0];
        acc[2 + rb][r] *= inv[1];
        acc[4 + rb][r] *= inv[2];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bq[2 * rb + s2] = pack8(acc[rb], s2);
        *reinterpret_cast<bf16x8*>(Kn + tok * RS + (32 * rb + 16 * s2 + 8 * lhi) * 2) = pack8(acc[2 + rb], s2);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 v4;
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = (bf16)acc[4 + rb][4 * g + r];
        *reinterpret_cast<bf16x4*>(Vn + tok * RS + (32 * rb + 8 * g + 4 * lhi) * 2) = v4;
      }
    }
    __syncthreads();

    // ---- streamed attention for the wave's 32 queries
    float l = 0.f;
    f32x16 o[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 S = score_tile(Kn + (kt * 32 + l31) * RS + lhi * 16, bq);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(fmaf(S[r], sl2, -c0));
        if (!full) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          p = key < N ? p : 0.f;
        }
        S[r] = p;
        l += p;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack8(S, s2);
        const int row0 = kt * 32 + 16 * s2 + 4 * lhi + tr_row;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          const char* p0 = Vn + row0 * RS + dt * 64 + tr_col;
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(p0, p0 + 8 * RS), pb, o[dt], 0, 0, 0);
        }
      }
    }
    l += __shfl_xor(l, 32, 64);
    const float linv = 1.0f / l;
    if (tvalid) {
      bf16* dst = y + ((long)b * N + tok) * C + head * D;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (bf16)(o[dt][4 * g + r] * linv);
          *reinterpret_cast<bf16x4*>(dst + dt * 32 + 8 * g + 4 * lhi) = ov;
        }
      if (stat && lhi == 0) stat[((long)b * heads + head) * N + tok] = linv;
    }
    // (the next head's K / V writes come after NCH more barriers: every wave has left this head's attention by then)
  }
}

template <int NT>
size_t lds_fwd() { return (size_t)2 * NT * 32 * RS + (size_t)WRING * Streamed<NT, 6>::SLOTB; }

template <int NT>
void launch_fwd(const void* x, const void* Wqkv, void* y, float* stat, int B, int N, int heads, int HP, hipStream_t st) {
  auto kern = k_attn_qkv_fwd<NT>;
  EDM_MAX_LDS(kern, 160 * 1024);
  const int groups = heads / HP;
  const int grid = ((B + 7) / 8) * 8 * groups;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT * 64), lds_fwd<NT>(), st, (const bf16*)x, (const bf16*)Wqkv, (bf16*)y, stat,
                     (const bf16*)edm_zero_page(), B, N, heads, HP);
}

}  // namespace

// 1 if the fused kernels cover (C, heads, N): head_dim 64, C = 256, 33..256 tokens
extern "C" int edm_attention_qkv_supported(int N, int C, int heads) {
  return (C == 256 && heads == 4 && N > 32 && N <= 256) ? 1 : 0;
}

// y = cosine attention of qkv_conv(x); stat (B, heads, N) fp32 receives the softmax normaliser the backward needs
extern "C" int edm_attention_qkv_fwd(const void* x, const void* Wqkv, void* y, void* stat, int B, int N, int C, int heads,
                                     int hp, hipStream_t st) {
  EDM_REQUIRE(x && Wqkv && y, "attention_qkv_fwd: null pointer");
  EDM_REQUIRE(B > 0 && edm_attention_qkv_supported(N, C, heads), "attention_qkv_fwd: C = 256, 4 heads, 33..256 tokens only "
              "(got C %d, heads %d, N %d)", C, heads, N);
  EDM_ZERO_PAGE(zero_page_, "attention_qkv_fwd");
  (void)zero_page_;
  const int nt = (N + 31) / 32;
  int HP = hp > 0 ? hp : (nt > 4 ? 2 : 1);
  EDM_REQUIRE(heads % HP == 0, "attention_qkv_fwd: heads per workgroup must divide heads");
  if (nt <= 2) launch_fwd<2>(x, Wqkv, y, (float*)stat, B, N, heads, HP, st);
  else if (nt <= 4) launch_fwd<4>(x, Wqkv, y, (float*)stat, B, N, heads, HP, st);
  else launch_fwd<8>(x, Wqkv, y, (float*)stat, B, N, heads, HP, st);
  EDM_CHECK_LAUNCH("attention_qkv_fwd");
  return EDM_OK;
}

// CosineAttention with the qkv projection INSIDE the attention kernel (round 5) -- reference networks.py:191-202:
//   qkv = qkv_conv(x) -> view(b, heads, d, 3, hw) -> pixel_norm over d -> softmax(q k^T / sqrt(d)) v
// for the shape the benchmarked nets run it at: C = 256 channels, 4 heads of head_dim 64, N = H*W <= 256 tokens.
// The qkv tensor (3C channels per token) never exists in HBM: a workgroup owns `HP` heads of one sample,
//   1. every wave keeps the 32 tokens x 256 channels of its token block as MFMA B fragments in 64 registers (loaded once),
//   2. the head's 192 weight rows stream through an LDS ring by LDS-DMA (4 or 7 slots, counted vmcnt, one barrier per
//      32-channel chunk, taken in the MIDDLE of the previous chunk's MFMAs) and are the A operand: the product lands TOKEN-ON-THE-LANE, channels in the accumulator registers -- the
//      layout in which the pixel norm is a register sum + one lane^32 exchange,
//   3. normalised K / V go to LDS images (K in accumulator order: a row read of it matches the k order of the Q fragments
//      built from the accumulators, common `pack8`; V in channel order for the transposing reads), Q stays in registers,
//   4. the scores are STREAMED: cosine attention bounds every logit by |q.k| / sqrt(d) <= sqrt(d) = 8 (q and k are
//      pixel-normalised), so softmax needs no running maximum -- p = exp(s - 8) is in [e^-16, ~1], summed in fp32 -- and a
//      32-key tile goes QK^T -> exp -> P.V with only the next tile's scores live beside it (32 score registers instead of
//      128).
// The backward (k_attn_qkv_bwd) recomputes q, k, v the same way from x, forms dO = b * gout . W_out for its head in the same
// layout, and runs attention.hip's two passes from LDS images; it writes gqkv (the qkv weight gradient and the input
// gradient are GEMMs over all heads / samples: separate launches).
//
// x     [B*N, C]   bf16 NHWC tokens          Wqkv [3C, C] bf16 forward pack, rows in [head][q|k|v][d] order (edm_weight_prep)
// y     [B*N, C]   bf16, channel = head*64 + d
// stat  [B, heads, N] fp32: 1 / sum_j exp(s_ij - 8) (saved for the backward)
#include "common.h"
#include <stdlib.h>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef short4v __attribute__((address_space(3))) * lds_s4p;

#ifdef EDM_AF_TIMELINE   // diagnostic build only (tools/af_timeline.py): per-wave timestamps of the kernels' phases
__device__ unsigned long long* g_af_timeline = nullptr;
#define AF_STAMP(slot)                                                                                       \
  if (g_af_timeline && (threadIdx.x & 63) == 0)                                                              \
    g_af_timeline[((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime()
#else
#define AF_STAMP(slot)
#endif

constexpr int D = 64, C = 256;
constexpr int RS = 2 * D + 16;          // padded image row (bytes): conflict-free ds_read_b128 / ds_write_b128 at a row per lane
constexpr int KC = 32;                  // channels per streamed weight chunk
constexpr int NCH = C / KC;             // 8 chunks per head
constexpr int WROWB = KC * 2;           // 64-byte LDS rows of a chunk
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
// the same product on top of an initial accumulator (row / column constants ride in for free: the softmax offset, -delta)
__device__ __forceinline__ f32x16 score_tile_init(const char* a_rows, const bf16x8 (&bq)[D / 16], f32x16 acc) {
#pragma unroll
  for (int s = 0; s < D / 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld128(a_rows + s * 32), bq[s], acc, 0, 0, 0);
  return acc;
}

// ---- streamed GEMM  acc[rb] (32 rows x 32 tokens) = Wrows[rb*32 .., 0..255] . X^T   for NRB row blocks
// The ROWS rows x 256 channels of the A operand go through an LDS ring in chunks of KCB channels by LDS-DMA.  KCB = 32:
// 64-byte rows, 16-byte pieces XOR-swizzled with (row >> 2) & 3 on the DMA source and on the fragment read
// (conv_igemm2.hip's layout); KCB = 64: 128-byte rows, swizzle row & 7.  A ring slot is SLOTB bytes (>= ROWS * 2 * KCB).
// (k is a loop counter of a fully unrolled loop: the chain folds to one s_waitcnt)
template <class SA, int KMAX>
__device__ __forceinline__ void leave_in_flight_k(int k, int wave);

template <int NT, int NRB, int KCB, int SLOTB_>
struct Streamed {
  static constexpr int NTW = NT;
  static constexpr int ROWS = NRB * 32;
  static constexpr int ROWB = KCB * 2;               // bytes of an LDS row
  static constexpr int RPS = 1024 / ROWB;            // rows per 1-KiB DMA instruction
  static constexpr int SLOTS = ROWS / RPS;           // DMA instructions per chunk
  static constexpr int SLOTB = SLOTB_;
  static constexpr int PER_WAVE = (SLOTS + NT - 1) / NT;
  static constexpr int PPR = ROWB / 16;              // 16-byte pieces per row
  static_assert(SLOTB_ >= ROWS * ROWB, "ring slot too small");

  static __device__ __forceinline__ int swz(int row) { return KCB == 32 ? ((row >> 2) & 3) : (row & 7); }

  // wave `wave` issues its share of chunk kc of the rows starting at `wrows` (row stride C elements) into ring slot `slot`
  static __device__ __forceinline__ void issue(const bf16* wrows, int kc, char* ring, int slot, int wave, int lane) {
    const int drow = lane / PPR, dp = lane % PPR;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int s = wave + NT * i;
      if (SLOTS % NT == 0 || s < SLOTS) {
        const int row = s * RPS + drow;
        const int c = dp ^ swz(row);
        dma16(wrows + (long)row * C + kc * KCB + c * 8, ring + slot * SLOTB + s * 1024);
      }
    }
  }
  // byte offset (inside a ring slot) of the fragment of k-step s (16 channels) for the lane's row of row block 0
  static __device__ __forceinline__ int frag_off(int l31, int lhi, int s) {
    return l31 * ROWB + (((2 * s + lhi) ^ swz(l31)) << 4);
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

constexpr int QKV_SLOTB = 192 * 64;                  // ring slot of the qkv projection (192 rows x 32 channels)

// s_waitcnt that leaves the wave's DMA instructions of KA chunks of stream SA (and KB chunks of stream SB) in flight
template <class SA, int KA, class SB = SA, int KB = 0>
__device__ __forceinline__ void leave_in_flight(int wave) {
  constexpr int NTW = SA::NTW;
  constexpr bool evenA = SA::SLOTS % NTW == 0 || KA == 0, evenB = SB::SLOTS % NTW == 0 || KB == 0;
  if constexpr (evenA && evenB) {
    wait_vmcnt<KA * (SA::SLOTS / NTW) + KB * (SB::SLOTS / NTW)>();
  } else {
    // uneven share (12 one-KiB pieces over 8 waves): waves below SLOTS % NT issue one more.  Only ONE uneven stream occurs.
    constexpr int cut = evenA ? SB::SLOTS % NTW : SA::SLOTS % NTW;
    constexpr int hi_ = KA * SA::PER_WAVE + KB * SB::PER_WAVE;
    constexpr int lo_ = KA * (evenA ? SA::PER_WAVE : SA::PER_WAVE - 1) + KB * (evenB ? SB::PER_WAVE : SB::PER_WAVE - 1);
    static_assert(hi_ < 64, "vmcnt immediate");
    if (wave < cut) wait_vmcnt<hi_>();
    else wait_vmcnt<lo_>();
  }
}

template <class SA, int KMAX>
__device__ __forceinline__ void leave_in_flight_k(int k, int wave) {
  if constexpr (KMAX == 0) {
    leave_in_flight<SA, 0>(wave);
  } else {
    if (k >= KMAX) leave_in_flight<SA, KMAX>(wave);
    else leave_in_flight_k<SA, KMAX - 1>(k, wave);
  }
}

template <class SA, class SB, int KA, int KBMAX>
__device__ __forceinline__ void leave_ab_b(int kb, int wave) {
  if constexpr (KBMAX == 0) {
    leave_in_flight<SA, KA, SB, 0>(wave);
  } else {
    if (kb >= KBMAX) leave_in_flight<SA, KA, SB, KBMAX>(wave);
    else leave_ab_b<SA, SB, KA, KBMAX - 1>(kb, wave);
  }
}
// leave ka chunks of stream SA and kb chunks of stream SB in flight (ka, kb: counters of fully unrolled loops)
template <class SA, class SB, int KAMAX, int KBMAX>
__device__ __forceinline__ void leave_ab(int ka, int kb, int wave) {
  if constexpr (KAMAX == 0) {
    leave_ab_b<SA, SB, 0, KBMAX>(kb, wave);
  } else {
    if (ka >= KAMAX) leave_ab_b<SA, SB, KAMAX, KBMAX>(kb, wave);
    else leave_ab<SA, SB, KAMAX - 1, KBMAX>(ka, kb, wave);
  }
}

// ---- round 6: the wave's 32 tokens x 256 channels as MFMA B fragments, fed ROW-CONTIGUOUSLY.  Loading a fragment straight
// from global memory (lane = token: 32 rows x two 16-byte pieces per instruction) is bound by the texture addresser, not by
// bytes -- 15.3 us for the backward's three operands where 5.5 us move the same bytes as whole rows
// (tools/frag_load, profiles/r05_frag_load.txt).  Here a HALF row (128 channels = 256 bytes) of each of the wave's tokens is
// brought into a wave-private 8 KB of LDS by LDS-DMA (1 KB per instruction = four whole half rows; the 16-byte pieces
// XOR-swizzled with the token index on the DMA source so that the fragment reads are conflict-free) and the eight fragments
// of that half are ds_read_b128s.  The staging bytes lie in regions the kernel writes only behind later workgroup barriers
// (the K / V images).  No workgroup barrier: a wave stages and reads its OWN tokens.
__device__ __forceinline__ void stage_issue(const bf16* rows /* token 0 of the sample */, const bf16* zeros, int N, int half,
                                            char* stg, int wave, int lane) {
  const int rr = lane >> 4, pp = lane & 15;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int r = 4 * s + rr;                       // token of the wave's block
    const int tok = wave * 32 + r;
    const int lp = pp ^ (r & 15);
    const bf16* src = (tok < N ? rows + (long)tok * C + half * 128 : zeros) + lp * 8;
    dma16(src, stg + s * 1024);
  }
}
__device__ __forceinline__ void stage_read(const char* stg, int l31, int lhi, bf16x8* frag /* 8 fragments of the half */) {
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) frag[kk] = ld128(stg + l31 * 256 + (((2 * kk + lhi) ^ (l31 & 15)) << 4));
}

// RING: slots of the weight ring.  A chunk needs 12 MFMAs per wave (0.2 us) and an L2 -> LDS round trip of > 1 us: with two
// chunks in flight the projection ran at 16 % of the MFMA rate (tools/af_timeline.py: 12 us for a head's 8 chunks); RING - 1
// chunks are kept in flight -- with 7 slots nearly the whole head's 96 KB of weights.
template <int NT, int RING, bool STAGE = true>
__global__ __launch_bounds__(NT * 64) void k_attn_qkv_fwd(const bf16* __restrict__ x, const bf16* __restrict__ Wqkv,
                                                            bf16* __restrict__ y, float* __restrict__ stat,
                                                            const bf16* __restrict__ zeros, int B, int N, int heads,
                                                            int HP) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT * 32;
  using G = Streamed<NT, 6, KC, QKV_SLOTB>;
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

  AF_STAMP(0);
  // ---- the wave's token block as B fragments: bx[kk] = x[tok][16 kk + 8 lhi .. +8]
  bf16x8 bx[C / 16];
  const int head0 = hg * HP;
  constexpr int AHEAD = RING - 1;                              // chunks in flight behind the one being consumed
  static_assert(AHEAD >= 2 && AHEAD < NCH, "ring depth");
  if constexpr (STAGE) {
    // row-contiguous feed (see stage_issue): half rows through the wave's 8 KB of the K | V image region, two rounds; the
    // weight ring's first chunks are issued behind the first round and land under the second
    static_assert(2 * NP * RS >= NT * 8192, "the staging area must fit the K | V image region");
    char* const stg = smem + wave * 8192;
    const bf16* const rows = x + (long)b * N * C;
    stage_issue(rows, zeros, N, 0, stg, wave, lane);
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) G::issue(Wqkv + (long)head0 * 3 * D * C, j, ring, j, wave, lane);
    leave_in_flight<G, AHEAD>(wave);                           // the eight staging pieces are older than the ring's
    stage_read(stg, l31, lhi, bx);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the fragments are in registers before the area is rewritten
    stage_issue(rows, zeros, N, 1, stg, wave, lane);
    wait_vmcnt<0>();
    stage_read(stg, l31, lhi, bx + 8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
    {
      const bf16* xr = (tvalid ? x + ((long)b * N + tok) * C : zeros) + lhi * 8;      // (padding tokens read the zero page)
#pragma unroll
      for (int kk = 0; kk < C / 16; ++kk) bx[kk] = *reinterpret_cast<const bf16x8*>(xr + kk * 16);
    }
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) G::issue(Wqkv + (long)head0 * 3 * D * C, j, ring, j, wave, lane);
  }

  const int a_off0 = G::frag_off(l31, lhi, 0), a_off1 = G::frag_off(l31, lhi, 1);
  const float sl2 = 0.125f * LOG2E, c0 = 8.0f * LOG2E;       // p = exp2(s * sl2 - c0) = exp(s / sqrt(d) - 8)
  const bool full = N == NP;
  const int tr_row = (lane & 15) >> 2;
  const int tr_col = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  int base = 0;                                               // ring slot of this head's chunk 0
  for (int hi = 0; hi < HP; ++hi) {
    const int head = head0 + hi;
    const bf16* wrows = Wqkv + (long)head * 3 * D * C;
    const bool next_head = hi + 1 < HP;
    f32x16 acc[6];
#pragma unroll
    for (int rb = 0; rb < 6; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
    // The chunk loop is software-pipelined by half a chunk (see k_attn_qkv_bwd): chunk kc + 1 becomes visible (wait + barrier)
    // in the MIDDLE of chunk kc's MFMAs and its first fragments are read under chunk kc's last MFMAs.
    if (hi > 0) wait_vmcnt<0>();                              // (the previous head's y stores sit behind the prefetched chunks)
    else leave_in_flight<G, AHEAD - 1>(wave);                 // chunk 0 of AHEAD issued
    __builtin_amdgcn_s_barrier();
    bf16x8 fa[6];
#pragma unroll
    for (int rb = 0; rb < 6; ++rb) fa[rb] = ld128(ring + base * G::SLOTB + rb * 32 * WROWB + a_off0);
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      int cur = base + kc % RING, nxt = base + (kc + 1) % RING, sl = base + (kc + AHEAD) % RING;
      cur -= cur >= RING ? RING : 0;
      nxt -= nxt >= RING ? RING : 0;
      sl -= sl >= RING ? RING : 0;
      const char* wt = ring + cur * G::SLOTB;
      const char* wn = ring + nxt * G::SLOTB;
#pragma unroll
      for (int rb = 0; rb < 6; ++rb) {               // the second k-step's read goes out right behind the MFMA of the first
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rb], bx[2 * kc], acc[rb], 0, 0, 0);
        fa[rb] = ld128(wt + rb * 32 * WROWB + a_off1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kc + 1 < NCH) {
        // retire chunk kc + 1: chunks up to kc + AHEAD - 1 have been issued (all of them exist if another head follows)
        if (next_head) leave_in_flight<G, AHEAD - 2>(wave);
        else leave_in_flight_k<G, AHEAD - 2>(NCH - 2 - kc, wave);
        __builtin_amdgcn_s_barrier();
        if (kc + AHEAD < NCH) G::issue(wrows, kc + AHEAD, ring, sl, wave, lane);
        else if (next_head) G::issue(wrows + (long)3 * D * C, kc + AHEAD - NCH, ring, sl, wave, lane);
      } else if (next_head) {
        // last chunk of the head: nothing to retire, but the slot of chunk kc - 1 is only free once EVERY wave has consumed
        // the fragments it read from it (i.e. has arrived here)
        __builtin_amdgcn_s_barrier();
        G::issue(wrows + (long)3 * D * C, kc + AHEAD - NCH, ring, sl, wave, lane);
      }
#pragma unroll
      for (int rb = 0; rb < 6; ++rb) {
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rb], bx[2 * kc + 1], acc[rb], 0, 0, 0);
        if (kc + 1 < NCH) fa[rb] = ld128(wn + rb * 32 * WROWB + a_off0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    base = (base + NCH) % RING;
    if (hi == 0) { AF_STAMP(1); }

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
    if (hi == 0) { AF_STAMP(2); }

    // ---- streamed attention for the wave's 32 queries
    float l = 0.f;
    f32x16 o[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    // software-pipelined by one key block: the QK^T MFMAs of block kt + 1 are in the matrix pipe while the vector ALU turns
    // block kt's scores into probabilities
    f32x16 Sn = score_tile(Kn + l31 * RS + lhi * 16, bq);
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 S = Sn;
      if (kt + 1 < NT) Sn = score_tile(Kn + ((kt + 1) * 32 + l31) * RS + lhi * 16, bq);
      __builtin_amdgcn_sched_barrier(0);
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
      __builtin_amdgcn_sched_barrier(0);
    }
    l += __shfl_xor(l, 32, 64);
    const float linv = 1.0f / l;
    if (hi == 0) { AF_STAMP(3); }
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
#ifdef EDM_AF_TIMELINE
  __builtin_amdgcn_s_waitcnt(0);
  AF_STAMP(4);
#endif
}

// EDM_ATTN_STAGE=0: the round-5 operand loads (fragments straight from global memory) in the BACKWARD kernel; A/B runs.
// Measured (tools/microbench_attnblock.py, one call, profiles/r06_attn_stage_ab.txt): backward 77.6 -> 71.5 us at 16x16 x 128,
// 20.8 -> 18.4 at 8x8, 282 -> 265 at batch 512; the training step 12.70 -> 12.66 ms.
static bool attn_stage() {       // (read per call: tests switch it inside one process)
  const char* e = getenv("EDM_ATTN_STAGE");
  return !(e && e[0] == '0');
}
// The FORWARD kernel's staged form is built and parity-tested (EDM_ATTN_STAGE_FWD=1) but off: its single operand costs two
// DMA round trips through the 8-KB staging area where the fragment-shaped loads cost about the same (37.5 vs 37.1 us at
// 16x16 x 128, 120.3 vs 121.3 at batch 512) -- the addresser-bound pattern only hurts once three operands queue behind it.
static bool attn_stage_fwd() {
  const char* e = getenv("EDM_ATTN_STAGE_FWD");
  return e && e[0] == '1';
}

template <int NT, int RING>
void launch_fwd(const void* x, const void* Wqkv, void* y, float* stat, int B, int N, int heads, int HP, hipStream_t st) {
  const size_t lds = (size_t)2 * NT * 32 * RS + (size_t)RING * QKV_SLOTB;
  static_assert((size_t)2 * NT * 32 * RS + (size_t)RING * QKV_SLOTB <= 160 * 1024, "LDS budget");
  const int groups = heads / HP;
  const int grid = ((B + 7) / 8) * 8 * groups;
  if (attn_stage_fwd()) {
    auto kern = k_attn_qkv_fwd<NT, RING, true>;
    EDM_MAX_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT * 64), lds, st, (const bf16*)x, (const bf16*)Wqkv, (bf16*)y, stat,
                       (const bf16*)edm_zero_page(), B, N, heads, HP);
  } else {
    auto kern = k_attn_qkv_fwd<NT, RING, false>;
    EDM_MAX_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT * 64), lds, st, (const bf16*)x, (const bf16*)Wqkv, (bf16*)y, stat,
                       (const bf16*)edm_zero_page(), B, N, heads, HP);
  }
}

// ====================================================================================================================
// backward: gqkv (packed order [head][q|k|v][d]) from x, y, gout -- the qkv tensor and dO = b * gout . W_out are rebuilt
// in registers / LDS.  Per head: (B) dO^T = alpha * Wd_out[64 h .. +64, :] . gout^T through the ring (64-channel chunks),
// delta = <dO, O>; (A) q, k, v as in the forward; images of Qn, Kn, Vn, dO (channel order) + the norm denominators in
// LDS; then attention.hip's two passes (query-major: dQ; key-major: dK, dV), with the forward's saved normaliser instead
// of a softmax sweep.  For 256 tokens the ring lies over the Q, K, V images' regions (written after the last chunk).
// ====================================================================================================================

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

// a 32-channel x 32-token accumulator block -> rows of a channel-ordered LDS image (lane = token; 8-byte pieces)
__device__ __forceinline__ void image_store(char* img_row, int rb, int lhi, const f32x16& a) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4 v4;
#pragma unroll
    for (int r = 0; r < 4; ++r) v4[r] = (bf16)a[4 * g + r];
    *reinterpret_cast<bf16x4*>(img_row + (32 * rb + 8 * g + 4 * lhi) * 2) = v4;
  }
}

// RING slots of 12 KB; OVERLAY: the ring occupies the Q | K | V image regions (all three are written after the last chunk
// has been consumed) instead of LDS of its own.  The 12 chunks of a head (4 of the out conv's dgrad rows, 64 channels each;
// 8 of the qkv rows, 32 channels each) are one stream: RING - 1 of them are kept in flight.
template <int NT, int RING, bool OVERLAY, bool STAGE = true>
__global__ __launch_bounds__(NT * 64) void k_attn_qkv_bwd(const bf16* __restrict__ x, const bf16* __restrict__ y,
                                                            const bf16* __restrict__ gout, const float* __restrict__ stat,
                                                            const bf16* __restrict__ Wqkv, const bf16* __restrict__ Wdo,
                                                            bf16* __restrict__ gqkv, const bf16* __restrict__ zeros,
                                                            float alpha, int B, int N, int heads, int HP) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT * 32;
  using GA = Streamed<NT, 6, KC, QKV_SLOTB>;       // qkv rows, 32-channel chunks
  using GB = Streamed<NT, 2, 64, QKV_SLOTB>;       // the head's 64 rows of the out conv's dgrad pack, 64-channel chunks
  constexpr int NCHB = C / 64, TOT = NCHB + NCH, AHEAD = RING - 1;
  static_assert(AHEAD >= 2 && AHEAD < TOT, "ring depth");
  static_assert(!OVERLAY || RING * QKV_SLOTB <= 3 * NP * RS, "the overlaid ring must fit the Q, K, V image regions");
  char* Qn = smem;
  char* Kn = Qn + NP * RS;
  char* Vn = Kn + NP * RS;
  char* dO = Vn + NP * RS;
  float* dsave = reinterpret_cast<float*>(dO + NP * RS);     // [3][NP]
  float* st_l = dsave + 3 * NP;                               // [NP] softmax normaliser (forward's)
  float* st_d = st_l + NP;                                    // [NP] delta = <dO, O>
  char* ring = OVERLAY ? smem : reinterpret_cast<char*>(st_d + NP);
  const int groups = heads / HP;
  const int id = blockIdx.x, xcd = id & 7, kk_ = id >> 3;
  const int b = (kk_ / groups) * 8 + xcd, hg = kk_ % groups;
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int tok = wave * 32 + l31;
  const bool tvalid = tok < N;
  const long trow = (long)b * N + tok;
  const float scale = 0.125f;
  const float sl2 = 0.125f * LOG2E, c0 = 8.0f * LOG2E;
  const bool full = N == NP;
  const int tr_row = (lane & 15) >> 2;
  const int tr_col = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int a_off0 = GA::frag_off(l31, lhi, 0), a_off1 = GA::frag_off(l31, lhi, 1);

  {
    // ONE head per workgroup (HP = 1): with a loop over heads hipcc hoists ~1 400 instructions of loop-invariant address
    // arithmetic in front of it and spills 43 registers of it
    const int head = hg;
    AF_STAMP(0);
    // ---- operands of both projections as B fragments (token on the lane), the head's slice of y for delta: all loads first
    bf16x8 bg[C / 16], bx[C / 16];
    bf16x4 yv[2][4];
    const bf16* wdo = Wdo + (long)head * D * C;
    const bf16* wrows = Wqkv + (long)head * 3 * D * C;
    // chunk j of the stream: j < NCHB -> rows of the out conv (64 channels), else chunk j - NCHB of the qkv rows; slot j % RING
    auto issue = [&](int j) {
      if (j < NCHB) GB::issue(wdo, j, ring, j % RING, wave, lane);
      else GA::issue(wrows, j - NCHB, ring, j % RING, wave, lane);
    };
    if constexpr (STAGE) {
      // row-contiguous feed of gout and x (stage_issue): a whole tensor's rows of the wave (two 8-KB halves) per round, in
      // the image regions (written only behind later barriers).  The ring has LDS of its own for <= 128 tokens and is
      // started first; for 256 tokens it lies over the Q | K | V regions -- over other waves' staging areas -- and starts
      // when every wave has its fragments in registers (one barrier; its first chunk's latency is exposed: ~1.5 us against
      // the ~9 us the fragment-shaped global loads cost).
      static_assert(4 * NP * RS >= NT * 16384, "the staging areas must fit the image regions");
      char* const stg = smem + wave * 16384;
      if constexpr (!OVERLAY) {
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) issue(j);
      }
      stage_issue(gout + (long)b * N * C, zeros, N, 0, stg, wave, lane);
      stage_issue(gout + (long)b * N * C, zeros, N, 1, stg + 8192, wave, lane);
      {
        const bf16* yr = (tvalid ? y + trow * C + head * D : zeros) + 4 * lhi;         // (padding tokens read the zero page)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) yv[dt][g] = *reinterpret_cast<const bf16x4*>(yr + dt * 32 + 8 * g);
      }
      wait_vmcnt<0>();
      stage_read(stg, l31, lhi, bg);
      stage_read(stg + 8192, l31, lhi, bg + 8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stage_issue(x + (long)b * N * C, zeros, N, 0, stg, wave, lane);
      stage_issue(x + (long)b * N * C, zeros, N, 1, stg + 8192, wave, lane);
      wait_vmcnt<0>();
      stage_read(stg, l31, lhi, bx);
      stage_read(stg + 8192, l31, lhi, bx + 8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (OVERLAY) {
        __builtin_amdgcn_s_barrier();      // the ring's slots cover the other waves' staging areas
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) issue(j);
      }
    } else {
      const bf16* gr = (tvalid ? gout + trow * C : zeros) + lhi * 8;                  // (padding tokens read the zero page)
      const bf16* yr = (tvalid ? y + trow * C + head * D : zeros) + 4 * lhi;
      const bf16* xr = (tvalid ? x + trow * C : zeros) + lhi * 8;
#pragma unroll
      for (int kk = 0; kk < C / 16; ++kk) bg[kk] = *reinterpret_cast<const bf16x8*>(gr + kk * 16);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) yv[dt][g] = *reinterpret_cast<const bf16x4*>(yr + dt * 32 + 8 * g);
#pragma unroll
      for (int kk = 0; kk < C / 16; ++kk) bx[kk] = *reinterpret_cast<const bf16x8*>(xr + kk * 16);
    }
    // The chunk loop is software-pipelined by half a chunk: the wait + barrier that make chunk j visible sit in the MIDDLE of
    // chunk j-1's MFMAs (which keep the matrix pipe busy while the wave waits), and chunk j's first fragments are read under
    // chunk j-1's last MFMAs.  (Per-chunk lockstep -- barrier, all waves read, all waves multiply -- ran both projections at
    // ~40 % of the LDS / MFMA bound: tools/af_timeline.py, 0.75 us per 12-MFMA chunk.)
    // retire(j): chunks 0 .. j + AHEAD - 2 have been issued when chunk j >= 1 is retired (0 .. AHEAD - 1 for j = 0)
    auto retire = [&](int j) {
      const int issued = (j == 0 ? AHEAD - 1 : j + AHEAD - 2);
      const int last = issued < TOT - 1 ? issued : TOT - 1;
      const int kb = (last < NCHB ? last : NCHB - 1) - j;               // out-conv chunks among j+1 .. last
      const int nb = kb > 0 ? kb : 0;
      leave_ab<GA, GB, NCH, NCHB - 1>(last - j - nb, nb, wave);
    };
    if constexpr (!STAGE) {
#pragma unroll
      for (int j = 0; j < AHEAD; ++j) issue(j);
    }

    // ---- phase B: dO^T (64 channels x 32 tokens) = alpha * Wd_out rows . gout^T
    f32x16 ad[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ad[rb][r] = 0.f;
    bf16x8 f[2], fa[6];
    retire(0);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) f[rb] = ld128(ring + rb * 32 * GB::ROWB + GB::frag_off(l31, lhi, 0));
#pragma unroll
    for (int t = 0; t < NCHB; ++t) {
      const char* wt = ring + (t % RING) * QKV_SLOTB;
      const char* wn = ring + ((t + 1) % RING) * QKV_SLOTB;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          ad[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[rb], bg[4 * t + s], ad[rb], 0, 0, 0);
          f[rb] = ld128(wt + rb * 32 * GB::ROWB + GB::frag_off(l31, lhi, s + 1));
        }
      __builtin_amdgcn_sched_barrier(0);
      retire(t + 1);
      __builtin_amdgcn_s_barrier();
      if (t + AHEAD < TOT) issue(t + AHEAD);            // (the slot of chunk t - 1: every wave has consumed it)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        ad[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[rb], bg[4 * t + 2], ad[rb], 0, 0, 0);
        f[rb] = ld128(wt + rb * 32 * GB::ROWB + GB::frag_off(l31, lhi, 3));
      }
      if (t + 1 < NCHB) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          ad[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[rb], bg[4 * t + 3], ad[rb], 0, 0, 0);
          f[rb] = ld128(wn + rb * 32 * GB::ROWB + GB::frag_off(l31, lhi, 0));
        }
      } else {
#pragma unroll
        for (int rb = 0; rb < 6; ++rb) fa[rb] = ld128(wn + rb * 32 * WROWB + a_off0);     // phase A's first fragments
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) ad[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[rb], bg[4 * t + 3], ad[rb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    AF_STAMP(1);
    // dO rounded like the unfused path's gy (a bf16 tensor) -> its LDS image (channel order); delta = <dO, O>
    float delta = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 v4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bf16 v = (bf16)(ad[dt][4 * g + r] * alpha);
          v4[r] = v;
          delta += (float)v * (float)yv[dt][g][r];
        }
        *reinterpret_cast<bf16x4*>(dO + tok * RS + (32 * dt + 8 * g + 4 * lhi) * 2) = v4;
      }
    delta += __shfl_xor(delta, 32, 64);
    // per-query constants of the two passes, in the form the score accumulators are initialised with:
    //   st_l = (log2(1 / l) - 8 log2 e) / (log2 e / sqrt(d))  ->  exp2((S + st_l) * sl2) = the normalised probability
    //   st_d = -delta                                          ->  (dP + st_d) = dP - delta
    // (a padding query has 1 / l = 0: st_l = -inf, its probabilities are exp2(-inf) = 0)
    const float linv = tvalid ? stat[((long)b * heads + head) * N + tok] : 0.f;
    const float s0 = (__builtin_amdgcn_logf(linv) - c0) * (1.0f / sl2);
    if (lhi == 0) {
      st_d[tok] = -delta;
      st_l[tok] = s0;
    }

    // ---- phase A: q, k, v of the head (as the forward)
    f32x16 acc[6];
#pragma unroll
    for (int rb = 0; rb < 6; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      constexpr int T0 = NCHB;
      const char* wt = ring + ((T0 + kc) % RING) * QKV_SLOTB;
      const char* wn = ring + ((T0 + kc + 1) % RING) * QKV_SLOTB;
#pragma unroll
      for (int rb = 0; rb < 6; ++rb) {               // the second k-step's read goes out right behind the MFMA of the first
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rb], bx[2 * kc], acc[rb], 0, 0, 0);
        fa[rb] = ld128(wt + rb * 32 * WROWB + a_off1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kc + 1 < NCH) {
        retire(T0 + kc + 1);
        __builtin_amdgcn_s_barrier();
        if (T0 + kc + AHEAD < TOT) issue(T0 + kc + AHEAD);
      }
#pragma unroll
      for (int rb = 0; rb < 6; ++rb) {
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rb], bx[2 * kc + 1], acc[rb], 0, 0, 0);
        if (kc + 1 < NCH) fa[rb] = ld128(wn + rb * 32 * WROWB + a_off0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    AF_STAMP(2);
    // every wave must be past its last ring read before the Q image overwrites the ring
    __syncthreads();
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
      const float dn = NORM_EPS + sqrtf(ss) * 0.125f;
      const float inv = 1.0f / dn;
      if (lhi == 0) dsave[w * NP + tok] = dn;
      char* img = (w == 0 ? Qn : w == 1 ? Kn : Vn) + tok * RS;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[2 * w + rb][r] *= inv;
        image_store(img, rb, lhi, acc[2 * w + rb]);
      }
    }
    __syncthreads();

    AF_STAMP(3);
    // ================= pass 1: query-major -> dQ =================
    {
      const int qi = tok;
      bf16x8 bq[D / 16], bdo[D / 16];
#pragma unroll
      for (int s = 0; s < D / 16; ++s) {
        bq[s] = ld128(Qn + qi * RS + s * 32 + lhi * 16);
        bdo[s] = ld128(dO + qi * RS + s * 32 + lhi * 16);
      }
      // dS = P * scale * (dP - delta): the query's constants start the accumulators (log2(scale) / sl2 = -3 / sl2 folds
      // the 1 / sqrt(d) of the chain rule into the exponent), three vector instructions per score are left: mul, exp2, mul
      f32x16 s_init, d_init;
      {
        const float si = s0 - 3.0f / sl2, di = -delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s_init[r] = si;
          d_init[r] = di;
        }
      }
      f32x16 accq[D / 32];
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accq[dt][r] = 0.f;
      // software-pipelined by one key block: the score / dP MFMAs of block kt + 1 run while the vector ALU works on block kt
      f32x16 Sn = score_tile_init(Kn + l31 * RS + lhi * 16, bq, s_init);
      f32x16 dPn = score_tile_init(Vn + l31 * RS + lhi * 16, bdo, d_init);   // dP^T tile: keys x queries
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        f32x16 S = Sn, dP = dPn;
        if (kt + 1 < NT) {
          Sn = score_tile_init(Kn + ((kt + 1) * 32 + l31) * RS + lhi * 16, bq, s_init);
          dPn = score_tile_init(Vn + ((kt + 1) * 32 + l31) * RS + lhi * 16, bdo, d_init);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = __builtin_amdgcn_exp2f(S[r] * sl2);
          if (!full) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            p = key < N ? p : 0.f;
          }
          dP[r] = p * dP[r];
        }
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
        __builtin_amdgcn_sched_barrier(0);
      }
      norm_bwd_store(accq, Qn + qi * RS, dsave[0 * NP + qi], gqkv + ((long)b * N + qi) * 3 * C + head * 3 * D, lhi, tvalid);
    }
    AF_STAMP(4);
    // ================= pass 2: key-major -> dK, dV =================
    {
      const int ki = tok;
      bf16x8 bk[D / 16], bv[D / 16];
#pragma unroll
      for (int s = 0; s < D / 16; ++s) {
        bk[s] = ld128(Kn + ki * RS + s * 32 + lhi * 16);
        bv[s] = ld128(Vn + ki * RS + s * 32 + lhi * 16);
      }
      f32x16 acck[D / 32], accv[D / 32];
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acck[dt][r] = 0.f;
          accv[dt][r] = 0.f;
        }
      // the queries' constants are the initial accumulators of a tile (rows = queries in the registers, columns = keys on
      // the lanes); software-pipelined by one query block like pass 1
      auto tiles = [&](int qt, f32x16& S, f32x16& dP) {
        f32x16 s_init, d_init;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int q0 = qt * 32 + 8 * g + 4 * lhi;
          const f32x4 ll = *reinterpret_cast<const f32x4*>(st_l + q0);
          const f32x4 dd = *reinterpret_cast<const f32x4*>(st_d + q0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s_init[4 * g + r] = ll[r];
            d_init[4 * g + r] = dd[r];
          }
        }
        S = score_tile_init(Qn + (qt * 32 + l31) * RS + lhi * 16, bk, s_init);
        dP = score_tile_init(dO + (qt * 32 + l31) * RS + lhi * 16, bv, d_init);
      };
      f32x16 Sn, dPn;
      tiles(0, Sn, dPn);
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        f32x16 S = Sn, dP = dPn;
        if (qt + 1 < NT) tiles(qt + 1, Sn, dPn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = __builtin_amdgcn_exp2f(S[r] * sl2);            // the normalised probability
          if (!full) p = tvalid ? p : 0.f;
          S[r] = p;
          dP[r] = p * dP[r];                                       // (the 1 / sqrt(d) of dS is applied to dK at the end)
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
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acck[dt][r] *= scale;
      bf16* dst = gqkv + ((long)b * N + ki) * 3 * C + head * 3 * D;
      norm_bwd_store(acck, Kn + ki * RS, dsave[1 * NP + ki], dst + D, lhi, tvalid);
      AF_STAMP(5);
      norm_bwd_store(accv, Vn + ki * RS, dsave[2 * NP + ki], dst + 2 * D, lhi, tvalid);
    }
  }
#ifdef EDM_AF_TIMELINE
  __builtin_amdgcn_s_waitcnt(0);
  AF_STAMP(6);
#endif
}

template <int NT, int RING, bool OVERLAY>
void launch_bwd(const void* x, const void* y, const void* gout, const void* stat, const void* Wqkv, const void* Wdo, void* gqkv,
                float alpha, int B, int N, int heads, int HP, hipStream_t st) {
  constexpr size_t lds = (size_t)4 * NT * 32 * RS + (size_t)5 * NT * 32 * sizeof(float) + (OVERLAY ? 0 : (size_t)RING * QKV_SLOTB);
  static_assert(lds <= 160 * 1024, "LDS budget");
  const int groups = heads / HP;
  const int grid = ((B + 7) / 8) * 8 * groups;
  if (attn_stage()) {
    auto kern = k_attn_qkv_bwd<NT, RING, OVERLAY, true>;
    EDM_MAX_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT * 64), lds, st, (const bf16*)x, (const bf16*)y, (const bf16*)gout,
                       (const float*)stat, (const bf16*)Wqkv, (const bf16*)Wdo, (bf16*)gqkv, (const bf16*)edm_zero_page(), alpha,
                       B, N, heads, HP);
  } else {
    auto kern = k_attn_qkv_bwd<NT, RING, OVERLAY, false>;
    EDM_MAX_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT * 64), lds, st, (const bf16*)x, (const bf16*)y, (const bf16*)gout,
                       (const float*)stat, (const bf16*)Wqkv, (const bf16*)Wdo, (bf16*)gqkv, (const bf16*)edm_zero_page(), alpha,
                       B, N, heads, HP);
  }
}

}  // namespace

#ifdef EDM_AF_TIMELINE
extern "C" int edm_af_set_timeline(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_af_timeline), &buf, sizeof(buf)) == hipSuccess ? EDM_OK : EDM_ERR_LAUNCH;
}
#endif

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
  // heads per workgroup: two = one round of 256 workgroups at batch 128 (16x16: 40.9 / 34.6 / 55.5 us for 1 / 2 / 4 heads);
  // at sampler batches (>= 512 samples) all four: x is loaded once per sample and the grid still fills the chip twice
  // (B = 512: 130.9 -> 121.1 us)
  int HP = hp > 0 ? hp : (nt > 4 ? (B >= 512 ? 4 : 2) : 1);
  EDM_REQUIRE(heads % HP == 0, "attention_qkv_fwd: heads per workgroup must divide heads");
  // (64-token maps: 66 KB of LDS per 128-thread workgroup, two per CU; larger maps: the seven-slot ring, one per CU)
  if (nt <= 2) launch_fwd<2, 4>(x, Wqkv, y, (float*)stat, B, N, heads, HP, st);
  else if (nt <= 4) launch_fwd<4, 7>(x, Wqkv, y, (float*)stat, B, N, heads, HP, st);
  else launch_fwd<8, 7>(x, Wqkv, y, (float*)stat, B, N, heads, HP, st);
  EDM_CHECK_LAUNCH("attention_qkv_fwd");
  return EDM_OK;
}

// gqkv [B*N, 3C] (packed order) = d loss / d qkv_conv(x), given gout = d loss / d (out_conv output) [B*N, C]:
// dO = alpha * gout . W_out (Wd_out = the out conv's dgrad pack [C(ci), C(co)]) is formed inside (networks.py:203-205 backward)
extern "C" int edm_attention_qkv_bwd(const void* x, const void* y, const void* gout, const void* stat, const void* Wqkv,
                                     const void* Wd_out, void* gqkv, float alpha, int B, int N, int C, int heads, int hp,
                                     hipStream_t st) {
  EDM_REQUIRE(x && y && gout && stat && Wqkv && Wd_out && gqkv, "attention_qkv_bwd: null pointer");
  EDM_REQUIRE(B > 0 && edm_attention_qkv_supported(N, C, heads), "attention_qkv_bwd: C = 256, 4 heads, 33..256 tokens only "
              "(got C %d, heads %d, N %d)", C, heads, N);
  EDM_ZERO_PAGE(zero_page_, "attention_qkv_bwd");
  (void)zero_page_;
  const int nt = (N + 31) / 32;
  const int HP = 1;           // (one head per workgroup: see the kernel)
  EDM_REQUIRE(hp == 0 || hp == 1, "attention_qkv_bwd: the backward runs one head per workgroup (heads_per_wg 0 or 1, got %d)", hp);
  // 256-token maps: nine ring slots laid over the Q | K | V image regions (152 KB in all); smaller maps: a ring of its own
  if (nt <= 2) launch_bwd<2, 3, false>(x, y, gout, stat, Wqkv, Wd_out, gqkv, alpha, B, N, heads, HP, st);
  else if (nt <= 4) launch_bwd<4, 6, false>(x, y, gout, stat, Wqkv, Wd_out, gqkv, alpha, B, N, heads, HP, st);
  else launch_bwd<8, 9, true>(x, y, gout, stat, Wqkv, Wd_out, gqkv, alpha, B, N, heads, HP, st);
  EDM_CHECK_LAUNCH("attention_qkv_bwd");
  return EDM_OK;
}

// 3x3 implicit-GEMM convolution, the static-schedule kernel of conv_igemm4.hip on v_mfma_f32_16x16x32_bf16.
//
//   Y[p, co] = alpha * sum_{tap, ci} X[p + off(tap), ci] * Wp[tap, co, ci]  (+ beta * R[p, co])     (networks.py:37)
//
// Why a second MFMA shape.  The device is power-limited in MFMA-dense code: a bare loop that re-reads every operand
// from LDS sustains 1.52 PFLOP/s at 1.61 GHz on v_mfma_f32_32x32x16_bf16 and 1.66 PFLOP/s at 1.80 GHz on
// v_mfma_f32_16x16x32_bf16 -- equal cycles per FLOP, a higher held clock (tools/mfma_shape, MI355X_MICROARCH.md "DVFS
// give-back" item 7).  Geometry, LDS images, DMA plan, ring depths and counted vmcnt waits are those of
// k_conv3x3_v4 (512 pixels x 128 or 64 channels per workgroup, 8 waves x (all channels x 64 pixels), 64-byte LDS rows
// with the 16-byte pieces XOR-swizzled by (row >> 2) & 3); what changes is everything fragment-shaped:
//  * a fragment is 16 rows x 32 k: lane -> row (lane & 15), piece (lane >> 4).  A step (tap, 32-channel chunk) is ONE
//    k-step: 2 NI weight fragments + 4 pixel fragments (the same 12 ds_read_b128 as two 32x32x16 k-steps) feeding
//    8 NI MFMAs of 16 cycles; accumulators are 8 NI blocks of 4 registers.
//  * pixel-fragment addresses: the swizzle term of row q + 16 j + shift(tap) does not depend on j, so the four pixel
//    blocks of a wave are ONE per-lane address per tap plus an immediate -- 9 address registers instead of the 36 the
//    folded-mask scheme of v4 would need here.  Image borders: a 16-pixel block lies in one image row (W % 16 == 0), so
//    the 32 (non-centre tap, block) validity bits of a lane's pixels live in ONE register; a fragment read expands its
//    bit to a word mask (v_bfe_i32) and selects the tap address or a zero row with it (v_bfi_b32) -- two vector
//    instructions per fragment (four zero rows 16 apart, so the block immediate still applies).  (Scalar lane masks in
//    SGPR pairs + v_cndmask were tried first: hoisted they need 64 SGPRs, re-derived at each use 4-6 scalar
//    instructions per fragment, 10 % of the 64-channel kernel.)
//  * fragment registers as v4 (48): weight fragments double-buffered by channel half, pixel fragments single-buffered
//    with rolling reuse -- an MFMA has read its operands 8 cycles after issue, so block j of the NEXT step is loaded
//    right behind block j's last MFMA of this step.
#include "common.h"
#include <stdlib.h>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int BM = 512, KC = 32, TAPS = 9;
constexpr int ROWB = KC * 2;      // 64-byte LDS rows
constexpr int WRING = 6, D = WRING - 1;
constexpr int ZERO_PAGE = 4096;

__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
struct IC { static constexpr int value = N; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(IC<B>{});
    static_for<B + 1, E>(f);
  }
}
#define LDS_RD128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
template <int N>
__device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

#ifdef EDM_V6_TIMELINE   // diagnostic build only (tools/v6_timeline.py): per-(workgroup, tile) timestamps of the kernel's phases
__device__ unsigned long long* g_v6_timeline = nullptr;
#define V6_ROW ((long)blockIdx.x + (long)tl_iter * gridDim.x)
#define V6_STAMP(slot)                                                                                          \
  if (g_v6_timeline && threadIdx.x == 0) g_v6_timeline[V6_ROW * 8 + (slot)] = __builtin_amdgcn_s_memrealtime()
// the SHADER clock (s_memtime) at the same point: (d memtime) / (d memrealtime) x 100 MHz is the clock the chip holds there
#define V6_CLK(slot)                                                                                            \
  if (g_v6_timeline && threadIdx.x == 0) g_v6_timeline[V6_ROW * 8 + (slot)] = __builtin_amdgcn_s_memtime()
#else
#define V6_STAMP(slot)
#define V6_CLK(slot)
#endif

template <int NX, int EPI = 0, int NI = 4, bool WPTR64 = true, int WB = 0, bool LATE_DMA = false, bool FOLD = false,
          bool PERSIST = false>
__global__ __launch_bounds__(512, 2) void k_conv3x3_v6(const bf16* __restrict__ X, const bf16* __restrict__ Wp,
                                                         bf16* __restrict__ Y, const bf16* __restrict__ R,
                                                         const char* __restrict__ zeros, float alpha, float beta,
                                                         int Npix, int H, int W, int Cin, int Cout, int tiles_m,
                                                         int tiles_n, ModEpilogue mod) {
  apply_dyn(mod);
#ifdef EDM_V6_TIMELINE
  int tl_iter = 0;
#endif
  constexpr int NJ = 2;               // 32-pixel blocks per wave (epilogue units)
  constexpr int NA = 2 * NI, NAH = NI;  // 16-channel weight fragments per step / per half
  constexpr int NB = 4;               // 16-pixel fragments per wave
  constexpr int XROWS = NX * 8 * 16;  // LDS rows per slab buffer; rows >= xrows are zero rows
  constexpr int BMW = 512;
  constexpr int XBYTES = XROWS * ROWB;
  constexpr int BNW = 32 * NI;
  constexpr int WTILE = BNW * ROWB;
  constexpr int ZROW = XROWS - 1 - 48;  // zero rows ZROW + 16 j (host-checked: ZROW >= xrows)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xb = smem;
  char* const Wb = smem + 2 * XBYTES;
  // modulation backward: the per-(sample, channel) sums of the eight waves meet in LDS (one row of BNW floats per wave,
  // behind the slab and ring buffers) and leave as one global atomic per sample and channel of the tile
  float* const gmred = reinterpret_cast<float*>(smem + 2 * XBYTES + WRING * WTILE);

  static_assert(!PERSIST || (!FOLD && !LATE_DMA && NX == 5), "persistent form: 5-slot slab, no folded phase");
  const int id = blockIdx.x;
  const int xcd = id & 7, k = id >> 3;
  const int tn = k % tiles_n;
  int tm = (k / tiles_n) * 8 + xcd;
  if (tm >= tiles_m) return;
  // PERSIST (round 6): a workgroup walks the pixel tiles tm, tm + tm_step, ... of ONE channel tile (host: gridDim.x / 8 is a
  // multiple of tiles_n, so the XCD and tn of a workgroup's tiles do not change).  The weight ring simply keeps turning
  // across the tile boundary (tile (chunk 0, tap t) of the next pixel tile IS the ring's next tile) and the next tile's
  // first slab is issued at tap 0 of this tile's last chunk, so both land under the epilogue: every tile but a
  // workgroup's first starts with its prologue already done (tools/v6_timeline.py: 3.1 us of a 57-us tile).
  const int tm_step = PERSIST ? ((int)(gridDim.x >> 3) / tiles_n) * 8 : 0;
  int m0 = tm * BMW;
  const int n0 = tn * BNW;
  const int HALO = W + 1;
  const int xrows = BMW + 2 * HALO;

  int tid_v = threadIdx.x;
  bool first_tile = true;
  for (;;) {   // the pixel tiles of this workgroup (exactly one unless PERSIST)
  // PERSIST: every per-lane address below is re-derived per tile from an opaque copy of the thread index, so that none of
  // them (~45 registers) stays live across the epilogue (kept live, the 128-channel forms spill 28-123 registers)
  if constexpr (PERSIST) asm volatile("" : "+v"(tid_v));
  V6_STAMP(0);
#ifdef EDM_V6_TIMELINE
  if (g_v6_timeline && threadIdx.x == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_v6_timeline[V6_ROW * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  const bool next_tile = PERSIST ? (tm + tm_step < tiles_m) : false;
  const int tid = tid_v, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = pixel octant of this wave
  const int l15 = lane & 15, lq = lane >> 4;
  const int drow = lane >> 2, dp = lane & 3;  // DMA lane -> (row in 16-row slot, physical 16-B piece)
  const bool late = LATE_DMA && wave >= 4;    // see the DMA issue in the step body

  // ---- DMA sources (as k_conv3x3_v4: per lane, computed once, advancing by constants)
  constexpr int WROWS = NI == 4 ? 16 : 8;
  const bool w_lane = NI == 4 || lane < 32;
  // weight tile (chunk, tap): ONE 32-bit per-lane byte offset; chunk and tap enter through the uniform base, so the DMA
  // takes the SGPR-base + VGPR-offset form and no per-tap 64-bit pointer is kept in vector registers (host-checked:
  // the packed weights are < 4 GiB)
  unsigned woff;
  {
    const int row = wave * WROWS + (drow & (WROWS - 1));
    const int co = min(n0 + row, Cout - 1);
    const int c = dp ^ ((row >> 2) & 3);
    woff = (unsigned)(((long)co * Cin + c * 8) * 2);
  }
  const char* const wbase = reinterpret_cast<const char*>(Wp);
  const char* const wsrc = wbase + woff;      // NI == 2 (registers to spare): plain per-lane pointer, as k_conv3x3_v4
  const long tap_stride = (long)Cout * Cin * 2;
  const char* xsrc[NX];
  auto set_xsrc = [&](int m0v, const char* (&xs)[NX]) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int row = (wave + 8 * i) * 16 + drow;
      const int c = dp ^ ((row >> 2) & 3);
      bool real = row < xrows;
      if constexpr (WB >= 2) {
        // images start and end on tile boundaries (H*W % 512 == 0): the rows above the first / below the last image row
        // are fed as zeros, which is all the vertical border handling this form needs
        const int hw = H * W;
        if (m0v % hw == 0 && row < HALO) real = false;
        if ((m0v + BMW) % hw == 0 && row >= HALO + BMW) real = false;
      }
      if (real) {
        long pix = (long)m0v - HALO + row;
        pix = pix < 0 ? 0 : (pix >= Npix ? Npix - 1 : pix);  // out-of-range rows only feed masked taps
        xs[i] = reinterpret_cast<const char*>(X + pix * (mod.ldX ? mod.ldX : Cin) + c * 8);
      } else {
        xs[i] = zeros + c * 16;
      }
    }
  };
  set_xsrc(m0, xsrc);

  // ---- pixel-fragment addresses: ONE per tap (block 0 of this wave; block j = + j * 16 rows = + 1024 bytes, the
  // swizzle term of the row is unchanged by 16 j), the zero row, and the border masks
  const unsigned xb_off = (unsigned)(uintptr_t)(lds_char*)Xb;
  unsigned bp[TAPS];
  {
    const int q0 = wave * 64 + l15 + HALO;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int r = q0 + (t / 3 - 1) * W + (t % 3 - 1);
      bp[t] = xb_off + r * ROWB + ((lq ^ ((r >> 2) & 3)) << 4);
    }
  }
  const unsigned bz = xb_off + ZROW * ROWB + (lq << 4);
  // WB != 0 (W = 16 WB, images aligned to tiles): block j starts an image row when j % WB == 0 and ends one when
  // j % WB == WB - 1 -- compile-time facts -- so the left / right border is two more address sets in which the
  // border lane already points at the zero row: no instruction per fragment at all
  unsigned bpS[3], bpE[3];
  if constexpr (WB != 0) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      bpS[dy] = l15 == 0 ? bz : bp[3 * dy];
      bpE[dy] = l15 == 15 ? bz : bp[3 * dy + 2];
    }
  }
  // WB == 1 (16x16 images, two per tile, a wave = four whole image rows): block 0 of waves 0 and 4 is an image's first
  // row, block 3 of waves 3 and 7 its last -- wave-uniform, so the taps that would leave the image get their own
  // address registers (all lanes at the zero row in exactly those waves)
  unsigned bpT[3], bpB[3];
  if constexpr (WB == 1) {
    const bool top = (wave & 3) == 0, bot = (wave & 3) == 3;
    bpT[0] = top ? bz : bpS[0]; bpT[1] = top ? bz : bp[1]; bpT[2] = top ? bz : bpE[0];
    bpB[0] = bot ? bz : bpS[2]; bpB[1] = bot ? bz : bp[7]; bpB[2] = bot ? bz : bpE[2];
  }
  // border validity of this lane's pixel in each of the four blocks, one bit per (non-centre tap, block): bit 4 t8 + j
  // with t8 = tap (tap < 4) or tap - 1.  A fragment read turns its bit into an all-ones / all-zeros word (v_bfe_i32) and
  // picks the tap address or the zero row with it (v_bfi_b32): two vector instructions, no scalar work, no VCC.
  unsigned vbits = 0;
  auto set_vbits = [&](int m0v) {
    vbits = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int m = m0v + wave * 64 + 16 * j + l15;
      const int w = m % W, h = (m / W) % H;
#pragma unroll
      for (int t8 = 0; t8 < 8; ++t8) {
        const int t = t8 < 4 ? t8 : t8 + 1;
        const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) vbits |= 1u << (4 * t8 + j);
      }
    }
  };
  if constexpr (WB == 0) set_vbits(m0);
  // weight-fragment rows 16 i + l15: the swizzle term depends on l15 only
  const unsigned wb_off = (unsigned)(uintptr_t)(lds_char*)Wb;
  const unsigned ap = wb_off + l15 * ROWB + ((lq ^ ((l15 >> 2) & 3)) << 4);

  f32x4 acc[NA][NB];

  const int nchunks = Cin / KC;  // even (host-checked)
  // split-bf16 evaluation (mod.kwrap != 0): K chunk c reads the X channels of chunk (c < kwrap ? c : c - kwrap) -- X rows
  // are [hi | lo] pairs, the pack [w_hi | w_lo | w_hi] (common.h)
  const int kwrap = mod.kwrap ? mod.kwrap : (1 << 30);

  // ---- prologue: slab 0, then weight tiles 0..D-1 (issue order fixes the counted waits).  PERSIST: a workgroup's later
  // tiles find both issued by the tile before them
  if (!PERSIST || first_tile) {
#pragma unroll
    for (int i = 0; i < NX; ++i) dma16(xsrc[i], Xb + (wave + 8 * i) * 1024);
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (w_lane) dma16(wbase + d * tap_stride + woff, Wb + d * WTILE + wave * (WROWS * ROWB));
  }

  u32x4 fa[2][NAH], fb[NB];
  // (macros, not nested lambdas: clang rejects implicit captures of the kernel's locals from a generic lambda nested in
  // the generic step lambda below)
  // weight fragments of channel half `half` (blocks NAH*half ..) of ring slot `slot`
#define RD_A(half, slot, kk) \
  if constexpr (NAH > kk) LDS_RD128(fa[half][kk], ap, (slot) * WTILE + ((half) * NAH + kk) * 16 * ROWB);
#define READ_A(half, slot) { RD_A(half, slot, 0) RD_A(half, slot, 1) RD_A(half, slot, 2) RD_A(half, slot, 3) }
  // pixel fragment of block j at tap `tap` (constant expressions) in slab buffer `cpar`
#define READ_B(j, tap, cpar)                                                                  \
  {                                                                                           \
    unsigned addr_ = bp[tap];                                                                 \
    if constexpr (WB != 0) {                                                                  \
      if constexpr ((tap) % 3 == 0 && (j) % WB == 0) addr_ = bpS[(tap) / 3];                  \
      if constexpr ((tap) % 3 == 2 && (j) % WB == WB - 1) addr_ = bpE[(tap) / 3];             \
      if constexpr (WB == 1 && (tap) / 3 == 0 && (j) == 0) addr_ = bpT[(tap) % 3];            \
      if constexpr (WB == 1 && (tap) / 3 == 2 && (j) == 3) addr_ = bpB[(tap) % 3];            \
    } else if constexpr ((tap) != 4) {                                                        \
      constexpr int bit_ = 4 * ((tap) < 4 ? (tap) : (tap) - 1) + (j);                          \
      unsigned msk_;                                                                          \
      asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(msk_) : "v"(vbits), "n"(bit_));           \
      asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(addr_) : "v"(msk_), "v"(bp[tap]), "v"(bz)); \
    }                                                                                         \
    LDS_RD128(fb[j], addr_, (cpar) * XBYTES + (j) * 16 * ROWB);                               \
  }
  // the NAH MFMAs of channel half h on pixel block j.  Inline asm with the accumulator tied ("+v"): with the builtin the
  // register allocator lets the 32 accumulator blocks wander (vdst != srcC) and ends up 40-75 registers over budget.
  // The asm is invisible to hipcc's hazard recogniser: an accumulator block is touched once per 32 MFMAs (no
  // back-to-back dependency), and the epilogue pads the MFMA-write -> VALU-read distance by hand.
#define MFMA1(h, j, kk)                                                                                          \
  if constexpr (NAH > kk) {                                                                                      \
    if constexpr (NI == 4)                                                                                       \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(h) * NAH + kk][j])                      \
                   : "v"(fa[h][kk]), "v"(fb[j]));                                                                \
    else /* 64-channel tile: the builtin fits (202 registers) and measures 10 % faster than the tied asm form */ \
      acc[(h) * NAH + kk][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                          \
          __builtin_bit_cast(bf16x8, fa[h][kk]), __builtin_bit_cast(bf16x8, fb[j]), acc[(h) * NAH + kk][j], 0, 0, 0); \
  }
#define MFMA_BLOCK(h, j)                                              \
  {                                                                   \
    MFMA1(h, j, 0) MFMA1(h, j, 1) MFMA1(h, j, 2) MFMA1(h, j, 3)       \
    __builtin_amdgcn_sched_barrier(0);                                \
  }

#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  for (int chunk2 = 0; chunk2 < nchunks; chunk2 += 2) {
    static_for<0, 2 * TAPS>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int tap = u % TAPS, cpar = u / TAPS;
      const int chunk = chunk2 + cpar;
      const bool more_chunks = chunk + 1 < nchunks;
      const bool more = more_chunks || next_tile;   // PERSIST: the next tile's DMAs take the place of the next chunk's
      // ---- retire weight tiles t and t+1 (and slab chunk+1 before its first read, issued at tap 8): as v4
      if (more) {
        if (tap >= 1 && tap <= D - 1) wait_vmcnt<D - 2 + NX>();
        else wait_vmcnt<D - 2>();
      } else {
        switch ((TAPS - 2 - tap) < (D - 2) ? (TAPS - 2 - tap) : (D - 2)) {  // last chunk: the ring drains
          case 3: wait_vmcnt<3>(); break;
          case 2: wait_vmcnt<2>(); break;
          case 1: wait_vmcnt<1>(); break;
          default: wait_vmcnt<0>(); break;
        }
      }
      __builtin_amdgcn_s_barrier();
#ifdef EDM_V6_TIMELINE
      if (u == 0 && chunk2 == 0) { V6_STAMP(1); V6_CLK(6); }
#endif
      // ---- this step's DMAs: weight tile t+D into the ring slot read at step t-1, and (tap 0) the next slab.  An LDS-DMA
      // instruction holds the issuing wave for 60-180 cycles and the two waves of a SIMD leave the barrier together.
      // LATE_DMA (waves 4-7 issue in the middle of the step instead, so that one partner multiplies while the other
      // issues; per-wave issue ORDER unchanged, so the counted vmcnt waits hold) was measured in round 2: 0.5 % SLOWER in
      // the training step (14.78 vs 14.68 ms), mixed in isolation -- off.
      auto issue_dma = [&]() {
        constexpr int tq = tap + D;
        constexpr int cq = cpar + (tq >= TAPS ? 1 : 0);
        constexpr int tapq = tq >= TAPS ? tq - TAPS : tq;
        int cidx = chunk2 + cq;
        bool wok = cidx < nchunks;
        if constexpr (PERSIST) {
          if (!wok && next_tile) { cidx = 0; wok = true; }   // (chunk2 + cq == nchunks: chunk 0 of the next tile)
        }
        if constexpr (NI == 2 && WPTR64) {
          if (wok && w_lane)
            dma16(wsrc + (long)cidx * (KC * 2) + tapq * tap_stride,
                  Wb + ((u + D) % WRING) * WTILE + wave * (WROWS * ROWB));
        } else if (wok && w_lane) {
          const char* ub = wbase + ((long)cidx * (KC * 2) + tapq * tap_stride);
          asm volatile("" : "+s"(ub));   // keep (chunk, tap) in the scalar base: SGPR-base + VGPR-offset DMA, no per-tap
                                         // 64-bit vector pointers hoisted out of the loop
          dma16(ub + woff, Wb + ((u + D) % WRING) * WTILE + wave * (WROWS * ROWB));
        }
        if (tap == 0 && more_chunks) {
          const int xc_next = chunk + 1 >= kwrap ? chunk + 1 - kwrap : chunk + 1;
#pragma unroll
          for (int i = 0; i < NX; ++i)
            dma16(xsrc[i] + (long)xc_next * (KC * 2), Xb + (cpar ^ 1) * XBYTES + (wave + 8 * i) * 1024);
        }
        if constexpr (PERSIST && tap == 0 && cpar == 1) {
          // last chunk of this tile (nchunks is even, so it sits in slab buffer 1): the next tile's chunk 0 goes to buffer 0
          // (read last in chunk nchunks - 2: the barrier above is behind it)
          if (!more_chunks && next_tile) {
            const char* xn[NX];
            set_xsrc((tm + tm_step) * BMW, xn);
#pragma unroll
            for (int i = 0; i < NX; ++i) dma16(xn[i], Xb + (wave + 8 * i) * 1024);
          }
        }
      };
      if (!late) issue_dma();
      // ---- 8 NI MFMAs.  LDS reads in issue order: A1(u) | A0(u+1) | B(u+1, 0..3); at the first step A0(0), B(0, *) first.
      if (u == 0 && chunk2 == 0) {
        READ_A(0, 0);
        READ_B(0, 0, 0); READ_B(1, 0, 0); READ_B(2, 0, 0); READ_B(3, 0, 0);
      }
      READ_A(1, u % WRING);
      __builtin_amdgcn_sched_barrier(0);
      // channel half 0 on blocks 0..3: block j needs A0(u) and B(u, j); younger: B(u, j+1..3) and A1(u)
      lgkm_wait<3 + NAH>(); MFMA_BLOCK(0, 0);
      lgkm_wait<2 + NAH>(); MFMA_BLOCK(0, 1);
      lgkm_wait<1 + NAH>(); MFMA_BLOCK(0, 2);
      lgkm_wait<0 + NAH>(); MFMA_BLOCK(0, 3);
      if (late) issue_dma();
      __builtin_amdgcn_sched_barrier(0);
      constexpr int tapn = (tap + 1) % TAPS, cparn = (u + 1) / TAPS % 2, slotn = (u + 1) % WRING;
      if constexpr (NI == 2 && WPTR64) {
        // 64-channel tile (registers to spare): the last step skips its prefetch
        const bool has_next = more_chunks || tap + 1 < TAPS;
        if (has_next) {
          READ_A(0, slotn);
          lgkm_wait<NAH>();
        } else {
          lgkm_wait<0>();
        }
        MFMA_BLOCK(1, 0); if (has_next) READ_B(0, tapn, cparn); __builtin_amdgcn_sched_barrier(0);
        MFMA_BLOCK(1, 1); if (has_next) READ_B(1, tapn, cparn); __builtin_amdgcn_sched_barrier(0);
        MFMA_BLOCK(1, 2); if (has_next) READ_B(2, tapn, cparn); __builtin_amdgcn_sched_barrier(0);
        MFMA_BLOCK(1, 3); if (has_next) READ_B(3, tapn, cparn); __builtin_amdgcn_sched_barrier(0);
      } else {
        // The next step's fragments are prefetched unconditionally: behind the very last step they come from ring slot /
        // slab positions that hold stale but in-bounds data and are never multiplied (no branch in the step body: each
        // one costs registers here).
        READ_A(0, slotn);
        lgkm_wait<NAH>();
        // channel half 1; behind block j's MFMAs its register takes the NEXT step's block j
        MFMA_BLOCK(1, 0); READ_B(0, tapn, cparn); __builtin_amdgcn_sched_barrier(0);
        MFMA_BLOCK(1, 1); READ_B(1, tapn, cparn); __builtin_amdgcn_sched_barrier(0);
        MFMA_BLOCK(1, 2); READ_B(2, tapn, cparn); __builtin_amdgcn_sched_barrier(0);
        MFMA_BLOCK(1, 3); READ_B(3, tapn, cparn); __builtin_amdgcn_sched_barrier(0);
      }
    });
  }

  // ---- FOLD (round 6): the decoder block's 512 -> 256 skip projection as a second reduction on the same accumulators
  // (common.h, ModEpilogue::X2).  A chunk of it is ONE step (centre tap only): a fresh 512-row slab (32 KB, no halo) and a
  // weight tile per 8 NI MFMAs -- nine times the staging traffic per MFMA of the 3x3 part -- so this phase runs at what the
  // L2 -> LDS path delivers, not at the matrix pipe's rate (wait + barrier per chunk, three chunks in flight).  It still beats
  // the separate launch it replaces (68 us at 32x32 x 128): no second kernel boundary, no [pixels][Cout] round trip through HBM.
  if constexpr (FOLD) {
    lgkm_wait<0>();                // the last step's (unused) prefetch
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMA results -> first read by ordinary code
    __builtin_amdgcn_s_barrier();  // every wave is done with the slab / weight ring of the nine taps
    {
      const float fs = mod.fold_scale;
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] *= fs;
    }
    const int n2 = mod.C2 / KC;                     // even (host-checked)
    const int kw2 = mod.kwrap2 ? mod.kwrap2 : (1 << 30);
    const char* x2src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave + 8 * i) * 16 + drow;
      const int c = dp ^ ((row >> 2) & 3);
      long pix = (long)m0 + row;
      pix = pix >= Npix ? Npix - 1 : pix;           // rows behind the last pixel feed outputs that are never stored
      x2src[i] = reinterpret_cast<const char*>(mod.X2 + pix * mod.ldX2 + c * 8);
    }
    unsigned woff2;
    {
      const int row = wave * WROWS + (drow & (WROWS - 1));
      const int co = min(n0 + row, Cout - 1);
      const int c = dp ^ ((row >> 2) & 3);
      woff2 = (unsigned)(((long)co * mod.C2 + c * 8) * 2);
    }
    const char* const w2base = reinterpret_cast<const char*>(mod.W2);
    // phase-2 LDS layout: four slots of [512-row slab (32 KB) | weight tile] over the slab + ring regions (the launch asks for
    // the larger of the two phases' LDS), THREE chunks in flight behind the one being multiplied (with one, every chunk waited
    // a whole L2 -> LDS round trip and the phase took as long as the launch it replaces; two: 1.4 us per chunk)
    constexpr int SLAB2 = BMW * ROWB, SLOT2 = SLAB2 + WTILE, NS2 = 4;   // (four slots = all 160 KB of a CU's LDS at NI = 4)
    static_assert(NS2 * SLOT2 <= 160 * 1024, "phase-2 slots must fit LDS (launch6 asks for the larger of the two phases)");
    const int q2 = wave * 64 + l15;
    const unsigned bp2 = xb_off + q2 * ROWB + ((lq ^ ((q2 >> 2) & 3)) << 4);
    const unsigned ap2 = xb_off + SLAB2 + l15 * ROWB + ((lq ^ ((l15 >> 2) & 3)) << 4);
    auto issue2 = [&](int c) {
      const int xc = c >= kw2 ? c - kw2 : c;
      char* const slot = smem + (c % NS2) * SLOT2;
#pragma unroll
      for (int i = 0; i < 4; ++i) dma16(x2src[i] + (long)xc * (KC * 2), slot + (wave + 8 * i) * 1024);
      if (w_lane) dma16(w2base + (long)c * (KC * 2) + woff2, slot + SLAB2 + wave * (WROWS * ROWB));
    };
    issue2(0);
    if (n2 > 1) issue2(1);
    if (n2 > 2) issue2(2);
#pragma unroll 1
    for (int c = 0; c < n2; ++c) {
      // this wave's five pieces of chunk c (chunks c + 1 and c + 2 may still be in flight) ...
      if (c + 2 < n2) wait_vmcnt<10>();
      else if (c + 1 < n2) wait_vmcnt<5>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();      // ... and everybody's; every wave has read its fragments of chunk c - 1
      if (c + 3 < n2) issue2(c + 3);     // into the slot chunk c - 1 was read from
      const unsigned so = (unsigned)((c % NS2) * SLOT2);
      const unsigned aa = ap2 + so, bb = bp2 + so;
      u32x4 fa2[NA], fb2[NB];
#define RD2A(i) if constexpr (NA > (i)) LDS_RD128(fa2[i], aa, (i) * 16 * ROWB);
#define RD2B(j) LDS_RD128(fb2[j], bb, (j) * 16 * ROWB);
      RD2A(0) RD2A(1) RD2A(2) RD2A(3) RD2A(4) RD2A(5) RD2A(6) RD2A(7)
      RD2B(0) RD2B(1) RD2B(2) RD2B(3)
#undef RD2A
#undef RD2B
      static_assert(NA <= 8 && NB == 4, "fragment reads above are written out");
      lgkm_wait<0>();
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          if constexpr (NI == 4)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fa2[i]), "v"(fb2[j]));
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa2[i]),
                                                                 __builtin_bit_cast(bf16x8, fb2[j]), acc[i][j], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: transposed through wave-private LDS (common.h: store_tile_transposed16)
  lgkm_wait<0>();                // the last step's (unused) prefetch
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMA results -> first read by ordinary code
  __builtin_amdgcn_s_barrier();  // every wave is done with the slab / weight ring
  V6_STAMP(2);
  V6_CLK(7);
  // wave-private staging of the epilogue: over the slab buffers -- or (PERSIST: buffer 0 and ring slots 0..D-1 are being
  // filled for the next tile) over slab buffer 1 for the first waves and ring slot D + the LDS behind the ring for the rest
  constexpr int STG = EPI == 4 ? 16 * (NI * 128 + 16) : 32 * (NI * 64 + 16);
  constexpr int NFIT = XBYTES / STG < 8 ? XBYTES / STG : 8;
  char* stage = smem + wave * STG;
  float* gmr = gmred;
  if constexpr (PERSIST) {
    static_assert(2 * XBYTES + D * WTILE + (8 - NFIT) * STG + (EPI == 1 ? 8 * BNW * 4 : 0) <= 160 * 1024,
                  "persistent form: epilogue staging must fit beside the next tile's prefetch");
    stage = wave < NFIT ? Xb + XBYTES + wave * STG : Wb + D * WTILE + (wave - NFIT) * STG;
    gmr = reinterpret_cast<float*>(Wb + D * WTILE + (8 - NFIT) * STG);
  }
  if constexpr (EPI == 1) {
    const bool wave_rows = mod.HW % (32 * NJ) == 0;   // a wave's 64 pixels lie in one sample
    store_tile_transposed16<NI, NJ, EPI>(acc, stage, Y, R, alpha, beta,
                                         (long)m0 + wave * (32 * NJ), Npix, n0, Cout, mod,
                                         wave_rows ? gmr + wave * BNW : nullptr);
    if (wave_rows) {
      __syncthreads();
      if (tid < BNW && n0 + tid < Cout) {
        const long gstride = mod.gm_stride ? mod.gm_stride : (long)Cout;
        int cur = m0 / mod.HW;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {                   // fixed order within the tile
          const int pw = m0 + w * (32 * NJ);
          if (pw >= Npix) break;
          const int sm = pw / mod.HW;
          if (sm != cur) {
            atomicAdd(mod.gm + cur * gstride + n0 + tid, sum);
            sum = 0.f;
            cur = sm;
          }
          sum += gmr[w * BNW + tid];
        }
        atomicAdd(mod.gm + cur * gstride + n0 + tid, sum);
      }
    }
  } else if constexpr (EPI == 4) {
    // (round 6: row-contiguous through wave-private LDS.  tools/microbench_split_epi.py, 32x32 x 512: the two pairs buffers of
    // the copy-free concat 1 774 -> 1 522 us, fp32 + pairs 1 578 -> 1 494, a single fp32 stream 1 411 -> 1 446 in isolation --
    // but in the solve staging EVERY form is the fastest policy (A/B/C in one call: 206.6 / 207.6 / 209.8 img/s for never /
    // only multi-stream tiles / always; the modulation factors are read once per lane instead of once per accumulator block).
    // mod.wfrag carries the A/B switch EDM_F32_EPI_STAGED=0, set by the dispatcher)
    if (mod.wfrag)
      store_tile_f32_16<NI, NJ>(acc, reinterpret_cast<float*>(Y), reinterpret_cast<const float*>(R), alpha, beta,
                                (long)m0 + wave * (32 * NJ), Npix, n0, Cout, mod);
    else
      store_tile_f32_16_staged<NI, NJ>(acc, stage, reinterpret_cast<float*>(Y),
                                       reinterpret_cast<const float*>(R), alpha, beta, (long)m0 + wave * (32 * NJ), Npix, n0,
                                       Cout, mod);
  } else {
    store_tile_transposed16<NI, NJ, EPI>(acc, stage, Y, R, alpha, beta,
                                         (long)m0 + wave * (32 * NJ), Npix, n0, Cout, mod);
  }
  V6_STAMP(3);
#ifdef EDM_V6_TIMELINE
  if (next_tile) { V6_STAMP(4); }   // (no store drain between a workgroup's tiles)
  else {
    __builtin_amdgcn_s_waitcnt(0);  // stores retired
    V6_STAMP(4);
  }
  ++tl_iter;
#endif
  if (!next_tile) break;
  tm += tm_step;
  m0 = tm * BMW;
  first_tile = false;
  }   // tiles of this workgroup
#undef READ_A
#undef RD_A
#undef READ_B
#undef MFMA_BLOCK
#undef MFMA1
}

long g_persistent_launches = 0;   // (include/tinyedm_hip_diag.h: edm_v6_persistent_launches)

// Workgroups of a persistent launch: one per CU, rounded down to whole (XCD, channel tile) sets.
int persist_grid(int tiles_n) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      n = 256;
    return n < 8 ? 8 : n;
  }();
  const int per_xcd = cus / 8;
  return 8 * tiles_n * (per_xcd / tiles_n > 0 ? per_xcd / tiles_n : 1);
}

template <int NX, int EPI, int NI, int WB = 0, bool FOLD = false, bool PERSIST = false>
void launch6(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int Npix, int H, int W,
             int Cin, int Cout, const ModEpilogue& mod, hipStream_t st) {
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + 32 * NI - 1) / (32 * NI);
  size_t lds = (size_t)2 * NX * 8 * 16 * ROWB + WRING * (32 * NI * ROWB) + (EPI == 1 ? 8 * 32 * NI * 4 : 0);
  if (FOLD && lds < (size_t)4 * (BM * ROWB + 32 * NI * ROWB)) lds = (size_t)4 * (BM * ROWB + 32 * NI * ROWB);   // phase 2: four slots
  int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  if (PERSIST) {
    // epilogue staging beside the next tile's prefetch (see the kernel): waves that do not fit over slab buffer 1 stage
    // behind ring slot D
    constexpr int XBYTES = NX * 8 * 16 * ROWB, WTILE = 32 * NI * ROWB;
    constexpr int STG = EPI == 4 ? 16 * (NI * 128 + 16) : 32 * (NI * 64 + 16);
    constexpr int NFIT = XBYTES / STG < 8 ? XBYTES / STG : 8;
    const size_t need = (size_t)2 * XBYTES + D * WTILE + (8 - NFIT) * STG + (EPI == 1 ? 8 * 32 * NI * 4 : 0);
    if (lds < need) lds = need;
    grid = persist_grid(tiles_n);
    __atomic_add_fetch(&g_persistent_launches, 1L, __ATOMIC_RELAXED);
  }
  auto kern = k_conv3x3_v6<NX, EPI, NI, true, WB, false, FOLD, PERSIST>;
  EDM_MAX_LDS(kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const bf16*)X, (const bf16*)Wp, (bf16*)Y, (const bf16*)R,
                     (const char*)edm_zero_page(), alpha, beta, Npix, H, W, Cin, Cout, tiles_m, tiles_n, mod);
}

}  // namespace

// Shapes this kernel covers: those of edm_conv_igemm_v4_ex with 49 spare zero rows in the slab buffer (any W <= 64: the
// static border forms need W = 32 / 64 with images aligned to the tiles, or 16x16 images; everything else takes the
// per-lane validity bits).  Returns EDM_ERR_UNSUPPORTED (-3) otherwise.
int edm_conv_igemm_v6_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                         int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st) {
  EDM_REQUIRE(X && Wp && (Y || mod.Y2), "conv_igemm_v6: null pointer");
  EDM_REQUIRE(!mod.wfrag || mod.mode == 4, "conv_igemm_v6: fragment-major weight packs are read by k_conv3x3_s only");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "conv_igemm_v6: bad B/H/W");
  EDM_REQUIRE(Cout > 0 && Cout % 8 == 0, "conv_igemm_v6: Cout %% 8 required");
  if (taps != 9 || Cin <= 0 || Cin % 64 != 0 || Cin * 2 + 64 > ZERO_PAGE || W > 64) return EDM_ERR_UNSUPPORTED;
  EDM_ZERO_PAGE(zero_page_, "conv_igemm_v6");
  (void)zero_page_;
  const int Npix = B * H * W;
  const int xrows = BM + 2 * (W + 1);
  const bool nx5 = xrows <= 5 * 128 - 49;
  if (!nx5 && xrows > 6 * 128 - 49) return EDM_ERR_UNSUPPORTED;
  const long tiles4 = (long)((Npix + BM - 1) / BM) * ((Cout + 127) / 128);
  // 128-channel tiles when they fill the chip -- unless they would mostly multiply padding: Cout = 192 (the 64x64 layers of
  // the default ImageNet Denoiser) is 1.5 tiles of 128 but 3 of 64, and the 64-channel form's lower matrix-pipe occupancy
  // (69 % against 80 %, profiles/r05_v6_timeline_16x16.txt) costs less than a quarter of the MFMAs spent on zeros (round 6;
  // EDM_V6_NARROW=0: the round-5 rule)
  static const bool narrow_ok = !(getenv("EDM_V6_NARROW") && getenv("EDM_V6_NARROW")[0] == '0');
  const int pad4 = (Cout + 127) / 128 * 128, pad2 = (Cout + 63) / 64 * 64;
  const bool wide = tiles4 >= 512 && !(narrow_ok && pad2 * 80 < pad4 * 69);
  // images aligned to the 512-pixel tiles (W = 32 / 64), or 16x16 images (two per tile): border handling without
  // per-fragment instructions
  const int wb = ((H * W) % BM == 0 && (W == 32 || W == 64)) ? W / 16 : (H == 16 && W == 16) ? 1 : 0;
  // (W = 64 needs the 6-slot slab, W = 16 / 32 fit the 5-slot one: only the combinations that can occur are built)
#define L6A(EPIV, NIV)                                                                                     \
  (wb == 1   ? launch6<5, EPIV, NIV, 1>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st)          \
   : wb == 2 ? launch6<5, EPIV, NIV, 2>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st)          \
             : launch6<5, EPIV, NIV, 0>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st))
#define L6B(EPIV, NIV)                                                                                     \
  (wb == 4 ? launch6<6, EPIV, NIV, 4>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st)            \
           : launch6<6, EPIV, NIV, 0>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st))
#define L6(EPIV) (nx5 ? (wide ? L6A(EPIV, 4) : L6A(EPIV, 2)) : (wide ? L6B(EPIV, 4) : L6B(EPIV, 2)))
  if (mod.X2) {
    // folded skip projection: the 5-slot slab forms only (W <= 32: every shipped decoder level), plain and split epilogues
    if (!nx5 || (mod.mode != 0 && mod.mode != 3 && mod.mode != 4) || R) return EDM_ERR_UNSUPPORTED;
    // (split-bf16 form: K wraps over the [hi | lo] halves of X2, so the reduction is longer than a row)
    EDM_REQUIRE(mod.W2 && mod.C2 > 0 && mod.C2 % 64 == 0 && mod.ldX2 % 8 == 0 &&
                    (mod.kwrap2 ? mod.ldX2 >= 64 * mod.kwrap2 : mod.ldX2 >= mod.C2),
                "conv_igemm_v6: folded projection needs C2 %% 64 == 0 and rows that hold the reduction (C2 %d, ldX2 %d)", mod.C2, mod.ldX2);
#define L6F(EPIV, NIV)                                                                                           \
  (wb == 1   ? launch6<5, EPIV, NIV, 1, true>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st)          \
   : wb == 2 ? launch6<5, EPIV, NIV, 2, true>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st)          \
             : launch6<5, EPIV, NIV, 0, true>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st))
    if (mod.mode == 4) { if (wide) L6F(4, 4); else L6F(4, 2); }
    else { if (wide) L6F(0, 4); else L6F(0, 2); }
#undef L6F
    EDM_CHECK_LAUNCH("conv_igemm_v6 (folded projection)");
    return EDM_OK;
  }
  // persistent form (round 6), OPT-IN (EDM_V6_PERSIST=1, read per call: A/B in one process): 128-channel tiles of the
  // tile-aligned 16x16 / 32x32 layers when a CU gets two or more of them (B = 128: the 32x32 layers; the samplers' B = 512:
  // 16x16 too), plain / forward-modulation / split-bf16 epilogues.  Inside a launch it does what it was built for (every tile
  // after a workgroup's first starts 2.3 us earlier, 499 -> 475 us for the 2048 tiles of a B = 512 layer,
  // profiles/r06_v6_persistent.txt) -- and in a sustained stream of such launches (the samplers) it changes nothing
  // (587.8 vs 590.5 img/s bf16, 213.9 vs 215.4 split): the device is power-limited there, and the idle microseconds the
  // form removes are what let the clock recover.  The modulation-backward / mp_silu-backward epilogues are not built in
  // this form: their epilogues need the registers the walk keeps live (22 / 42 spilled).
  if (nx5 && wide && (wb == 1 || wb == 2) && mod.mode != 1 && mod.mode != 2) {
    const int tiles_n4 = (Cout + 127) / 128;
    const char* e = getenv("EDM_V6_PERSIST");
    if (e && e[0] == '1' && (long)((Npix + BM - 1) / BM + 7) / 8 * 8 * tiles_n4 >= 2L * persist_grid(tiles_n4)) {
#define L6P(EPIV)                                                                                                  \
  (wb == 1 ? launch6<5, EPIV, 4, 1, false, true>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st)         \
           : launch6<5, EPIV, 4, 2, false, true>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st))
      if (mod.mode == 4) L6P(4);
      else L6P(0);
#undef L6P
      EDM_CHECK_LAUNCH("conv_igemm_v6 (persistent)");
      return EDM_OK;
    }
  }
  if (mod.mode == 1) L6(1);
  else if (mod.mode == 2) L6(2);
  else if (mod.mode == 4) L6(4);
  else L6(0);
#undef L6
#undef L6A
#undef L6B
  EDM_CHECK_LAUNCH("conv_igemm_v6");
  return EDM_OK;
}

extern "C" long edm_v6_persistent_launches(void) { return __atomic_load_n(&g_persistent_launches, __ATOMIC_RELAXED); }

#ifdef EDM_V6_TIMELINE
extern "C" int edm_v6_set_timeline(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_v6_timeline), &buf, sizeof(buf)) == hipSuccess ? EDM_OK : EDM_ERR_LAUNCH;
}
#endif

extern "C" int edm_conv_igemm_v6(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B,
                                 int H, int W, int Cin, int Cout, int taps, hipStream_t st) {
  EDM_REQUIRE(Y, "conv_igemm_v6: null pointer");
  return edm_conv_igemm_v6_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, ModEpilogue{}, st);
}

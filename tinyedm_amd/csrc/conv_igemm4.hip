// 3x3 implicit-GEMM convolution, fourth generation ("static schedule"): the tall-tile LDS-DMA kernel of
// conv_igemm3.hip with every LDS address made a compile-time offset of a per-lane register.
//
//   Y[p, co] = alpha * sum_{tap, ci} X[p + off(tap), ci] * Wp[tap, co, ci]  (+ beta * R[p, co])
//
// Why: generations 2/3 spend ~4 vector-ALU instructions per MFMA on swizzle/tap address arithmetic, border
// masks and 64-bit DMA source addresses; with 2-4 waves per SIMD the shared vector issue port, not the matrix
// pipe, is what saturates (measured: fragment reads + MFMA alone reach only ~51 % of the MFMA rate).  Here:
//  * the (tap, 2 ci-chunks) loop is unrolled 18x, so weight-ring slot, slab buffer and tap are constants:
//    every ds_read_b128 is `base register + immediate`;
//  * the 36 pixel-fragment addresses (9 taps x 2 pixel blocks x 2 k-steps) are computed ONCE per workgroup;
//    border masks are folded into them: a masked lane simply reads a zero row of the slab buffer;
//  * DMA source pointers advance by constants (one 64-bit add per weight tile); out-of-range rows are clamped
//    to valid memory (their products are masked) and zero rows walk inside a 4 KiB zero page -- no selects.
//  * 6-deep weight ring (5 tiles in flight), counted vmcnt immediates, raw s_barrier.
// Geometry as generation 3: 512 pixels x 128 channels per workgroup, 8 waves x (128 co x 64 px), Cin % 64 == 0.
#include "common.h"
#include <stdlib.h>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 512, BN = 128, KC = 32, TAPS = 9;
constexpr int ROWB = KC * 2;      // 64-byte LDS rows
// NI = 32-channel blocks per wave: 4 -> 512x128 workgroup tile (BN above); 2 -> 512x64, for layers too small to give
// every CU a 512x128 tile (the 16x16 layers at batch 128: 64 pixel tiles x 2 channel tiles = 128 workgroups).
constexpr int WRING = 6, D = WRING - 1;

constexpr int ZERO_PAGE = 4096;

__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ bf16x8 lds128(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
typedef __attribute__((address_space(3))) char lds_char;
// read 16 B at an LDS byte offset held as an integer (keeps the access a ds_read_b128 after integer arithmetic)
__device__ __forceinline__ bf16x8 lds128_at(unsigned off) {
  return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((lds_char*)(uintptr_t)off);
}

// compile-time loop: f(IC<B>{}) ... f(IC<E-1>{}); the index is a constant expression inside f (inline-asm immediates)
template <int N>
struct IC { static constexpr int value = N; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(IC<B>{});
    static_for<B + 1, E>(f);
  }
}
// Fragment reads are issued from inline asm and retired by hand-counted lgkmcnt: left to itself hipcc answers a
// "6 older + 6 younger reads in flight" state with s_waitcnt lgkmcnt(0), which serialises the two halves again.
#define LDS_RD128_(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define LDS_RD128(dst, addr, imm)                                   \
  do {                                                              \
    if constexpr (ABL & 16) asm volatile("" : "=v"(dst) : "v"(addr)); \
    else LDS_RD128_(dst, addr, imm);                                \
  } while (0)
#define LGKM_WAIT(n)                                       \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");  \
  __builtin_amdgcn_sched_barrier(0)
template <int N>
__device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ABL: timing-only ablation builds for tools/ (outputs wrong by construction): bit0 no MFMA, bit1 no DMA issue in the
// loop, bit2 no barrier, bit4 no fragment reads.  ABL = 0 is the product.
// NJ = 32-pixel blocks per wave: 2 -> 512-pixel tile, 254 registers, one workgroup per CU (the product).  NJ = 1 (256x64
// tile, <= 128 registers, TWO workgroups per CU so that one's prologue / epilogue / reads run under the other's MFMAs)
// was built and measured in round 2: correct, but 20 % SLOWER on the 32x32 layers (177 vs 147 us) and equal on the
// 16x16 ones -- the smaller tile doubles the weight staging and the barriers per MFMA; it is not dispatched.
template <int NX, int EPI = 0, int NI = 4, int ABL = 0, int NJ = 2>
__global__ __launch_bounds__(512, NJ == 1 ? 4 : 2) void k_conv3x3_v4(const bf16* __restrict__ X, const bf16* __restrict__ Wp,
                                                         bf16* __restrict__ Y, const bf16* __restrict__ R,
                                                         const char* __restrict__ zeros, float alpha, float beta,
                                                         int Npix, int H, int W, int Cin, int Cout, int tiles_m,
                                                         int tiles_n, int stagger, ModEpilogue mod) {
  apply_dyn(mod);
  constexpr int XROWS = NX * 8 * 16;  // LDS rows per slab buffer; rows >= xrows are zero rows
  constexpr int BMW = 8 * 32 * NJ;    // pixels per workgroup
  constexpr int XBYTES = XROWS * ROWB;
  constexpr int BNW = 32 * NI;        // output channels per workgroup
  constexpr int WTILE = BNW * ROWB;   // 8 KiB (NI = 4) or 4 KiB weight tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xb = smem;
  char* const Wb = smem + 2 * XBYTES;

  const int id = blockIdx.x;
  const int xcd = id & 7, k = id >> 3;
  const int tn = k % tiles_n, tm = (k / tiles_n) * 8 + xcd;
  if (tm >= tiles_m) return;
  const int m0 = tm * BMW, n0 = tn * BNW;
  const int HALO = W + 1;
  const int xrows = BMW + 2 * HALO;  // < XROWS (host-checked): row XROWS-1 is always a zero row

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = pixel octant of this wave
  const bool late = stagger && wave >= 4;                       // second-dispatched half of the workgroup (see below)
  const int l31 = lane & 31, lhi = lane >> 5;
  const int drow = lane >> 2, dp = lane & 3;  // DMA lane -> (row in 16-row slot, physical 16-B chunk)

  // ---- DMA sources (per lane, computed once; they then advance by constants)
  // weight tile (chunk, tap): rows n0 + 16*wave + drow (clamped: rows >= Cout feed discarded outputs)
  // NI = 2: the tile has 64 rows, every wave still issues ONE instruction per tile (uniform vmcnt) with its lower
  // 32 lanes = 8 rows
  constexpr int WROWS = NI == 4 ? 16 : 8;          // weight-tile rows per wave instruction
  const bool w_lane = NI == 4 || lane < 32;
  const char* wsrc;
  {
    const int row = wave * WROWS + (drow & (WROWS - 1));
    const int co = min(n0 + row, Cout - 1);
    const int c = dp ^ ((row >> 2) & 3);
    wsrc = reinterpret_cast<const char*>(Wp + (long)co * Cin + c * 8);
  }
  const long tap_stride = (long)Cout * Cin * 2;  // bytes between taps of the packed weights
  // slab slot i: rows (wave + 8i)*16 + drow; chunk advances the pointer by 64 B
  const char* xsrc[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int row = (wave + 8 * i) * 16 + drow;
    const int c = dp ^ ((row >> 2) & 3);
    if (row < xrows) {
      long pix = (long)m0 - HALO + row;
      pix = pix < 0 ? 0 : (pix >= Npix ? Npix - 1 : pix);  // out-of-range rows only feed masked taps
      xsrc[i] = reinterpret_cast<const char*>(X + pix * Cin + c * 8);
    } else {
      xsrc[i] = zeros + c * 16;  // zero rows: the pointer walks inside the 4 KiB zero page
    }
  }

  // ---- the 36 pixel-fragment LDS addresses (tap, pixel block, k-step), border masks folded in
  const unsigned xb_off = (unsigned)(uintptr_t)(lds_char*)Xb;
  unsigned bp[TAPS][NJ];  // LDS byte offsets for k-step 0; k-step 1 = same offset with bit 5 flipped (chunk ^ 2)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int ml = wave * (32 * NJ) + j * 32 + l31;
    const int m = m0 + ml;
    const int w = m % W, h = (m / W) % H;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
      const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
      const int r = ml + HALO + (t / 3 - 1) * W + (t % 3 - 1);
      const int sw = (r >> 2) & 3;
      bp[t][j] = xb_off + (ok ? r * ROWB + ((lhi ^ sw) << 4) : (XROWS - 1) * ROWB);
    }
  }
  // weight-fragment rows i*32 + l31: swizzle term depends on l31 only
  const int a_sw = (l31 >> 2) & 3;
  const unsigned wb_off = (unsigned)(uintptr_t)(lds_char*)Wb;
  const unsigned ap[2] = {wb_off + l31 * ROWB + (((0 + lhi) ^ a_sw) << 4), wb_off + l31 * ROWB + (((2 + lhi) ^ a_sw) << 4)};

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = Cin / KC;  // even (host-checked)

  // ---- prologue: slab 0, then weight tiles 0..D-1 (issue order fixes the counted waits)
#pragma unroll
  for (int i = 0; i < NX; ++i) dma16(xsrc[i], Xb + (wave + 8 * i) * 1024);
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (w_lane) dma16(wsrc + d * tap_stride, Wb + d * WTILE + wave * (WROWS * ROWB));

  // Fragment pipeline: the 16 MFMAs of a step run as two halves (k-step 0 / k-step 1, 8 MFMAs each) and the 6 reads
  // of the NEXT half are in flight while the current half computes, so the LDS pipe and the matrix pipe overlap
  // inside one wave instead of alternating (all 8 waves of the group are barrier-aligned, so they would otherwise
  // all read, then all compute).  The barrier at the top of step u therefore retires tile u+1 as well as tile u.
  u32x4 fa[2][NI], fb[2][NJ];
  auto mfma_half = [&](int set) {
    if constexpr (ABL & 1) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][i]),
                                                           __builtin_bit_cast(bf16x8, fb[set][j]), acc[i][j], 0, 0, 0);
    }
  };
#define READ_HALF(set, ks, tap, cpar, slot)                                               \
  {                                                                                       \
    unsigned o0 = bp[tap][0], o1 = bp[tap][NJ - 1];                                       \
    if (ks) { /* opaque xor: keep 18, not 36, address registers */                        \
      asm volatile("v_xor_b32 %0, 32, %1" : "=v"(o0) : "v"(bp[tap][0]));                  \
      if constexpr (NJ == 2) asm volatile("v_xor_b32 %0, 32, %1" : "=v"(o1) : "v"(bp[tap][NJ - 1])); \
    }                                                                                     \
    LDS_RD128(fb[set][0], o0, (cpar) * XBYTES);                                           \
    if constexpr (NJ == 2) LDS_RD128(fb[set][NJ - 1], o1, (cpar) * XBYTES);               \
    LDS_RD128(fa[set][0], ap[ks], (slot) * WTILE + 0 * 32 * ROWB);                        \
    LDS_RD128(fa[set][1], ap[ks], (slot) * WTILE + 1 * 32 * ROWB);                        \
    if constexpr (NI == 4) {                                                              \
      LDS_RD128(fa[set][2], ap[ks], (slot) * WTILE + 2 * 32 * ROWB);                      \
      LDS_RD128(fa[set][3], ap[ks], (slot) * WTILE + 3 * 32 * ROWB);                      \
    }                                                                                     \
  }

  for (int chunk2 = 0; chunk2 < nchunks; chunk2 += 2) {
    static_for<0, 2 * TAPS>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int tap = u % TAPS, cpar = u / TAPS;
      const int chunk = chunk2 + cpar;
      const bool more_chunks = chunk + 1 < nchunks;
      // ---- retire weight tiles t and t+1 (and slab chunk+1 before its first read at tap 8); younger DMAs stay in
      // flight across the barrier: D-2 younger weight tiles, plus slab chunk+1 while it is younger than W(t+1)
      if (more_chunks) {
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
      if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
      // ---- issue weight tile t+D into the ring slot read at iteration t-1, then (tap 0) the next slab
      if constexpr (!(ABL & 2)) {
        constexpr int tq = tap + D;                       // tap index of tile t+D, maybe in the next chunk
        constexpr int cq = cpar + (tq >= TAPS ? 1 : 0);   // chunk offset from chunk2 (0, 1 or 2)
        constexpr int tapq = tq >= TAPS ? tq - TAPS : tq;
        if (chunk2 + cq < nchunks && w_lane)
          dma16(wsrc + (long)(chunk2 + cq) * (KC * 2) + tapq * tap_stride,
                Wb + ((u + D) % WRING) * WTILE + wave * (WROWS * ROWB));
      }
      if (tap == 0 && more_chunks && !(ABL & 2)) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
          dma16(xsrc[i] + (long)(chunk + 1) * (KC * 2), Xb + (cpar ^ 1) * XBYTES + (wave + 8 * i) * 1024);
      }
      // ---- 2 x 8 MFMAs; all addresses = register + immediate.
      // Stagger (MI355X_MICROARCH.md, "Two waves per SIMD", item 9): the two waves of a SIMD run this same program in
      // lockstep -- both read, then both multiply.  Waves 4-7 therefore DEFER the second half-step's 8 MFMAs (their
      // operands are already in registers) to just behind the next barrier, where waves 0-3 are issuing DMA and
      // fragment reads: after every barrier one partner feeds the matrix pipe while the other feeds the LDS pipe.
      // The deferred MFMAs touch registers only, so the ring-slot reuse rules are unchanged (every read of tile t is
      // still retired before the barrier of step t+1).
      if (late && !(u == 0 && chunk2 == 0)) {
        mfma_half(1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (u == 0 && chunk2 == 0) READ_HALF(0, 0, 0, 0, 0);  // pipeline fill (first step of the kernel only)
      READ_HALF(1, 1, tap, cpar, u % WRING);
      lgkm_wait<NJ + NI>();
      mfma_half(0);
      __builtin_amdgcn_sched_barrier(0);
      if (more_chunks || tap + 1 < TAPS) {
        READ_HALF(0, 0, (tap + 1) % TAPS, (u + 1) / TAPS % 2, (u + 1) % WRING);
        lgkm_wait<NJ + NI>();
      } else {
        lgkm_wait<0>();
      }
      if (!late) {
        mfma_half(1);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  }
  if (late) mfma_half(1);   // the last step's deferred half
#undef READ_HALF

  // ---- epilogue: transposed through wave-private LDS so that residual reads and stores are whole rows
  // (common.h: store_tile_transposed; direct stores from the MFMA layout cost 1.7x the HBM write bytes)
  __builtin_amdgcn_s_barrier();  // every wave is done with the slab / weight ring
  store_tile_transposed<NI, NJ, EPI>(acc, smem + wave * (32 * (NI * 64 + 16)), Y, R, alpha, beta, (long)m0 + wave * (32 * NJ), Npix,
                                    n0, Cout, mod);
}

template <int NX, int EPI = 0, int NI = 4, int ABL = 0, int NJ = 2>
void launch4(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int Npix, int H, int W,
             int Cin, int Cout, const ModEpilogue& mod, hipStream_t st) {
  constexpr int BMW = 8 * 32 * NJ;
  const int tiles_m = (Npix + BMW - 1) / BMW, tiles_n = (Cout + 32 * NI - 1) / (32 * NI);
  const size_t lds = (size_t)2 * NX * 8 * 16 * ROWB + WRING * (32 * NI * ROWB);
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  auto kern = k_conv3x3_v4<NX, EPI, NI, ABL, NJ>;
  // wave stagger (see the kernel): measured null on this kernel (+-1 % in A/B runs on one device), off by default
  static const int stagger = [] { const char* e = getenv("EDM_V4_STAGGER"); return e ? atoi(e) : 0; }();
  static std::atomic<bool> attr_set{false};  // (idempotent call: a race only repeats it)
  if (!attr_set.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const bf16*)X, (const bf16*)Wp, (bf16*)Y, (const bf16*)R,
                     (const char*)edm_zero_page(), alpha, beta, Npix, H, W, Cin, Cout, tiles_m, tiles_n, stagger, mod);
}

}  // namespace

// the static-schedule kernel pays off when it can give every CU a tile: >= 512 tiles of 512x128, or >= 256 of 512x64
bool edm_conv_v4_worthwhile(long npix, int Cout) {
  const long tm = (npix + BM - 1) / BM;
  return tm * ((Cout + 127) / 128) >= 512 || tm * ((Cout + 63) / 64) >= 256;
}

// conv_igemm6.hip: the same kernel on v_mfma_f32_16x16x32_bf16 (W % 16 == 0)
int edm_conv_igemm_v6_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                         int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st);

// 3x3 only.  Same contract as edm_conv_igemm; returns EDM_ERR_UNSUPPORTED (-3) for shapes it does not cover
// (taps != 9, Cin % 64 != 0, Cin > 2048, W > 64, fewer than 9*Cin/32 >= 18 tiles ...).
int edm_conv_igemm_v4_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                         int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st) {
  EDM_REQUIRE(X && Wp && (Y || mod.Y2), "conv_igemm_v4: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "conv_igemm_v4: bad B/H/W");
  EDM_REQUIRE(Cout > 0 && Cout % 8 == 0, "conv_igemm_v4: Cout %% 8 required");
  if (taps != 9 || Cin <= 0 || Cin % 64 != 0 || Cin * 2 + 64 > ZERO_PAGE || W > 64) return EDM_ERR_UNSUPPORTED;
  EDM_ZERO_PAGE(zero_page_, "conv_igemm_v4");
  (void)zero_page_;
  // MFMA shape: the shapes it covers go to the 16x16x32 form of this kernel (conv_igemm6.hip: the device holds a higher
  // clock on that shape; 32x32 layers +11-19 %, 16x16 layers +2-3 %); EDM_V4_MFMA16=0 keeps the 32x32x16 form
  static const int mfma16 = [] { const char* e = getenv("EDM_V4_MFMA16"); return e ? atoi(e) : 1; }();
  if (mfma16) {
    const int rc = edm_conv_igemm_v6_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  const int Npix = B * H * W;
  const int xrows = BM + 2 * (W + 1);
  // 512x128 workgroup tiles when that still gives every CU two of them, else 512x64 (small feature maps)
  const long tiles4 = (long)((Npix + BM - 1) / BM) * ((Cout + 127) / 128);
  const bool wide = tiles4 >= 512;
#define L4(NXV, EPIV)                                                                                   \
  (wide ? launch4<NXV, EPIV, 4>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st)               \
        : launch4<NXV, EPIV, 2>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st))
  // backward epilogues: their own instantiations (keep the common kernels free of their registers)
  const bool nx5 = xrows < 5 * 128;
  static const int abl = [] { const char* e = getenv("EDM_V4_ABLATE"); return e ? atoi(e) : 0; }();   // tools only
  if (abl && wide && nx5 && mod.mode == 0) {
    switch (abl) {
      case 1: launch4<5, 0, 4, 1>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st); break;
      case 2: launch4<5, 0, 4, 2>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st); break;
      case 4: launch4<5, 0, 4, 4>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st); break;
      case 6: launch4<5, 0, 4, 6>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st); break;
      case 16: launch4<5, 0, 4, 16>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st); break;
      case 18: launch4<5, 0, 4, 18>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st); break;
      default: launch4<5, 0, 4, 22>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st); break;
    }
    EDM_CHECK_LAUNCH("conv_igemm_v4");
    return EDM_OK;
  }
  if (mod.mode == 1) { if (nx5) L4(5, 1); else L4(6, 1); }
  else if (mod.mode == 2) { if (nx5) L4(5, 2); else L4(6, 2); }
  else { if (nx5) L4(5, 0); else L4(6, 0); }
#undef L4
  EDM_CHECK_LAUNCH("conv_igemm_v4");
  return EDM_OK;
}

extern "C" int edm_conv_igemm_v4(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B,
                                 int H, int W, int Cin, int Cout, int taps, hipStream_t st) {
  EDM_REQUIRE(Y, "conv_igemm_v4: null pointer");
  return edm_conv_igemm_v4_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, ModEpilogue{}, st);
}

int edm_conv_igemm_v1_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                         int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st);
// conv_igemm5.hip: small feature maps (reduction split over the waves of a workgroup)
bool edm_conv_s_worthwhile(long npix, int W, int Cin, int Cout);
int edm_conv_igemm_s_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                        int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st);

// 3x3 conv with the fused embedding modulation epilogue (networks.py:253-260 / 317-324):
//   u  = conv3x3(X, Wp)                               -> Y  (bf16; may be null when the caller does not need it: eval)
//   a2 = dropout(mp_silu(u * (lin[b,:]*gain + 1)))     -> Y2 (bf16)   [same values as edm_mod_silu_drop_fwd on u]
// mark_dropped != 0: the elements of Y the dropout removed are written as NaN (bf16 | 0x7FFF) instead of u -- their value is
// never needed again, and edm_conv3x3_modbwd(u_marked = 1) / edm_mod_silu_drop_bwd then read the mask from U.
// Picks the static-schedule kernel for layers with >= 512 tall tiles and the 128x128-tile kernel otherwise.
extern "C" int edm_conv3x3_mod(const void* X, const void* Wp, void* Y, void* Y2, const float* lin, long lin_stride,
                               const float* gain, float pdrop, unsigned long long seed, unsigned sub, unsigned step,
                               int mark_dropped, int B, int H, int W, int Cin, int Cout, const void* dyn, hipStream_t st) {
  EDM_REQUIRE(X && Wp && Y2 && lin && gain, "conv3x3_mod: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0 && lin_stride >= Cout && pdrop >= 0.f && pdrop < 1.f,
              "conv3x3_mod: bad args");
  ModEpilogue mod{lin, gain, (bf16*)Y2, lin_stride, H * W, pdrop, (uint32_t)seed, (uint32_t)(seed >> 32), sub, step,
                  nullptr, nullptr, nullptr, 0.f, 0, (const StepParams*)dyn, 0, (mark_dropped && pdrop > 0.f) ? 1 : 0};
  if (edm_conv_v4_worthwhile((long)B * H * W, Cout)) {
    const int rc = edm_conv_igemm_v4_ex(X, Wp, Y, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  if (edm_conv_s_worthwhile((long)B * H * W, W, Cin, Cout)) {
    const int rc = edm_conv_igemm_s_ex(X, Wp, Y, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  return edm_conv_igemm_v1_ex(X, Wp, Y, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
}

// Backward counterpart: the dgrad of a block's second 3x3 conv with the modulation backward fused into its epilogue
// (networks.py:253-263 under autograd).  With ga = alpha * conv3x3(dY, Wd) (bf16, never written):
//   gr = ga * keep * mp_silu'(u*m) * m   -> GR (bf16),     gm[b,c] += sum_px ga * keep * mp_silu'(u*m) * u   (fp32 atomics)
// where m = lin*gain + 1 and u = U is the forward's pre-activation.  gm must be zero-filled [B][Cout]; follow with
// edm_mod_finish.  Same values as edm_conv_igemm + edm_mod_silu_drop_bwd (gm up to summation order).
// Returns EDM_ERR_UNSUPPORTED (-3) when H*W is not a multiple of 32 (a 32-pixel block would straddle images).
// gm_stride: row stride of gm in floats (>= Cout; 0 = Cout): gm may be a column slice of a buffer shared by all blocks of
// a network, finished by ONE edm_mod_finish_multi launch at the end of the backward pass.
// u_marked != 0: U comes from edm_conv3x3_mod(mark_dropped = 1) with the same pdrop -- an element is dropped iff U holds a
// NaN there, and no Philox stream is regenerated (seed / sub / step are ignored).
extern "C" int edm_conv3x3_modbwd(const void* dY, const void* Wd, float alpha, const void* U, const float* lin,
                                  long lin_stride, const float* gain, void* GR, float* gm, long gm_stride, float pdrop,
                                  unsigned long long seed, unsigned sub, unsigned step, int u_marked, int B, int H, int W,
                                  int Cin, int Cout, const void* dyn, hipStream_t st) {
  EDM_REQUIRE(dY && Wd && U && lin && gain && GR && gm, "conv3x3_modbwd: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0 && lin_stride >= Cout && pdrop >= 0.f && pdrop < 1.f &&
                  (gm_stride == 0 || gm_stride >= Cout),
              "conv3x3_modbwd: bad args");
  if ((H * W) % 32 != 0) return EDM_ERR_UNSUPPORTED;
  ModEpilogue mod{lin, gain, (bf16*)GR, lin_stride, H * W, pdrop, (uint32_t)seed, (uint32_t)(seed >> 32), sub, step,
                  (const bf16*)U, gm, nullptr, 0.f, 1, (const StepParams*)dyn, gm_stride, (u_marked && pdrop > 0.f) ? 1 : 0};
  if (edm_conv_v4_worthwhile((long)B * H * W, Cout)) {
    const int rc = edm_conv_igemm_v4_ex(dY, Wd, nullptr, nullptr, alpha, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  if (edm_conv_s_worthwhile((long)B * H * W, W, Cin, Cout)) {
    const int rc = edm_conv_igemm_s_ex(dY, Wd, nullptr, nullptr, alpha, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  return edm_conv_igemm_v1_ex(dY, Wd, nullptr, nullptr, alpha, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
}

// dgrad of a block's FIRST 3x3 conv with the mp_silu backward of the block input fused into its epilogue
// (networks.py:249-252 / 313-316 under autograd): with g = conv3x3(dY, Wd) (bf16, never written),
//   GX = mp_silu'(Xpre) * g + add_scale * ADD        (ADD optional: the residual-path gradient)
// Same values as edm_conv_igemm followed by edm_silu_bwd.
extern "C" int edm_conv3x3_silubwd(const void* dY, const void* Wd, const void* Xpre, const void* ADD, float add_scale,
                                   void* GX, int B, int H, int W, int Cin, int Cout, hipStream_t st) {
  EDM_REQUIRE(dY && Wd && Xpre && GX, "conv3x3_silubwd: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0, "conv3x3_silubwd: bad args");
  ModEpilogue mod{nullptr, nullptr, (bf16*)GX, 0, H * W, 0.f, 0u, 0u, 0u, 0u, (const bf16*)Xpre, nullptr, (const bf16*)ADD,
                  add_scale, 2, nullptr};
  if (edm_conv_v4_worthwhile((long)B * H * W, Cout)) {
    const int rc = edm_conv_igemm_v4_ex(dY, Wd, nullptr, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  if (edm_conv_s_worthwhile((long)B * H * W, W, Cin, Cout)) {
    const int rc = edm_conv_igemm_s_ex(dY, Wd, nullptr, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  return edm_conv_igemm_v1_ex(dY, Wd, nullptr, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
}

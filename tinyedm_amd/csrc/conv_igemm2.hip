// Implicit-GEMM convolution, second generation: same math and interface as conv_igemm.hip
// ("flat-M, masked halo"), rebuilt around the CDNA4 async global->LDS path.
//
//   Y[p, co] = alpha * sum_{tap, ci} X[p + off(tap), ci] * Wp[tap, co, ci]  (+ beta * R[p, co])
//
// What changed against generation 1 (measured 0.70-0.88 PFLOP/s, 38 % of wave time issuing):
//  * operands go HBM/L2 -> LDS by `global_load_lds_dwordx4` (LDS-DMA): no VGPR staging, no ds_write
//    pass, no per-load predication branches (out-of-range rows read a 16-byte zero page instead);
//  * weight tiles run through a 3-deep LDS ring and the pixel slab is double buffered, with COUNTED
//    `s_waitcnt vmcnt(N)` + raw `s_barrier`: two tiles stay in flight across every barrier;
//  * tile 256 pixels x 128 channels, 8 waves, 32-channel K-steps: 65-73 KB LDS and <= 128 VGPRs, so two
//    workgroups (16 waves, 4 per SIMD) share a CU and one group's barrier hides under the other's MFMAs;
//    each weight tile now serves 256 pixels (half the L2->LDS weight traffic per FLOP);
//  * LDS rows are unpadded 64-byte rows (LDS-DMA writes 1 KiB lane-linear), bank conflicts are removed by
//    an XOR swizzle chunk' = chunk ^ ((row >> 2) & 3) applied to the DMA *source* address and to the
//    ds_read_b128 address.
#include "common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 256, BN = 128, KC = 32;
constexpr int ROWB = KC * 2;           // 64-byte LDS rows
constexpr int WTILE = BN * ROWB;       // 8 KiB weight tile
constexpr int WRING = 3;

__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// diagnostic stamps (STAMP=true builds only; never used by the product path): cycles per loop segment
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

// EPI 4 (round 6, 1x1 only): the split-bf16 evaluation's operands and fp32 epilogue (common.h, mode 4) -- X rows are [hi | lo]
// pairs of ldX elements, K chunk c reads X chunk (c < kwrap ? c : c - kwrap), the result leaves as floats / pairs.
template <int TAPS, int NX, bool STAMP = false, int ABL = 0, int EPI = 0>
__global__ __launch_bounds__(512, EPI == 4 ? 2 : 4) void k_conv_igemm2(const bf16* __restrict__ X, const bf16* __restrict__ Wp,
                                                          bf16* __restrict__ Y, const bf16* __restrict__ R,
                                                          const bf16* __restrict__ zeros, float alpha, float beta,
                                                          int Npix, int H, int W, int Cin, int Cout, int tiles_m,
                                                          int tiles_n, unsigned long long* dbg = nullptr,
                                                          ModEpilogue mod = ModEpilogue{}) {
  constexpr int XBUFS = (TAPS == 9) ? 2 : 3;
  constexpr int XROWS = NX * 8 * 16;  // LDS rows per slab buffer (every DMA slot is backed by LDS)
  constexpr int XBYTES = XROWS * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + XBUFS * XBYTES;

  const int id = blockIdx.x;
  const int xcd = id & 7, k = id >> 3;
  const int tn = k % tiles_n, tm = (k / tiles_n) * 8 + xcd;
  if (tm >= tiles_m) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int HALO = (TAPS == 9) ? (W + 1) : 0;
  const int xrows = BM + 2 * HALO;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;  // 2 channel halves x 4 pixel quarters
  const int l31 = lane & 31, lhi = lane >> 5;
  const int drow = lane >> 2, dp = lane & 3;  // DMA lane -> (row in 16-row slot, physical 16-B chunk)
  const long ldX = (EPI == 4 && mod.ldX) ? mod.ldX : Cin;
  const int kwrap = (EPI == 4 && mod.kwrap) ? mod.kwrap : (1 << 30);

  // ---- DMA issue helpers (one wave-instruction = 16 rows x 64 B = 1 KiB, lane-linear in LDS)
  auto issue_x = [&](int chunk, int buf) {
    const int xc = chunk >= kwrap ? chunk - kwrap : chunk;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int slot = wave + 8 * i;
      const int row = slot * 16 + drow;
      const long pix = (long)m0 - HALO + row;
      const int c = dp ^ ((row >> 2) & 3);
      const bool ok = row < xrows && pix >= 0 && pix < Npix;
      const bf16* src = ok ? X + pix * ldX + xc * KC + c * 8 : zeros;
      dma16(src, Xb + buf * XBYTES + slot * 1024);
    }
  };
  auto issue_w = [&](int chunk, int tap, int buf) {
    const int row = wave * 16 + drow;
    const int co = n0 + row;
    const int c = dp ^ ((row >> 2) & 3);
    const bf16* src = (co < Cout) ? Wp + ((long)tap * Cout + co) * Cin + chunk * KC + c * 8 : zeros;
    dma16(src, Wb + buf * WTILE + wave * 1024);
  };

  // ---- per-lane tap masks / slab rows of the two pixel blocks this wave multiplies
  unsigned mask[2];
  int brow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ml = wn * 64 + j * 32 + l31;
    const int m = m0 + ml;
    brow[j] = ml + HALO;
    unsigned mk = 0;
    if (TAPS == 9) {
      const int w = m % W, h = (m / W) % H;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) mk |= 1u << t;
      }
    } else {
      mk = 1;
    }
    mask[j] = mk;
  }
  // weight-fragment rows: wm*64 + i*32 + l31 -> swizzle term depends on l31 only
  const int a_sw = (l31 >> 2) & 3;
  const int a_off0 = (wm * 64 + l31) * ROWB + (((0 + lhi) ^ a_sw) << 4);
  const int a_off1 = (wm * 64 + l31) * ROWB + (((2 + lhi) ^ a_sw) << 4);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = Cin / KC;
  const int T = nchunks * TAPS;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- prologue
  if (TAPS == 9) {
    issue_x(0, 0);
    issue_w(0, 0, 0);
    if (T > 1) issue_w(0, 1, 1);
  } else {
    issue_x(0, 0);
    issue_w(0, 0, 0);
    if (T > 1) {
      issue_x(1, 1);
      issue_w(1, 0, 1);
    }
  }

  unsigned long long seg[5] = {0, 0, 0, 0, 0}, tprev = 0;
  if (STAMP) tprev = stamp();
  int chunk = 0, tap = 0;
  for (int t = 0; t < T; ++t) {
    // ---- retire tile t (counted: the younger tile(s) stay in flight across the barrier)
    if (ABL & 2) {
    } else if (t + 1 >= T) {
      wait_vmcnt<0>();
    } else if (TAPS == 9) {
      if ((tap == 1 || tap == 2) && chunk + 1 < nchunks) wait_vmcnt<NX + 1>();
      else wait_vmcnt<1>();
    } else {
      wait_vmcnt<NX + 1>();
    }
    if (STAMP) { unsigned long long n = stamp(); seg[0] += n - tprev; tprev = n; }
    if (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    if (STAMP) { unsigned long long n = stamp(); seg[1] += n - tprev; tprev = n; }
    // ---- issue tile t+2 (its ring slot was last read at iteration t-1: every wave is past that barrier)
    if (ABL & 2) {
    } else if (TAPS == 9) {
      if (t + 2 < T) {
        int tp2 = tap + 2, ch2 = chunk;
        if (tp2 >= TAPS) { tp2 -= TAPS; ++ch2; }
        issue_w(ch2, tp2, (t + 2) % WRING);
      }
      if (tap == 0 && chunk + 1 < nchunks) issue_x(chunk + 1, (chunk + 1) & 1);
    } else {
      if (t + 2 < T) {
        issue_w(t + 2, 0, (t + 2) % WRING);
        issue_x(t + 2, (t + 2) % XBUFS);
      }
    }
    if (STAMP) { unsigned long long n = stamp(); seg[2] += n - tprev; tprev = n; }
    // ---- MFMA over this (chunk, tap)
    const int toff = (TAPS == 9) ? ((tap / 3 - 1) * W + (tap % 3 - 1)) : 0;
    const char* wt = Wb + (t % WRING) * WTILE;
    const char* xs = Xb + ((TAPS == 9) ? (chunk & 1) : (t % XBUFS)) * XBYTES;
    const int r0 = brow[0] + toff, r1 = brow[1] + toff;
    const int s0 = (r0 >> 2) & 3, s1 = (r1 >> 2) & 3;
    const bool v0 = (mask[0] >> tap) & 1, v1 = (mask[1] >> tap) & 1;
    if (!(ABL & 4)) {
      bf16x8 a0 = *reinterpret_cast<const bf16x8*>(wt + a_off0);
      bf16x8 a1 = *reinterpret_cast<const bf16x8*>(wt + 32 * ROWB + a_off0);
      bf16x8 b0 = *reinterpret_cast<const bf16x8*>(xs + r0 * ROWB + (((0 + lhi) ^ s0) << 4));
      bf16x8 b1 = *reinterpret_cast<const bf16x8*>(xs + r1 * ROWB + (((0 + lhi) ^ s1) << 4));
      bf16x8 a2 = *reinterpret_cast<const bf16x8*>(wt + a_off1);
      bf16x8 a3 = *reinterpret_cast<const bf16x8*>(wt + 32 * ROWB + a_off1);
      bf16x8 b2 = *reinterpret_cast<const bf16x8*>(xs + r0 * ROWB + (((2 + lhi) ^ s0) << 4));
      bf16x8 b3 = *reinterpret_cast<const bf16x8*>(xs + r1 * ROWB + (((2 + lhi) ^ s1) << 4));
      b0 = v0 ? b0 : zero8;
      b1 = v1 ? b1 : zero8;
      b2 = v0 ? b2 : zero8;
      b3 = v1 ? b3 : zero8;
      if (STAMP) {
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        unsigned long long n = stamp(); seg[3] += n - tprev; tprev = n;
      }
      if (ABL & 1) {
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3));
        asm volatile("" ::"v"(b0), "v"(b1), "v"(b2), "v"(b3));
      } else {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b3, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b2, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b3, acc[1][1], 0, 0, 0);
      }
    }
    if (STAMP) { unsigned long long n = stamp(); seg[4] += n - tprev; tprev = n; }
    if (++tap == TAPS) { tap = 0; ++chunk; }
  }
  if (STAMP && dbg && lane == 0) {
#pragma unroll
    for (int q = 0; q < 5; ++q) atomicAdd(dbg + q, seg[q]);
    atomicAdd(dbg + 5, 1ull);
  }

  // ---- epilogue: transposed through wave-private LDS (common.h: store_tile_transposed)
  __syncthreads();  // every wave is done with the staged tiles
  if constexpr (EPI == 4)
    store_tile_f32<2, 2>(acc, reinterpret_cast<float*>(Y), reinterpret_cast<const float*>(R), alpha, beta,
                         (long)m0 + wn * 64, Npix, n0 + wm * 64, Cout, mod);
  else
  store_tile_transposed<2, 2>(acc, smem + (wm * 4 + wn) * (32 * (2 * 64 + 16)), Y, R, alpha, beta, (long)m0 + wn * 64, Npix,
                              n0 + wm * 64, Cout, mod);
}

template <int TAPS, int NX, int EPI = 0>
void launch2(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int Npix, int H, int W,
             int Cin, int Cout, hipStream_t st, const ModEpilogue& mod) {
  constexpr int XBUFS = (TAPS == 9) ? 2 : 3;
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + BN - 1) / BN;
  const size_t lds = (size_t)XBUFS * NX * 8 * 16 * ROWB + WRING * WTILE;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  auto kern = k_conv_igemm2<TAPS, NX, false, 0, EPI>;
  EDM_MAX_LDS(kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const bf16*)X, (const bf16*)Wp, (bf16*)Y, (const bf16*)R,
                     (const bf16*)edm_zero_page(), alpha, beta, Npix, H, W, Cin, Cout, tiles_m, tiles_n,
                     (unsigned long long*)nullptr, mod);
}

}  // namespace

// Diagnostic build (tools only): 3x3 shape, accumulates per-wave segment cycles into dbg[0..4], wave count in dbg[5].
// Segments: 0 vmcnt wait, 1 barrier, 2 DMA issue, 3 fragment reads landed, 4 MFMA issue.
extern "C" int edm_conv_igemm_v2_stamp(const void* X, const void* Wp, void* Y, int B, int H, int W, int Cin, int Cout,
                                       unsigned long long* dbg, hipStream_t st) {
  EDM_REQUIRE(X && Wp && Y && dbg && edm_zero_page() && W <= 32, "conv_igemm_v2_stamp: bad args (run the product kernel once first)");
  const int Npix = B * H * W;
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + BN - 1) / BN;
  const size_t lds = (size_t)2 * 3 * 8 * 16 * ROWB + WRING * WTILE;
  auto kern = k_conv_igemm2<9, 3, true>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(((tiles_m + 7) / 8) * 8 * tiles_n), dim3(512), lds, st, (const bf16*)X, (const bf16*)Wp,
                     (bf16*)Y, (const bf16*)nullptr, (const bf16*)edm_zero_page(), 1.f, 0.f, Npix, H, W, Cin, Cout, tiles_m,
                     tiles_n, dbg, ModEpilogue{});
  EDM_CHECK_LAUNCH("conv_igemm_v2_stamp");
  return EDM_OK;
}

// Diagnostic (tools only): timing-only ablations of the 3x3 v2 kernel (outputs are wrong by construction).
// mode bit0 no MFMA, bit1 no DMA/waits, bit2 no fragment reads + MFMA, bit3 no barrier.
template <int ABL>
static void launch_abl(const void* X, const void* Wp, void* Y, int Npix, int H, int W, int Cin, int Cout, hipStream_t st) {
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + BN - 1) / BN;
  const size_t lds = (size_t)2 * 3 * 8 * 16 * ROWB + WRING * WTILE;
  auto kern = k_conv_igemm2<9, 3, false, ABL>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(((tiles_m + 7) / 8) * 8 * tiles_n), dim3(512), lds, st, (const bf16*)X, (const bf16*)Wp,
                     (bf16*)Y, (const bf16*)nullptr, (const bf16*)edm_zero_page(), 1.f, 0.f, Npix, H, W, Cin, Cout, tiles_m,
                     tiles_n, (unsigned long long*)nullptr, ModEpilogue{});
}
extern "C" int edm_conv_igemm_v2_ablate(const void* X, const void* Wp, void* Y, int B, int H, int W, int Cin, int Cout,
                                        int mode, hipStream_t st) {
  EDM_REQUIRE(X && Wp && Y && edm_zero_page() && W <= 32, "conv_igemm_v2_ablate: bad args");
  const int Npix = B * H * W;
  switch (mode) {
    case 0: launch_abl<0>(X, Wp, Y, Npix, H, W, Cin, Cout, st); break;
    case 1: launch_abl<1>(X, Wp, Y, Npix, H, W, Cin, Cout, st); break;
    case 2: launch_abl<2>(X, Wp, Y, Npix, H, W, Cin, Cout, st); break;
    case 4: launch_abl<4>(X, Wp, Y, Npix, H, W, Cin, Cout, st); break;
    case 6: launch_abl<6>(X, Wp, Y, Npix, H, W, Cin, Cout, st); break;
    case 8: launch_abl<8>(X, Wp, Y, Npix, H, W, Cin, Cout, st); break;
    case 10: launch_abl<10>(X, Wp, Y, Npix, H, W, Cin, Cout, st); break;
    default: edm_set_error("conv_igemm_v2_ablate: unknown mode %d", mode); return EDM_ERR_ARG;
  }
  EDM_CHECK_LAUNCH("conv_igemm_v2_ablate");
  return EDM_OK;
}

// Same contract as edm_conv_igemm (conv_igemm.hip); returns EDM_ERR_UNSUPPORTED for shapes it does not cover so the
// dispatcher can fall back to generation 1.
int edm_conv_igemm_v2_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                         int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st) {
  EDM_REQUIRE(X && Wp && (Y || (mod.mode == 4 && mod.Y2)), "conv_igemm_v2: null pointer");
  EDM_REQUIRE(!mod.wfrag, "conv_igemm_v2: fragment-major weight packs are read by k_conv3x3_s only");
  EDM_REQUIRE(mod.mode == 0 || mod.mode == 3 || (mod.mode == 4 && taps == 1),
              "conv_igemm_v2: plain / strided-output epilogues (and the split-bf16 form of the 1x1 kernel) only");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "conv_igemm_v2: bad B/H/W");
  EDM_REQUIRE(taps == 1 || taps == 9, "conv_igemm_v2: taps must be 1 or 9");
  EDM_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 8 == 0, "conv_igemm_v2: Cin %% 32, Cout %% 8 required");
  if (taps == 9 && W > 64) return EDM_ERR_UNSUPPORTED;
  EDM_ZERO_PAGE(zero_page_, "conv_igemm_v2");
  (void)zero_page_;
  const int Npix = B * H * W;
  if (taps == 1) {
    if (mod.mode == 4) launch2<1, 2, 4>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod);
    else launch2<1, 2>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod);
  } else {
    const int xrows = BM + 2 * (W + 1);
    const int need = (xrows + 127) / 128;
    if (need <= 3) launch2<9, 3>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod);
    else launch2<9, 4>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod);
  }
  EDM_CHECK_LAUNCH("conv_igemm_v2");
  return EDM_OK;
}

extern "C" int edm_conv_igemm_v2(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B,
                                 int H, int W, int Cin, int Cout, int taps, hipStream_t st) {
  return edm_conv_igemm_v2_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, ModEpilogue{}, st);
}

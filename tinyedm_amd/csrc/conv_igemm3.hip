// Implicit-GEMM convolution, third generation ("tall tile"): same math and interface as conv_igemm2.hip.
//
//   Y[p, co] = alpha * sum_{tap, ci} X[p + off(tap), ci] * Wp[tap, co, ci]  (+ beta * R[p, co])
//
// Timing-only ablations of generation 2 on MI355X (tools/ablate_igemm.py; 3x3 256->256 @32x32, B=128:
// full 179 us; DMA+barrier only 88 us; fragment reads+MFMA only 131 us; ideal MFMA 71 us) showed three
// comparable costs: L2->LDS streaming, LDS fragment reads per MFMA, and read->MFMA latency per barrier.
// This generation changes the geometry to cut all three:
//  * tile 512 pixels x 128 channels, 8 waves, each wave a 128(co) x 64(px) block = 4x2 MFMA 32x32 tiles:
//    a weight tile serves 512 pixels (L2->LDS bytes per FLOP -41 %), fragment reads per MFMA drop from 1.0
//    to 0.75, and every barrier is followed by 16 MFMAs per wave instead of 8;
//  * LDS-DMA staging, XOR swizzle, 3-deep weight ring, double-buffered pixel slab and the counted-vmcnt
//    protocol are those of generation 2.
// One workgroup per CU (104-120 KB LDS, 8 waves, <= 256 VGPRs); chosen when the layer has >= 256 such tiles.
#include "common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 512, BN = 128, KC = 32;
constexpr int ROWB = KC * 2;      // 64-byte LDS rows
constexpr int WTILE = BN * ROWB;  // 8 KiB weight tile
constexpr int WRING = 6;   // weight tiles resident: 1 being read + 5 in flight (L2 latency under load ~1-2 us)

__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// counted wait with a run-time (wave-uniform) count: vmcnt takes an immediate, so dispatch over the possible values
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 11: wait_vmcnt<11>(); break;
    case 12: wait_vmcnt<12>(); break;
    default: wait_vmcnt<0>(); break;
  }
}
__device__ __forceinline__ bf16x8 lds128(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <int TAPS, int NX, bool CLK = false>
__global__ __launch_bounds__(512, 2) void k_conv_igemm3(const bf16* __restrict__ X, const bf16* __restrict__ Wp,
                                                          bf16* __restrict__ Y, const bf16* __restrict__ R,
                                                          const bf16* __restrict__ zeros, float alpha, float beta,
                                                          int Npix, int H, int W, int Cin, int Cout, int tiles_m,
                                                          int tiles_n, unsigned long long* dbg) {
  unsigned long long c0 = 0, r0t = 0;
  if (CLK) {  // diagnostic build only: shader-clock vs 100 MHz reference around the whole kernel
    c0 = __builtin_amdgcn_s_memtime();
    r0t = __builtin_amdgcn_s_memrealtime();
  }
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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = pixel octant of this wave
  const int l31 = lane & 31, lhi = lane >> 5;
  const int drow = lane >> 2, dp = lane & 3;  // DMA lane -> (row in 16-row slot, physical 16-B chunk)

  auto issue_x = [&](int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int slot = wave + 8 * i;
      const int row = slot * 16 + drow;
      const long pix = (long)m0 - HALO + row;
      const int c = dp ^ ((row >> 2) & 3);
      const bool ok = row < xrows && pix >= 0 && pix < Npix;
      const bf16* src = ok ? X + pix * Cin + chunk * KC + c * 8 : zeros;
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
    const int ml = wave * 64 + j * 32 + l31;
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
  // weight-fragment rows i*32 + l31 (all 128 channels per wave): swizzle term depends on l31 only
  const int a_sw = (l31 >> 2) & 3;
  const int a_off[2] = {l31 * ROWB + (((0 + lhi) ^ a_sw) << 4), l31 * ROWB + (((2 + lhi) ^ a_sw) << 4)};

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = Cin / KC;
  const int T = nchunks * TAPS;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- prologue (issue order matters for the counted waits below)
  constexpr int D = (TAPS == 9) ? WRING - 1 : 2;  // prefetch distance in (chunk,tap) tiles
  issue_x(0, 0);
  if (TAPS == 9) {
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < T) issue_w(d / TAPS, d % TAPS, d % WRING);
  } else {
    issue_w(0, 0, 0);
    if (T > 1) {
      issue_x(1, 1);
      issue_w(1, 0, 1);
    }
  }

  int chunk = 0, tap = 0;
  for (int t = 0; t < T; ++t) {
    // ---- retire tile t; the younger tile(s) stay in flight across the barrier
    if (TAPS == 9) {
      // DMAs issued after W(t): the younger weight tiles, plus the next slab while it is younger than W(t)
      // (slab c+1 is issued at tap 0 right after W(t+D), so it is younger than W(t) for taps 1..D of chunk c)
      const int n_w = min(D - 1, T - 1 - t);
      const int n_x = (tap >= 1 && tap <= D && chunk + 1 < nchunks) ? NX : 0;
      wait_vmcnt_dyn(n_w + n_x);
    } else if (t + 1 >= T) {
      wait_vmcnt<0>();
    } else {
      wait_vmcnt<NX + 1>();
    }
    __builtin_amdgcn_s_barrier();
    // ---- issue tile t+D (its ring slot was last read at iteration t-1: every wave is past that barrier)
    if (TAPS == 9) {
      if (t + D < T) {
        const int td = t + D;
        issue_w(td / TAPS, td % TAPS, td % WRING);
      }
      if (tap == 0 && chunk + 1 < nchunks) issue_x(chunk + 1, (chunk + 1) & 1);
    } else {
      if (t + 2 < T) {
        issue_w(t + 2, 0, (t + 2) % 3);
        issue_x(t + 2, (t + 2) % XBUFS);
      }
    }
    // ---- 16 MFMAs over this (chunk, tap): 2 k-steps x (4 weight blocks x 2 pixel blocks)
    const int toff = (TAPS == 9) ? ((tap / 3 - 1) * W + (tap % 3 - 1)) : 0;
    const char* wt = Wb + (t % ((TAPS == 9) ? WRING : 3)) * WTILE;
    const char* xs = Xb + ((TAPS == 9) ? (chunk & 1) : (t % XBUFS)) * XBYTES;
    const int r0 = brow[0] + toff, r1 = brow[1] + toff;
    const int s0 = (r0 >> 2) & 3, s1 = (r1 >> 2) & 3;
    const bool v0 = (mask[0] >> tap) & 1, v1 = (mask[1] >> tap) & 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 b0 = lds128(xs + r0 * ROWB + (((2 * ks + lhi) ^ s0) << 4));
      bf16x8 b1 = lds128(xs + r1 * ROWB + (((2 * ks + lhi) ^ s1) << 4));
      bf16x8 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = lds128(wt + i * 32 * ROWB + a_off[ks]);
      b0 = v0 ? b0 : zero8;
      b1 = v1 ? b1 : zero8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b1, acc[i][1], 0, 0, 0);
      }
    }
    if (++tap == TAPS) { tap = 0; ++chunk; }
  }

  if (CLK && dbg && tid == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1t = __builtin_amdgcn_s_memrealtime();
    atomicAdd(dbg + 0, c1 - c0);
    atomicAdd(dbg + 1, r1t - r0t);
    atomicAdd(dbg + 2, 1ull);
  }
  // ---- epilogue: lane holds, per (i,j), pixel = col(l31) and 4x4 consecutive output channels
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long m = (long)m0 + wave * 64 + j * 32 + l31;
    if (m >= Npix) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = n0 + i * 32 + 8 * g + 4 * lhi;
        if (co < Cout) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = alpha * acc[i][j][4 * g + r];
          if (R) {
            bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + m * Cout + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += beta * (float)rv[r];
          }
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
          *reinterpret_cast<bf16x4*>(Y + m * Cout + co) = o;
        }
      }
    }
  }
}

template <int TAPS, int NX>
void launch3(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int Npix, int H, int W,
             int Cin, int Cout, hipStream_t st) {
  constexpr int XBUFS = (TAPS == 9) ? 2 : 3;
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + BN - 1) / BN;
  const size_t lds = (size_t)XBUFS * NX * 8 * 16 * ROWB + WRING * WTILE;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  auto kern = k_conv_igemm3<TAPS, NX>;
  static std::atomic<bool> attr_set{false};  // (idempotent call: a race only repeats it)
  if (!attr_set.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const bf16*)X, (const bf16*)Wp, (bf16*)Y, (const bf16*)R,
                     (const bf16*)edm_zero_page(), alpha, beta, Npix, H, W, Cin, Cout, tiles_m, tiles_n,
                     (unsigned long long*)nullptr);
}

}  // namespace

// Diagnostic (tools only): the product 3x3 kernel bracketed by s_memtime / s_memrealtime -> dbg[0] shader cycles,
// dbg[1] 100 MHz ticks, dbg[2] workgroups.
extern "C" int edm_conv_igemm_v3_clock(const void* X, const void* Wp, void* Y, int B, int H, int W, int Cin, int Cout,
                                       unsigned long long* dbg, hipStream_t st) {
  EDM_REQUIRE(X && Wp && Y && dbg && edm_zero_page() && W <= 32, "conv_igemm_v3_clock: bad args");
  const int Npix = B * H * W;
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + BN - 1) / BN;
  const size_t lds = (size_t)2 * 5 * 8 * 16 * ROWB + WRING * WTILE;
  auto kern = k_conv_igemm3<9, 5, true>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(((tiles_m + 7) / 8) * 8 * tiles_n), dim3(512), lds, st, (const bf16*)X, (const bf16*)Wp,
                     (bf16*)Y, (const bf16*)nullptr, (const bf16*)edm_zero_page(), 1.f, 0.f, Npix, H, W, Cin, Cout, tiles_m,
                     tiles_n, dbg);
  EDM_CHECK_LAUNCH("conv_igemm_v3_clock");
  return EDM_OK;
}

// Same contract as edm_conv_igemm; returns EDM_ERR_UNSUPPORTED (-3) for shapes it does not cover.
extern "C" int edm_conv_igemm_v3(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B,
                                 int H, int W, int Cin, int Cout, int taps, hipStream_t st) {
  EDM_REQUIRE(X && Wp && Y, "conv_igemm_v3: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "conv_igemm_v3: bad B/H/W");
  EDM_REQUIRE(taps == 1 || taps == 9, "conv_igemm_v3: taps must be 1 or 9");
  EDM_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 8 == 0, "conv_igemm_v3: Cin %% 32, Cout %% 8 required");
  if (taps == 9 && W > 64) return EDM_ERR_UNSUPPORTED;
  EDM_ZERO_PAGE(zero_page_, "conv_igemm_v3");
  (void)zero_page_;
  const int Npix = B * H * W;
  if (taps == 1) {
    launch3<1, 4>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st);
  } else {
    const int xrows = BM + 2 * (W + 1);
    const int need = (xrows + 127) / 128;  // 16-row DMA slots per wave
    if (need <= 5) launch3<9, 5>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st);
    else launch3<9, 6>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st);
  }
  EDM_CHECK_LAUNCH("conv_igemm_v3");
  return EDM_OK;
}

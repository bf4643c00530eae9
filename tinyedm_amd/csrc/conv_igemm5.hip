// 3x3 implicit-GEMM convolution for SMALL feature maps (the 8x8 layers: 8192 pixels at batch 128).
//
//   Y[p, co] = alpha * sum_{tap, ci} X[p + off(tap), ci] * Wp[tap, co, ci]  (+ beta * R[p, co])
//
// Why a kernel of its own: 8192 x 256 outputs over 256 CUs is 8192 outputs -- two 32x32 MFMA blocks -- per wave.
// Tiled the usual way (one output sub-tile per wave, conv_igemm.hip: 64 px x 128 co per workgroup) every wave reads
// 1.5 fragments per MFMA, every workgroup streams the whole 2304-deep weight slice of its channels through L2 -> LDS
// (256 x 590 KB = the aggregate L2 rate for the whole ideal MFMA time) and one wave per SIMD hides nothing:
// r01 measured 0.27 PFLOP/s.  Here the reduction dimension is split over the four waves of a workgroup instead:
//  * workgroup tile 128 px x 64 co (256 workgroups at 8x8 x 128); wave w computes the WHOLE tile for the input
//    channel chunks c = w (mod 4): 2 x 4 accumulator blocks, 6 fragment reads per 8 MFMAs (as the 32x32 layers'
//    kernel), 376 KB staged per workgroup instead of 632 KB;
//  * each wave stages ITS OWN operands (its chunk's pixel slab, double buffered, and a 4-deep ring of its chunk's
//    weight tiles) by LDS-DMA into a private LDS region, so the main loop has NO workgroup barrier: only counted
//    vmcnt / lgkmcnt waits of the wave itself;
//  * the four partial tiles meet once, through LDS, in a fixed order (bit-reproducible), and every wave finishes a
//    32-pixel block of the tile through the shared transposed epilogue (residual, alpha/beta, fused modulation
//    forward/backward, mp_silu backward: common.h).
// Geometry: Cin % 256 == 0 (rounds of 4 chunks, unrolled in pairs), W <= 16, Cout % 8 == 0.
#include "common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int BM = 128, BNW = 64, KC = 32, TAPS = 9;
constexpr int ROWB = KC * 2;                 // 64-byte LDS rows
constexpr int NXI = 11;                      // 16-row slab slots per wave: 176 rows >= 128 + 2*(16+1), last rows zero
constexpr int XROWS = NXI * 16;
constexpr int XBYTES = XROWS * ROWB;         // 11 KiB
constexpr int WTILE = BNW * ROWB;            // 4 KiB weight tile (64 co x 32 ci)
constexpr int NWI = WTILE / 1024;            // 4 DMA instructions per weight tile
constexpr int DRING = 4;                     // weight ring depth (3 tiles in flight)
constexpr int WAVE_LDS = 2 * XBYTES + DRING * WTILE;   // 38 KiB per wave
constexpr int ESTAGE = 32 * (2 * 64 + 16);   // transposed-epilogue stage per wave (common.h)

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

template <int EPI>
__global__ __launch_bounds__(256, 1) void k_conv3x3_s(const bf16* __restrict__ X, const bf16* __restrict__ Wp,
                                                        bf16* __restrict__ Y, const bf16* __restrict__ R,
                                                        const char* __restrict__ zeros, float alpha, float beta,
                                                        int Npix, int H, int W, int Cin, int Cout, int tiles_m,
                                                        int tiles_n, ModEpilogue mod) {
  apply_dyn(mod);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int id = blockIdx.x;
  const int xcd = id & 7, k = id >> 3;
  const int tn = k % tiles_n, tm = (k / tiles_n) * 8 + xcd;   // the column tiles of one pixel tile share an XCD
  if (tm >= tiles_m) return;
  const int m0 = tm * BM, n0 = tn * BNW;
  const int HALO = W + 1;
  const int xrows = BM + 2 * HALO;            // <= 162 < XROWS: rows >= xrows are zero rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = this wave's chunk residue (mod 4)
  const int l31 = lane & 31, lhi = lane >> 5;
  const int drow = lane >> 2, dp = lane & 3;
  char* const Xb = smem + wave * WAVE_LDS;     // [2][XROWS][64 B]
  char* const Wb = Xb + 2 * XBYTES;            // [DRING][64 co][64 B]

  // ---- DMA sources.  Chunk c (32 input channels) of round r is c = 4 r + wave: byte offset (4 r + wave) * 64.
  const long tap_stride = (long)Cout * Cin * 2;
  const char* wsrc[NWI];
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int row = i * 16 + drow;
    const int co = min(n0 + row, Cout - 1);                     // rows >= Cout feed discarded outputs
    const int c = dp ^ ((row >> 2) & 3);
    wsrc[i] = reinterpret_cast<const char*>(Wp + (long)co * Cin + wave * KC + c * 8);
  }
  const char* xsrc[NXI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int row = i * 16 + drow;
    const int c = dp ^ ((row >> 2) & 3);
    if (row < xrows) {
      long pix = (long)m0 - HALO + row;
      pix = pix < 0 ? 0 : (pix >= Npix ? Npix - 1 : pix);       // out-of-range rows only feed masked taps
      xsrc[i] = reinterpret_cast<const char*>(X + pix * Cin + wave * KC + c * 8);
    } else {
      xsrc[i] = zeros + c * 16;                                 // zero rows: the pointer walks inside the zero page
    }
  }

  // ---- fragment addresses: 9 taps x 4 pixel blocks (border masks folded in: a masked lane reads a zero row)
  const unsigned xb_off = (unsigned)(uintptr_t)(lds_char*)Xb;
  unsigned bp[TAPS][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ml = j * 32 + l31;
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
  const int a_sw = (l31 >> 2) & 3;
  const unsigned wb_off = (unsigned)(uintptr_t)(lds_char*)Wb;
  const unsigned ap0 = wb_off + l31 * ROWB + (((0 + lhi) ^ a_sw) << 4);   // k-step 0; k-step 1 = ap0 ^ 32

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nrounds = Cin / (4 * KC);          // even (host-checked)
  const int nsteps = nrounds * TAPS;

  // ---- prologue: slab of round 0, weight tiles of steps 0 .. DRING-2
#pragma unroll
  for (int i = 0; i < NXI; ++i) dma16(xsrc[i], Xb + i * 1024);
#pragma unroll
  for (int d = 0; d < DRING - 1; ++d)
#pragma unroll
    for (int i = 0; i < NWI; ++i) dma16(wsrc[i] + d * tap_stride, Wb + d * WTILE + i * 1024);

  u32x4 fa[2][2], fb[2][4];
  auto mfma_half = [&](int set) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][i]),
                                                           __builtin_bit_cast(bf16x8, fb[set][j]), acc[i][j], 0, 0, 0);
  };
  // read one half-step's fragments: pixel rows of tap `tap` in slab buffer `xpar`, weight tile in ring slot `wslot`
#define READ_HALF(set, ks, tap, xpar, wslot)                                                        \
  {                                                                                                 \
    const unsigned a_ = (ap0 ^ ((ks) ? 32u : 0u)) + (wslot) * WTILE;                                \
    LDS_RD128(fb[set][0], bp[tap][0] ^ ((ks) ? 32u : 0u), (xpar) * XBYTES);                         \
    LDS_RD128(fb[set][1], bp[tap][1] ^ ((ks) ? 32u : 0u), (xpar) * XBYTES);                         \
    LDS_RD128(fb[set][2], bp[tap][2] ^ ((ks) ? 32u : 0u), (xpar) * XBYTES);                         \
    LDS_RD128(fb[set][3], bp[tap][3] ^ ((ks) ? 32u : 0u), (xpar) * XBYTES);                         \
    LDS_RD128(fa[set][0], a_, 0);                                                                   \
    LDS_RD128(fa[set][1], a_, 32 * ROWB);                                                           \
  }

  int u = 0;   // global step = round * 9 + tap; ring slot = u & 3
  for (int r2 = 0; r2 < nrounds; r2 += 2) {
    static_for<0, 2 * TAPS>([&](auto uc) {
      constexpr int v = decltype(uc)::value;
      constexpr int tap = v % TAPS, xpar = v / TAPS;
      const int round = r2 + xpar;
      const bool more_rounds = round + 1 < nrounds;
      // ---- retire W(u), W(u+1) (the second half of this step already reads tile u+1's successor pattern: the
      // first half-step of step u+1 is read during step u) and, from tap DRING-1 on, the next round's slab
      if (more_rounds) {
        if (tap >= 1 && tap <= DRING - 2) wait_vmcnt<NWI * (DRING - 3) + NXI>();
        else wait_vmcnt<NWI * (DRING - 3)>();
      } else {
        if (u + 2 <= nsteps - 1) wait_vmcnt<NWI * (DRING - 3)>();
        else wait_vmcnt<0>();
      }
      // ---- issue W(u + DRING - 1) into the slot tile u-1 was read from, then (tap 0) the next round's slab
      {
        const int uq = u + DRING - 1;
        if (uq < nsteps) {
          const int rq = uq / TAPS, tq = uq - rq * TAPS;
          const long off = (long)rq * (4 * KC * 2) + tq * tap_stride;
          char* dst = Wb + (uq & (DRING - 1)) * WTILE;
#pragma unroll
          for (int i = 0; i < NWI; ++i) dma16(wsrc[i] + off, dst + i * 1024);
        }
      }
      if (tap == 0 && more_rounds) {
        const long off = (long)(round + 1) * (4 * KC * 2);
#pragma unroll
        for (int i = 0; i < NXI; ++i)
          dma16(xsrc[i] + off, Xb + (xpar ^ 1) * XBYTES + i * 1024);   // zero-row lanes walk inside the zero page
      }
      const int wslot = u & (DRING - 1), wslot1 = (u + 1) & (DRING - 1);
      if (v == 0 && r2 == 0) READ_HALF(0, 0, 0, 0, wslot);      // pipeline fill (first step of the kernel only)
      READ_HALF(1, 1, tap, xpar, wslot);
      lgkm_wait<6>();
      mfma_half(0);
      __builtin_amdgcn_sched_barrier(0);
      if (more_rounds || tap + 1 < TAPS) {
        READ_HALF(0, 0, (tap + 1) % TAPS, (v + 1) / TAPS % 2, wslot1);
        lgkm_wait<6>();
      } else {
        lgkm_wait<0>();
      }
      mfma_half(1);
      __builtin_amdgcn_sched_barrier(0);
      ++u;
    });
  }
#undef READ_HALF

  // ---- reduce the four K-slices through LDS in a fixed order: wave w ends up with pixel block w (32 px x 64 co)
  // layout [source wave][pixel block j][co block i][lane][16 floats]
  __builtin_amdgcn_s_barrier();                 // every wave is done with its private staging region
  float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // (every block goes through LDS, the wave's own one included: selecting "acc[i][wave]" would index the
    // accumulator array dynamically and push all 128 accumulator registers into scratch memory)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* dst = red + ((((wave * 4 + j) * 2 + i) * 64 + lane) << 4);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
    }
  }
  __syncthreads();
  f32x16 out[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int src = 0; src < 4; ++src) {         // fixed order 0,1,2,3
      const float* p = red + ((((src * 4 + wave) * 2 + i) * 64 + lane) << 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
        s[4 * q] += t[0]; s[4 * q + 1] += t[1]; s[4 * q + 2] += t[2]; s[4 * q + 3] += t[3];
      }
    }
    out[i][0] = s;
  }
  __syncthreads();                              // the epilogue stage overlays the reduction buffer
  store_tile_transposed<2, 1, EPI>(out, smem + wave * ESTAGE, Y, R, alpha, beta, (long)m0 + wave * 32, Npix, n0, Cout, mod);
}

template <int EPI>
void launch5(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int Npix, int H, int W,
             int Cin, int Cout, const ModEpilogue& mod, hipStream_t st) {
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + BNW - 1) / BNW;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  auto kern = k_conv3x3_s<EPI>;
  static std::atomic<bool> attr_set{false};  // (idempotent call: a race only repeats it)
  if (!attr_set.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)4 * WAVE_LDS, st, (const bf16*)X, (const bf16*)Wp, (bf16*)Y,
                     (const bf16*)R, (const char*)edm_zero_page(), alpha, beta, Npix, H, W, Cin, Cout, tiles_m, tiles_n, mod);
}

}  // namespace

// small feature maps whose pixel count cannot give every CU one of the larger tiles
bool edm_conv_s_worthwhile(long npix, int W, int Cin, int Cout) {
  const long tiles = ((npix + BM - 1) / BM) * ((Cout + BNW - 1) / BNW);
  return W <= 16 && Cin % 256 == 0 && tiles >= 128 && tiles <= 1024;
}

// 3x3 only; same contract as edm_conv_igemm; EDM_ERR_UNSUPPORTED (-3) for shapes it does not cover.
int edm_conv_igemm_s_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                        int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st) {
  EDM_REQUIRE(X && Wp && (Y || mod.Y2), "conv_igemm_s: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "conv_igemm_s: bad B/H/W");
  EDM_REQUIRE(Cout > 0 && Cout % 8 == 0, "conv_igemm_s: Cout %% 8 required");
  if (taps != 9 || Cin <= 0 || Cin % 256 != 0 || W > 16 || Cin * 2 + 64 > 4096) return EDM_ERR_UNSUPPORTED;
  if (mod.mode == 1 && (H * W) % 32 != 0) return EDM_ERR_UNSUPPORTED;
  EDM_ZERO_PAGE(zero_page_, "conv_igemm_s");
  (void)zero_page_;
  const int Npix = B * H * W;
  static_assert(4 * WAVE_LDS <= 160 * 1024 && 4 * 4 * 2 * 64 * 16 * 4 <= 4 * WAVE_LDS, "LDS budget");
  if (mod.mode == 1) launch5<1>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st);
  else if (mod.mode == 2) launch5<2>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st);
  else launch5<0>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st);
  EDM_CHECK_LAUNCH("conv_igemm_s");
  return EDM_OK;
}

extern "C" int edm_conv_igemm_s(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B,
                                int H, int W, int Cin, int Cout, int taps, hipStream_t st) {
  EDM_REQUIRE(Y, "conv_igemm_s: null pointer");
  return edm_conv_igemm_s_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, ModEpilogue{}, st);
}

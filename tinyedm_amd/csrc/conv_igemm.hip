// Implicit-GEMM convolution (3x3 "same" and 1x1) on bf16 MFMA for gfx950.
//
//   Y[p, co] = alpha * sum_{tap, ci} X[p + off(tap), ci] * Wp[tap, co, ci]  (+ beta * R[p, co])
//
// Replaces the F.conv2d call of the reference's weight-normalised Conv2d (networks.py:35-37) and,
// with the flipped/transposed weight pack, its autograd dgrad.  Activations are NHWC bf16
// viewed as a flat [pixels][channels] matrix; weights are the packed bf16 [tap][co][ci]
// produced by edm_weight_prep.
//
// Design ("flat-M, masked halo"):
//  * a workgroup owns BM consecutive *flat* pixels x BN output channels.  The input slab for
//    all 9 taps is the contiguous flat range [m0-(W+1), m0+BM+(W+1)) -- staged ONCE per
//    ci-chunk into LDS (coalesced 16 B/lane reads), then reused by the 9 taps as a constant row
//    shift.  Image-border taps are zeroed per lane with a precomputed 9-bit mask (v_cndmask on
//    the B fragment), so any B,H,W works and no padded copy of the activation exists in HBM.
//  * MFMA roles: A = weights (rows = co), B = pixels (cols = pixel).  The 32x32 accumulator then
//    holds 4 consecutive co per register group for one pixel per lane -> 8-byte packed stores.
//  * weights stream through a 2-deep LDS ring, prefetched global->VGPR one (chunk,tap) ahead
//    (issue early / write late), one barrier per tap.
//  * LDS rows are padded by 16 B so the ds_read_b128 fragment reads are bank-conflict free.
#include "common.h"

namespace {

template <int KC>
struct Cfg {
  static constexpr int ROWB = KC * 2 + 16;  // padded LDS row bytes
  static constexpr int CPR = KC / 8;        // 16-B chunks per row
};

template <int TAPS, int KC, int XL, int NJ = 2, int EPI = 0>  // NJ = 32-pixel blocks per wave: tile = (64*NJ) pixels x 128 channels
__global__ __launch_bounds__(256, 2) void k_conv_igemm(const bf16* __restrict__ X, const bf16* __restrict__ Wp,
                                                         bf16* __restrict__ Y, const bf16* __restrict__ R,
                                                         float alpha, float beta, int Npix, int H, int W, int Cin,
                                                         int Cout, int tiles_m, int tiles_n, ModEpilogue mod) {
  apply_dyn(mod);
  constexpr int BM = 64 * NJ, BN = 128;
  constexpr int ROWB = Cfg<KC>::ROWB, CPR = Cfg<KC>::CPR;
  constexpr int WL = BN * CPR / 256;  // W loads per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- XCD-aware tile mapping: the tiles_n column tiles of one pixel tile run back to back on one XCD
  const int id = blockIdx.x;
  const int xcd = id & 7, k = id >> 3;
  const int tn = k % tiles_n, tm = (k / tiles_n) * 8 + xcd;
  if (tm >= tiles_m) return;
  const int m0 = tm * BM, n0 = tn * BN;

  const int HALO = (TAPS == 9) ? (W + 1) : 0;
  const int xrows = BM + 2 * HALO;
  char* Xs = smem;
  char* Ws = smem + ((xrows * ROWB + 15) & ~15);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  // ---- per-lane tap masks for the two pixel blocks this wave multiplies
  unsigned mask[NJ];
  int brow[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int ml = wn * (32 * NJ) + j * 32 + l31;
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

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = Cin / KC;
  // split-bf16 evaluation (common.h, mode 4): X rows are [hi | lo] pairs of ldX elements; K chunk c reads X chunk
  // (c < kwrap ? c : c - kwrap) (kwrap counted in chunks of KC channels here)
  const long ldX = mod.ldX ? mod.ldX : Cin;
  const int kwrapc = mod.kwrap ? mod.kwrap * 32 / KC : (1 << 30);
  auto xchunk = [&](int c) { return c >= kwrapc ? c - kwrapc : c; };
  const int T = nchunks * TAPS;
  const int xchunks = xrows * CPR;

  bf16x8 wreg[WL];
  bf16x8 xreg[XL];
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_w = [&](int chunk, int tap) {
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int q = tid + 256 * i;
      const int row = q / CPR, kc = q % CPR;
      const int co = n0 + row;
      wreg[i] = (co < Cout) ? *reinterpret_cast<const bf16x8*>(Wp + ((long)tap * Cout + co) * Cin + chunk * KC + kc * 8)
                            : zero8;
    }
  };
  auto load_x = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int q = tid + 256 * i;
      const int row = q / CPR, kc = q % CPR;
      const long pix = (long)m0 - HALO + row;
      xreg[i] = (q < xchunks && pix >= 0 && pix < Npix)
                    ? *reinterpret_cast<const bf16x8*>(X + pix * ldX + xchunk(chunk) * KC + kc * 8)
                    : zero8;
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int q = tid + 256 * i;
      const int row = q / CPR, kc = q % CPR;
      *reinterpret_cast<bf16x8*>(Ws + buf * (BN * ROWB) + row * ROWB + kc * 16) = wreg[i];
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int q = tid + 256 * i;
      const int row = q / CPR, kc = q % CPR;
      if (q < xchunks) *reinterpret_cast<bf16x8*>(Xs + row * ROWB + kc * 16) = xreg[i];
    }
  };

  load_w(0, 0);
  load_x(0);

  int chunk = 0, tap = 0;
  for (int t = 0; t < T; ++t) {
    if (tap == 0) {
      if (t > 0) __syncthreads();  // every wave is done reading the previous ci-chunk's slab
      store_x();
    }
    store_w(t & 1);
    __syncthreads();
    // prefetch the next (chunk, tap) into registers; consumed at the top of the next iteration
    int ntap = tap + 1, nchunk = chunk;
    if (ntap == TAPS) { ntap = 0; ++nchunk; }
    if (t + 1 < T) {
      load_w(nchunk, ntap);
      if (ntap == 0) load_x(nchunk);
    }
    // ---- MFMA over this tap's KC-deep slice
    const int toff = (TAPS == 9) ? ((tap / 3 - 1) * W + (tap % 3 - 1)) : 0;
    const char* wbase = Ws + (t & 1) * (BN * ROWB) + (wm * 64 + l31) * ROWB + lhi * 16;
    const char* xb[NJ];
    bool vv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      xb[j] = Xs + (brow[j] + toff) * ROWB + lhi * 16;
      vv[j] = (mask[j] >> tap) & 1;
    }
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      bf16x8 a0 = *reinterpret_cast<const bf16x8*>(wbase + ks * 32);
      bf16x8 a1 = *reinterpret_cast<const bf16x8*>(wbase + 32 * ROWB + ks * 32);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        bf16x8 b = *reinterpret_cast<const bf16x8*>(xb[j] + ks * 32);
        b = vv[j] ? b : zero8;
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1][j], 0, 0, 0);
      }
    }
    tap = ntap;
    chunk = nchunk;
  }

  // ---- epilogue: transposed through wave-private LDS (common.h: store_tile_transposed)
  __syncthreads();  // every wave is done with the staged tiles
  if constexpr (EPI == 4) {
    // (round 6: staged through wave-private LDS -- 4 waves x 8.7 KB of the slab / ring area -- like k_conv3x3_v6's;
    // mod.wfrag = the A/B switch EDM_F32_EPI_STAGED=0)
    if (mod.wfrag)
      store_tile_f32<2, NJ>(acc, reinterpret_cast<float*>(Y), reinterpret_cast<const float*>(R), alpha, beta,
                            (long)m0 + wn * (32 * NJ), Npix, n0 + wm * 64, Cout, mod);
    else
      store_tile_f32_staged<2, NJ>(acc, smem + (wm * 2 + wn) * (32 * (2 * 128 + 16)), reinterpret_cast<float*>(Y),
                                   reinterpret_cast<const float*>(R), alpha, beta, (long)m0 + wn * (32 * NJ), Npix,
                                   n0 + wm * 64, Cout, mod);
  } else
  store_tile_transposed<2, NJ, EPI>(acc, smem + (wm * 2 + wn) * (32 * (2 * 64 + 16)), Y, R, alpha, beta,
                               (long)m0 + wn * (32 * NJ), Npix, n0 + wm * 64, Cout, mod);
}

template <int TAPS, int KC, int XL, int NJ = 2, int EPI = 0>
int launch(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int Npix, int H, int W,
           int Cin, int Cout, hipStream_t st, const ModEpilogue& mod = ModEpilogue{}) {
  constexpr int BM = 64 * NJ, BN = 128;
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + BN - 1) / BN;
  const int HALO = (TAPS == 9) ? (W + 1) : 0;
  const int xrows = BM + 2 * HALO;
  size_t lds = ((xrows * Cfg<KC>::ROWB + 15) & ~15) + 2 * BN * Cfg<KC>::ROWB;
  if (EPI == 4 && lds < (size_t)4 * 32 * (2 * 128 + 16)) lds = (size_t)4 * 32 * (2 * 128 + 16);   // the staged fp32 epilogue's area
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  auto kern = k_conv_igemm<TAPS, KC, XL, NJ, EPI>;
  EDM_MAX_LDS(kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const bf16*)X, (const bf16*)Wp, (bf16*)Y, (const bf16*)R,
                     alpha, beta, Npix, H, W, Cin, Cout, tiles_m, tiles_n, mod);
  return 0;
}

// 3x3 launches come in two epilogue flavours: plain / forward modulation, and modulation backward (mod.U set)
#define LAUNCH9(KC, XL, NJ)                                                                                      \
  (mod.mode == 1   ? launch<9, KC, XL, NJ, 1>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod)          \
   : mod.mode == 2 ? launch<9, KC, XL, NJ, 2>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod)          \
   : mod.mode == 4 ? launch<9, KC, XL, NJ, 4>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod)          \
                   : launch<9, KC, XL, NJ, 0>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod))
#define LAUNCH1(KC, XL, NJ)                                                                                      \
  (mod.mode == 4 ? launch<1, KC, XL, NJ, 4>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod)            \
                 : launch<1, KC, XL, NJ, 0>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, st, mod))

}  // namespace

// X [B*H*W, Cin] bf16, Wp [taps, Cout, Cin] bf16, Y/R [B*H*W, Cout] bf16.  taps in {1, 9}.
// internal form with the optional fused modulation epilogue (Y may be null when only mod.Y2 is wanted)
int edm_conv_igemm_v1_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                         int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st) {
  EDM_REQUIRE(X && Wp && (Y || mod.Y2), "conv_igemm: null pointer");
  EDM_REQUIRE(!mod.wfrag || mod.mode == 4, "conv_igemm: fragment-major weight packs are read by k_conv3x3_s only");
  EDM_REQUIRE(mod.mode == 0 || mod.mode == 3 || mod.mode == 4 || taps == 9, "conv_igemm: the backward epilogues are 3x3 only");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "conv_igemm: bad B/H/W");
  EDM_REQUIRE(taps == 1 || taps == 9, "conv_igemm: taps must be 1 or 9 (got %d)", taps);
  EDM_REQUIRE(Cin > 0 && Cin % 32 == 0, "conv_igemm: Cin=%d must be a multiple of 32", Cin);
  EDM_REQUIRE(Cout > 0 && Cout % 8 == 0, "conv_igemm: Cout=%d must be a multiple of 8", Cout);
  EDM_REQUIRE(W <= 64 || taps == 1, "conv_igemm: W=%d > 64 unsupported for 3x3", W);
  const int Npix = B * H * W;
  // small feature maps (fewer than ~1.5 tiles per CU at 128 pixels): halve the pixel tile to fill the chip
  const long tiles128 = (long)((Npix + 127) / 128) * ((Cout + 127) / 128);
  if (Cin % 64 == 0 && tiles128 < 384 && W <= 30) {
    if (taps == 1) LAUNCH1(64, 2, 1);
    else LAUNCH9(64, 4, 1);   // (64 + 2*(W+1)) * 8 <= 1024 chunks
    EDM_CHECK_LAUNCH("conv_igemm");
    return EDM_OK;
  }
  const int xrows = 128 + (taps == 9 ? 2 * (W + 1) : 0);
  if (Cin % 64 == 0) {
    const int need = (xrows * 8 + 255) / 256;
    if (taps == 1) LAUNCH1(64, 4, 2);
    else if (need <= 7) LAUNCH9(64, 7, 2);
    else LAUNCH9(64, 9, 2);
  } else {
    const int need = (xrows * 4 + 255) / 256;
    if (taps == 1) LAUNCH1(32, 2, 2);
    else if (need <= 4) LAUNCH9(32, 4, 2);
    else LAUNCH9(32, 5, 2);
  }
  EDM_CHECK_LAUNCH("conv_igemm");
  return EDM_OK;
}

extern "C" int edm_conv_igemm(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B,
                              int H, int W, int Cin, int Cout, int taps, hipStream_t st) {
  EDM_REQUIRE(Y, "conv_igemm: null pointer");
  return edm_conv_igemm_v1_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, ModEpilogue{}, st);
}

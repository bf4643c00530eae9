// Weight-gradient of the 3x3 / 1x1 convolution on bf16 MFMA (gfx950):
//
//   dWp[tap, co, ci] = sum_p dY[p, co] * X[p + off(tap), ci]          (fp32, split-K slabs)
//
// This is the autograd wgrad of the reference's F.conv2d call (networks.py:37).  The reduction
// runs over pixels, which are the *rows* of both NHWC operands, so both MFMA operands are read
// from LDS with the CDNA4 transposing read ds_read_b64_tr_b16.
//
// Design ("zero-padded flat K"):
//  * the reduction index is a padded flat pixel index kp = R*PW + w with PW = W+1 and one shared
//    zero row between images (PH = H+1): every 3x3 neighbour is then a CONSTANT shift
//    (kh-1)*PW + (kw-1) in kp, and border taps hit zero pads -- no masks in the MFMA loop.
//    The pads exist only in the LDS image (decoded while staging); HBM stays plain NHWC.
//  * a workgroup (4 waves) owns a 64(co) x 64(ci) tile for ALL taps and one K-split; each wave
//    owns a 32x32 (co,ci) block and keeps 9 accumulators (one per tap, 144 registers): the dY
//    fragment is read once per k-step and reused by the 9 taps, X fragments are shifted reads
//    of one slab.  Two workgroups fit a CU, so one's barrier hides under the other's MFMAs.
//  * LDS images are [rows][32 ch] (64-B rows) so each transposing read touches 256 contiguous
//    bytes per half-wave: bank-conflict free without padding.
//  * staging is register-prefetched one stage ahead (issue early / write late), 2-deep LDS ring,
//    one barrier per 64-row stage (36 MFMAs per wave per barrier).
//  * partial sums leave as plain coalesced fp32 stores into per-split slabs (no float atomics:
//    the chip-wide atomic rate would bind); edm_wgrad_finish reduces them deterministically.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef short4v __attribute__((address_space(3))) * lds_s4p;

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
  short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p0));
  short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p1));
  short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}

constexpr int KP = 64;  // padded pixels per LDS stage

struct RowState {  // decode state of one staged row (padded flat index -> pixel)
  int w, h, n;
};

template <int TAPS, int XL>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad(const bf16* __restrict__ X, const bf16* __restrict__ dY,
                                                         float* __restrict__ slabs, int B, int H, int W, int Cin,
                                                         int Cout, int tiles_ci, long L, long kbeg0, long kend) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PW = (TAPS == 9) ? W + 1 : W;
  const int PH = (TAPS == 9) ? H + 1 : H;
  const int HALOX = (TAPS == 9) ? PW + 1 : 0;
  const int xrows = KP + 2 * HALOX;
  const int DYS_BYTES = 2 * KP * 64;        // 2 sub-images of [KP][32ch]
  const int XS_BYTES = 2 * xrows * 64;      // 2 sub-images of [xrows][32ch]
  char* dYs = smem;                         // [2][DYS_BYTES]
  char* Xs = smem + 2 * DYS_BYTES;          // [2][XS_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tco = blockIdx.x / tiles_ci, tci = blockIdx.x % tiles_ci;
  const int s = blockIdx.y;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int cb = wave & 1, ib = wave >> 1;  // co block, ci block (32 wide each)
  const long ks0 = kbeg0 + (long)s * L;
  const long ks1 = (ks0 + L < kend) ? ks0 + L : kend;
  const int nstages = (ks1 > ks0) ? (int)((ks1 - ks0 + KP - 1) / KP) : 0;

  // ---- staging assignment: dY 2 chunks/thread, X up to XL chunks/thread (16 B each)
  const int xchunks = xrows * 8;
  RowState dst[2], xst[XL];
  auto decode = [&](long kp) {
    RowState r;
    if (TAPS == 9) {
      long R = kp / PW;
      r.w = (int)(kp - R * PW);
      long r2 = R + PH - 2;
      r.n = (int)(r2 / PH) - 1;
      r.h = (int)(r2 % PH);
    } else {
      r.w = (int)kp;  // TAPS==1: w holds the flat pixel itself
      r.h = 0;
      r.n = 0;
    }
    return r;
  };
  const int adv_w = KP % PW, adv_h = KP / PW;
  auto advance = [&](RowState& r) {
    if (TAPS == 9) {
      r.w += adv_w;
      r.h += adv_h;
      if (r.w >= PW) { r.w -= PW; r.h += 1; }
      while (r.h >= PH) { r.h -= PH; r.n += 1; }
    } else {
      r.w += KP;
    }
  };
  const long Npix = (long)B * H * W;
  auto pixel_of = [&](const RowState& r, bool& valid) -> long {
    if (TAPS == 9) {
      valid = r.n >= 0 && r.n < B && r.h < H && r.w < W;
      return ((long)r.n * H + r.h) * W + r.w;
    } else {
      valid = (long)r.w < Npix;
      return r.w;
    }
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) dst[i] = decode(ks0 + ((tid + 256 * i) >> 3));
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    long kp = ks0 - HALOX + ((tid + 256 * i) >> 3);
    xst[i] = decode(kp < 0 ? 0 : kp);  // kbeg0 = PW+1 guarantees kp >= 0; clamp is belt and braces
  }

  bf16x8 dreg[2], xreg[XL];
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  auto load_stage = [&](int stage) {
    const long kbase = ks0 + (long)stage * KP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 256 * i;
      const int row = q >> 3, c16 = q & 7;
      bool valid;
      long pix = pixel_of(dst[i], valid);
      const int co = co0 + c16 * 8;
      valid = valid && (kbase + row < ks1) && co < Cout;
      dreg[i] = valid ? *reinterpret_cast<const bf16x8*>(dY + pix * Cout + co) : zero8;
      advance(dst[i]);
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int q = tid + 256 * i;
      const int c16 = q & 7;
      bool valid;
      long pix = pixel_of(xst[i], valid);
      const int ci = ci0 + c16 * 8;
      valid = valid && q < xchunks && ci < Cin;
      xreg[i] = valid ? *reinterpret_cast<const bf16x8*>(X + pix * Cin + ci) : zero8;
      advance(xst[i]);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 256 * i;
      const int row = q >> 3, c16 = q & 7;
      *reinterpret_cast<bf16x8*>(dYs + buf * DYS_BYTES + (c16 >> 2) * (KP * 64) + row * 64 + (c16 & 3) * 16) = dreg[i];
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int q = tid + 256 * i;
      const int row = q >> 3, c16 = q & 7;
      if (q < xchunks)
        *reinterpret_cast<bf16x8*>(Xs + buf * XS_BYTES + (c16 >> 2) * (xrows * 64) + row * 64 + (c16 & 3) * 16) = xreg[i];
    }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // per-lane transposing-read geometry: 16-lane group g supplies rows q=(l&15)>>2, cols 4*(l&3)
  const int krow_l = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int chan_b = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const bool wave_active = (co0 + cb * 32 < Cout) && (ci0 + ib * 32 < Cin);

  if (nstages > 0) load_stage(0);
  for (int st = 0; st < nstages; ++st) {
    const int buf = st & 1;
    store_stage(buf);
    __syncthreads();
    if (st + 1 < nstages) load_stage(st + 1);
    if (wave_active) {
      const char* abase = dYs + buf * DYS_BYTES + cb * (KP * 64) + krow_l * 64 + chan_b;
      const char* bbase = Xs + buf * XS_BYTES + ib * (xrows * 64) + (krow_l + HALOX) * 64 + chan_b;
#pragma unroll 1
      for (int ks = 0; ks < KP / 16; ++ks) {
        bf16x8 a = tr_frag(abase + ks * 16 * 64, abase + ks * 16 * 64 + 4 * 64);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int off = (TAPS == 9) ? ((t / 3 - 1) * PW + (t % 3 - 1)) : 0;
          const char* bp = bbase + (ks * 16 + off) * 64;
          bf16x8 b = tr_frag(bp, bp + 4 * 64);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // ---- store the wave's 9 (or 1) 32x32 fp32 blocks into this split's slab
  if (wave_active) {
    const int l31 = lane & 31, lhi = lane >> 5;
    const int ci = ci0 + ib * 32 + l31;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      float* base = slabs + (((long)s * TAPS + t) * Cout) * Cin;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        base[(long)co * Cin + ci] = acc[t][r];
      }
    }
  }
}

template <int TAPS, int XL>
void launch(const void* X, const void* dY, float* slabs, int B, int H, int W, int Cin, int Cout, int S, long L,
            long kbeg0, long kend, hipStream_t st) {
  const int PW = (TAPS == 9) ? W + 1 : W;
  const int HALOX = (TAPS == 9) ? PW + 1 : 0;
  const int xrows = KP + 2 * HALOX;
  const size_t lds = 2 * (2 * KP * 64) + 2 * (2 * xrows * 64);
  const int tiles_co = (Cout + 63) / 64, tiles_ci = (Cin + 63) / 64;
  auto kern = k_conv_wgrad<TAPS, XL>;
  EDM_MAX_LDS(kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(tiles_co * tiles_ci, S), dim3(256), lds, st, (const bf16*)X, (const bf16*)dY, slabs,
                     B, H, W, Cin, Cout, tiles_ci, L, kbeg0, kend);
}

}  // namespace

// How many K-splits edm_conv_wgrad will use for this shape (callers size the slab workspace
// as nsplit * taps * Cout * Cin floats).
extern "C" int edm_conv_wgrad_nsplit(int B, int H, int W, int Cin, int Cout, int taps) {
  const long PW = (taps == 9) ? W + 1 : W, PH = (taps == 9) ? H + 1 : H;
  const long ktot = (taps == 9) ? (2 + (long)B * PH) * PW - (PW + 1) : (long)B * H * W;
  const int tiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
  long S = (256 + tiles - 1) / tiles;
  const long max_s = (ktot + 4 * KP - 1) / (4 * KP);  // at least ~4 stages per split
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  if (S > 128) S = 128;
  return (int)S;
}

// X [B*H*W, Cin] bf16, dY [B*H*W, Cout] bf16 -> slabs [nsplit, taps, Cout, Cin] fp32 (fully overwritten).
extern "C" int edm_conv_wgrad(const void* X, const void* dY, float* slabs, int B, int H, int W, int Cin, int Cout,
                              int taps, int nsplit, hipStream_t st) {
  EDM_REQUIRE(X && dY && slabs, "conv_wgrad: null pointer");
  EDM_REQUIRE(taps == 1 || taps == 9, "conv_wgrad: taps must be 1 or 9");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * (H + 1) * (W + 1) < (1L << 30), "conv_wgrad: bad B/H/W");
  EDM_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 32 == 0, "conv_wgrad: Cin=%d Cout=%d must be multiples of 32", Cin, Cout);
  EDM_REQUIRE(W <= 64 || taps == 1, "conv_wgrad: W=%d > 64 unsupported for 3x3", W);
  EDM_REQUIRE(nsplit == edm_conv_wgrad_nsplit(B, H, W, Cin, Cout, taps), "conv_wgrad: nsplit mismatch");
  const long PW = (taps == 9) ? W + 1 : W, PH = (taps == 9) ? H + 1 : H;
  const long kbeg0 = (taps == 9) ? PW + 1 : 0;
  const long kend = (taps == 9) ? (2 + (long)B * PH) * PW : (long)B * H * W;
  long L = (kend - kbeg0 + nsplit - 1) / nsplit;
  L = (L + KP - 1) / KP * KP;
  if (taps == 1) {
    launch<1, 2>(X, dY, slabs, B, H, W, Cin, Cout, nsplit, L, kbeg0, kend, st);
  } else {
    const int xrows = KP + 2 * (int)(PW + 1);
    const int need = (xrows * 8 + 255) / 256;
    if (need <= 5) launch<9, 5>(X, dY, slabs, B, H, W, Cin, Cout, nsplit, L, kbeg0, kend, st);
    else launch<9, 7>(X, dY, slabs, B, H, W, Cin, Cout, nsplit, L, kbeg0, kend, st);
  }
  EDM_CHECK_LAUNCH("conv_wgrad");
  return EDM_OK;
}

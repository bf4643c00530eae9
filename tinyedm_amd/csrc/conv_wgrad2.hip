// Weight-gradient convolution, second generation: same math and slab interface as conv_wgrad.hip
// ("zero-padded flat K" + transposing LDS reads), rebuilt around the CDNA4 async global->LDS path.
//
//   dWp[tap, co, ci] = sum_p dY[p, co] * X[p + off(tap), ci]          (fp32, split-K slabs)
//
// What changed against generation 1 (0.51 PFLOP/s; 40 % of wave time parked on vmcnt/barrier, every
// 64-row stage re-staged its 2x34-row halo, staging went HBM -> VGPR -> ds_write with per-row decode):
//  * both operands are staged by `global_load_lds_dwordx4` (LDS-DMA), 16 rows x 64 B per wave-instruction,
//    straight into the [rows][32 ch] images the transposing reads want; pad rows read a zero page;
//  * X lives in a rolling LDS window (4 stage slots + 2 mirror slots so every 3x3 window is contiguous):
//    each 64-row stage loads only its 64 NEW rows (was 132): -52 % X traffic, -35 % total;
//  * dY stage t+2 and X stage t+3 are issued while stage t computes; counted `s_waitcnt vmcnt(2)` + raw
//    `s_barrier` keep the younger stage in flight across the barrier;
//  * one padded-row decode per lane per stage serves BOTH operands (dY stage t+2 and X stage t+3 are the
//    same padded rows).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef short4v __attribute__((address_space(3))) * lds_s4p;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int KP = 64;                 // padded rows per stage
constexpr int SUBB = KP * 64;          // bytes of one [64 rows][32 ch] sub-image stage
// LEADS = X stages of lead-in before the split's first dY stage = how many 64-row stages the 3x3 halo (PW+1 rows) can
// reach: 1 for W <= 62 (X ring of 4 slots + 2 mirrors, 3 dY stages), 2 for W <= 126 (ring of 8 + 4 mirrors, 4 dY
// stages in both: compute stage t then keeps X stages t .. t+4 resident while X(t+5) is in flight).
template <int LEADS>
struct Ring {
  static constexpr int XR = LEADS == 1 ? 4 : 8;       // ring slots
  static constexpr int MIRR = LEADS == 1 ? 2 : 4;     // slots mirrored behind the ring (contiguous windows at the wrap)
  static constexpr int XSLOTS = XR + MIRR;
  static constexpr int DYRING = 3;
};

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
  short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p0));
  short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p1));
  short8v c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}
__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- hand-scheduled fragment pipeline (3x3 path).  hipcc guards every ds_read_tr *builtin* that follows an LDS-DMA
// with `s_waitcnt vmcnt(0)` (it cannot prove the read does not alias the DMA's LDS destination), which drains the
// prefetched stages on every k-step.  The reads are therefore issued from inline asm (invisible to that pass) and
// ordered by hand: LDS completion by counted `lgkmcnt`, DMA completion by the stage-level vmcnt + barrier.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#define TR_RD(dst, base, imm) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(imm))
#define LGKM_WAIT(n)                                        \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");   \
  __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ bf16x8 frag_of(const u32x2_t& lo, const u32x2_t& hi) {
  u32x4 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
  return __builtin_bit_cast(bf16x8, v);
}

struct Geo {  // padded flat-K geometry (host-computed)
  int PW, PH, lead_rows, kmult;
  long kbeg0, kend;
};

template <int TAPS, int LEADS = 1>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad2(const bf16* __restrict__ X, const bf16* __restrict__ dY,
                                                          float* __restrict__ slabs, const bf16* __restrict__ zeros,
                                                          int B, int H, int W, int Cin, int Cout, int tiles_ci, long L,
                                                          Geo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XR = Ring<LEADS>::XR, MIRR = Ring<LEADS>::MIRR, XSLOTS = Ring<LEADS>::XSLOTS, DYRING = Ring<LEADS>::DYRING;
  char* dYb = smem;                          // [DYRING][2 sub][64 rows][64 B]
  char* Xb = smem + DYRING * 2 * SUBB;       // [2 sub][XSLOTS][64 rows][64 B]
  const int PW = g.PW, PH = g.PH;
  const int HALOX = (TAPS == 9) ? PW + 1 : 0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tco = blockIdx.x / tiles_ci, tci = blockIdx.x % tiles_ci;
  const int s = blockIdx.y;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int cb = wave & 1, ib = wave >> 1;  // this wave's 32x32 (co, ci) block
  const long ks0 = g.kbeg0 + (long)s * L;
  const long ks1 = (ks0 + L < g.kend) ? ks0 + L : g.kend;
  const int nst = (ks1 > ks0) ? (int)((ks1 - ks0 + KP - 1) / KP) : 0;
  const long Npix = (long)B * H * W;

  // ---- per-lane decode state of the padded row this lane stages: row = base + 16*wave + (lane>>2)
  const int drow = lane >> 2, dp = lane & 3;
  long kp = ks0 - LEADS * KP + wave * 16 + drow;  // rows of X stage 0 (LEADS stages of lead-in before the split)
  int sw, sh, sn;                         // (w, h, n) of kp  [TAPS==9]
  if (TAPS == 9) {
    const long R = kp / PW;
    sw = (int)(kp - R * PW);
    const long r2 = R - g.lead_rows + (long)g.kmult * PH;
    sn = (int)(r2 / PH) - g.kmult;
    sh = (int)(r2 % PH);
  } else {
    sw = sh = sn = 0;
  }
  const int adv_w = KP % PW, adv_h = KP / PW;
  auto advance = [&]() {
    kp += KP;
    if (TAPS == 9) {
      sw += adv_w;
      sh += adv_h;
      if (sw >= PW) { sw -= PW; sh += 1; }
      while (sh >= PH) { sh -= PH; sn += 1; }
    }
  };
  auto pixel = [&](bool& valid) -> long {
    if (TAPS == 9) {
      valid = sn >= 0 && sn < B && sh < H && sw < W;
      return ((long)sn * H + sh) * W + sw;
    } else {
      const long p = kp - KP;
      valid = p >= 0 && p < Npix;
      return p;
    }
  };
  // stage X stage j (and its mirror) / dY stage t from the CURRENT decode state
  auto issue_x = [&](int j) {
    bool valid;
    const long pix = pixel(valid);
    const int slot = j & (XR - 1);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int ci = ci0 + sub * 32;
      const bf16* src = (valid && ci < Cin) ? X + pix * Cin + ci + dp * 8 : zeros;
      dma16(src, Xb + sub * (XSLOTS * SUBB) + slot * SUBB + wave * 1024);
      if (TAPS == 9 && slot < MIRR) dma16(src, Xb + sub * (XSLOTS * SUBB) + (XR + slot) * SUBB + wave * 1024);
    }
  };
  auto issue_dy = [&](int t) {
    bool valid;
    const long pix = pixel(valid);
    valid = valid && kp < ks1;
    const int buf = t % DYRING;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int co = co0 + sub * 32;
      const bf16* src = (valid && co < Cout) ? dY + pix * Cout + co + dp * 8 : zeros;
      dma16(src, dYb + buf * (2 * SUBB) + sub * SUBB + wave * 1024);
    }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int krow_l = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int chan_b = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const bool wave_active = (co0 + cb * 32 < Cout) && (ci0 + ib * 32 < Cin);

  if (nst > 0) {
    // ---- prologue, issue order X(0) .. X(2*LEADS), dY(0), dY(1): dY stage t shares its rows with X stage t+LEADS,
    // so the decode states of X stages LEADS and LEADS+1 are kept for the two dY issues (issue ORDER fixes the waits)
    long kpa = 0, kpb = 0;
    int wa = 0, ha = 0, na = 0, wb = 0, hb = 0, nb = 0;
#pragma unroll
    for (int j = 0; j <= 2 * LEADS; ++j) {
      if (j > 0) advance();
      if (j == LEADS) { kpa = kp; wa = sw; ha = sh; na = sn; }
      if (j == LEADS + 1) { kpb = kp; wb = sw; hb = sh; nb = sn; }
      issue_x(j);
    }
    const long kpl = kp;
    const int wl = sw, hl = sh, nl = sn;
    kp = kpa; sw = wa; sh = ha; sn = na;
    issue_dy(0);
    if (nst > 1) {
      kp = kpb; sw = wb; sh = hb; sn = nb;
      issue_dy(1);
    }
    kp = kpl; sw = wl; sh = hl; sn = nl;
  }
  for (int t = 0; t < nst; ++t) {
    // ---- retire dY(t) and X(t+2*LEADS); dY(t+1) (2 DMAs per wave, issued last) may stay in flight
    if (t + 1 < nst) wait_vmcnt<2>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // ---- issue X(t+2*LEADS+1) then dY(t+2).  dY(t+2) has the rows of X(t+2+LEADS): the stage just issued when
    // LEADS == 1, the one issued a stage earlier (= the state before this advance) when LEADS == 2.
    if (t + 2 <= nst) {
      const long kpp = kp;
      const int wp = sw, hp = sh, np_ = sn;
      advance();
      issue_x(t + 2 * LEADS + 1);
      if (t + 2 < nst) {
        if (LEADS == 2) {
          const long kpn = kp;
          const int wn = sw, hn = sh, nn = sn;
          kp = kpp; sw = wp; sh = hp; sn = np_;
          issue_dy(t + 2);
          kp = kpn; sw = wn; sh = hn; sn = nn;
        } else {
          issue_dy(t + 2);
        }
      }
    }
    // ---- 36 (or 4) MFMAs over this stage
    if (TAPS == 9 && wave_active) {
      const int slot = (t + LEADS) & (XR - 1);                        // X stage t+LEADS is the centre of the window
      const int base_row = (slot < LEADS ? slot + XR : slot) * KP;    // mirrored position when the window would wrap
      const unsigned a_u = (unsigned)(uintptr_t)(lds_char*)dYb + (t % DYRING) * (2 * SUBB) + cb * SUBB + krow_l * 64 + chan_b;
      const unsigned b_u = (unsigned)(uintptr_t)(lds_char*)Xb + ib * (XSLOTS * SUBB) + (base_row + krow_l) * 64 + chan_b;
      unsigned tb[9];
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) tb[tp] = b_u + ((tp / 3 - 1) * PW + (tp % 3 - 1)) * 64;
      u32x2_t A[2][2], Bf[2][3][2];
      // item q = (k-step ks = q/3, tap group g = q%3) : 3 MFMAs; its reads are issued one item ahead
#define RD_A(ks) TR_RD(A[(ks) & 1][0], a_u, (ks) * 1024); TR_RD(A[(ks) & 1][1], a_u, (ks) * 1024 + 256)
#define RD_B(set, ks, g)                                                                                   \
  TR_RD(Bf[set][0][0], tb[3 * (g) + 0], (ks) * 1024); TR_RD(Bf[set][0][1], tb[3 * (g) + 0], (ks) * 1024 + 256); \
  TR_RD(Bf[set][1][0], tb[3 * (g) + 1], (ks) * 1024); TR_RD(Bf[set][1][1], tb[3 * (g) + 1], (ks) * 1024 + 256); \
  TR_RD(Bf[set][2][0], tb[3 * (g) + 2], (ks) * 1024); TR_RD(Bf[set][2][1], tb[3 * (g) + 2], (ks) * 1024 + 256)
#define MM3(set, ks, g)                                                                                           \
  acc[3 * (g) + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(A[(ks) & 1][0], A[(ks) & 1][1]),             \
                                                            frag_of(Bf[set][0][0], Bf[set][0][1]), acc[3 * (g) + 0], 0, 0, 0); \
  acc[3 * (g) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(A[(ks) & 1][0], A[(ks) & 1][1]),             \
                                                            frag_of(Bf[set][1][0], Bf[set][1][1]), acc[3 * (g) + 1], 0, 0, 0); \
  acc[3 * (g) + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(A[(ks) & 1][0], A[(ks) & 1][1]),             \
                                                            frag_of(Bf[set][2][0], Bf[set][2][1]), acc[3 * (g) + 2], 0, 0, 0)
      RD_A(0); RD_B(0, 0, 0);
      RD_B(1, 0, 1); LGKM_WAIT(6); MM3(0, 0, 0);
      RD_B(0, 0, 2); LGKM_WAIT(6); MM3(1, 0, 1);
      RD_A(1); RD_B(1, 1, 0); LGKM_WAIT(8); MM3(0, 0, 2);
      RD_B(0, 1, 1); LGKM_WAIT(6); MM3(1, 1, 0);
      RD_B(1, 1, 2); LGKM_WAIT(6); MM3(0, 1, 1);
      RD_A(2); RD_B(0, 2, 0); LGKM_WAIT(8); MM3(1, 1, 2);
      RD_B(1, 2, 1); LGKM_WAIT(6); MM3(0, 2, 0);
      RD_B(0, 2, 2); LGKM_WAIT(6); MM3(1, 2, 1);
      RD_A(3); RD_B(1, 3, 0); LGKM_WAIT(8); MM3(0, 2, 2);
      RD_B(0, 3, 1); LGKM_WAIT(6); MM3(1, 3, 0);
      RD_B(1, 3, 2); LGKM_WAIT(6); MM3(0, 3, 1);
      LGKM_WAIT(0); MM3(1, 3, 2);
#undef RD_A
#undef RD_B
#undef MM3
    }
    if (TAPS != 9 && wave_active) {
      const int slot = (t + LEADS) & (XR - 1);
      const int base_row = slot * KP;
      const char* abase = dYb + (t % DYRING) * (2 * SUBB) + cb * SUBB + krow_l * 64 + chan_b;
      const char* bbase = Xb + ib * (XSLOTS * SUBB) + (base_row + krow_l) * 64 + chan_b;
#pragma unroll 1
      for (int ks = 0; ks < KP / 16; ++ks) {
        bf16x8 a = tr_frag(abase + ks * 16 * 64, abase + ks * 16 * 64 + 4 * 64);
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          const int off = (TAPS == 9) ? ((tp / 3 - 1) * PW + (tp % 3 - 1)) : 0;
          const char* bp = bbase + (ks * 16 + off) * 64;
          bf16x8 b = tr_frag(bp, bp + 4 * 64);
          acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[tp], 0, 0, 0);
        }
      }
    }
  }

  // ---- store the wave's 9 (or 1) 32x32 fp32 blocks into this split's slab
  if (wave_active) {
    const int l31 = lane & 31, lhi = lane >> 5;
    const int ci = ci0 + ib * 32 + l31;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
      float* base = slabs + (((long)s * TAPS + tp) * Cout) * Cin;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        base[(long)co * Cin + ci] = acc[tp][r];
      }
    }
  }
  (void)HALOX;
}

Geo make_geo(int B, int H, int W, int taps, int leads) {
  Geo g;
  if (taps == 9) {
    g.PW = W + 1;
    g.PH = H + 1;
    g.lead_rows = 2 + (leads * KP + g.PW - 1) / g.PW;
    g.kmult = (g.lead_rows + g.PH - 1) / g.PH;
    g.kbeg0 = (long)(g.lead_rows - 1) * g.PW + 1;
    g.kend = ((long)g.lead_rows + (long)B * g.PH) * g.PW;
  } else {
    g.PW = W;
    g.PH = H;
    g.lead_rows = 0;
    g.kmult = 0;
    g.kbeg0 = KP;
    g.kend = KP + (long)B * H * W;
  }
  return g;
}

}  // namespace

// How many K-splits edm_conv_wgrad_v2 uses for this shape (callers size the slab workspace as nsplit * taps * Cout * Cin floats).
// (Round 6: the first-generation kernel k_conv_wgrad -- register-staged, 64 x 64 tiles, conv_wgrad.hip -- was retired: no
// default dispatch reached it since round 2 and it had the same Cin / Cout % 32 contract as this one.)
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

// X [B*H*W, Cin] bf16, dY [B*H*W, Cout] bf16 -> slabs [nsplit, taps, Cout, Cin] fp32 (fully overwritten); returns -3 for
// shapes outside its coverage (3x3 with W > 126).
extern "C" int edm_conv_wgrad_v2(const void* X, const void* dY, float* slabs, int B, int H, int W, int Cin, int Cout,
                                 int taps, int nsplit, hipStream_t st) {
  EDM_REQUIRE(X && dY && slabs, "conv_wgrad_v2: null pointer");
  EDM_REQUIRE(taps == 1 || taps == 9, "conv_wgrad_v2: taps must be 1 or 9");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * (H + 1) * (W + 1) < (1L << 30), "conv_wgrad_v2: bad B/H/W");
  EDM_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 32 == 0, "conv_wgrad_v2: Cin, Cout must be multiples of 32");
  EDM_REQUIRE(nsplit == edm_conv_wgrad_nsplit(B, H, W, Cin, Cout, taps), "conv_wgrad_v2: nsplit mismatch");
  if (taps == 9 && W + 2 > 2 * KP) return EDM_ERR_UNSUPPORTED;
  EDM_ZERO_PAGE(zero_page_, "conv_wgrad_v2");
  (void)zero_page_;
  const int leads = (taps == 9 && W + 2 > KP) ? 2 : 1;   // 3x3 halo of PW+1 rows: within one 64-row stage, or two
  const Geo g = make_geo(B, H, W, taps, leads);
  long L = (g.kend - g.kbeg0 + nsplit - 1) / nsplit;
  L = (L + KP - 1) / KP * KP;
  const int tiles_co = (Cout + 63) / 64, tiles_ci = (Cin + 63) / 64;
  auto launch = [&](auto kern, size_t lds, std::atomic<unsigned long long>& done) {
    edm_max_lds_once(reinterpret_cast<const void*>(kern), 160 * 1024, done);
    hipLaunchKernelGGL(kern, dim3(tiles_co * tiles_ci, nsplit), dim3(256), lds, st, (const bf16*)X, (const bf16*)dY,
                       slabs, (const bf16*)edm_zero_page(), B, H, W, Cin, Cout, tiles_ci, L, g);
  };
  const size_t lds1 = (size_t)Ring<1>::DYRING * 2 * SUBB + (size_t)2 * Ring<1>::XSLOTS * SUBB;
  const size_t lds2 = (size_t)Ring<2>::DYRING * 2 * SUBB + (size_t)2 * Ring<2>::XSLOTS * SUBB;
  static std::atomic<unsigned long long> set1{0}, set9{0}, set9w{0};   // (a bit per device: common.h edm_max_lds_once)
  if (taps == 1) launch(k_conv_wgrad2<1, 1>, lds1, set1);
  else if (leads == 1) launch(k_conv_wgrad2<9, 1>, lds1, set9);
  else launch(k_conv_wgrad2<9, 2>, lds2, set9w);
  EDM_CHECK_LAUNCH("conv_wgrad_v2");
  return EDM_OK;
}

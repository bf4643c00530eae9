// Weight gradients of the 3x3 convolutions, third generation: a GROUP of layers per launch.
//
//   dWp[tap, co, ci] = sum_p dY[p, co] * X[p + off(tap), ci]     (autograd wgrad of F.conv2d, networks.py:35-37)
//
// Same math as conv_wgrad2.hip ("zero-padded flat K", LDS-DMA staging, rolling X window, transposing LDS reads);
// what changed, and why (r01 profile: 43 launches x 81 us + 115 finish launches, 4.2 GB of split-K slab traffic):
//  * workgroup tile 128(co) x 64(ci) x 9 taps, ONE wave per SIMD with the whole 512-register file: a wave owns
//    64(co) x 32(ci) x 9 taps = 18 accumulator blocks, so a k-step is 2 dY + 9 X fragments for 18 MFMAs
//    (1.2 transposing reads per MFMA instead of 2.2: the LDS pipe was busier than the matrix pipe);
//  * the reduction dimension of ALL layers of the group is laid end to end ("stream-K"), and the unit that walks
//    it is a TEAM of 8 workgroups resident on ONE XCD (blockIdx % 8 is the XCD): a layer's tiles are cut into
//    groups of 8 (gco co-tiles x gci ci-tiles, member i of a team owns tile i of the group), the sequence is
//    (layer, tile group, stage), and team t takes the stages [t*q, (t+1)*q) of it.  The 8 members therefore read
//    the SAME pixel rows at the same time -- the X slice of a ci-tile is shared by the gco members above it, the dY
//    slice of a co-tile by the gci members beside it -- so the second..eighth read of a row hits that XCD's L2
//    instead of HBM (round 2, first form: every workgroup walked its own range and the kernel fetched 3.1x its
//    algorithmic bytes).  A member keeps its accumulators across consecutive stages of its tile and flushes a
//    partial tile only when the team crosses a group boundary or its range ends; a partial's position is static:
//    slot(member, team, group) = member * (teams + groups) + team + group;
//  * `k_wgrad3_finish` (one launch per group, one workgroup per weight row) sums a row's partials, applies the
//    weight-normalisation projection and accumulates into the gradient arena;
//  * layers whose tiles do not fill a team (round 6: 128 channels = 1 x 2 tiles, 192 = 2 x 3, 384 = 3 x 6 ...) split their
//    K RANGE as well: with `ksplit` = S the layer has T * S VIRTUAL tiles (v = ks * T + tile; group = v / 8, member = v % 8),
//    a group is ceil(nst / S) stages long and member m walks the stages of its K share ks -- every member of the team has
//    a tile where 6 of 8 (MNIST's 28x28 layers) or 2 of 8 (192-channel 64x64 layers) sat idle.  Members of one group no
//    longer all read the same pixel rows (different shares), which these layers' few tiles could not share much anyway;
//    each share is one more 288-KiB partial tile for the finish pass, which make_plan prices (W3_FLUSH_STAGES).
#include "common.h"
#include <stdlib.h>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

constexpr int KP = 64;                 // padded pixel rows per stage
constexpr int SUBB = KP * 64;          // bytes of one [64 rows][32 ch] sub-image
constexpr int TCO = 128, TCI = 64;     // workgroup tile
constexpr int TILE_FLOATS = 9 * TCO * TCI;
constexpr int MAXL = 48;               // layers per group (the launch table lives in device memory: common.h EDM_UPLOAD_TABLE)
constexpr int DYRING = 3;

template <int LEADS>
struct Ring {  // as conv_wgrad2.hip: LEADS = 64-row stages the 3x3 halo (W + 2 rows) can reach
  static constexpr int XR = LEADS == 1 ? 4 : 8;
  static constexpr int MIRR = LEADS == 1 ? 2 : 4;
  static constexpr int XSLOTS = XR + MIRR;
  static constexpr size_t LDS = (size_t)DYRING * 4 * SUBB + (size_t)2 * XSLOTS * SUBB;
};

struct W3Layer {
  const bf16* X;
  const bf16* dY;
  int B, H, W, Cin, Cout;
  int PW, PH, lead_rows, kmult;
  int tiles_ci, tiles_co, nst;
  int gci, gco, nblk_ci, ngroups, group0;   // tile groups of 8 = gco x gci (see the header)
  int ksplit;                               // > 1: K shares per tile (virtual tiles; nst = stages of ONE share)
  long stage0, kbeg0, kend;
};
struct W3Group {
  W3Layer L[MAXL];
  int nlayers, nwg;
  int tpx, slots_per_member;   // teams per XCD; partial-tile slots of one member = teams + groups
  long total, q;
  float* work;
  const bf16* zeros;
};

__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#define TR_RD_(dst, base, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(imm))
// timing-only builds with ABL bit 4 keep the registers "written" without reading LDS
#define TR_RD(dst, base, imm)                                     \
  do {                                                            \
    if constexpr (ABL & 16) asm volatile("" : "=v"(dst) : "v"(base)); \
    else TR_RD_(dst, base, imm);                                  \
  } while (0)
#define LGKM_WAIT(n)                                       \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");  \
  __builtin_amdgcn_sched_barrier(0)
template <int N>
struct IC { static constexpr int value = N; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(IC<B>{});
    static_for<B + 1, E>(f);
  }
}
__device__ __forceinline__ bf16x8 frag_of(const u32x2_t& lo, const u32x2_t& hi) {
  u32x4 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
  return __builtin_bit_cast(bf16x8, v);
}

// ABL: timing-only ablation builds for tools/ (outputs wrong by construction): bit0 no MFMA, bit1 no DMA issue,
// bit2 no barrier, bit3 no fragment reads + MFMA.  ABL = 0 is the product.
// MF16: the MFMAs are v_mfma_f32_16x16x32_bf16 (the device holds a higher clock on that shape: tools/mfma_shape,
// conv_igemm6.hip): same LDS images and bytes, same DMA plan; a fragment is 16 channels x 32 pixel rows -- two
// ds_read_b64_tr_b16 of rows 8g..8g+3 and 8g+4..8g+7 per 16-lane group g -- and the wave's 64(co) x 32(ci) x 9 taps are
// 72 accumulator blocks of 4 registers (64 in AGPRs, tap 8 in VGPRs, as before).
template <int LEADS, int ABL = 0, bool MF16 = false>
__global__ __launch_bounds__(256, 1) void k_wgrad3(const W3Group* __restrict__ gp) {
  const W3Group& g = *gp;   // (device memory: uniform scalar loads, as from the kernel-argument segment)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XR = Ring<LEADS>::XR, MIRR = Ring<LEADS>::MIRR, XSLOTS = Ring<LEADS>::XSLOTS;
  char* dYb = smem;                          // [DYRING][4 sub][64 rows][64 B]
  char* Xb = smem + DYRING * 4 * SUBB;       // [2 sub][XSLOTS][64 rows][64 B]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave & 1, ib = wave >> 1;   // this wave's 64(co) x 32(ci) block of the tile
  const int drow = lane >> 2, dp = lane & 3;
  const int krow_l = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int chan_b = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int l31 = lane & 31, lhi = lane >> 5;
  // 16x16x32 operands: lane 4q+p of group g = lane >> 4 supplies row 8g + q, columns 4p..4p+3 of a 16-column block
  const int krow16 = 8 * (lane >> 4) + ((lane & 15) >> 2);
  const int chan16 = (4 * (lane & 3)) * 2;
  const int l15 = lane & 15, lq = lane >> 4;

  // blockIdx -> (XCD, local index on it) -> (team, member): the 8 members of a team share an XCD (and its L2)
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int member = local & 7, team = xcd * g.tpx + (local >> 3);
  long pos = (long)team * g.q;
  const long pend = (pos + g.q < g.total) ? pos + g.q : g.total;
  int li = 0;
  {   // the layer of the team's first stage: bisection over the layers' first stages (up to 48 layers: 6 dependent scalar loads)
    int hi = g.nlayers - 1;
    while (li < hi) {
      const int mid = (li + hi + 1) >> 1;
      if (pos >= g.L[mid].stage0) li = mid;
      else hi = mid - 1;
    }
  }

  while (pos < pend) {
    // ---- one segment: `nst` consecutive stages of tile `tl` of layer `li`, starting at stage st0 of the tile
    const bf16* __restrict__ X = g.L[li].X;
    const bf16* __restrict__ dY = g.L[li].dY;
    const int B = g.L[li].B, H = g.L[li].H, W = g.L[li].W, Cin = g.L[li].Cin, Cout = g.L[li].Cout;
    const int PW = g.L[li].PW, PH = g.L[li].PH;
    const long rel = pos - g.L[li].stage0;
    const int grp = (int)(rel / g.L[li].nst);
    const int st0 = (int)(rel - (long)grp * g.L[li].nst);
    int nst = g.L[li].nst - st0;
    if ((long)nst > pend - pos) nst = (int)(pend - pos);
    int tco, tci, ksh = 0;
    if (g.L[li].ksplit > 1) {   // virtual tile v = K share * tiles + tile (see the header); all of this is workgroup-uniform
      const int T = g.L[li].tiles_co * g.L[li].tiles_ci, v = grp * 8 + member;
      ksh = v / T;
      const int tl = v - ksh * T;
      tco = ksh < g.L[li].ksplit ? tl / g.L[li].tiles_ci : g.L[li].tiles_co;
      tci = tl % g.L[li].tiles_ci;
    } else {
      const int bco = grp / g.L[li].nblk_ci, bci = grp - bco * g.L[li].nblk_ci;
      tco = bco * g.L[li].gco + member / g.L[li].gci;
      tci = bci * g.L[li].gci + member % g.L[li].gci;
    }
    if (tco >= g.L[li].tiles_co || tci >= g.L[li].tiles_ci) {   // this member has no tile in the group (workgroup-uniform)
      pos += nst;
      if (pos >= g.L[li].stage0 + (long)g.L[li].ngroups * g.L[li].nst) ++li;
      continue;
    }
    const int co0 = tco * TCO, ci0 = tci * TCI;
    const long ks0 = g.L[li].kbeg0 + ((long)ksh * g.L[li].nst + st0) * KP;   // (a share's last stages may lie past kend: zeros)
    const long ks1 = (ks0 + (long)nst * KP < g.L[li].kend) ? ks0 + (long)nst * KP : g.L[li].kend;
    float* __restrict__ out = g.work + ((long)member * g.slots_per_member + team + g.L[li].group0 + grp) * TILE_FLOATS;

    // per-lane decode state of the padded row this lane stages: row = base + 16*wave + (lane>>2)
    long kp = ks0 - LEADS * KP + wave * 16 + drow;
    int sw, sh, sn;
    {
      const long R = kp / PW;
      sw = (int)(kp - R * PW);
      const long r2 = R - g.L[li].lead_rows + (long)g.L[li].kmult * PH;
      sn = (int)(r2 / PH) - g.L[li].kmult;
      sh = (int)(r2 % PH);
    }
    const int adv_w = KP % PW, adv_h = KP / PW;
    auto advance = [&]() {
      kp += KP;
      sw += adv_w;
      sh += adv_h;
      if (sw >= PW) { sw -= PW; sh += 1; }
      while (sh >= PH) { sh -= PH; sn += 1; }
    };
    auto pixel = [&](bool& valid) -> long {
      valid = sn >= 0 && sn < B && sh < H && sw < W;
      return ((long)sn * H + sh) * W + sw;
    };
    auto issue_x = [&](int j) {
      bool valid;
      const long pix = pixel(valid);
      const int slot = j & (XR - 1);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int ci = ci0 + sub * 32;
        const bf16* src = (valid && ci < Cin) ? X + pix * Cin + ci + dp * 8 : g.zeros;
        dma16(src, Xb + sub * (XSLOTS * SUBB) + slot * SUBB + wave * 1024);
        if (slot < MIRR) dma16(src, Xb + sub * (XSLOTS * SUBB) + (XR + slot) * SUBB + wave * 1024);
      }
    };
    auto issue_dy = [&](int t) {
      bool valid;
      const long pix = pixel(valid);
      valid = valid && kp < ks1;
      const int buf = t % DYRING;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int co = co0 + sub * 32;
        const bf16* src = (valid && co < Cout) ? dY + pix * Cout + co + dp * 8 : g.zeros;
        dma16(src, dYb + buf * (4 * SUBB) + sub * SUBB + wave * 1024);
      }
    };

    f32x16 acc[MF16 ? 1 : 2][MF16 ? 1 : 8], accv[MF16 ? 1 : 2];   // taps 0..7 (AGPRs) and tap 8 (VGPRs, see MMV)
    f32x4 acc16[MF16 ? 4 : 1][2][MF16 ? 8 : 1], accv16[MF16 ? 4 : 1][2];   // [co block of 16][ci block of 16][tap]
    if constexpr (MF16) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[c][b][t][r] = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) accv16[c][b][r] = 0.f;
        }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[i][r] = 0.f;
      }
    }

    {
      // prologue, issue order X(0) .. X(2*LEADS), dY(0), dY(1): dY stage t shares its rows with X stage t+LEADS
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
      // retire dY(t) and X(t+2*LEADS); dY(t+1) (4 DMAs per wave, issued last) may stay in flight
      if (t + 1 < nst) wait_vmcnt<4>();
      else wait_vmcnt<0>();
      if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
      // ---- sources of this stage's DMAs: X(t+2*LEADS+1), then dY(t+2) (dY(t+2) has the rows of X(t+2+LEADS): the
      // stage decoded here when LEADS == 1, the one decoded a stage earlier when LEADS == 2).  They are ISSUED further
      // down, spread between the MFMAs (an LDS-DMA instruction costs the issuing wave 60-180 cycles: issued in a block
      // in front of the MFMAs they took 20 % of the kernel).
      const bool do_x = t + 2 <= nst && !(ABL & 2), do_dy = t + 2 < nst && !(ABL & 2);
      const bf16* xsrc0 = g.zeros;
      const bf16* xsrc1 = g.zeros;
      const bf16* dsrc[4] = {g.zeros, g.zeros, g.zeros, g.zeros};
      const int xslot = (t + 2 * LEADS + 1) & (XR - 1);
      if (do_x) {
        const long kpp = kp;
        const int wp = sw, hp = sh, np_ = sn;
        advance();
        bool valid;
        long pix = pixel(valid);
        if (valid && ci0 < Cin) xsrc0 = X + pix * Cin + ci0 + dp * 8;
        if (valid && ci0 + 32 < Cin) xsrc1 = X + pix * Cin + ci0 + 32 + dp * 8;
        if (do_dy) {
          long kq = kp;
          if (LEADS == 2) {
            // decode state of the previous stage
            const long kpn = kp;
            const int wn = sw, hn = sh, nn = sn;
            kp = kpp; sw = wp; sh = hp; sn = np_;
            pix = pixel(valid);
            kq = kp;
            kp = kpn; sw = wn; sh = hn; sn = nn;
          }
          valid = valid && kq < ks1;
#pragma unroll
          for (int sub = 0; sub < 4; ++sub)
            if (valid && co0 + sub * 32 < Cout) dsrc[sub] = dY + pix * Cout + co0 + sub * 32 + dp * 8;
        }
      }
      char* const xdst = Xb + xslot * SUBB + wave * 1024;
      char* const ddst = dYb + ((t + 2) % DYRING) * (4 * SUBB) + wave * 1024;
      if constexpr (MF16) {
        const int slot = (t + LEADS) & (XR - 1);
        const int base_row = (slot < LEADS ? slot + XR : slot) * KP;
        const unsigned a_u = (unsigned)(uintptr_t)(lds_char*)dYb + (t % DYRING) * (4 * SUBB) + (cb * 2) * SUBB + krow16 * 64 + chan16;
        const unsigned b_u = (unsigned)(uintptr_t)(lds_char*)Xb + ib * (XSLOTS * SUBB) + (base_row + krow16) * 64 + chan16;
        unsigned tb[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) tb[tp] = b_u + ((tp / 3 - 1) * PW + (tp % 3 - 1)) * 64;
        // ---- 144 MFMAs of this stage: k-step ks2 (32 rows) x tap tp x ci block b; X fragment n = 18 ks2 + 2 tp + b feeds
        // four MFMAs (the wave's four co blocks).  One MFMA, then ONE fragment read for three X fragments ahead (behind the
        // first two MFMAs of a fragment); the four dY fragments of k-step 1 are read behind fragments 8, 10, 12, 14.
        // dY fragment c (co block c of 16): sub-image c >> 1, columns 16 (c & 1) ..; k-step: + 32 rows.
        u32x2_t A[2][4][2], Bf[6][2];
#define A_RD(set, c, ks2_)                                                                    \
  TR_RD(A[set][c][0], a_u, ((c) >> 1) * SUBB + ((c) & 1) * 32 + (ks2_) * 2048);               \
  TR_RD(A[set][c][1], a_u, ((c) >> 1) * SUBB + ((c) & 1) * 32 + (ks2_) * 2048 + 256);
        A_RD(0, 0, 0) A_RD(0, 1, 0) A_RD(0, 2, 0) A_RD(0, 3, 0)
        TR_RD(Bf[0][0], tb[0], 0); TR_RD(Bf[0][1], tb[0], 256);
        TR_RD(Bf[1][0], tb[0], 32); TR_RD(Bf[1][1], tb[0], 32 + 256);
        TR_RD(Bf[2][0], tb[1], 0); TR_RD(Bf[2][1], tb[1], 256);
        static_for<0, 36>([&](auto nc) {
          constexpr int n = decltype(nc)::value;
          constexpr int ks2 = n / 18, f = n % 18, tp = f / 2, b = f % 2;
          const unsigned au_ = a_u;
          // reads younger than fragment n: fragments n+1, n+2 and the dY reads issued in slots n-3 .. n-1 (slots 8, 10, 12, 14)
          constexpr int na = ((n - 3 <= 8 && 8 <= n - 1) ? 1 : 0) + ((n - 3 <= 10 && 10 <= n - 1) ? 1 : 0) +
                             ((n - 3 <= 12 && 12 <= n - 1) ? 1 : 0) + ((n - 3 <= 14 && 14 <= n - 1) ? 1 : 0);
          constexpr int cnt = n >= 33 ? (35 - n) * 2 : 4 + 2 * na;
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(cnt) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (n == 1) { if (do_x) { dma16(xsrc0, xdst); if (xslot < MIRR) dma16(xsrc0, xdst + XR * SUBB); } __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 5) { if (do_x) { dma16(xsrc1, xdst + XSLOTS * SUBB); if (xslot < MIRR) dma16(xsrc1, xdst + (XSLOTS + XR) * SUBB); } __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 10) { if (do_dy) dma16(dsrc[0], ddst); __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 14) { if (do_dy) dma16(dsrc[1], ddst + SUBB); __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 19) { if (do_dy) dma16(dsrc[2], ddst + 2 * SUBB); __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 23) { if (do_dy) dma16(dsrc[3], ddst + 3 * SUBB); __builtin_amdgcn_sched_barrier(0); }
          const bf16x8 bfr = frag_of(Bf[n % 6][0], Bf[n % 6][1]);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const bf16x8 afr = frag_of(A[ks2 & 1][c][0], A[ks2 & 1][c][1]);
            if constexpr (tp < 8) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[c][b][tp < 8 ? tp : 0]) : "v"(afr), "v"(bfr));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accv16[c][b]) : "v"(afr), "v"(bfr));
            if constexpr (n + 3 < 36) {
              constexpr int m = n + 3, ksm = m / 18, fm = m % 18;
              if (c < 2) TR_RD(Bf[m % 6][c < 2 ? c : 0], tb[fm / 2], ksm * 2048 + (fm % 2) * 32 + (c < 2 ? c : 0) * 256);
            }
          }
          if constexpr (ks2 == 0 && b == 0 && tp >= 4 && tp <= 7) {   // dY fragments of k-step 1: co block tp - 4
            constexpr int ca = tp - 4;
            TR_RD(A[1][ca][0], au_, (ca >> 1) * SUBB + (ca & 1) * 32 + 2048);
            TR_RD(A[1][ca][1], au_, (ca >> 1) * SUBB + (ca & 1) * 32 + 2048 + 256);
          }
        });
#undef A_RD
      } else if constexpr (!(ABL & 8)) {
        const int slot = (t + LEADS) & (XR - 1);                        // X stage t+LEADS is the centre of the window
        const int base_row = (slot < LEADS ? slot + XR : slot) * KP;    // mirrored position when the window would wrap
        unsigned a_u = (unsigned)(uintptr_t)(lds_char*)dYb + (t % DYRING) * (4 * SUBB) + (cb * 2) * SUBB + krow_l * 64 + chan_b;
        const unsigned b_u = (unsigned)(uintptr_t)(lds_char*)Xb + ib * (XSLOTS * SUBB) + (base_row + krow_l) * 64 + chan_b;
        unsigned tb[9];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) tb[tp] = b_u + ((tp / 3 - 1) * PW + (tp % 3 - 1)) * 64;
        // ---- 72 MFMAs of this stage: k-step ks (16 rows) x tap p; fragment n = 9 ks + p.  Everything is inline asm in
        // program order: one MFMA, then ONE fragment read for three fragments ahead (the wave issues in order: reads
        // placed in a block in front of a block of MFMAs do not overlap with them -- measured: reads+waits alone 281 us,
        // MFMAs alone 264 us, together 638 us), waits counted per position.  The dY fragments of k-step ks+1 are read
        // behind taps 4 and 5.  At most 10 LDS reads are outstanding (lgkmcnt is a 4-bit counter).
        u32x2_t A[2][2][2], Bf[6][2];
        TR_RD(A[0][0][0], a_u, 0); TR_RD(A[0][0][1], a_u, 256);
        TR_RD(A[0][1][0], a_u, SUBB); TR_RD(A[0][1][1], a_u, SUBB + 256);
        TR_RD(Bf[0][0], tb[0], 0); TR_RD(Bf[0][1], tb[0], 256);
        TR_RD(Bf[1][0], tb[1], 0); TR_RD(Bf[1][1], tb[1], 256);
        TR_RD(Bf[2][0], tb[2], 0); TR_RD(Bf[2][1], tb[2], 256);
        static_for<0, 36>([&](auto nc) {
          constexpr int n = decltype(nc)::value;
          constexpr int ks = n / 9, p = n % 9;
          const unsigned au_ = a_u;      // (named outside the `if constexpr` below so that the generic lambda captures it)
          // reads younger than fragment n at this point: fragments n+1, n+2 (if any) + the next k-step's dY reads
          constexpr int cnt = ks < 3 ? (p == 5 || p == 8 ? 6 : (p == 6 || p == 7 ? 8 : 4))
                                     : (p <= 6 ? 4 : (p == 7 ? 2 : 0));
          if constexpr (!(ABL & 32)) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(cnt) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          // DMA issue points (X first, then dY: the order the stage-level vmcnt counts on)
          if constexpr (n == 1) { if (do_x) { dma16(xsrc0, xdst); if (xslot < MIRR) dma16(xsrc0, xdst + XR * SUBB); } __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 5) { if (do_x) { dma16(xsrc1, xdst + XSLOTS * SUBB); if (xslot < MIRR) dma16(xsrc1, xdst + (XSLOTS + XR) * SUBB); } __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 10) { if (do_dy) dma16(dsrc[0], ddst); __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 14) { if (do_dy) dma16(dsrc[1], ddst + SUBB); __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 19) { if (do_dy) dma16(dsrc[2], ddst + 2 * SUBB); __builtin_amdgcn_sched_barrier(0); }
          if constexpr (n == 23) { if (do_dy) dma16(dsrc[3], ddst + 3 * SUBB); __builtin_amdgcn_sched_barrier(0); }
          const bf16x8 bfr = frag_of(Bf[n % 6][0], Bf[n % 6][1]);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const bf16x8 afr = frag_of(A[ks & 1][i][0], A[ks & 1][i][1]);
            if constexpr (!(ABL & 1)) {
              // taps 0..7 accumulate in AGPRs; tap 8 in architectural VGPRs (256 AGPRs hold 16 of the 18 blocks)
              if constexpr (p < 8) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][p < 8 ? p : 0]) : "v"(afr), "v"(bfr));
              else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[i]) : "v"(afr), "v"(bfr));
            }
            if constexpr (n + 3 < 36) {      // one read of fragment n+3 behind each MFMA
              constexpr int m = n + 3;
              TR_RD(Bf[m % 6][i], tb[m % 9], (m / 9) * 1024 + i * 256);
            }
          }
          if constexpr (ks < 3 && (p == 4 || p == 5)) {   // dY fragments of k-step ks+1: co block p-4
            constexpr int ia = p - 4;
            TR_RD(A[(ks + 1) & 1][ia][0], au_, ia * SUBB + (ks + 1) * 1024);
            TR_RD(A[(ks + 1) & 1][ia][1], au_, ia * SUBB + (ks + 1) * 1024 + 256);
          }
        });
      } else {
        if (do_x) { dma16(xsrc0, xdst); if (xslot < MIRR) dma16(xsrc0, xdst + XR * SUBB); dma16(xsrc1, xdst + XSLOTS * SUBB); if (xslot < MIRR) dma16(xsrc1, xdst + (XSLOTS + XR) * SUBB); }
        if (do_dy) { dma16(dsrc[0], ddst); dma16(dsrc[1], ddst + SUBB); dma16(dsrc[2], ddst + 2 * SUBB); dma16(dsrc[3], ddst + 3 * SUBB); }
      }
    }
    // ---- flush the partial tile: [tap][128 co][64 ci] fp32, always whole (inactive blocks hold zeros).
    // The asm MFMAs are invisible to hipcc's hazard recogniser: pad their write -> VMEM-read distance by hand.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if constexpr (MF16) {   // block (c, b): rows = co 16 c + 4 (lane >> 4) + r, columns = ci 16 b + (lane & 15)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            float* base = out + ((long)tp * TCO + cb * 64 + c * 16 + 4 * lq) * TCI + ib * 32 + b * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) base[r * TCI] = (tp < 8) ? acc16[c][b][tp < 8 ? tp : 0][r] : accv16[c][b][r];
          }
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        float* base = out + ((long)tp * TCO + cb * 64 + i * 32) * TCI + ib * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          base[((r & 3) + 8 * (r >> 2) + 4 * lhi) * TCI] = (tp < 8) ? acc[i][tp < 8 ? tp : 0][r] : accv[i][r];
      }
    }
    pos += nst;
    if (pos >= g.L[li].stage0 + (long)g.L[li].ngroups * g.L[li].nst) ++li;
    // every wave must be past its LDS reads before the next segment's prologue overwrites the rings
    __builtin_amdgcn_s_barrier();
  }
}

// ---- finish: one workgroup per packed weight row.  grad[mo, i, t] (=|+=) projection(scale * sum of the row's
// partial tiles) through w_hat = w / (d sqrt(n)) -- the arithmetic of k_wgrad_finish (weights.hip).
struct F3Layer {
  const float* w;
  float* grad;
  const int* perm;
  int O, I, Cin, tiles_ci, nst, row0;
  int gci, gco, nblk_ci, group0;
  int ksplit, tiles_co;
  float scale;
  int accumulate;
  long stage0;
};
struct F3Group {
  F3Layer L[MAXL];
  int nlayers, slots_per_member;
  long q;
  const float* work;
};

__device__ __forceinline__ float block_sum_f(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

__global__ __launch_bounds__(256) void k_wgrad3_finish(const F3Group* __restrict__ gp) {
  const F3Group& g = *gp;
  extern __shared__ __attribute__((aligned(16))) float gsm[];  // [n] gradient row in master order (i*9 + t)
  __shared__ float red[8];
  int li = 0;
  {   // bisection over the layers' first rows (a walk is up to 47 dependent scalar loads at the head of a short workgroup)
    int hi = g.nlayers - 1;
    while (li < hi) {
      const int mid = (li + hi + 1) >> 1;
      if ((int)blockIdx.x >= g.L[mid].row0) li = mid;
      else hi = mid - 1;
    }
  }
  const int r = blockIdx.x - g.L[li].row0;
  const int I = g.L[li].I, tiles_ci = g.L[li].tiles_ci, nst = g.L[li].nst;
  const int n = I * 9;
  const int mo = g.L[li].perm ? g.L[li].perm[r] : r;
  const float* __restrict__ row = g.L[li].w + (long)mo * n;
  const int tco = r / TCO, rl = r - tco * TCO;
  const float scale = g.L[li].scale;
  // the row of the master weight (projection below): loaded NOW, while the partial tiles are in flight -- the kernel is a
  // chain of memory round trips per workgroup (partials -> row -> block sums -> store) and this takes one of them out
  constexpr int WPF = 28;                       // covers n <= 7168 (Cin <= 796) with 256 threads; longer rows load late
  const bool pf = n <= WPF * 256;
  float wpre[WPF];
  if (pf) {
#pragma unroll
    for (int k = 0; k < WPF; ++k) {
      const int e = threadIdx.x + k * 256;
      wpre[k] = e < n ? row[e] : 0.f;
    }
  }
  // work items: (ci tile, tap, float4 of the 64 ci) -> 16 per (tile, tap)
  const int items = tiles_ci * 9 * 16;
  for (int idx = threadIdx.x; idx < items; idx += blockDim.x) {
    const int c4 = idx & 15, tt = idx >> 4;
    const int tp = tt % 9, tci = tt / 9;
    // the tile's group and owner, then the teams whose stage ranges touch the group: one partial each -- and that for
    // every K share of the tile when the layer's K range is split (fixed order: share, then team)
    const int gco = g.L[li].gco, gci = g.L[li].gci, ksplit = g.L[li].ksplit;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int ksh = 0; ksh < ksplit; ++ksh) {
      int grp, member;
      if (ksplit > 1) {
        const int v = ksh * (g.L[li].tiles_co * tiles_ci) + tco * tiles_ci + tci;
        grp = v >> 3;
        member = v & 7;
      } else {
        grp = (tco / gco) * g.L[li].nblk_ci + tci / gci;
        member = (tco % gco) * gci + tci % gci;
      }
      const long s0 = g.L[li].stage0 + (long)grp * nst;
      const int w_lo = (int)(s0 / g.q), w_hi = (int)((s0 + nst - 1) / g.q);
      const float* p = g.work + ((long)member * g.slots_per_member + w_lo + g.L[li].group0 + grp) * TILE_FLOATS +
                       ((long)tp * TCO + rl) * TCI + c4 * 4;
      for (int w = w_lo; w <= w_hi; ++w) {
        a += *reinterpret_cast<const f32x4*>(p);
        p += TILE_FLOATS;
      }
    }
    const int i0 = tci * TCI + c4 * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j < I) gsm[(i0 + j) * 9 + tp] = a[j] * scale;
  }
  __syncthreads();
  float dot = 0.f, ss = 0.f;
  if (pf) {
#pragma unroll
    for (int k = 0; k < WPF; ++k) {
      const int e = threadIdx.x + k * 256;
      if (e < n) {
        dot += gsm[e] * wpre[k];
        ss += wpre[k] * wpre[k];
      }
    }
  } else {
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
      const float wv = row[e];
      dot += gsm[e] * wv;
      ss += wv * wv;
    }
  }
  dot = block_sum_f(dot, red);
  ss = block_sum_f(ss, red);
  const float rn = sqrtf(ss);
  const float sqn = sqrtf((float)n);
  const float d = NORM_EPS + rn / sqn;
  const float c0 = 1.0f / (d * sqn);
  const float c1 = rn > 0.f ? dot / (d * rn * sqn) : 0.f;
  float* __restrict__ outp = g.L[li].grad + (long)mo * n;
  const int accumulate = g.L[li].accumulate;
  if (pf) {
#pragma unroll
    for (int k = 0; k < WPF; ++k) {
      const int e = threadIdx.x + k * 256;
      if (e < n) {
        const float v = c0 * (gsm[e] - wpre[k] * c1);
        outp[e] = accumulate ? outp[e] + v : v;
      }
    }
  } else {
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
      const float v = c0 * (gsm[e] - row[e] * c1);
      outp[e] = accumulate ? outp[e] + v : v;
    }
  }
}

struct Plan {
  W3Group wg;
  F3Group fg;
  int leads, rows_total, ngroups_total, max_n;
  long work_floats;
};

}  // namespace

// One layer of a group (plain C struct, mirrored by tinyedm_amd/ops.py).
extern "C" {
typedef struct {
  const void* X;       // bf16 [B*H*W][Cin]   the layer's input
  const void* dY;      // bf16 [B*H*W][Cout]  gradient of its output
  const float* w;      // fp32 master weight [Cout][I][3][3]
  float* grad;         // fp32 gradient, same layout
  const int* perm;     // packed row -> master row (NULL = identity)
  int B, H, W, Cin, Cout, I;   // I <= Cin: real input channels of the master weight (Cin may be zero-padded)
  float scale;
  int accumulate;      // 0: grad = ..., 1: grad += ...
} edm_wgrad3_item;
}


namespace {

int make_plan(const edm_wgrad3_item* it, int n, Plan& P) {
  EDM_REQUIRE(it && n > 0 && n <= MAXL, "wgrad3: need 1..%d layers per group, got %d", MAXL, n);
  P.leads = 0;
  long stage = 0;
  int group = 0, row = 0;
  P.max_n = 0;
  for (int k = 0; k < n; ++k) {
    const edm_wgrad3_item& a = it[k];
    EDM_REQUIRE(a.X && a.dY && a.w && a.grad, "wgrad3: null pointer in layer %d", k);
    EDM_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && (long)a.B * (a.H + 1) * (a.W + 1) < (1L << 30), "wgrad3: bad B/H/W");
    EDM_REQUIRE(a.Cin > 0 && a.Cin % 32 == 0 && a.Cout > 0 && a.Cout % 32 == 0 && a.I > 0 && a.I <= a.Cin,
                "wgrad3: Cin=%d Cout=%d must be multiples of 32 (I=%d <= Cin)", a.Cin, a.Cout, a.I);
    EDM_REQUIRE(a.W + 2 <= 2 * KP, "wgrad3: W=%d > %d unsupported", a.W, 2 * KP - 2);
    EDM_REQUIRE((long)a.I * 9 * 4 <= 64 * 1024, "wgrad3: fan_in %d too large for the LDS row buffer", a.I * 9);
    const int leads = (a.W + 2 > KP) ? 2 : 1;
    EDM_REQUIRE(P.leads == 0 || P.leads == leads, "wgrad3: layers of one group must share the halo class (W <= 62 or W > 62)");
    P.leads = leads;
    W3Layer& L = P.wg.L[k];
    L.X = (const bf16*)a.X;
    L.dY = (const bf16*)a.dY;
    L.B = a.B; L.H = a.H; L.W = a.W; L.Cin = a.Cin; L.Cout = a.Cout;
    L.PW = a.W + 1;
    L.PH = a.H + 1;
    L.lead_rows = 2 + (leads * KP + L.PW - 1) / L.PW;
    L.kmult = (L.lead_rows + L.PH - 1) / L.PH;
    L.kbeg0 = (long)(L.lead_rows - 1) * L.PW + 1;
    L.kend = ((long)L.lead_rows + (long)a.B * L.PH) * L.PW;
    L.tiles_ci = (a.Cin + TCI - 1) / TCI;
    L.tiles_co = (a.Cout + TCO - 1) / TCO;
    L.gco = L.tiles_co >= 8 ? 8 : L.tiles_co >= 4 ? 4 : L.tiles_co >= 2 ? 2 : 1;   // members along co: each X slice is shared by gco
    {
      // ... unless another shape of the team of eight leaves fewer members without a tile (round 6): 3 x 6 tiles (384
      // channels) fill 18 of 32 slots as 2 x 4 teams and 18 of 24 as 1 x 8; 5 x 9 (576 channels) 45 of 80 as 4 x 2 and 45 of 72
      // as 2 x 4.  The rule above stays the tie-break (256 channels: 2 x 4 tiles, full either way -- plans unchanged).
      auto slots = [&](int gco) {
        const int gci = 8 / gco;
        return ((L.tiles_co + gco - 1) / gco) * ((L.tiles_ci + gci - 1) / gci);
      };
      static const bool search = [] { const char* e = getenv("EDM_W3_TEAMS"); return !(e && e[0] == '0'); }();   // tools only (A/B)
      int best = L.gco;
      for (int gco = 1; search && gco <= 8; gco *= 2)
        if (slots(gco) < slots(best)) best = gco;
      L.gco = best;
    }
    L.gci = 8 / L.gco;                                                             // ... and each dY slice by gci members
    L.nblk_ci = (L.tiles_ci + L.gci - 1) / L.gci;
    L.ngroups = ((L.tiles_co + L.gco - 1) / L.gco) * L.nblk_ci;
    L.group0 = group;
    L.nst = (int)((L.kend - L.kbeg0 + KP - 1) / KP);
    L.ksplit = 1;
    {
      // K shares (see the header): the team walks ceil(T * S / 8) groups of ceil(nst / S) stages instead of `ngroups` groups of
      // nst; every group costs each member the flush of a 288-KiB partial tile (and the finish pass its read), priced at
      // W3_FLUSH_STAGES stages (a stage is ~2 us of a CU, 256 partial tiles at once ~25 us of HBM writes + their read back).
      // 256-channel layers (2 x 4 or 2 x 8 tiles: no idle member) never split; of the CIFAR-10 net only conv_in does (2 x 1 tiles).
      constexpr int W3_FLUSH_STAGES = 16;
      static const int smax = [] { const char* e = getenv("EDM_W3_KSPLIT"); return e ? atoi(e) : 8; }();   // 0 / 1: off (tools, A/B)
      const int T = L.tiles_co * L.tiles_ci;
      long best = (long)L.ngroups * (L.nst + W3_FLUSH_STAGES);
      for (int sp = 2; sp <= smax && sp <= 8; ++sp) {
        const int nst_s = (L.nst + sp - 1) / sp;
        if (nst_s < W3_FLUSH_STAGES) break;   // a share shorter than its own flush: more partial tiles for nothing
        const long cost = (long)((T * sp + 7) / 8) * (nst_s + W3_FLUSH_STAGES);
        if (cost < best) { best = cost; L.ksplit = sp; }
      }
      if (L.ksplit > 1) {
        L.ngroups = (T * L.ksplit + 7) / 8;
        L.nst = (L.nst + L.ksplit - 1) / L.ksplit;
      }
    }
    L.stage0 = stage;
    F3Layer& F = P.fg.L[k];
    F.w = a.w; F.grad = a.grad; F.perm = a.perm;
    F.O = a.Cout; F.I = a.I; F.Cin = a.Cin; F.tiles_ci = L.tiles_ci; F.nst = L.nst; F.row0 = row;
    F.gci = L.gci; F.gco = L.gco; F.nblk_ci = L.nblk_ci; F.group0 = group;
    F.ksplit = L.ksplit; F.tiles_co = L.tiles_co;
    F.scale = a.scale; F.accumulate = a.accumulate; F.stage0 = stage;
    stage += (long)L.ngroups * L.nst;
    group += L.ngroups;
    row += a.Cout;
    if (a.I * 9 > P.max_n) P.max_n = a.I * 9;
  }
  // MI355X: 8 XCDs x 32 CUs, one workgroup per CU -> up to 4 teams of 8 per XCD; at least QMIN stages per team
  constexpr int NXCD = 8, TPX_MAX = 4, QMIN = 8;
  int tpx = TPX_MAX;
  while (tpx > 1 && stage < (long)QMIN * NXCD * tpx) --tpx;
  const int teams = NXCD * tpx;
  const long q = (stage + teams - 1) / teams;
  P.wg.nlayers = P.fg.nlayers = n;
  P.wg.nwg = teams * 8;
  P.wg.tpx = tpx;
  P.wg.slots_per_member = P.fg.slots_per_member = teams + group;
  P.wg.total = stage;
  P.wg.q = P.fg.q = q;
  P.rows_total = row;
  P.ngroups_total = group;
  P.work_floats = (long)8 * (teams + group) * TILE_FLOATS;
  return EDM_OK;
}

}  // namespace

// Bytes of workspace edm_wgrad3_group needs for this group (fp32 partial tiles; fully overwritten before being read).
extern "C" long edm_wgrad3_workspace(const edm_wgrad3_item* items, int n) {
  Plan P;
  if (make_plan(items, n, P) != EDM_OK) return -1;
  return P.work_floats * 4;
}

// Diagnostics (include/tinyedm_hip_diag.h): the K shares per tile the plan of this group gives each layer (1 = not split).
extern "C" int edm_wgrad3_plan_ksplit(const edm_wgrad3_item* items, int n, int* ksplit_out) {
  Plan P;
  const int rc = make_plan(items, n, P);
  if (rc != EDM_OK) return rc;
  EDM_REQUIRE(ksplit_out, "wgrad3_plan_ksplit: null output");
  for (int k = 0; k < n; ++k) ksplit_out[k] = P.wg.L[k].ksplit;
  return EDM_OK;
}

// Weight gradients of up to 48 3x3 layers (edm_wgrad3_max_layers): one stream-K launch + one finish launch.  `items` is HOST memory (read
// during the call only); workspace is device memory of at least edm_wgrad3_workspace(items, n) bytes.
extern "C" long edm_wgrad3_table_bytes(void) { return (long)(sizeof(W3Group) + sizeof(F3Group)); }
extern "C" int edm_wgrad3_max_layers(void) { return MAXL; }

// Measurement hook (bench.py's roofline leg; include/tinyedm_hip_diag.h): the NEXT edm_wgrad3_group call of this thread
// records `ev_start` right before and `ev_end` right behind its k_wgrad3 launch -- the grouped entry point also uploads its
// launch table and runs k_wgrad3_finish, so a pair of events around the CALL times three things (round 5: 888 us per call
// against 813-831 us for the kernel in the rocprofv3 traces).  One-shot: cleared by that call.
static thread_local hipEvent_t g_probe_start = nullptr, g_probe_end = nullptr;
extern "C" int edm_wgrad3_probe(void* ev_start, void* ev_end) {
  g_probe_start = (hipEvent_t)ev_start;
  g_probe_end = (hipEvent_t)ev_end;
  return EDM_OK;
}

extern "C" int edm_wgrad3_group(const edm_wgrad3_item* items, int n, void* workspace, long workspace_bytes,
                                void* table_host, void* table_dev, int defer_upload, hipStream_t st) {
  Plan P;
  const int rc = make_plan(items, n, P);
  if (rc != EDM_OK) return rc;
  EDM_REQUIRE(workspace && workspace_bytes >= P.work_floats * 4, "wgrad3: workspace too small (%ld < %ld bytes)",
              workspace_bytes, P.work_floats * 4);
  const bf16* zeros = (const bf16*)edm_zero_page();
  EDM_REQUIRE(zeros, "wgrad3: edm_init() has not been called on this device");
  P.wg.work = (float*)workspace;
  P.wg.zeros = zeros;
  P.fg.work = (const float*)workspace;
  // the two layer tables go to device memory with ONE stream-ordered copy (common.h: EDM_UPLOAD_TABLE)
  static_assert(sizeof(W3Group) % 16 == 0, "F3Group must start aligned behind W3Group");
  {
    char stage[sizeof(W3Group) + sizeof(F3Group)];
    memcpy(stage, &P.wg, sizeof(W3Group));
    memcpy(stage + sizeof(W3Group), &P.fg, sizeof(F3Group));
    EDM_UPLOAD_TABLE(table_dev, table_host, stage, sizeof(stage), st, "wgrad3", defer_upload);
  }
  const W3Group* const wgp = (const W3Group*)table_dev;
  const F3Group* const fgp = (const F3Group*)((const char*)table_dev + sizeof(W3Group));
  static std::atomic<unsigned long long> set1{0}, set2{0};   // (a bit per device: common.h edm_max_lds_once)
  static const int abl = [] { const char* e = getenv("EDM_W3_ABLATE"); return e ? atoi(e) : 0; }();   // tools only
  if (g_probe_start) (void)hipEventRecord(g_probe_start, st);
  if (P.leads == 1 && abl) {
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(kern, dim3(P.wg.nwg), dim3(256), Ring<1>::LDS, st, wgp);
    };
    switch (abl) {
      case 1: go(k_wgrad3<1, 1>); break;
      case 2: go(k_wgrad3<1, 2>); break;
      case 4: go(k_wgrad3<1, 4>); break;
      case 6: go(k_wgrad3<1, 6>); break;
      case 8: go(k_wgrad3<1, 8>); break;
      case 10: go(k_wgrad3<1, 10>); break;
      case 16: go(k_wgrad3<1, 16>); break;
      case 34: go(k_wgrad3<1, 34>); break;
      case 38: go(k_wgrad3<1, 38>); break;
      case 18: go(k_wgrad3<1, 18>); break;
      case 22: go(k_wgrad3<1, 22>); break;
      default: go(k_wgrad3<1, 14>); break;
    }
  } else {
    // MFMA shape (see the kernel): 16x16x32 by default (32x32 layers +4 %, 16x16 +2 %, 8x8 -2 % inside a group; the
    // training step gains 0.03 ms -- the kernel waits on its LDS-DMA issue, not on the matrix pipe); EDM_W3_MFMA16=0 keeps
    // v_mfma_f32_32x32x16_bf16
    static const int mf16 = [] { const char* e = getenv("EDM_W3_MFMA16"); return e ? atoi(e) : 1; }();
    auto go = [&](auto kern, size_t lds, std::atomic<unsigned long long>& once) {
      edm_max_lds_once(reinterpret_cast<const void*>(kern), 160 * 1024, once);
      hipLaunchKernelGGL(kern, dim3(P.wg.nwg), dim3(256), lds, st, wgp);
    };
    static std::atomic<unsigned long long> set1m{0}, set2m{0};
    if (P.leads == 1) { if (mf16) go(k_wgrad3<1, 0, true>, Ring<1>::LDS, set1m); else go(k_wgrad3<1>, Ring<1>::LDS, set1); }
    else { if (mf16) go(k_wgrad3<2, 0, true>, Ring<2>::LDS, set2m); else go(k_wgrad3<2>, Ring<2>::LDS, set2); }
  }
  if (g_probe_end) (void)hipEventRecord(g_probe_end, st);
  g_probe_start = g_probe_end = nullptr;
  EDM_CHECK_LAUNCH("wgrad3");
  EDM_MAX_LDS(k_wgrad3_finish, 64 * 1024);
  hipLaunchKernelGGL(k_wgrad3_finish, dim3(P.rows_total), dim3(256), (size_t)P.max_n * 4, st, fgp);
  EDM_CHECK_LAUNCH("wgrad3_finish");
  return EDM_OK;
}

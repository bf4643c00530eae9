// Dispatch of the convolution entry points that carry an epilogue descriptor: the fused forms of the residual blocks
// (modulation forward / backward, mp_silu backward) and -- round 4 -- the strided / split output forms that make the
// decoder's torch.cat((input, skip * gate)) (networks.py:311) copy-free.  The kernels themselves live in conv_igemm*.hip;
// every one ends in common.h's store_tile_core, which is where the descriptor (ModEpilogue) is interpreted.
#include "common.h"
#include <stdlib.h>

int edm_conv_igemm_v1_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                         int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st);
int edm_conv_igemm_v2_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                         int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st);
int edm_conv_igemm_v6_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                         int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st);
// conv_igemm5.hip: small feature maps (reduction split over the waves of a workgroup)
bool edm_conv_s_worthwhile(long npix, int W, int Cin, int Cout);
int edm_conv_igemm_s_ex(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                        int W, int Cin, int Cout, int taps, const ModEpilogue& mod, hipStream_t st);

// the static-schedule kernel pays off when it can give every CU a tile: >= 512 tiles of 512x128, or >= 256 of 512x64
bool edm_conv_tall_worthwhile(long npix, int Cout) {
  const long tm = (npix + 511) / 512;
  return tm * ((Cout + 127) / 128) >= 512 || tm * ((Cout + 63) / 64) >= 256;
}

// Y = alpha * conv(X, Wp) + beta * R with an OUTPUT DESCRIPTOR (same operand contract as edm_conv_igemm otherwise):
//   ldY    row stride of Y in elements (0 = Cout): Y may be a column block of a wider [pixels][ldY] buffer;
//   Ysilu  optional: mp_silu(Y) (of the bf16-rounded result) at the same offsets of a second buffer with the same stride;
//   Yb     optional: output channels >= split go to Yb[pixel * ldYb + channel - split] instead (split % 8 == 0);
//   wfrag  the pack is fragment-major (edm_weight_prep_multi, bits 8 / 9 of a record's taps field): kernel 5 only;
//   kernel which generation runs: 1 = k_conv_igemm, 2 = k_conv_igemm2, 5 = k_conv3x3_s, 6 = k_conv3x3_v6 (the caller picks
//          per shape exactly as for the plain entry points; -3 = shape not covered by that generation).
// Used by the decoder blocks: the producer of a block's `input` writes it (and mp_silu of it) into the left half of the
// next block's concatenated operands, and the 1x1 dgrad that produces d loss / d cat writes its two halves to the
// gradient of `input` and to the raw gradient of the gated skip (reference: networks.py:306-316 and its autograd).
extern "C" int edm_conv_igemm_o(const void* X, const void* Wp, void* Y, long ldY, void* Ysilu, void* Yb, long ldYb,
                                int split, const void* R, float alpha, float beta, int B, int H, int W, int Cin, int Cout,
                                int taps, int kernel, int wfrag, hipStream_t st) {
  EDM_REQUIRE(X && Wp && Y, "conv_igemm_o: null pointer");
  EDM_REQUIRE(!Yb || (split > 0 && split < Cout && split % 8 == 0 && ldYb >= Cout - split && ldYb % 8 == 0),
              "conv_igemm_o: bad split output");
  EDM_REQUIRE(ldY == 0 || (ldY >= (Yb ? split : Cout) && ldY % 8 == 0),
              "conv_igemm_o: ldY must be 0 or a multiple of 8 >= the columns written to Y");
  EDM_REQUIRE(!(Yb && Ysilu), "conv_igemm_o: the split form has no mp_silu output");
  ModEpilogue mod{};
  mod.Y2 = (bf16*)Ysilu;
  mod.mode = Ysilu ? 3 : 0;
  mod.HW = H * W;
  mod.ldY = ldY ? ldY : (long)(Yb ? split : (Ysilu ? Cout : 0));   // (a split destination's first half is [pixels][split]
                                                                    // unless told otherwise; 0 = the plain contiguous form)
  mod.Yb = (bf16*)Yb;
  mod.ldYb = ldYb;
  mod.split = split;
  mod.wfrag = wfrag;
  EDM_REQUIRE(!wfrag || kernel == 5, "conv_igemm_o: a fragment-major pack (wfrag) is read by kernel 5 (k_conv3x3_s) only");
  switch (kernel) {
    case 1: return edm_conv_igemm_v1_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, mod, st);
    case 2: return edm_conv_igemm_v2_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, mod, st);
    case 5: return edm_conv_igemm_s_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, mod, st);
    case 6: return edm_conv_igemm_v6_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, mod, st);
    default: edm_set_error("conv_igemm_o: unknown kernel id %d", kernel); return EDM_ERR_ARG;
  }
}

// Folded skip projection (round 6; k_conv3x3_v6<..., FOLD>, common.h ModEpilogue::X2):
//   Y = alpha3 * conv3x3(X, Wp) + alpha1 * conv1x1(X2, W2p)
// X [pixels][Cin], Wp [9][Cout][Cin]; X2 [pixels][ldX2] of which the first C2 channels are read, W2p [Cout][C2] (the 1x1
// conv's forward pack); Y / ldY / Ysilu: the output descriptor of edm_conv_igemm_o.  Replaces, in a decoder block with a
// U-Net skip, conv_1x1(cat) + the second 3x3 conv with that result as its residual (networks.py:313, 325-327): one launch
// instead of two and no [pixels][Cout] round trip; the projection's result is NOT rounded to bf16 on the way (the unfused
// pair rounds it once more).  edm_conv3x3_fold_supported: 1 when the shape runs on the static-schedule kernel's 5-slot
// forms (W <= 38, Cin % 64 == 0, C2 % 64 == 0, enough tiles to fill the chip); the entry point returns
// EDM_ERR_UNSUPPORTED (-3) otherwise and the caller keeps the two launches.
extern "C" int edm_conv3x3_fold_supported(int B, int H, int W, int Cin, int Cout, int C2) {
  const long npix = (long)B * H * W;
  return B > 0 && H > 0 && W > 0 && npix < (1L << 31) && Cin > 0 && Cin % 64 == 0 && Cin * 2 + 64 <= 4096 && Cout > 0 &&
         Cout % 8 == 0 && C2 > 0 && C2 % 64 == 0 && 512 + 2 * (W + 1) <= 5 * 128 - 49 && edm_conv_tall_worthwhile(npix, Cout);
}
extern "C" int edm_conv3x3_fold(const void* X, const void* Wp, const void* X2, long ldX2, const void* W2p, int C2, void* Y,
                                long ldY, void* Ysilu, float alpha3, float alpha1, int B, int H, int W, int Cin, int Cout,
                                hipStream_t st) {
  EDM_REQUIRE(X && Wp && X2 && W2p && Y, "conv3x3_fold: null pointer");
  EDM_REQUIRE(alpha1 != 0.f && ldX2 >= C2 && ldX2 % 8 == 0 && ldX2 < (1L << 30), "conv3x3_fold: bad alpha1 / ldX2");
  EDM_REQUIRE(ldY == 0 || (ldY >= Cout && ldY % 8 == 0), "conv3x3_fold: ldY must be 0 or a multiple of 8 >= Cout");
  if (!edm_conv3x3_fold_supported(B, H, W, Cin, Cout, C2)) return EDM_ERR_UNSUPPORTED;
  ModEpilogue mod{};
  mod.Y2 = (bf16*)Ysilu;
  mod.mode = Ysilu ? 3 : 0;
  mod.HW = H * W;
  mod.ldY = ldY ? ldY : (long)(Ysilu ? Cout : 0);
  mod.X2 = (const bf16*)X2;
  mod.W2 = (const bf16*)W2p;
  mod.C2 = C2;
  mod.ldX2 = (int)ldX2;
  mod.fold_scale = alpha3 / alpha1;
  return edm_conv_igemm_v6_ex(X, Wp, Y, nullptr, alpha1, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
}

// Split-bf16 convolution of the reference-precision evaluation path (round 4; csrc/eval_f32.hip "split"): an fp32-accurate
// conv at the rate of the bf16 kernels / 3.  Xp: [pixels][2 C] bf16 = [hi | lo] pairs of the fp32 activation
// (edm_f32_to_pairs), Wp3: [taps][Cout][3 C] bf16 = [w_hi | w_lo | w_hi] of the fp32 effective weight (edm_split_pack);
// three MFMA passes hi.w_hi + hi.w_lo + lo.w_hi accumulate in fp32 (the dropped lo.w_lo term is 2^-18 of a product);
// Y = alpha * conv + beta * R, or Y = mp_silu(conv * (lin[b,:] * gain + 1)) when lin is given -- Y, R FLOATS; Ypairs
// (optional, Y may then be NULL): the same result as [pixels][2 Cout] bf16 pairs, ready to be the next conv's Xp.
// C % 32 == 0 (3x3 on the static-schedule kernel: C % 64 == 0, W <= 64; anything else on k_conv_igemm).
// EDM_F32_EPI_STAGED=0: the fp32 epilogue of k_conv3x3_v6 straight from the accumulator layout (round 4/5; A/B runs)
static int split_epi_unstaged() {     // -> ModEpilogue.wfrag of an EPI-4 launch of k_conv3x3_v6: 1 = straight from the accumulators
  const char* e = getenv("EDM_F32_EPI_STAGED");
  return (e && e[0] == '0') ? 1 : 0;
}

// edm_split_conv_o: the same with an OUTPUT DESCRIPTOR for the pairs (round 6): rows of ld_pairs elements (0 = 2 Cout) with
// the lo halves lo_off elements behind the hi halves (0 = Cout) -- the left column blocks of the next decoder block's
// concatenated operand [hi(Ci + Cs) | lo(Ci + Cs)] -- and Ysilu_pairs (optional): mp_silu of the result as pairs at the same
// offsets of a second buffer (that block's first 3x3 conv reads it).  Any of Y / Ypairs / Ysilu_pairs may be NULL, not all.
static int split_conv_impl(const void* Xp, const void* Wp3, float* Y, void* Ypairs, long ld_pairs, long lo_off,
                           void* Ysilu_pairs, const float* R, float alpha, float beta, const float* lin, long lin_stride,
                           const float* gain, int B, int H, int W, int C, int Cout, int taps, hipStream_t st);
extern "C" int edm_split_conv(const void* Xp, const void* Wp3, float* Y, void* Ypairs, const float* R, float alpha, float beta,
                              const float* lin, long lin_stride, const float* gain, int B, int H, int W, int C, int Cout,
                              int taps, hipStream_t st) {
  return split_conv_impl(Xp, Wp3, Y, Ypairs, 0, 0, nullptr, R, alpha, beta, lin, lin_stride, gain, B, H, W, C, Cout, taps, st);
}
extern "C" int edm_split_conv_o(const void* Xp, const void* Wp3, float* Y, void* Ypairs, long ld_pairs, long lo_off,
                                void* Ysilu_pairs, const float* R, float alpha, float beta, const float* lin,
                                long lin_stride, const float* gain, int B, int H, int W, int C, int Cout, int taps,
                                hipStream_t st) {
  EDM_REQUIRE((ld_pairs == 0 && lo_off == 0) || (lo_off >= Cout && ld_pairs >= lo_off + Cout && ld_pairs % 4 == 0 && lo_off % 4 == 0),
              "split_conv_o: pairs rows need lo_off >= Cout, ld_pairs >= lo_off + Cout, both multiples of 4");
  return split_conv_impl(Xp, Wp3, Y, Ypairs, ld_pairs, lo_off, Ysilu_pairs, R, alpha, beta, lin, lin_stride, gain, B, H, W, C,
                         Cout, taps, st);
}
static int split_conv_impl(const void* Xp, const void* Wp3, float* Y, void* Ypairs, long ld_pairs, long lo_off,
                           void* Ysilu_pairs, const float* R, float alpha, float beta, const float* lin, long lin_stride,
                           const float* gain, int B, int H, int W, int C, int Cout, int taps, hipStream_t st) {
  EDM_REQUIRE(Xp && Wp3 && (Y || Ypairs || Ysilu_pairs), "split_conv: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && Cout > 0 && Cout % 8 == 0 && (taps == 1 || taps == 9),
              "split_conv: bad args (C %% 32, Cout %% 8, taps 1 or 9)");
  EDM_REQUIRE(!lin || (gain && lin_stride >= Cout), "split_conv: the modulation epilogue needs gain and lin_stride >= Cout");
  ModEpilogue mod{};
  mod.mode = 4;
  mod.Y2 = (bf16*)Ypairs;
  mod.Yb = (bf16*)Ysilu_pairs;
  mod.ldY = ld_pairs;
  mod.ldYb = lo_off;
  mod.lin = lin;
  mod.gain = gain;
  mod.lin_stride = lin_stride;
  mod.HW = H * W;
  mod.ldX = 2 * C;
  mod.kwrap = C / 32;
  const int K = 3 * C;
  const long npix = (long)B * H * W;
  if (taps == 9 && C % 64 == 0 && W <= 64 && edm_conv_tall_worthwhile(npix, Cout)) {
    ModEpilogue m6 = mod;
    m6.wfrag = split_epi_unstaged();     // (A/B switch of the staged fp32 epilogue; not a weight-pack matter here)
    const int rc = edm_conv_igemm_v6_ex(Xp, Wp3, Y, R, alpha, beta, B, H, W, K, Cout, 9, m6, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  // round 6: the small-map kernel for the 8x8-class 3x3 layers that are too small for the static-schedule kernel (batch 256:
  // 82 vs 108 us per 256->256 layer; at batch 512 those layers give every CU a 512x64 tile and take the branch above).
  // The LDS-DMA 256x128 kernel for the 1x1 layers (k_conv_igemm2, EPI 4) is built and parity-tested but NOT dispatched by
  // default: in the solve it measured 5.33 vs 5.11 ms per evaluation for the 1x1 family (its fp32 epilogue with a residual
  // is slower than the register-staged kernel's: profiles/r06_split_dispatch.txt).  EDM_SPLIT_V2=1 selects it (A/B runs);
  // EDM_SPLIT_FAST=0: the round-5 dispatch.
  // (read per call: tests switch them inside one process)
  const char* const e_fast = getenv("EDM_SPLIT_FAST");
  const char* const e_v2 = getenv("EDM_SPLIT_V2");
  const bool fast = !(e_fast && e_fast[0] == '0'), v2 = e_v2 && e_v2[0] == '1';
  if (fast && taps == 9 && edm_conv_s_worthwhile(npix, W, K, Cout)) {
    const int rc = edm_conv_igemm_s_ex(Xp, Wp3, Y, R, alpha, beta, B, H, W, K, Cout, 9, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  if (v2 && taps == 1 && ((npix + 255) / 256) * ((Cout + 127) / 128) >= 512)
    return edm_conv_igemm_v2_ex(Xp, Wp3, Y, R, alpha, beta, B, H, W, K, Cout, 1, mod, st);
  ModEpilogue m1 = mod;
  m1.wfrag = split_epi_unstaged();
  return edm_conv_igemm_v1_ex(Xp, Wp3, Y, R, alpha, beta, B, H, W, K, Cout, taps, m1, st);
}

// The folded skip projection (edm_conv3x3_fold) for the split-bf16 evaluation: Y = alpha3 * conv3x3(Xp, Wp3) + alpha1 *
// conv1x1(X2p, W2p3), every product in three bf16 passes.  X2p [pixels][2 C2] pairs, W2p3 [Cout][3 C2] (edm_split_pack of the
// 1x1 conv); outputs as edm_split_conv_o.  -3 for shapes the static-schedule kernel's folded form does not cover.
extern "C" int edm_split_conv_fold(const void* Xp, const void* Wp3, const void* X2p, const void* W2p3, int C2, float* Y,
                                   void* Ypairs, long ld_pairs, long lo_off, void* Ysilu_pairs, float alpha3, float alpha1,
                                   int B, int H, int W, int C, int Cout, hipStream_t st) {
  EDM_REQUIRE(Xp && Wp3 && X2p && W2p3 && (Y || Ypairs || Ysilu_pairs), "split_conv_fold: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && C2 > 0 && C2 % 32 == 0 && Cout > 0 && Cout % 8 == 0 && alpha1 != 0.f,
              "split_conv_fold: bad args (C %% 32, C2 %% 32, Cout %% 8)");
  EDM_REQUIRE((ld_pairs == 0 && lo_off == 0) || (lo_off >= Cout && ld_pairs >= lo_off + Cout && ld_pairs % 4 == 0 && lo_off % 4 == 0),
              "split_conv_fold: pairs rows need lo_off >= Cout, ld_pairs >= lo_off + Cout, both multiples of 4");
  if (C % 64 != 0 || !edm_conv3x3_fold_supported(B, H, W, 3 * C, Cout, 3 * C2)) return EDM_ERR_UNSUPPORTED;
  ModEpilogue mod{};
  mod.mode = 4;
  mod.Y2 = (bf16*)Ypairs;
  mod.Yb = (bf16*)Ysilu_pairs;
  mod.ldY = ld_pairs;
  mod.ldYb = lo_off;
  mod.HW = H * W;
  mod.ldX = 2 * C;
  mod.kwrap = C / 32;
  mod.X2 = (const bf16*)X2p;
  mod.W2 = (const bf16*)W2p3;
  mod.C2 = 3 * C2;
  mod.ldX2 = 2 * C2;
  mod.kwrap2 = C2 / 32;
  mod.fold_scale = alpha3 / alpha1;
  mod.wfrag = split_epi_unstaged();
  return edm_conv_igemm_v6_ex(Xp, Wp3, Y, nullptr, alpha1, 0.0f, B, H, W, 3 * C, Cout, 9, mod, st);
}

// 3x3 conv with the fused embedding modulation epilogue (networks.py:253-260 / 317-324):
//   u  = conv3x3(X, Wp)                               -> Y  (bf16; may be null when the caller does not need it: eval)
//   a2 = dropout(mp_silu(u * (lin[b,:]*gain + 1)))     -> Y2 (bf16)   [same values as edm_mod_silu_drop_fwd on u]
// mark_dropped != 0: the elements of Y the dropout removed are written as NaN (bf16 | 0x7FFF) instead of u -- their value is
// never needed again, and edm_conv3x3_modbwd(u_marked = 1) / edm_mod_silu_drop_bwd then read the mask from U.
// Picks the static-schedule kernel (k_conv3x3_v6) for layers that give every CU a tall tile, the split-K small-map kernel
// (k_conv3x3_s) for 8x8-class layers and the 128x128-tile kernel otherwise.
extern "C" int edm_conv3x3_mod(const void* X, const void* Wp, void* Y, void* Y2, const float* lin, long lin_stride,
                               const float* gain, float pdrop, unsigned long long seed, unsigned sub, unsigned step,
                               int mark_dropped, int B, int H, int W, int Cin, int Cout, const void* dyn, int wfrag,
                               hipStream_t st) {
  EDM_REQUIRE(X && Wp && Y2 && lin && gain, "conv3x3_mod: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0 && lin_stride >= Cout && pdrop >= 0.f && pdrop < 1.f,
              "conv3x3_mod: bad args");
  ModEpilogue mod{lin, gain, (bf16*)Y2, lin_stride, H * W, pdrop, (uint32_t)seed, (uint32_t)(seed >> 32), sub, step,
                  nullptr, nullptr, nullptr, 0.f, 0, (const StepParams*)dyn, 0, (mark_dropped && pdrop > 0.f) ? 1 : 0};
  mod.wfrag = wfrag;
  if (wfrag)      // fragment-major pack: the caller has established that this shape runs on k_conv3x3_s
    return edm_conv_igemm_s_ex(X, Wp, Y, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
  if (edm_conv_tall_worthwhile((long)B * H * W, Cout)) {
    const int rc = edm_conv_igemm_v6_ex(X, Wp, Y, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
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
                                  int Cin, int Cout, const void* dyn, int wfrag, hipStream_t st) {
  EDM_REQUIRE(dY && Wd && U && lin && gain && GR && gm, "conv3x3_modbwd: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0 && lin_stride >= Cout && pdrop >= 0.f && pdrop < 1.f &&
                  (gm_stride == 0 || gm_stride >= Cout),
              "conv3x3_modbwd: bad args");
  if ((H * W) % 32 != 0) return EDM_ERR_UNSUPPORTED;
  ModEpilogue mod{lin, gain, (bf16*)GR, lin_stride, H * W, pdrop, (uint32_t)seed, (uint32_t)(seed >> 32), sub, step,
                  (const bf16*)U, gm, nullptr, 0.f, 1, (const StepParams*)dyn, gm_stride, (u_marked && pdrop > 0.f) ? 1 : 0};
  mod.wfrag = wfrag;
  if (wfrag) return edm_conv_igemm_s_ex(dY, Wd, nullptr, nullptr, alpha, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
  if (edm_conv_tall_worthwhile((long)B * H * W, Cout)) {
    const int rc = edm_conv_igemm_v6_ex(dY, Wd, nullptr, nullptr, alpha, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
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
                                   void* GX, int B, int H, int W, int Cin, int Cout, int wfrag, hipStream_t st) {
  EDM_REQUIRE(dY && Wd && Xpre && GX, "conv3x3_silubwd: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0, "conv3x3_silubwd: bad args");
  ModEpilogue mod{nullptr, nullptr, (bf16*)GX, 0, H * W, 0.f, 0u, 0u, 0u, 0u, (const bf16*)Xpre, nullptr, (const bf16*)ADD,
                  add_scale, 2, nullptr};
  mod.wfrag = wfrag;
  if (wfrag) return edm_conv_igemm_s_ex(dY, Wd, nullptr, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
  if (edm_conv_tall_worthwhile((long)B * H * W, Cout)) {
    const int rc = edm_conv_igemm_v6_ex(dY, Wd, nullptr, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  if (edm_conv_s_worthwhile((long)B * H * W, W, Cin, Cout)) {
    const int rc = edm_conv_igemm_s_ex(dY, Wd, nullptr, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
    if (rc != EDM_ERR_UNSUPPORTED) return rc;
  }
  return edm_conv_igemm_v1_ex(dY, Wd, nullptr, nullptr, 1.0f, 0.0f, B, H, W, Cin, Cout, 9, mod, st);
}

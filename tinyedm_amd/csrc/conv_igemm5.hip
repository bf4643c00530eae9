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
//  * each wave stages ITS OWN pixel slab (its chunk's rows, double buffered) by LDS-DMA into a private LDS region, so the
//    main loop has NO workgroup barrier: only counted vmcnt / lgkmcnt waits of the wave itself;
//  * (round 4) the WEIGHT fragments never touch LDS: a lane's 16 bytes of an A fragment (row l31, k-group lhi) are loaded
//    from the [tap][co][ci] pack straight into the register it is multiplied from, SIX steps (96 registers, 24 loads) ahead
//    -- the single wave of a SIMD owns all 512 registers, so the ring that hides the L2 latency lives there instead of in
//    LDS.  Round 2-3 staged 4-KiB weight tiles through a 4-deep LDS ring (12 KiB in flight per wave, two steps of
//    look-ahead against ~1.1 us of L2 -> LDS latency): every step stalled on its tile, 22 us per 8x8 256->256 layer of
//    which 3.8 are MFMA.  The ring in registers also removes a third of the fragment reads (4 instead of 6 per 8 MFMAs);
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
constexpr int NWI = 4;                       // loads per step: 2 channel blocks x 2 k-steps, 16 B per lane each
constexpr int P = 6;                         // steps of weight fragments in flight per wave (18 % P == 0: static ring slots)
constexpr int WAVE_LDS = 2 * XBYTES;         // 22 KiB per wave
constexpr int RED_BYTES = 4 * 4 * 2 * 64 * 16 * 4;   // the K-slice reduction buffer overlays the slabs (128 KiB)
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
// 16 bytes per lane from (uniform 64-bit base) + (per-lane 32-bit byte offset) + immediate, straight into a fragment register
#define GLD128(dst, voff, sbase, imm) \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(imm) : "memory")
template <int N>
__device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

#ifdef EDM_S_TIMELINE   // diagnostic build only (tools/s_timeline.py): per-wave timestamps of the kernel's phases
__device__ unsigned long long* g_s_timeline = nullptr;
#define S_STAMP(slot)                                                                                      \
  if (g_s_timeline && (threadIdx.x & 63) == 0)                                                             \
    g_s_timeline[((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime()
#else
#define S_STAMP(slot)
#endif

template <int EPI, bool FRAG = false>
__global__ __launch_bounds__(256, 1) void k_conv3x3_s(const bf16* __restrict__ X, const bf16* __restrict__ Wp,
                                                        bf16* __restrict__ Y, const bf16* __restrict__ R,
                                                        const char* __restrict__ zeros, float alpha, float beta,
                                                        int Npix, int H, int W, int Cin, int Cout, int tiles_m,
                                                        int tiles_n, ModEpilogue mod) {
  apply_dyn(mod);
  S_STAMP(0);
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

  // ================= prologue.  With ONE workgroup per CU nothing hides this workgroup's set-up: the loads go out
  // first (slab of round 0, then P steps of weight fragments -- that issue order fixes the counted waits), and the ~700
  // vector instructions of the fragment-address set-up (borders, swizzles) run while they are in flight (round 4: they
  // came first and the first MFMA issued 6.5 us after the wave started, tools/s_timeline.py).
  // ---- slab rows of this wave's chunk.  Chunk c (32 input channels) of round r is c = 4 r + wave: byte offset
  // (4 r + wave) * 64.
  // (branch-free and in 32-bit arithmetic: this code is on the critical path of a workgroup that has the CU to itself; its
  // first form -- 64-bit clamps and multiplies inside per-lane branches -- was 1 700 instructions, 5 us, before the first
  // load was even issued; host-checked: the activation tensor is < 4 GiB)
  // (EPI 4, the split-bf16 evaluation: X rows are [hi | lo] pairs of ldX elements and K round r reads the X channels of
  // round (r < kw4 ? r : r - kw4), kw4 = kwrap / 4 rounds -- common.h, mode 4)
  const int kw4 = (EPI == 4 && mod.kwrap) ? mod.kwrap >> 2 : (1 << 30);
  const char* xsrc[NXI];
  {
    const unsigned cbyte = (unsigned)(wave * KC * 2);
    const unsigned rowb = (unsigned)((EPI == 4 && mod.ldX) ? mod.ldX : Cin) * 2u;
    const char* const xbase = reinterpret_cast<const char*>(X);
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int row = i * 16 + drow;
      const unsigned c16 = (unsigned)((dp ^ ((row >> 2) & 3)) << 4);
      int pix = m0 - HALO + row;
      pix = pix < 0 ? 0 : (pix > Npix - 1 ? Npix - 1 : pix);    // out-of-range rows only feed masked taps
      const char* real = xbase + ((unsigned)pix * rowb + cbyte + c16);
      const char* zero = zeros + c16;                            // zero rows: the pointer walks inside the zero page
      const bool isreal = row < xrows;
      // (64-bit select written as two 32-bit selects: no branch)
      const unsigned long long rv = (unsigned long long)(uintptr_t)real, zv = (unsigned long long)(uintptr_t)zero;
      const unsigned lo = isreal ? (unsigned)rv : (unsigned)zv, hi = isreal ? (unsigned)(rv >> 32) : (unsigned)(zv >> 32);
      xsrc[i] = reinterpret_cast<const char*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
    }
  }
#pragma unroll
  for (int i = 0; i < NXI; ++i) dma16(xsrc[i], Xb + i * 1024);
  // ---- weight fragments: lane (l31, lhi) holds row co = n0 + 32 i + l31, k-group lhi (8 channels = 16 bytes) of k-step
  // ks: one 32-bit byte offset per channel block i; round, tap and k-step enter through the uniform base / the immediate
  // (host-checked: the pack is < 4 GiB)
  const long tap_stride = (long)Cout * Cin * 2;
  const char* const wbase = reinterpret_cast<const char*>(Wp);
  unsigned woff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int co = min(n0 + 32 * i + l31, Cout - 1);            // rows >= Cout feed discarded outputs
    woff[i] = (unsigned)(((long)co * Cin + wave * KC + lhi * 8) * 2);
  }
  u32x4 fa[P][2][2], fb[2][4];
  // the two loads of k-step ks of global step (round rq, tap tq) into ring slot `slot` (constants)
  // FRAG (fragment-major pack, weights.hip): the four fragments of a step are ONE contiguous 4 KiB at
  // [tap][chunk = 4 round + wave][cb = 2 tn .. 2 tn + 1][ks][lane][8]: lane offset 16 lane, immediates 0 / 1 / 2 / 3 KiB.
  // Otherwise ([tap][co][ci] pack): row co, 16 bytes at this wave's chunk: 32 rows x 32 bytes per instruction.
  const unsigned lane16 = lane * 16;
  const int ncb = Cout >> 5, nck = Cin / KC;
#define LOAD_A2(slot, ks, rq, tq)                                                            \
  {                                                                                          \
  if constexpr (FRAG) {                                                                      \
    const char* ub_ = wbase + ((((long)(tq) * nck + (rq) * 4 + wave) * ncb + 2 * tn) * 2048); \
    asm volatile("" : "+s"(ub_));                                                            \
    GLD128(fa[slot][0][ks], lane16, ub_, (ks) * 1024);                                       \
    GLD128(fa[slot][1][ks], lane16, ub_, 2048 + (ks) * 1024);                                \
  } else {                                                                                   \
    const char* ub_ = wbase + ((long)(rq) * (4 * KC * 2) + (tq) * tap_stride);               \
    asm volatile("" : "+s"(ub_));                                                            \
    GLD128(fa[slot][0][ks], woff[0], ub_, (ks) * 32);                                        \
    GLD128(fa[slot][1][ks], woff[1], ub_, (ks) * 32);                                        \
  }                                                                                          \
  }
#define LOAD_A(slot, rq, tq) { LOAD_A2(slot, 0, rq, tq) LOAD_A2(slot, 1, rq, tq) }
  static_assert(P == 6 && P < TAPS, "prologue below is written out for P = 6");
  LOAD_A(0, 0, 0) LOAD_A(1, 0, 1) LOAD_A(2, 0, 2) LOAD_A(3, 0, 3) LOAD_A(4, 0, 4) LOAD_A(5, 0, 5)   // (nsteps >= 18 > P)
  // ---- cold start: the first touch of a weight line by an XCD is an L2 miss (2-4 us), and a ring of P steps looks ahead
  // less than that -- steps 6 .. 8 of round 0 paid it a second time (tools/s_timeline.py: round 0 took 6 us, round 1 3).
  // One 4-byte load per lane touches every line of a (round, tap) tile (row co, this wave's 64 bytes of it); the tiles of
  // steps P .. 17 are touched here, into a register nobody reads, so that all of the first two rounds' misses overlap.
  // The loads are counted (they sit between A(5) and A(6) in the in-order completion queue): steps 0 .. P-1 of the first
  // pair of rounds allow NPF more in flight.
  constexpr int NPF = 2 * TAPS - P;
  // (the destination register must stay reserved until the loads have landed -- they write it when their data arrives,
  // microseconds after issue: one register, tied through every load ("+v"), kept alive up to the end of the main loop)
  unsigned pf_dummy = 0;
  {
    // (FRAG: a tile is 32 lines of one contiguous 4 KiB; else: 64 rows x this wave's 64 bytes)
    const unsigned poff = FRAG ? (unsigned)((lane & 31) * 128) : (unsigned)(((long)min(n0 + lane, Cout - 1) * Cin + wave * KC) * 2);
#define PF_BASE(u_)                                                                                              \
  (FRAG ? wbase + ((((long)((u_) % TAPS) * nck + ((u_) / TAPS) * 4 + wave) * ncb + 2 * tn) * 2048)               \
        : wbase + ((long)((u_) / TAPS) * (4 * KC * 2) + ((u_) % TAPS) * tap_stride))
#define PF_TILE(u_)                                                                                         \
    {                                                                                                       \
      const char* ub_ = PF_BASE(u_);                                                                        \
      asm volatile("" : "+s"(ub_));                                                                         \
      asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_dummy) : "v"(poff), "s"(ub_) : "memory");       \
    }
    PF_TILE(6) PF_TILE(7) PF_TILE(8) PF_TILE(9) PF_TILE(10) PF_TILE(11) PF_TILE(12) PF_TILE(13) PF_TILE(14) PF_TILE(15)
    PF_TILE(16) PF_TILE(17)
#undef PF_TILE
#undef PF_BASE
    static_assert(NPF == 12, "prefetch list above is written out for P = 6");
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- fragment addresses: 9 taps x 4 pixel blocks (border masks folded in: a masked lane reads a zero row).  Block j of
  // tap t is block 0's address + 2 KiB j (32 rows: the swizzle term (row >> 2) & 3 does not change); what depends on j is
  // only whether the lane's pixel has that neighbour.
  const unsigned xb_off = (unsigned)(uintptr_t)(lds_char*)Xb;
  unsigned bp[TAPS][4];
  {
    const bool pow2 = ((W & (W - 1)) | (H & (H - 1))) == 0;     // uniform: shifts instead of ~35-instruction divisions
    const int lw = __builtin_ctz(W);
    unsigned base_t[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int r = l31 + HALO + (t / 3 - 1) * W + (t % 3 - 1);
      base_t[t] = xb_off + r * ROWB + ((lhi ^ ((r >> 2) & 3)) << 4);
    }
    const unsigned zaddr = xb_off + (XROWS - 1) * ROWB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + j * 32 + l31;
      int w, h;
      if (pow2) {
        w = m & (W - 1);
        h = (m >> lw) & (H - 1);
      } else {
        w = m % W;
        h = (m / W) % H;
      }
      const bool up = h >= 1, down = h + 1 < H, left = w >= 1, right = w + 1 < W;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const bool okh = t / 3 == 0 ? up : (t / 3 == 2 ? down : true);
        const bool okw = t % 3 == 0 ? left : (t % 3 == 2 ? right : true);
        bp[t][j] = (okh && okw) ? base_t[t] + j * (32 * ROWB) : zaddr;
      }
    }
  }

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nrounds = Cin / (4 * KC);          // even (host-checked)

  // ================= main loop.  A step (round, tap) is 16 MFMAs: k-step 0 then k-step 1, each over the 2 x 4 accumulator
  // blocks in pixel-block order.  ONE wave per SIMD means every instruction that is not an MFMA stalls the matrix pipe
  // unless it issues in the shadow of one, so everything else is spread BETWEEN the MFMA pairs (round 4: bunched between
  // the two halves they cost 30 % of the loop): behind the pair of pixel block j its fragment register takes the NEXT
  // step's block j (rolling reuse: an MFMA has read its operands a few cycles after issue); the next round's slab DMAs
  // (60-180 cycles of issue each) go three per pair at tap 0; the weight fragments of step u + P replace those of step u
  // as soon as its k-step has been multiplied.  Per-step conditions are compile-time except "this is the last pair of
  // rounds" -- the unrolled body exists twice and a uniform branch picks one.
#define RD_B(set, j, ks, tap, xpar) LDS_RD128(fb[set][j], bp[tap][j] ^ ((ks) ? 32u : 0u), (xpar) * XBYTES)
#define MFMA_PAIR(slot, ks, set, j)                                                                                     \
  {                                                                                                                     \
    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[slot][0][ks]),                   \
                                                        __builtin_bit_cast(bf16x8, fb[set][j]), acc[0][j], 0, 0, 0);    \
    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[slot][1][ks]),                   \
                                                        __builtin_bit_cast(bf16x8, fb[set][j]), acc[1][j], 0, 0, 0);    \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
  // pipeline fill: slab 0 has landed once at most the 4 P weight loads behind it are outstanding; both k-steps of step 0
  S_STAMP(6);
  wait_vmcnt<NWI * P + NPF>();
  RD_B(0, 0, 0, 0, 0); RD_B(0, 1, 0, 0, 0); RD_B(0, 2, 0, 0, 0); RD_B(0, 3, 0, 0, 0);
  RD_B(1, 0, 1, 0, 0); RD_B(1, 1, 1, 0, 0); RD_B(1, 2, 1, 0, 0); RD_B(1, 3, 1, 0, 0);
  S_STAMP(1);

  int r2 = 0;
  {
    auto step = [&](auto uc) {
      constexpr int idx = decltype(uc)::value;
      constexpr bool LAST = idx >= 2 * TAPS;             // the last pair of rounds (nothing follows step 17)
      constexpr int v = idx % (2 * TAPS);
      constexpr int tap = v % TAPS, xpar = v / TAPS, slot = v % P;
      constexpr int tapn = (v + 1) % TAPS, xparn = ((v + 1) / TAPS) % 2;
      constexpr bool has_next = !(LAST && v == 2 * TAPS - 1);
      constexpr bool next_round = xpar == 0 || !LAST;    // a round follows this one: its slab is issued at tap 0
      constexpr bool load_next = !LAST || v + P <= 2 * TAPS - 1;
      const int round = r2 + xpar;
      // ---- retire A(u) -- and with it everything older: this round's slab.  Vector-memory loads complete in order;
      // what may still be in flight is what was issued AFTER A(u) (in step u - P, behind that step's slab DMAs):
      // A(u+1 .. u+P-1), and the next round's slab (issued at tap 0 of this round: younger than A(u) while tap <= P - 1)
      constexpr int rem = LAST ? 2 * TAPS - 1 - v : P;   // steps after this one (P stands for "at least P - 1")
      constexpr int younger = NWI * (rem < P - 1 ? rem : P - 1) + ((next_round && tap >= 1 && tap <= P - 1) ? NXI : 0);
      if constexpr (v < P) {          // the prologue's NPF prefetch loads are younger than A(0 .. P-1) of the FIRST pair of rounds
        if (r2 == 0) wait_vmcnt<younger + NPF>();
        else wait_vmcnt<younger>();
      } else {
        wait_vmcnt<younger>();
      }
      lgkm_wait<4>();                                     // k-step 0's pixel fragments (k-step 1's may be in flight)
#ifdef EDM_S_TIMELINE
      if (v == 9 && r2 == 0) { S_STAMP(5); }
#endif
      // ---- k-step 0 (written out per pixel block: nested generic lambdas push the fragment arrays into scratch)
      const long xoff = (long)(round + 1 >= kw4 ? round + 1 - kw4 : round + 1) * (4 * KC * 2);
#define SLAB_DMA(i) if constexpr (tap == 0 && next_round && (i) < NXI) dma16(xsrc[i] + xoff, Xb + (xpar ^ 1) * XBYTES + (i) * 1024);
#define KSTEP0(j)                                                   \
      MFMA_PAIR(slot, 0, 0, j);                                     \
      if constexpr (has_next) RD_B(0, j, 0, tapn, xparn);           \
      SLAB_DMA(3 * (j)) SLAB_DMA(3 * (j) + 1) SLAB_DMA(3 * (j) + 2) \
      __builtin_amdgcn_sched_barrier(0);
      KSTEP0(0) KSTEP0(1) KSTEP0(2) KSTEP0(3)
#undef KSTEP0
#undef SLAB_DMA
      if constexpr (has_next) lgkm_wait<4>(); else lgkm_wait<0>();   // k-step 1's pixel fragments
      // ---- k-step 1
#define KSTEP1(j)                                                   \
      MFMA_PAIR(slot, 1, 1, j);                                     \
      if constexpr (has_next) RD_B(1, j, 1, tapn, xparn);           \
      if constexpr (load_next && (j) == 0) LOAD_A2(slot, 0, r2 + (v + P) / TAPS, (v + P) % TAPS); /* k-step 0's registers are free */ \
      __builtin_amdgcn_sched_barrier(0);
      KSTEP1(0) KSTEP1(1) KSTEP1(2) KSTEP1(3)
#undef KSTEP1
      if constexpr (load_next) LOAD_A2(slot, 1, r2 + (v + P) / TAPS, (v + P) % TAPS);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (; r2 + 2 < nrounds; r2 += 2) static_for<0, 2 * TAPS>(step);   // (an if / else of the two bodies inside ONE loop made the
    static_for<2 * TAPS, 4 * TAPS>(step);                              //  register allocator spill 263 registers)
  }
  asm volatile("" ::"v"(pf_dummy));   // (see the prefetch loads of the prologue)
#undef RD_B
#undef MFMA_PAIR
#undef LOAD_A
#undef LOAD_A2

  S_STAMP(2);
  // ---- reduce the four K-slices through LDS in a fixed order: wave w ends up with pixel block w (32 px x 64 co)
  // layout [source wave][pixel block j][co block i][q][lane][4 floats]: a lane's 16-byte pieces are 16 bytes apart from its
  // neighbours' (round 4; [lane][16 floats] put them 64 bytes apart: a 4-way bank conflict on every ds_write / ds_read_b128)
  __builtin_amdgcn_s_barrier();                 // every wave is done with its private staging region
  float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // (every block goes through LDS, the wave's own one included: selecting "acc[i][wave]" would index the
    // accumulator array dynamically and push all 128 accumulator registers into scratch memory)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* dst = red + ((((wave * 4 + j) * 2 + i) * 256 + lane) << 2);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(dst + 256 * q) = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
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
      const float* p = red + ((((src * 4 + wave) * 2 + i) * 256 + lane) << 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + 256 * q);
        s[4 * q] += t[0]; s[4 * q + 1] += t[1]; s[4 * q + 2] += t[2]; s[4 * q + 3] += t[3];
      }
    }
    out[i][0] = s;
  }
  __syncthreads();                              // the epilogue stage overlays the reduction buffer
  S_STAMP(3);
  if constexpr (EPI == 4)
    store_tile_f32<2, 1>(out, reinterpret_cast<float*>(Y), reinterpret_cast<const float*>(R), alpha, beta,
                         (long)m0 + wave * 32, Npix, n0, Cout, mod);
  else
  store_tile_transposed<2, 1, EPI>(out, smem + wave * ESTAGE, Y, R, alpha, beta, (long)m0 + wave * 32, Npix, n0, Cout, mod);
#ifdef EDM_S_TIMELINE
  __builtin_amdgcn_s_waitcnt(0);   // stores retired
  S_STAMP(4);
#endif
}

template <int EPI, bool FRAG>
void launch5(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int Npix, int H, int W,
             int Cin, int Cout, const ModEpilogue& mod, hipStream_t st) {
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + BNW - 1) / BNW;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  auto kern = k_conv3x3_s<EPI, FRAG>;
  EDM_MAX_LDS(kern, 160 * 1024);
  constexpr size_t lds = (size_t)RED_BYTES > (size_t)4 * WAVE_LDS ? (size_t)RED_BYTES : (size_t)4 * WAVE_LDS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const bf16*)X, (const bf16*)Wp, (bf16*)Y,
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
  // split-bf16 form: the hi / lo halves of X must be whole rounds of 4 chunks (C % 128 == 0)
  if (mod.mode == 4 && (mod.wfrag || mod.kwrap % 4 != 0)) return EDM_ERR_UNSUPPORTED;
  EDM_ZERO_PAGE(zero_page_, "conv_igemm_s");
  (void)zero_page_;
  const int Npix = B * H * W;
  static_assert(4 * WAVE_LDS <= 160 * 1024 && RED_BYTES <= 160 * 1024 && 4 * ESTAGE <= RED_BYTES, "LDS budget");
  EDM_REQUIRE((long)9 * Cout * Cin * 2 < (1L << 32) && (long)B * H * W * (mod.ldX ? mod.ldX : Cin) * 2 < (1L << 32),
              "conv_igemm_s: packed weights and the input tensor must be < 4 GiB each (32-bit lane offsets)");
  EDM_REQUIRE(!mod.wfrag || Cout % 64 == 0, "conv_igemm_s: a fragment-major pack needs Cout %% 64 == 0");
#define L5(EPIV) (mod.wfrag ? launch5<EPIV, true>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st) \
                            : launch5<EPIV, false>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st))
  if (mod.mode == 1) L5(1);
  else if (mod.mode == 2) L5(2);
  else if (mod.mode == 4) launch5<4, false>(X, Wp, Y, R, alpha, beta, Npix, H, W, Cin, Cout, mod, st);
  else L5(0);
#undef L5
  EDM_CHECK_LAUNCH("conv_igemm_s");
  return EDM_OK;
}

#ifdef EDM_S_TIMELINE
extern "C" int edm_s_set_timeline(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_s_timeline), &buf, sizeof(buf)) == hipSuccess ? EDM_OK : EDM_ERR_LAUNCH;
}
#endif

extern "C" int edm_conv_igemm_s(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B,
                                int H, int W, int Cin, int Cout, int taps, hipStream_t st) {
  EDM_REQUIRE(Y, "conv_igemm_s: null pointer");
  return edm_conv_igemm_s_ex(X, Wp, Y, R, alpha, beta, B, H, W, Cin, Cout, taps, ModEpilogue{}, st);
}

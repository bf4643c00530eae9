// Weight gradient of the 1x1 convolutions (skip projections, attention qkv/out):
//
//   dW[co, ci] = sum_p dY[p, co] * X[p, ci]            (fp32, split-K slabs [S][Cout][Cin])
//
// A 1x1 layer has 9x less arithmetic per operand byte than a 3x3 one, so the 64x64-output tiles of
// conv_wgrad.hip / conv_wgrad2.hip (32 FLOP per staged byte) saturate the L2 -> LDS path at ~0.19 PFLOP/s.
// This kernel keeps a 256(co) x 128(ci) output tile per workgroup (87 FLOP per staged byte):
//  * 8 waves, each a 64x64 block (2x2 MFMA 32x32x16 accumulators);
//  * both operands arrive by LDS-DMA (`global_load_lds_dwordx4`, 16 rows x 64 B per wave instruction) into
//    [64 pixel rows][32 ch] sub-images, 12 per stage (8 dY + 4 X) = 48 KiB, 3-stage ring (144 KiB);
//  * fragments come from `ds_read_b64_tr_b16` (K-major memory -> MFMA operand layout) issued from inline asm
//    one k-step ahead of the MFMAs, retired by counted lgkmcnt; all LDS addresses are register + immediate;
//  * counted `s_waitcnt vmcnt(6)` + raw s_barrier keep the next stage in flight across the barrier.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

constexpr int KP = 64;             // pixel rows per stage
constexpr int SUBB = KP * 64;      // bytes of one [64 rows][32 ch] sub-image
constexpr int NSUB = 12;           // 8 dY + 4 X sub-images per stage
constexpr int STAGE = NSUB * SUBB; // 48 KiB
constexpr int RING = 3;
constexpr int TCO = 256, TCI = 128;

__device__ __forceinline__ void dma16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#define TR_RD(dst, base, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(imm))
#define LGKM_WAIT(n)                                       \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");  \
  __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ bf16x8 frag_of(const u32x2_t& lo, const u32x2_t& hi) {
  u32x4 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void wgrad1x1_body(const bf16* __restrict__ X, const bf16* __restrict__ dY,
                                              float* __restrict__ slabs, const bf16* __restrict__ zeros, long Npix, int Cin,
                                              int Cout, int tiles_ci, long L, int tile, int s, bool spread) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tco = tile / tiles_ci, tci = tile % tiles_ci;
  const int co0 = tco * TCO, ci0 = tci * TCI;
  const int wr = wave >> 1, wc = wave & 1;  // this wave's 64(co) x 64(ci) block
  const long k0 = (long)s * L;
  const long k1 = (k0 + L < Npix) ? k0 + L : Npix;
  const int nst = (k1 > k0) ? (int)((k1 - k0 + KP - 1) / KP) : 0;

  // ---- DMA plan: instruction j = wave + 8*i (i = 0..5) stages sub-image j/4, rows 16*(j%4) .. +15
  const int drow = lane >> 2, dp = lane & 3;
  const char* src[6];
  long row_of[6];     // pixel row (within the stage) this lane stages: tail test
  int stride[6];      // bytes per stage advance
  bool chan_ok[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int j = wave + 8 * i;
    const int sub = j >> 2, rg = j & 3;
    const int row = rg * 16 + drow;
    row_of[i] = k0 + row;
    if (sub < 8) {
      const int co = co0 + sub * 32;
      chan_ok[i] = co < Cout;
      src[i] = reinterpret_cast<const char*>(dY + (k0 + row) * Cout + co + dp * 8);
      stride[i] = KP * Cout * 2;
    } else {
      const int ci = ci0 + (sub - 8) * 32;
      chan_ok[i] = ci < Cin;
      src[i] = reinterpret_cast<const char*>(X + (k0 + row) * Cin + ci + dp * 8);
      stride[i] = KP * Cin * 2;
    }
  }
  // DMAs i0 .. i1-1 of stage t from the current pointers, then advance them.  (Round 6: inside the loop the six DMAs of a
  // stage are issued in three pairs BETWEEN the k-steps instead of in one burst behind the barrier -- a wave sits 60-180
  // cycles in each LDS-DMA instruction, and both waves of a SIMD leave the barrier together, so the burst kept the matrix
  // pipe idle for the first ~0.3 us of every stage.  Issue order per wave is unchanged: the counted vmcnt waits hold.)
  auto issue_part = [&](int t, auto i0c, auto i1c) {
    constexpr int i0 = decltype(i0c)::value, i1 = decltype(i1c)::value;
    char* dst = smem + (t % RING) * STAGE;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const int j = wave + 8 * i;
      const bool ok = chan_ok[i] && row_of[i] < k1;
      dma16(ok ? (const void*)src[i] : (const void*)zeros, dst + (j >> 2) * SUBB + (j & 3) * 1024);
      src[i] += stride[i];
      row_of[i] += KP;
    }
  };
  auto issue = [&](int t) { issue_part(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{}); };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // transposing-read lane geometry (as conv_wgrad2.hip): 16 k-rows x 32 channels per fragment, two b64 reads
  const int krow_l = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int chan_b = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem + krow_l * 64 + chan_b;
  const unsigned a_rel = lds0 + (wr * 2) * SUBB;       // dY sub-images 2wr, 2wr+1
  const unsigned b_rel = lds0 + (8 + wc * 2) * SUBB;   // X sub-images 2wc, 2wc+1

  if (nst > 0) {
    issue(0);
    if (nst > 1) issue(1);
  }
  for (int t = 0; t < nst; ++t) {
    if (t + 1 < nst) wait_vmcnt<6>();  // stage t landed; stage t+1 (6 DMAs per wave) may stay in flight
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const bool more = t + 2 < nst;  // stage t+2 goes to the ring slot of stage t-1: every wave is past its reads (barrier above)
    if (more && !spread) issue(t + 2);
    const unsigned sb = (t % RING) * STAGE;
    const unsigned au = a_rel + sb, bu = b_rel + sb;
    u32x2_t A[2][2][2], Bf[2][2][2];  // [set][block][half]
#define RD(set, ks)                                                                                   \
  TR_RD(A[set][0][0], au, (ks) * 1024);         TR_RD(A[set][0][1], au, (ks) * 1024 + 256);           \
  TR_RD(A[set][1][0], au, SUBB + (ks) * 1024);  TR_RD(A[set][1][1], au, SUBB + (ks) * 1024 + 256);    \
  TR_RD(Bf[set][0][0], bu, (ks) * 1024);        TR_RD(Bf[set][0][1], bu, (ks) * 1024 + 256);          \
  TR_RD(Bf[set][1][0], bu, SUBB + (ks) * 1024); TR_RD(Bf[set][1][1], bu, SUBB + (ks) * 1024 + 256)
#define MM(set)                                                                                                  \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(A[set][0][0], A[set][0][1]),                       \
                                                     frag_of(Bf[set][0][0], Bf[set][0][1]), acc[0][0], 0, 0, 0); \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(A[set][0][0], A[set][0][1]),                       \
                                                     frag_of(Bf[set][1][0], Bf[set][1][1]), acc[0][1], 0, 0, 0); \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(A[set][1][0], A[set][1][1]),                       \
                                                     frag_of(Bf[set][0][0], Bf[set][0][1]), acc[1][0], 0, 0, 0); \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(A[set][1][0], A[set][1][1]),                       \
                                                     frag_of(Bf[set][1][0], Bf[set][1][1]), acc[1][1], 0, 0, 0)
    using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>; using I6 = std::integral_constant<int, 6>;
    RD(0, 0);
    RD(1, 1); LGKM_WAIT(8); MM(0); __builtin_amdgcn_sched_barrier(0);
    if (more && spread) issue_part(t + 2, I0{}, I2{});
    RD(0, 2); LGKM_WAIT(8); MM(1); __builtin_amdgcn_sched_barrier(0);
    if (more && spread) issue_part(t + 2, I2{}, I4{});
    RD(1, 3); LGKM_WAIT(8); MM(0); __builtin_amdgcn_sched_barrier(0);
    if (more && spread) issue_part(t + 2, I4{}, I6{});
    LGKM_WAIT(0); MM(1); __builtin_amdgcn_sched_barrier(0);
#undef RD
#undef MM
  }

  // ---- this split's slab: lane holds column ci = l31 of each block, rows (r&3) + 8*(r>>2) + 4*lhi
  const int l31 = lane & 31, lhi = lane >> 5;
  float* base = slabs + (long)s * Cout * Cin;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ci = ci0 + wc * 64 + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (co < Cout && ci < Cin) base[(long)co * Cin + ci] = acc[a][b][r];
      }
    }
}

__global__ __launch_bounds__(512, 1) void k_wgrad1x1(const bf16* __restrict__ X, const bf16* __restrict__ dY,
                                                       float* __restrict__ slabs, const bf16* __restrict__ zeros,
                                                       long Npix, int Cin, int Cout, int tiles_ci, long L) {
  wgrad1x1_body(X, dY, slabs, zeros, Npix, Cin, Cout, tiles_ci, L, blockIdx.x, blockIdx.y, true);
}

// A GROUP of 1x1 layers in one launch (round 3).  The ~31 1x1 weight gradients of a step were 31 launches of 9-70 us, the
// 8x8 / 16x16 ones launch- and latency-bound (one short burst of workgroups each).  Their results are only needed by the
// optimizer, so the layers of a stretch of the backward pass are queued and their (tile, split) workgroups laid end to
// end in ONE grid: a workgroup finds its layer by its block index and runs the same body.
constexpr int W1_MAX = 16;
struct W1Group {
  const bf16* X[W1_MAX];
  const bf16* dY[W1_MAX];
  float* slabs[W1_MAX];
  long npix[W1_MAX], L[W1_MAX];
  int Cin[W1_MAX], Cout[W1_MAX], tiles_ci[W1_MAX], tiles[W1_MAX], nsplit[W1_MAX];
  int wg_end[W1_MAX];     // exclusive prefix of workgroups: layer i owns blocks [wg_end[i-1], wg_end[i])
  int n;
  int spread;   // DMA issue spread over the k-steps (EDM_W1_SPREAD=0: one burst behind the barrier, the round-1..5 form)
  const bf16* zeros;
};
// Block -> (tile, split) inside a layer: the `tiles` workgroups of ONE split read the same pixel rows at the same time
// (a dY row is shared by the ci-tiles, an X row by the co-tiles), so they are put on ONE XCD (blockIdx % 8 under round-robin
// placement; layer ranges start at multiples of 8) and the second..last read of a row hits that XCD's L2 instead of HBM:
// within a chunk of 8 * tiles consecutive blocks, block j is split (chunk * 8 + j % 8), tile j / 8.  Splits past the
// layer's count (its block range is padded to whole chunks) exit at once.
__global__ __launch_bounds__(512, 1) void k_wgrad1x1_group(const W1Group* __restrict__ gp) {
  const W1Group& g = *gp;   // (device memory: common.h EDM_UPLOAD_TABLE)
  int i = 0;
  const int b = blockIdx.x;
  while (i + 1 < g.n && b >= g.wg_end[i]) ++i;
  i = __builtin_amdgcn_readfirstlane(i);
  const int lid = b - (i ? g.wg_end[i - 1] : 0);
  const int tiles = g.tiles[i], per = 8 * tiles;
  const int chunk = lid / per, j = lid - chunk * per;
  const int s = chunk * 8 + (j & 7), tile = j >> 3;
  if (s >= g.nsplit[i]) return;
  wgrad1x1_body(g.X[i], g.dY[i], g.slabs[i], g.zeros, g.npix[i], g.Cin[i], g.Cout[i], g.tiles_ci[i], g.L[i], tile, s, g.spread != 0);
}

}  // namespace

// Number of split-K slabs edm_conv_wgrad_1x1 writes for this shape (sizes the workspace [S][Cout][Cin] fp32).
extern "C" int edm_conv_wgrad_1x1_nsplit(long npix, int Cin, int Cout) {
  if (npix <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const int tiles = ((Cout + TCO - 1) / TCO) * ((Cin + TCI - 1) / TCI);
  long S = (256 + tiles - 1) / tiles;            // ~one workgroup per CU
  const long max_s = (npix + 4 * KP - 1) / (4 * KP);  // at least 4 stages per workgroup
  if (S > max_s) S = max_s;
  static const long max_split = [] { const char* e = getenv("EDM_W1_MAXSPLIT"); return e ? atol(e) : 64L; }();
  if (S > max_split) S = max_split;  // more splits buy no kernel time (r01 sweep) but every slab is re-read by the finish pass
  if (S < 1) S = 1;
  return (int)S;
}

// Splits of a layer inside a GROUPED launch: the group as a whole fills the chip, so a layer needs only enough splits to
// give each workgroup ~32 stages of 64 pixels -- every split is a slab written here and re-read by the finish pass
// (16x16 layers: 16 instead of 43-64 splits, 8x8 layers: 4 instead of 32; sweep 8 / 16 / 32 / 64 stages: 13.91 / 13.81 / 13.79 / 13.78 ms per step).
extern "C" int edm_conv_wgrad_1x1_nsplit_grouped(long npix, int Cin, int Cout) {
  const int single = edm_conv_wgrad_1x1_nsplit(npix, Cin, Cout);
  static const long stages = [] { const char* e = getenv("EDM_W1_STAGES"); return e ? atol(e) : 32L; }();   // tools only
  long S = (npix + stages * KP - 1) / (stages * KP);
  // the splits of a layer are what spreads it over the XCDs (k_wgrad1x1_group: split = block % 8 within a chunk): with fewer
  // than 8, the blocks of the missing splits exit at once and their XCDs idle through this layer's stretch of the grid (round 6,
  // tools/microbench_wgrad1x1.py: sixteen 8x8 layers of the training batch, 4 -> 8 splits: 150 -> 106 us (256 -> 768),
  // 105 -> 92 us (512 -> 256); 16-stage workgroups everywhere instead cost the 16x16 layers 15 %)
  static const bool min8 = [] { const char* e = getenv("EDM_W1_MIN8"); return !(e && e[0] == '0'); }();   // tools only (A/B)
  if (min8 && S < 8 && npix >= 8L * 12 * KP) S = 8;
  if (S > single) S = single;
  return S < 1 ? 1 : (int)S;
}

// X [npix, Cin] bf16, dY [npix, Cout] bf16 (NHWC flattened) -> slabs fp32 [nsplit][Cout][Cin]; Cin, Cout % 32 == 0.
extern "C" int edm_conv_wgrad_1x1(const void* X, const void* dY, float* slabs, long npix, int Cin, int Cout, int nsplit,
                                  hipStream_t st) {
  EDM_REQUIRE(X && dY && slabs, "conv_wgrad_1x1: null pointer");
  EDM_REQUIRE(npix > 0 && npix < (1L << 31), "conv_wgrad_1x1: bad pixel count");
  EDM_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 32 == 0, "conv_wgrad_1x1: Cin, Cout must be multiples of 32");
  EDM_REQUIRE(nsplit == edm_conv_wgrad_1x1_nsplit(npix, Cin, Cout), "conv_wgrad_1x1: nsplit mismatch");
  EDM_ZERO_PAGE(zero_page_, "conv_wgrad_1x1");
  (void)zero_page_;
  long L = (npix + nsplit - 1) / nsplit;
  L = (L + KP - 1) / KP * KP;
  const int tiles_co = (Cout + TCO - 1) / TCO, tiles_ci = (Cin + TCI - 1) / TCI;
  EDM_MAX_LDS(k_wgrad1x1, 160 * 1024);
  hipLaunchKernelGGL(k_wgrad1x1, dim3(tiles_co * tiles_ci, nsplit), dim3(512), (size_t)RING * STAGE, st, (const bf16*)X,
                     (const bf16*)dY, slabs, (const bf16*)edm_zero_page(), npix, Cin, Cout, tiles_ci, L);
  EDM_CHECK_LAUNCH("conv_wgrad_1x1");
  return EDM_OK;
}

// The same for up to 16 layers in ONE launch; `items` is a HOST array read during the call; every layer's slabs are
// [nsplit][Cout][Cin] fp32 with nsplit = edm_conv_wgrad_1x1_nsplit_grouped(npix, Cin, Cout).
struct edm_wgrad1_item_ {
  const void* X;
  const void* dY;
  float* slabs;
  long npix;
  int Cin, Cout, nsplit, pad;
};
extern "C" long edm_conv_wgrad_1x1_group_table_bytes(void) { return (long)sizeof(W1Group); }

extern "C" int edm_conv_wgrad_1x1_group(const void* items_, int n, void* table_host, void* table_dev, int defer_upload, hipStream_t st) {
  const edm_wgrad1_item_* items = (const edm_wgrad1_item_*)items_;
  EDM_REQUIRE(items && n > 0 && n <= W1_MAX, "conv_wgrad_1x1_group: 1..%d layers per group", W1_MAX);
  EDM_ZERO_PAGE(zero_page_, "conv_wgrad_1x1_group");
  W1Group g;
  g.n = n;
  {
    const char* e = getenv("EDM_W1_SPREAD");   // read per call (A/B in one process; tools only)
    g.spread = !(e && e[0] == '0');
  }
  g.zeros = (const bf16*)zero_page_;
  long total = 0;
  for (int i = 0; i < n; ++i) {
    const edm_wgrad1_item_& it = items[i];
    EDM_REQUIRE(it.X && it.dY && it.slabs, "conv_wgrad_1x1_group: null pointer (layer %d)", i);
    EDM_REQUIRE(it.npix > 0 && it.npix < (1L << 31), "conv_wgrad_1x1_group: bad pixel count (layer %d)", i);
    EDM_REQUIRE(it.Cin > 0 && it.Cin % 32 == 0 && it.Cout > 0 && it.Cout % 32 == 0,
                "conv_wgrad_1x1_group: Cin, Cout must be multiples of 32 (layer %d)", i);
    EDM_REQUIRE(it.nsplit == edm_conv_wgrad_1x1_nsplit_grouped(it.npix, it.Cin, it.Cout),
                "conv_wgrad_1x1_group: nsplit mismatch (layer %d)", i);
    long L = (it.npix + it.nsplit - 1) / it.nsplit;
    L = (L + KP - 1) / KP * KP;
    const int tiles_co = (it.Cout + TCO - 1) / TCO, tiles_ci = (it.Cin + TCI - 1) / TCI;
    g.X[i] = (const bf16*)it.X;
    g.dY[i] = (const bf16*)it.dY;
    g.slabs[i] = it.slabs;
    g.npix[i] = it.npix;
    g.L[i] = L;
    g.Cin[i] = it.Cin;
    g.Cout[i] = it.Cout;
    g.tiles_ci[i] = tiles_ci;
    g.tiles[i] = tiles_co * tiles_ci;
    g.nsplit[i] = it.nsplit;
    total += (long)g.tiles[i] * ((it.nsplit + 7) / 8 * 8);     // whole chunks of 8 splits: every layer starts on XCD 0
    EDM_REQUIRE(total < (1L << 30), "conv_wgrad_1x1_group: grid too large");
    g.wg_end[i] = (int)total;
  }
  for (int i = n; i < W1_MAX; ++i) {
    g.X[i] = g.dY[i] = nullptr;
    g.slabs[i] = nullptr;
    g.npix[i] = g.L[i] = 0;
    g.Cin[i] = g.Cout[i] = g.tiles_ci[i] = g.tiles[i] = g.nsplit[i] = 0;
    g.wg_end[i] = (int)total;
  }
  EDM_MAX_LDS(k_wgrad1x1_group, 160 * 1024);
  EDM_UPLOAD_TABLE(table_dev, table_host, &g, sizeof(W1Group), st, "conv_wgrad_1x1_group", defer_upload);
  hipLaunchKernelGGL(k_wgrad1x1_group, dim3((unsigned)total), dim3(512), (size_t)RING * STAGE, st, (const W1Group*)table_dev);
  EDM_CHECK_LAUNCH("conv_wgrad_1x1_group");
  return EDM_OK;
}

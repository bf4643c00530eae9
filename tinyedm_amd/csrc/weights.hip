// Weight-side kernels: forced weight normalisation + effective-weight packing, and the matching
// gradient finish (split-K slab reduction + projection through the normalisation).
//
// Reference semantics (networks.py:32-37 / 55-60, normalize at :17-19):
//   training forward:  w <- w / (eps + ||w_o|| / sqrt(n))            (in place, no grad)
//   always:            w_hat = w / (eps + ||w_o|| / sqrt(n)) / sqrt(n)   (differentiable)
// with n = fan_in = I*taps and the norm taken per output row o.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// one workgroup per packed output row r (master row mo = perm ? perm[r] : r)
__device__ __forceinline__ void prep_row(float* __restrict__ w, int O, int I, int taps, int Ipad,
                                         bf16* __restrict__ wp_fwd, bf16* __restrict__ wp_dgrad,
                                         float* __restrict__ w_hat, const int* __restrict__ perm,
                                         int normalize_inplace, int r, float* red) {
  const int mo = perm ? perm[r] : r;
  const int n = I * taps;
  float* row = w + (long)mo * n;
  const float rsn = rsqrtf((float)n);
  float ss = 0.f;
  for (int e = threadIdx.x; e < n; e += blockDim.x) ss += row[e] * row[e];
  ss = block_sum(ss, red);
  float d = NORM_EPS + sqrtf(ss) * rsn;
  float pre = 1.0f;  // factor applied to the stored master row
  if (normalize_inplace) {
    pre = 1.0f / d;
    // norm of the re-normalised row, recomputed from the rounded fp32 values like the reference does
    float ss2 = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
      float v = row[e] * pre;
      ss2 += v * v;
    }
    ss2 = block_sum(ss2, red);
    d = NORM_EPS + sqrtf(ss2) * rsn;
  }
  const float post = rsn / d;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const float wm = row[e] * pre;
    if (normalize_inplace) row[e] = wm;
    const float wh = wm * post;
    const int i = e / taps, t = e - i * taps;
    if (wp_fwd) wp_fwd[((long)t * O + r) * Ipad + i] = (bf16)wh;
    if (wp_dgrad) wp_dgrad[((long)(taps - 1 - t) * I + i) * O + r] = (bf16)wh;
    if (w_hat) w_hat[(long)mo * n + e] = wh;
  }
  if (wp_fwd && Ipad > I) {
    const int padn = (Ipad - I) * taps;
    for (int e = threadIdx.x; e < padn; e += blockDim.x) {
      const int t = e / (Ipad - I), i = I + e % (Ipad - I);
      wp_fwd[((long)t * O + r) * Ipad + i] = (bf16)0.f;
    }
  }
}

__global__ __launch_bounds__(256) void k_weight_prep(float* __restrict__ w, int O, int I, int taps, int Ipad,
                                                       bf16* __restrict__ wp_fwd, bf16* __restrict__ wp_dgrad,
                                                       float* __restrict__ w_hat, const int* __restrict__ perm,
                                                       int normalize_inplace) {
  __shared__ float red[8];
  prep_row(w, O, I, taps, Ipad, wp_fwd, wp_dgrad, w_hat, perm, normalize_inplace, blockIdx.x, red);
}

// multi-tensor form: ONE launch prepares every weight of the network (136 tensors for the CIFAR-10 U-Net).
// A workgroup owns `rb` consecutive packed rows of one tensor (host-chosen per tensor so that rb*n bf16 fit LDS):
// each wave normalises whole rows (coalesced fp32 reads/writes of the master row and w_hat), the bf16 effective
// weights meet in an LDS tile, and the two kernel-layout packs are then written from the tile so that every store
// is contiguous: the forward pack along ci, the dgrad pack ([tap][ci][co]: co is the fast index) rb rows at a time.
// (One-row-per-workgroup wrote the dgrad pack as isolated 2-byte stores: 0.51 ms per step for 35.6 M parameters.)
struct PrepDesc {  // mirrored by tinyedm_amd/networks.py (64 bytes)
  float* w;
  bf16* fwd;
  bf16* dgrad;
  float* hat;
  const int* perm;
  int O, I, taps, Ipad, row0, rb;
};
__global__ __launch_bounds__(256) void k_weight_prep_multi(const PrepDesc* __restrict__ descs,
                                                             const int2* __restrict__ groups, int normalize_inplace) {
  extern __shared__ __attribute__((aligned(16))) char tile_raw[];
  bf16* tile = reinterpret_cast<bf16*>(tile_raw);  // [rb][n]
  const int2 gr = groups[blockIdx.x];               // (descriptor, first packed row)
  PrepDesc d = descs[gr.x];
  // bits 8 / 9 of the taps field: the forward / dgrad pack of this tensor is written FRAGMENT-MAJOR (see the pack phase)
  const bool frag_fwd = (d.taps & 0x100) != 0, frag_dgrad = (d.taps & 0x200) != 0;
  d.taps &= 0xFF;
  const int r0 = gr.y;
  const int rbc = min(d.rb, d.O - r0);
  const int n = d.I * d.taps;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float rsn = rsqrtf((float)n);
  // normalisation phase: one wave per row.  The pass is latency-bound unless many loads are in flight, so rows of
  // up to 5120 elements are held in registers (<= 20 float4 per lane, all loads issued back to back, ONE trip to
  // HBM); longer or oddly sized rows take the three-pass loop.
  constexpr int K4MAX = 20;
  const bool in_regs = (n & 3) == 0 && n <= K4MAX * 256;
  for (int rr = wave; rr < rbc; rr += 4) {
    const int r = r0 + rr;
    const int mo = d.perm ? d.perm[r] : r;
    float* row = d.w + (long)mo * n;
    if (in_regs) {
      f32x4 v[K4MAX];
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < K4MAX; ++k) {
        const int e = k * 256 + lane * 4;
        v[k] = (e < n) ? *reinterpret_cast<const f32x4*>(row + e) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < K4MAX; ++k) ss += v[k][0] * v[k][0] + v[k][1] * v[k][1] + v[k][2] * v[k][2] + v[k][3] * v[k][3];
      ss = wave_sum(ss);
      float dn = NORM_EPS + sqrtf(ss) * rsn;
      float pre = 1.0f;
      if (normalize_inplace) {
        pre = 1.0f / dn;
        float ss2 = 0.f;
#pragma unroll
        for (int k = 0; k < K4MAX; ++k) {
          v[k] *= pre;
          ss2 += v[k][0] * v[k][0] + v[k][1] * v[k][1] + v[k][2] * v[k][2] + v[k][3] * v[k][3];
        }
        ss2 = wave_sum(ss2);
        dn = NORM_EPS + sqrtf(ss2) * rsn;
      }
      const float post = rsn / dn;
#pragma unroll
      for (int k = 0; k < K4MAX; ++k) {
        const int e = k * 256 + lane * 4;
        if (e < n) {
          if (normalize_inplace) *reinterpret_cast<f32x4*>(row + e) = v[k];
          const f32x4 wh = v[k] * post;
          if (d.hat) *reinterpret_cast<f32x4*>(d.hat + (long)mo * n + e) = wh;
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)wh[j];
          *reinterpret_cast<bf16x4*>(tile + (long)rr * n + e) = o;
        }
      }
      continue;
    }
    if ((n & 3) == 0) {
      // long rows (fan_in up to 1536*9 in the ImageNet nets): three passes, 8 float4 loads in flight per lane each
      const int n4 = n >> 2;
      const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
      float ss = 0.f;
      for (int k0 = 0; k0 < n4; k0 += 512) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + u * 64 + lane;
          v[u] = k < n4 ? row4[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) ss += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
      }
      ss = wave_sum(ss);
      float dn = NORM_EPS + sqrtf(ss) * rsn;
      float pre = 1.0f;
      if (normalize_inplace) {
        pre = 1.0f / dn;
        float ss2 = 0.f;
        for (int k0 = 0; k0 < n4; k0 += 512) {
          f32x4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * 64 + lane;
            v[u] = k < n4 ? row4[k] * pre : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) ss2 += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
        }
        ss2 = wave_sum(ss2);
        dn = NORM_EPS + sqrtf(ss2) * rsn;
      }
      const float post = rsn / dn;
      for (int k0 = 0; k0 < n4; k0 += 512) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + u * 64 + lane;
          if (k < n4) v[u] = row4[k] * pre;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + u * 64 + lane;
          if (k < n4) {
            if (normalize_inplace) reinterpret_cast<f32x4*>(row)[k] = v[u];
            const f32x4 wh = v[u] * post;
            if (d.hat) *reinterpret_cast<f32x4*>(d.hat + (long)mo * n + k * 4) = wh;
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)wh[j];
            *reinterpret_cast<bf16x4*>(tile + (long)rr * n + k * 4) = o;
          }
        }
      }
      continue;
    }
    float ss = 0.f;
    for (int e = lane; e < n; e += 64) ss += row[e] * row[e];
    ss = wave_sum(ss);
    float dn = NORM_EPS + sqrtf(ss) * rsn;
    float pre = 1.0f;
    if (normalize_inplace) {
      pre = 1.0f / dn;
      float ss2 = 0.f;  // norm of the re-normalised row, from the rounded fp32 values like the reference
      for (int e = lane; e < n; e += 64) {
        const float v = row[e] * pre;
        ss2 += v * v;
      }
      ss2 = wave_sum(ss2);
      dn = NORM_EPS + sqrtf(ss2) * rsn;
    }
    const float post = rsn / dn;
    for (int e = lane; e < n; e += 64) {
      const float wm = row[e] * pre;
      if (normalize_inplace) row[e] = wm;
      const float wh = wm * post;
      if (d.hat) d.hat[(long)mo * n + e] = wh;
      tile[(long)rr * n + e] = (bf16)wh;
    }
  }
  __syncthreads();
  // pack phase: index arithmetic is kept to adds (runtime integer division per element made this pass ALU-bound)
  const int taps = d.taps, I = d.I, Ipad = d.Ipad, O = d.O;
  // Fragment-major packs (round 4, the 8x8 layers' kernel k_conv3x3_s): the 16 bytes lane (l31, lhi) of a wave feeds to
  // v_mfma_f32_32x32x16_bf16 as its share of an A fragment -- row co = 32 cb + l31, channels 32 c + 16 ks + 8 lhi .. + 7 --
  // are stored at [tap][c][cb][ks][lane = 32 lhi + l31][8]: a fragment is ONE contiguous KiB, so the kernel loads its
  // weights straight into the registers it multiplies from with fully coalesced 1-KiB instructions (the [tap][co][ci] pack
  // gives such a load 32 rows x 32 bytes: 64 cache-line look-ups per instruction, which bound the kernel).
  // Host-checked for these tensors: I % 32 == 0, O % 32 == 0, Ipad == I, rb % 8 == 0, rb | 32.
  if (d.fwd && frag_fwd) {
    const int NC = I >> 5, NCB = O >> 5, cb = r0 >> 5, rl0 = r0 & 31;
    const int items = taps * NC * 4 * rbc;                   // (t, c, ks, g, rr), rr fastest: runs of 16 rbc bytes
    for (int idx = threadIdx.x; idx < items; idx += 256) {
      const int rr = idx % rbc, q = idx / rbc;
      const int g = q & 1, ks = (q >> 1) & 1, tc = q >> 2;
      const int c = tc % NC, t = tc / NC;
      const bf16* src = tile + (long)rr * n + (long)(c * 32 + ks * 16 + g * 8) * taps + t;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[e * taps];
      bf16* dst = d.fwd + ((((long)(t * NC + c) * NCB + cb) * 2 + ks) * 64 + g * 32 + rl0 + rr) * 8;
      *reinterpret_cast<bf16x8*>(dst) = v;
    }
  }
  if (d.dgrad && frag_dgrad) {
    // the dgrad conv: taps flipped, "input" channels = O (this workgroup's rows), "output" channels = I
    const int NC = O >> 5, NCB = I >> 5;
    const int items = taps * (rbc >> 3) * I;                 // (t, og, i), i fastest: runs of 512 bytes
    for (int idx = threadIdx.x; idx < items; idx += 256) {
      const int i = idx % I, q = idx / I;
      const int og = q % (rbc >> 3), t = q / (rbc >> 3);
      const int o0 = r0 + og * 8;
      const int c = o0 >> 5, ks = (o0 >> 4) & 1, g = (o0 >> 3) & 1;
      const bf16* src = tile + (long)(og * 8) * n + (long)i * taps + t;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[(long)e * n];
      bf16* dst = d.dgrad + ((((long)((taps - 1 - t) * NC + c) * NCB + (i >> 5)) * 2 + ks) * 64 + g * 32 + (i & 31)) * 8;
      *reinterpret_cast<bf16x8*>(dst) = v;
    }
  }
  if (d.fwd && !frag_fwd) {  // [t][O][Ipad], ci fast: one (row, tap) pair per wave pass, lanes along ci
    for (int pair = wave; pair < rbc * taps; pair += 4) {
      const int rr = pair / taps, t = pair - rr * taps;  // wave-uniform (scalar unit)
      bf16* dst = d.fwd + ((long)t * O + r0 + rr) * Ipad;
      const bf16* src = tile + (long)rr * n + t;
      if ((Ipad & 7) == 0) {
        for (int i8 = lane * 8; i8 < Ipad; i8 += 512) {
          bf16x8 v;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (i8 + j < I) ? src[(i8 + j) * taps] : (bf16)0.f;
          *reinterpret_cast<bf16x8*>(dst + i8) = v;
        }
      } else {
        for (int i = lane; i < Ipad; i += 64) dst[i] = i < I ? src[i * taps] : (bf16)0.f;
      }
    }
  }
  if (d.dgrad && !frag_dgrad) {  // [(taps-1-t)][ci][O], co fast: rbc contiguous values per (t, ci); threads along ci
    const bool vec = (rbc & 7) == 0 && (O & 7) == 0;
    for (int i = threadIdx.x; i < I; i += 256) {
      for (int t = 0; t < taps; ++t) {
        bf16* dst = d.dgrad + ((long)(taps - 1 - t) * I + i) * O + r0;
        const bf16* src = tile + i * taps + t;
        if (vec) {
          for (int c = 0; c < rbc; c += 8) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(long)(c + j) * n];
            *reinterpret_cast<bf16x8*>(dst + c) = v;
          }
        } else {
          for (int rr = 0; rr < rbc; ++rr) dst[rr] = src[(long)rr * n];
        }
      }
    }
  }
}

// grad[mo, i, t] (=|+=) projection( scale * sum_s slabs[s, t, r, i] ) through w_hat = w/(d*sqrt(n))
// One workgroup per packed row r.  The slab sum is a pure HBM stream of S*n floats per row whose rate is set by
// the bytes in flight: work item (e4, sg) sums slabs sg, sg+G, ... of one 16-byte vector with 8 loads in flight,
// G = s-groups per vector (host-chosen so that E4*G ~ one block), partials meet in LDS in a fixed order
// (deterministic).  part: [G][n] floats in packed (t major, i minor) order; g: [n] floats in master order.
__global__ __launch_bounds__(1024) void k_wgrad_finish(const float* __restrict__ slabs, int S, const float* __restrict__ w,
                                                         float* __restrict__ grad, const int* __restrict__ perm, int O,
                                                         int I, int Ipad, int taps, float scale, int accumulate, int G) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[16];
  const int r = blockIdx.x;
  const int mo = perm ? perm[r] : r;
  const int n = I * taps;
  float* part = sm;            // [G][n]
  float* g = sm + (long)G * n; // [n]
  const float* row = w + (long)mo * n;
  const long slab_stride = (long)taps * O * Ipad;
  if ((I & 3) == 0 && (Ipad & 3) == 0) {
    const int I4 = I >> 2, E4 = taps * I4;
    if (S <= 3) {
      // few slabs, long rows (the wide layers of the ImageNet nets: one work item per pass would have a single load
      // in flight): four vectors per thread per trip, all their loads issued together
      const int items = E4;  // G == 1 here (host: 2G <= S fails for S <= 1; S in 2..3 gives G <= 1 when E4 >= 512)
      for (int idx0 = threadIdx.x; idx0 < items * G; idx0 += 4 * blockDim.x) {
        f32x4 acc4[4];
        int e4s[4], sgs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = idx0 + u * blockDim.x;
          const bool ok = idx < items * G;
          const int sg = ok ? idx / E4 : 0, e4 = ok ? idx - sg * E4 : 0;
          e4s[u] = ok ? e4 : -1;
          sgs[u] = sg;
          const int t = e4 / I4, i = (e4 - t * I4) * 4;
          const float* sp = slabs + ((long)t * O + r) * Ipad + i;
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          if (ok)
            for (int sl = sg; sl < S; sl += G) a += *reinterpret_cast<const f32x4*>(sp + (long)sl * slab_stride);
          acc4[u] = a;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e4s[u] >= 0) *reinterpret_cast<f32x4*>(part + (long)sgs[u] * n + e4s[u] * 4) = acc4[u];
      }
    } else
    for (int idx = threadIdx.x; idx < E4 * G; idx += blockDim.x) {
      const int sg = idx / E4, e4 = idx - sg * E4;
      const int t = e4 / I4, i = (e4 - t * I4) * 4;
      const float* sp = slabs + ((long)t * O + r) * Ipad + i;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      int s = sg;
      for (; s + 7 * G < S; s += 8 * G) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(sp + (long)(s + u * G) * slab_stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
      }
      for (; s < S; s += G) a += *reinterpret_cast<const f32x4*>(sp + (long)s * slab_stride);
      *reinterpret_cast<f32x4*>(part + (long)sg * n + e4 * 4) = a;
    }
  } else {
    for (int idx = threadIdx.x; idx < n * G; idx += blockDim.x) {
      const int sg = idx / n, e = idx - sg * n;
      const int t = e / I, i = e - t * I;
      const float* sp = slabs + ((long)t * O + r) * Ipad + i;
      float a = 0.f;
      for (int s = sg; s < S; s += G) a += sp[(long)s * slab_stride];
      part[(long)sg * n + e] = a;
    }
  }
  __syncthreads();
  float dot = 0.f, ss = 0.f;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {  // e in packed order: (t, i)
    float a = 0.f;
    for (int sg = 0; sg < G; ++sg) a += part[(long)sg * n + e];
    a *= scale;
    const int t = e / I, i = e - t * I;
    const float wv = row[i * taps + t];
    g[i * taps + t] = a;
    dot += a * wv;
    ss += wv * wv;
  }
  dot = block_sum(dot, red);
  ss = block_sum(ss, red);
  const float rn = sqrtf(ss);
  const float sqn = sqrtf((float)n);
  const float d = NORM_EPS + rn / sqn;
  const float c0 = 1.0f / (d * sqn);
  const float c1 = rn > 0.f ? dot / (d * rn * sqn) : 0.f;
  __syncthreads();
  float* out = grad + (long)mo * n;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    float v = c0 * (g[e] - row[e] * c1);
    out[e] = accumulate ? out[e] + v : v;
  }
}

// ---- multi-tensor finish: the same reduction + projection for up to MAXF tensors in ONE launch (a training step has
// ~70 small weight gradients -- 1x1 convs, embedding / gate linears -- whose separate finish launches cost 8 us each).
constexpr int MAXF = 40;
struct FinItem {
  const float* slabs;
  const float* w;
  float* grad;
  const int* perm;
  int S, O, I, Ipad, taps, G, row0;
  float scale;
  int accumulate, pad;
};
struct FinGroup {
  FinItem it[MAXF];
  int n, prefetch;
};
__global__ __launch_bounds__(512) void k_wgrad_finish_multi(const FinGroup* __restrict__ gp) {
  const FinGroup& g = *gp;   // (device memory: common.h EDM_UPLOAD_TABLE)
  extern __shared__ __attribute__((aligned(16))) float smm[];
  __shared__ float red[16];
  int k = 0;
  while (k + 1 < g.n && (int)blockIdx.x >= g.it[k + 1].row0) ++k;
  const int r = blockIdx.x - g.it[k].row0;
  const float* __restrict__ slabs = g.it[k].slabs;
  const int S = g.it[k].S, O = g.it[k].O, I = g.it[k].I, Ipad = g.it[k].Ipad, taps = g.it[k].taps, G = g.it[k].G;
  const float scale = g.it[k].scale;
  const int mo = g.it[k].perm ? g.it[k].perm[r] : r;
  const int n = I * taps;
  float* part = smm;                 // [G][n] packed (t major, i minor)
  float* gm = smm + (long)G * n;     // [n] master order
  const float* row = g.it[k].w + (long)mo * n;
  const long slab_stride = (long)taps * O * Ipad;
  // The kernel is a chain of memory round trips per workgroup (table -> slabs -> master row -> old gradient -> store) over
  // ~10 rounds of workgroups per launch: the master row and (accumulate) the old gradient are loaded NOW, while the slab
  // loads are in flight -- as k_wgrad3_finish does (rows of up to 2 x 512 elements: every 1x1 conv / Linear of the nets here)
  constexpr int WPF = 2;
  const bool pf = g.prefetch && n <= WPF * 512;
  float* out = g.it[k].grad + (long)mo * n;
  const int accumulate = g.it[k].accumulate;
  float wpre[WPF] = {0.f, 0.f}, opre[WPF] = {0.f, 0.f};
  if (pf) {
#pragma unroll
    for (int q = 0; q < WPF; ++q) {
      const int e = threadIdx.x + q * 512;
      if (e < n) {
        wpre[q] = row[e];
        if (accumulate) opre[q] = out[e];
      }
    }
  }
  if ((I & 3) == 0 && (Ipad & 3) == 0) {
    const int I4 = I >> 2, E4 = taps * I4;
    for (int idx = threadIdx.x; idx < E4 * G; idx += blockDim.x) {
      const int sg = idx / E4, e4 = idx - sg * E4;
      const int t = e4 / I4, i = (e4 - t * I4) * 4;
      const float* sp = slabs + ((long)t * O + r) * Ipad + i;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      int s = sg;
      for (; s + 3 * G < S; s += 4 * G) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(sp + (long)(s + u * G) * slab_stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) a += v[u];
      }
      for (; s < S; s += G) a += *reinterpret_cast<const f32x4*>(sp + (long)s * slab_stride);
      *reinterpret_cast<f32x4*>(part + (long)sg * n + e4 * 4) = a;
    }
  } else {
    for (int idx = threadIdx.x; idx < n * G; idx += blockDim.x) {
      const int sg = idx / n, e = idx - sg * n;
      const int t = e / I, i = e - t * I;
      const float* sp = slabs + ((long)t * O + r) * Ipad + i;
      float a = 0.f;
      for (int s = sg; s < S; s += G) a += sp[(long)s * slab_stride];
      part[(long)sg * n + e] = a;
    }
  }
  __syncthreads();
  float dot = 0.f, ss = 0.f;
  if (pf && taps == 1) {   // packed order == master order: element e of the row is the prefetched one
#pragma unroll
    for (int q = 0; q < WPF; ++q) {
      const int e = threadIdx.x + q * 512;
      if (e < n) {
        float a = 0.f;
        for (int sg = 0; sg < G; ++sg) a += part[(long)sg * n + e];
        a *= scale;
        gm[e] = a;
        dot += a * wpre[q];
        ss += wpre[q] * wpre[q];
      }
    }
  } else {
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
      float a = 0.f;
      for (int sg = 0; sg < G; ++sg) a += part[(long)sg * n + e];
      a *= scale;
      const int t = e / I, i = e - t * I;
      const float wv = row[i * taps + t];
      gm[i * taps + t] = a;
      dot += a * wv;
      ss += wv * wv;
    }
  }
  dot = block_sum(dot, red);
  ss = block_sum(ss, red);
  const float rn = sqrtf(ss);
  const float sqn = sqrtf((float)n);
  const float d = NORM_EPS + rn / sqn;
  const float c0 = 1.0f / (d * sqn);
  const float c1 = rn > 0.f ? dot / (d * rn * sqn) : 0.f;
  __syncthreads();
  if (pf) {
#pragma unroll
    for (int q = 0; q < WPF; ++q) {
      const int e = threadIdx.x + q * 512;
      if (e < n) {
        const float v = c0 * (gm[e] - wpre[q] * c1);
        out[e] = accumulate ? opre[q] + v : v;
      }
    }
  } else {
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
      const float v = c0 * (gm[e] - row[e] * c1);
      out[e] = accumulate ? out[e] + v : v;
    }
  }
}

}  // namespace

extern "C" {
typedef struct {
  const float* slabs;   // fp32 [S][taps][O][Ipad] split-K partial sums (packed row order)
  const float* w;       // fp32 master weight [O][I][taps]
  float* grad;          // fp32 gradient, same layout
  const int* perm;      // packed row -> master row, or NULL
  int S, O, I, Ipad, taps;
  float scale;
  int accumulate;
} edm_finish_item;
}

// Reduce + project up to 40 weight gradients in one launch (`items` is HOST memory, read during the call).
extern "C" long edm_wgrad_finish_multi_table_bytes(void) { return (long)sizeof(FinGroup); }

extern "C" int edm_wgrad_finish_multi(const edm_finish_item* items, int n, void* table_host, void* table_dev, int defer_upload, hipStream_t st) {
  EDM_REQUIRE(items && n > 0 && n <= MAXF, "wgrad_finish_multi: need 1..%d tensors, got %d", MAXF, n);
  FinGroup g;
  g.n = n;
  static const int prefetch = [] { const char* e = getenv("EDM_FIN_PREFETCH"); return e ? atoi(e) : 1; }();   // tools only (A/B)
  g.prefetch = prefetch;
  int row = 0;
  size_t lds = 0;
  for (int k = 0; k < n; ++k) {
    const edm_finish_item& a = items[k];
    EDM_REQUIRE(a.slabs && a.w && a.grad && a.S > 0 && a.O > 0 && a.I > 0 && a.taps > 0 && a.Ipad >= a.I,
                "wgrad_finish_multi: bad item %d", k);
    const int nn = a.I * a.taps;
    const int E = ((a.I & 3) == 0 && (a.Ipad & 3) == 0) ? nn / 4 : nn;
    int G = 1;
    while (G < 8 && 2 * G <= a.S && E * 2 * G <= 512 && (long)(2 * G + 1) * nn * 4 <= 96 * 1024) G *= 2;
    EDM_REQUIRE((long)(G + 1) * nn * 4 <= 128 * 1024, "wgrad_finish_multi: fan_in %d too large for the LDS row buffer", nn);
    g.it[k] = FinItem{a.slabs, a.w, a.grad, a.perm, a.S, a.O, a.I, a.Ipad, a.taps, G, row, a.scale, a.accumulate, 0};
    row += a.O;
    const size_t need = (size_t)(G + 1) * nn * sizeof(float);
    if (need > lds) lds = need;
  }
  EDM_MAX_LDS(k_wgrad_finish_multi, 128 * 1024);
  EDM_UPLOAD_TABLE(table_dev, table_host, &g, sizeof(FinGroup), st, "wgrad_finish_multi", defer_upload);
  hipLaunchKernelGGL(k_wgrad_finish_multi, dim3(row), dim3(512), lds, st, (const FinGroup*)table_dev);
  EDM_CHECK_LAUNCH("wgrad_finish_multi");
  return EDM_OK;
}

// w [O, I, taps] fp32 master (taps = k*k, OIHW flattened).  Any of wp_fwd / wp_dgrad / w_hat may be null.
//   wp_fwd   bf16 [taps, O, Ipad]   (ci >= I zero-filled)      -> edm_conv_igemm forward
//   wp_dgrad bf16 [taps, I, O]      (taps flipped)             -> edm_conv_igemm as dgrad
//   w_hat    fp32 [O, I*taps]       (master order, unpermuted) -> fp32 linears / gates / conv_out
// perm (device int32 [O], nullable): packed row r reads master row perm[r].
extern "C" int edm_weight_prep(float* w, int O, int I, int taps, int Ipad, void* wp_fwd, void* wp_dgrad,
                               float* w_hat, const int* perm, int normalize_inplace, hipStream_t st) {
  EDM_REQUIRE(w && O > 0 && I > 0 && taps > 0 && Ipad >= I, "weight_prep: bad args O=%d I=%d taps=%d Ipad=%d", O, I, taps, Ipad);
  hipLaunchKernelGGL(k_weight_prep, dim3(O), dim3(256), 0, st, w, O, I, taps, Ipad, (bf16*)wp_fwd, (bf16*)wp_dgrad,
                     w_hat, perm, normalize_inplace);
  EDM_CHECK_LAUNCH("weight_prep");
  return EDM_OK;
}

// descs: device array of 64-byte records {w, fwd, dgrad, hat, perm (pointers), O, I, taps, Ipad, row0, rb (int32)};
// groups: device int32 [n_groups][2] = (record index, first packed row): one workgroup per group of <= rb rows;
// lds_bytes = max over records of rb * I * taps * 2.
extern "C" int edm_weight_prep_multi(const void* descs, const int* groups, int n_groups, int lds_bytes,
                                     int normalize_inplace, hipStream_t st) {
  EDM_REQUIRE(descs && groups && n_groups > 0 && lds_bytes > 0 && lds_bytes <= 128 * 1024, "weight_prep_multi: bad args");
  static_assert(sizeof(PrepDesc) == 64, "PrepDesc layout");
  EDM_MAX_LDS(k_weight_prep_multi, 128 * 1024);
  hipLaunchKernelGGL(k_weight_prep_multi, dim3(n_groups), dim3(256), (size_t)lds_bytes, st, (const PrepDesc*)descs,
                     (const int2*)groups, normalize_inplace);
  EDM_CHECK_LAUNCH("weight_prep_multi");
  return EDM_OK;
}

// slabs fp32 [S, taps, O, Ipad] (packed row order) -> grad fp32 [O, I, taps] (master order).
extern "C" int edm_wgrad_finish(const float* slabs, int S, const float* w, float* grad, const int* perm, int O, int I,
                                int Ipad, int taps, float scale, int accumulate, hipStream_t st) {
  EDM_REQUIRE(slabs && w && grad && S > 0 && O > 0 && I > 0 && taps > 0 && Ipad >= I, "wgrad_finish: bad args");
  const int n = I * taps;
  EDM_REQUIRE((long)n * 4 <= 64 * 1024, "wgrad_finish: fan_in %d too large for the LDS row buffer", n);
  // work items = (16-byte vectors of the row) x (s-groups): aim at one full block, at most 8 groups, G | nothing
  const int E = ((I & 3) == 0 && (Ipad & 3) == 0) ? n / 4 : n;
  int G = 1;
  while (G < 8 && 2 * G <= S && E * 2 * G <= 1024 && (long)(2 * G + 1) * n * 4 <= 96 * 1024) G *= 2;
  const int items = E * G;
  const int passes = (items + 1023) / 1024;
  int threads = ((items + passes - 1) / passes + 63) / 64 * 64;
  if (threads < 64) threads = 64;
  if (threads > 1024) threads = 1024;
  const size_t lds = (size_t)(G + 1) * n * sizeof(float);
  EDM_MAX_LDS(k_wgrad_finish, 128 * 1024);
  hipLaunchKernelGGL(k_wgrad_finish, dim3(O), dim3(threads), lds, st, slabs, S, w, grad, perm, O, I, Ipad, taps, scale,
                     accumulate, G);
  EDM_CHECK_LAUNCH("wgrad_finish");
  return EDM_OK;
}

// Reference-precision evaluation path: the denoiser forward with fp32 activations and EXACT fp32 products.
//
// The reference samples and validates in fp32 (generate.py:39-44, callbacks.py:41-49, solvers.py:43-59); the training
// path of this library computes in bf16 (1.3e-3 away from the fp32 trajectory after 32 Heun steps).  These kernels are the
// second, slower, exact evaluation: NHWC fp32 activations, fp32 effective weights (the `w_hat` arrays edm_weight_prep
// writes, master OIHW order, unpermuted), and the f32-input matrix instruction v_mfma_f32_32x32x2_f32 -- 157 TFLOP/s
// peak, 1/16 of the bf16 rate, every product and sum a plain fp32 fma -- for every convolution.  Forward only (no
// autograd), eval-mode semantics (no dropout, no forced weight normalisation).
//
//   edm_f32_conv        3x3 / 1x1 implicit GEMM, Y = alpha*conv + beta*R, or the block's modulation epilogue
//                       Y = mp_silu(conv * (lin*gain + 1))                        (networks.py:37, 253-260, 87-88)
//   edm_f32_attention   cosine attention on the qkv conv's OWN channel order (networks.py:194-202)
//   edm_f32_*           the elementwise steps between them (networks.py:9-14, 72, 80, 83-84, 112-118, 311, 578-603)
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
// the same four values as a (hi, lo) bf16 pair in a row [hi(C) | lo(C)] (split-bf16 operand format, see edm_f32_to_pairs)
__device__ __forceinline__ void st4_pairs(bf16* row, int C, int c, const f32x4& v) {
  bf16x4 hi, lo;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    hi[k] = (bf16)v[k];
    lo[k] = (bf16)(v[k] - (float)hi[k]);
  }
  *reinterpret_cast<bf16x4*>(row + c) = hi;
  *reinterpret_cast<bf16x4*>(row + C + c) = lo;
}
inline int cdivi(long a, long b) { return (int)((a + b - 1) / b); }
inline int gridf(long work, int block, int cap = 256 * 16) {
  long g = (work + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------ convolution
// GEMM view as the bf16 kernels: M = flat pixels (tile 256), N = output channels (tile 128), K = taps x Cin in chunks of
// KC channels.  8 waves = 4 pixel quarters x 2 channel halves, a wave owns 64 px x 64 ch = 2 x 2 accumulator blocks of
// 32 x 32.  MFMA roles: A = pixels (rows of the result), B = weights, so a lane holds ONE channel of 16 pixels and a
// store instruction covers two 128-byte row segments.  Operands are staged K-MAJOR in LDS ([k][row]): a fragment read
// is 64 consecutive words.  The pixel slab of a chunk (tile + halo rows) is staged once and shared by the nine taps as a
// constant row shift (border taps masked per lane); both operands are register-prefetched one chunk ahead.  WM = pixel
// quarters per workgroup: 4 (256-pixel tile, 512 threads) or 2 (128-pixel tile, 256 threads: small feature maps, where
// the big tile would leave CUs without a workgroup).  Weight rows are padded to BN + 1 words so that the scattered
// staging stores of one master row (18 lanes, same column, different (tap, k)) land in different banks.
constexpr int BN = 128, BNP = BN + 1;

template <int TAPS, int KC, int WM>
__global__ __launch_bounds__(128 * WM) void k_conv_f32(const float* __restrict__ X, const float* __restrict__ Wh,
                                                    float* __restrict__ Y, const float* __restrict__ R, float alpha,
                                                    float beta, const float* __restrict__ lin, long lin_stride,
                                                    const float* __restrict__ gain, int HW, int Npix, int H, int W,
                                                    int Cin, int I, int Cout, int tiles_m, int tiles_n, int XR) {
  constexpr int BM = 64 * WM, NT = 128 * WM;
  constexpr int UPRW = KC * TAPS / 4;                 // float4 units per weight row and chunk
  constexpr int NWU = (BN * UPRW + NT - 1) / NT;      // weight units per thread
  constexpr int NXU = TAPS == 9 ? (WM == 4 ? 2 : 3) : 4;   // slab units per thread: (BM + 2*65) * XQ <= NXU * NT
  constexpr int XQ = KC / 4;                          // float4 units per slab row
  extern __shared__ __attribute__((aligned(16))) float smf[];
  const int HALO = TAPS == 9 ? W + 1 : 0;
  const int xrows = BM + 2 * HALO;
  float* const Xs = smf;                              // [2][KC][XR]
  float* const Ws = smf + 2 * KC * XR;                // [2][TAPS][KC][BNP]

  const int id = blockIdx.x;
  const int xcd = id & 7, kk = id >> 3;
  const int tn = kk % tiles_n, tm = (kk / tiles_n) * 8 + xcd;
  if (tm >= tiles_m) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;
  const bool full_w = (I % KC == 0) && (I == Cin);   // weight rows hold whole chunks: aligned float4 loads

  // border masks of this lane's two pixels (bit t = tap t stays inside the image)
  unsigned mask[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + l31;
    unsigned mk = TAPS == 9 ? 0u : 1u;
    if (TAPS == 9) {
      const int w = m % W, h = (m / W) % H;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) mk |= 1u << t;
      }
    }
    mask[j] = m < Npix ? mk : 0u;
  }

  f32x4 xreg[NXU], wreg[NWU];                         // (xrows * XQ <= NXU * NT: host-checked)
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NXU; ++i) {
      xreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int u = tid + NT * i;
      if (u < xrows * XQ) {
        const int row = u / XQ, q = u % XQ;
        const long pix = (long)m0 - HALO + row;
        if (pix >= 0 && pix < Npix) xreg[i] = ld4(X + pix * Cin + chunk * KC + q * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < NWU; ++i) {
      wreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int u = tid + NT * i;
      if (u < BN * UPRW) {
        const int co = n0 + u / UPRW, f4 = u % UPRW;
        if (co < Cout) {
          if (full_w) {
            wreg[i] = ld4(Wh + ((long)co * I + chunk * KC) * TAPS + f4 * 4);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int f = f4 * 4 + e, ci = chunk * KC + f / TAPS;
              if (ci < I) wreg[i][e] = Wh[((long)co * I + ci) * TAPS + f % TAPS];
            }
          }
        }
      }
    }
  };
  auto store_chunk = [&](int buf) {
    float* xs = Xs + buf * KC * XR;
    float* ws = Ws + buf * TAPS * KC * BNP;
#pragma unroll
    for (int i = 0; i < NXU; ++i) {
      const int u = tid + NT * i;
      if (u < xrows * XQ) {
        const int row = u / XQ, q = u % XQ;
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[(q * 4 + e) * XR + row] = xreg[i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < NWU; ++i) {
      const int u = tid + NT * i;
      if (u < BN * UPRW) {
        const int col = u / UPRW, f4 = u % UPRW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = f4 * 4 + e;                   // flat (ci_local, tap) index of the master row
          ws[((f % TAPS) * KC + f / TAPS) * BNP + col] = wreg[i][e];
        }
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int nchunks = Cin / KC;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) load_chunk(c + 1);
    const float* xs = Xs + (c & 1) * KC * XR + wm * 64 + l31 + HALO;
    const float* ws = Ws + (c & 1) * TAPS * KC * BNP + wn * 64 + l31;
#pragma unroll 3
    for (int t = 0; t < TAPS; ++t) {
      const int shift = TAPS == 9 ? (t / 3 - 1) * W + (t % 3 - 1) : 0;
      const bool v0 = (mask[0] >> t) & 1, v1 = (mask[1] >> t) & 1;
#pragma unroll
      for (int ks = 0; ks < KC / 2; ++ks) {
        const int k = ks * 2 + lhi;
        float a0 = xs[k * XR + shift], a1 = xs[k * XR + 32 + shift];
        a0 = v0 ? a0 : 0.f;
        a1 = v1 ? a1 : 0.f;
        const float b0 = ws[(t * KC + k) * BNP], b1 = ws[(t * KC + k) * BNP + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    if (c + 1 < nchunks) {
      store_chunk((c + 1) & 1);   // the other buffer: its last readers finished before the previous barrier
      __syncthreads();
    }
  }

  // ---- epilogue: rows of the blocks = pixels 8 g + 4 lhi + r, columns = channels l31
  const float g = gain ? *gain : 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int co = n0 + wn * 64 + i * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long px = (long)m0 + wm * 64 + j * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
        if (px < Npix && co < Cout) {
          float v = alpha * acc[j][i][r];
          if (R) v += beta * R[px * Cout + co];
          if (lin) v = mp_silu_f(v * (lin[(px / HW) * lin_stride + co] * g + 1.0f));
          Y[px * Cout + co] = v;
        }
      }
    }
}

template <int TAPS, int KC, int WM>
void launch_conv(const float* X, const float* Wh, float* Y, const float* R, float alpha, float beta, const float* lin,
                 long lin_stride, const float* gain, int B, int H, int W, int Cin, int I, int Cout, hipStream_t st) {
  constexpr int BM = 64 * WM;
  const int Npix = B * H * W;
  const int tiles_m = (Npix + BM - 1) / BM, tiles_n = (Cout + BN - 1) / BN;
  const int xrows = BM + (TAPS == 9 ? 2 * (W + 1) : 0);
  const int XR = (xrows + 31) / 32 * 32 + 4;
  const size_t lds = ((size_t)2 * KC * XR + (size_t)2 * TAPS * KC * BNP) * sizeof(float);
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  auto kern = k_conv_f32<TAPS, KC, WM>;
  EDM_MAX_LDS(kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WM), lds, st, X, Wh, Y, R, alpha, beta, lin, lin_stride, gain, H * W,
                     Npix, H, W, Cin, I, Cout, tiles_m, tiles_n, XR);
}

// ------------------------------------------------------------------------------------------------ attention
// Any head_dim that is a multiple of 16, any token count.  FOUR threads share a query (each owns every fourth float4 of q
// and of the output accumulator; a logit is two lane shuffles away), a workgroup owns 64 queries of one (sample, head),
// and K / V stream through LDS in tiles of 64 keys, pixel-normalised as they are staged (four threads per key); online
// softmax in registers, plain fp32 fma chains.  The qkv tensor keeps the qkv conv's own channel order: channel
// head*3D + 3*dd + {0: q, 1: k, 2: v} (networks.py:194).  (A first form -- one query per thread, K and V of the whole
// head resident in LDS -- ran at 17 TF/s on the 16x16 layers and could not hold head_dim 144 / 192; this one: 30.)
template <int D>
__global__ __launch_bounds__(256) void k_attn_f32_g(const float* __restrict__ qkv, float* __restrict__ y, int N, int C,
                                                      int heads) {
  constexpr int TK = 64, NV = D / 16;      // keys per tile; float4s of q / acc per thread
  __shared__ __attribute__((aligned(16))) float Ks[TK * D];
  __shared__ __attribute__((aligned(16))) float Vs[TK * D];
  const int b = blockIdx.x / heads, head = blockIdx.x % heads;
  const float* base = qkv + ((long)b * N) * 3 * C + (long)head * 3 * D;
  const float rsd = 1.0f / sqrtf((float)D);
  const int part = threadIdx.x & 3, ql = threadIdx.x >> 2;
  const int qi = blockIdx.y * 64 + ql;
  const bool qok = qi < N;
  f32x4 q[NV], acc[NV];
  {
    const float* row = base + (long)(qok ? qi : 0) * 3 * C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int d0 = (i * 4 + part) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        q[i][e] = row[3 * (d0 + e)];
        ss += q[i][e] * q[i][e];
        acc[i][e] = 0.f;
      }
    }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    const float iq = rsd / (NORM_EPS + sqrtf(ss) * rsd);
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) q[i][e] *= iq;
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < N; k0 += TK) {
    __syncthreads();                       // the previous tile's readers are done
    {
      const int kj = k0 + ql;              // this thread stages its quarter of key kj (4 threads per key)
      const float* row = base + (long)(kj < N ? kj : 0) * 3 * C;
      f32x4 kv[NV], vv[NV];
      float ssk = 0.f, ssv = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int d0 = (i * 4 + part) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kv[i][e] = row[3 * (d0 + e) + 1];
          vv[i][e] = row[3 * (d0 + e) + 2];
          ssk += kv[i][e] * kv[i][e];
          ssv += vv[i][e] * vv[i][e];
        }
      }
      ssk += __shfl_xor(ssk, 1, 64);
      ssk += __shfl_xor(ssk, 2, 64);
      ssv += __shfl_xor(ssv, 1, 64);
      ssv += __shfl_xor(ssv, 2, 64);
      const float ik = 1.0f / (NORM_EPS + sqrtf(ssk) * rsd), iv = 1.0f / (NORM_EPS + sqrtf(ssv) * rsd);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int d0 = (i * 4 + part) * 4;
        f32x4 a = kv[i], c = vv[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] *= ik;
          c[e] *= iv;
        }
        st4(Ks + ql * D + d0, a);
        st4(Vs + ql * D + d0, c);
      }
    }
    __syncthreads();
    const int nk = min(TK, N - k0);
    for (int j = 0; j < nk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const f32x4 kk = ld4(Ks + j * D + (i * 4 + part) * 4);
        s += q[i][0] * kk[0] + q[i][1] * kk[1] + q[i][2] * kk[2] + q[i][3] * kk[3];
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), p = __expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const f32x4 vv = ld4(Vs + j * D + (i * 4 + part) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = acc[i][e] * corr + p * vv[e];
      }
      m = mn;
    }
  }
  if (qok) {
    const float il = 1.0f / l;
    float* dst = y + ((long)b * N + qi) * C + head * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 o = acc[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] *= il;
      st4(dst + (i * 4 + part) * 4, o);
    }
  }
}

// ------------------------------------------------------------------------------------------------ elementwise
// xn = x / (eps + |x| / sqrt(C)),  s = mp_silu(xn)        one wave per pixel
__global__ __launch_bounds__(256) void k_pnorm_silu_f32(const float* __restrict__ x, float* __restrict__ xn,
                                                          float* __restrict__ s, long P, int C, int s_pairs) {
  const int lane = threadIdx.x & 63;
  for (long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (long)gridDim.x * 4) {
    const float* xr = x + p * C;
    float ss = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 v = ld4(xr + c);
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / (NORM_EPS + sqrtf(ss) / sqrtf((float)C));
    for (int c = lane * 4; c < C; c += 256) {
      f32x4 v = ld4(xr + c);
      f32x4 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] *= inv;
        a[e] = mp_silu_f(v[e]);
      }
      st4(xn + p * C + c, v);
      if (s_pairs) st4_pairs(reinterpret_cast<bf16*>(s) + p * 2 * C, C, c, a);
      else st4(s + p * C + c, a);
    }
  }
}
__global__ void k_silu_f32(const float* __restrict__ x, float* __restrict__ s, long n4, int Cp) {   // Cp != 0: pairs out, rows of Cp
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = ld4(x + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = mp_silu_f(v[e]);
    if (Cp) {
      const long e0 = i * 4;
      st4_pairs(reinterpret_cast<bf16*>(s) + (e0 / Cp) * 2 * Cp, Cp, (int)(e0 % Cp), v);
    } else {
      st4(s + i * 4, v);
    }
  }
}
// y[b,h,w,:] = mean of the 2x2 block (H, W = output dims)
__global__ void k_pool2_f32(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long p = i / C4;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const long b = p / H;
    const long r0 = ((b * 2 * H + 2 * h) * 2 * W + 2 * w) * C4 + c4, rs = (long)2 * W * C4;
    const f32x4 a = ld4(x + r0 * 4), b1 = ld4(x + (r0 + C4) * 4), c = ld4(x + (r0 + rs) * 4), d = ld4(x + (r0 + rs + C4) * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = 0.25f * (a[e] + b1[e] + c[e] + d[e]);
    st4(y + i * 4, o);
  }
}
// nearest-exact x2 (H, W = output dims)
__global__ void k_up2_f32(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long p = i / C4;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const long b = p / H;
    st4(y + i * 4, ld4(x + (((b * (H / 2) + h / 2) * (W / 2) + w / 2) * C4 + c4) * 4));
  }
}
// ---- round 6: the launches that only existed because of where a tensor was materialised (the eval-shaped fusions) ----
// EncD blocks without a 1x1 conv (networks.py:246-252): 2x2 average pool -> pixel norm -> mp_silu in ONE pass; the pooled
// tensor is never written.  One wave per OUTPUT pixel; same arithmetic, in the same order, as k_pool2_f32 followed by
// k_pnorm_silu_f32 (bit-identical).  H, W = output dims; C <= 1024.
__global__ __launch_bounds__(256) void k_pool_pnorm_silu_f32(const float* __restrict__ x, float* __restrict__ xn,
                                                               float* __restrict__ s, long P, int H, int W, int C,
                                                               int s_pairs) {
  const int lane = threadIdx.x & 63;
  for (long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (long)gridDim.x * 4) {
    const int w = (int)(p % W);
    const long t = p / W;
    const int h = (int)(t % H);
    const long b = t / H;
    const float* r0 = x + ((b * 2 * H + 2 * h) * 2 * W + 2 * w) * C;
    const long rs = (long)2 * W * C;
    f32x4 v[4];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane * 4 + k * 256;
      if (c < C) {
        const f32x4 a = ld4(r0 + c), b1 = ld4(r0 + C + c), cc = ld4(r0 + rs + c), d = ld4(r0 + rs + C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k][e] = 0.25f * (a[e] + b1[e] + cc[e] + d[e]);
        ss += v[k][0] * v[k][0] + v[k][1] * v[k][1] + v[k][2] * v[k][2] + v[k][3] * v[k][3];
      }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / (NORM_EPS + sqrtf(ss) / sqrtf((float)C));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane * 4 + k * 256;
      if (c < C) {
        f32x4 o = v[k], a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] *= inv;
          a[e] = mp_silu_f(o[e]);
        }
        st4(xn + p * C + c, o);
        if (s_pairs) st4_pairs(reinterpret_cast<bf16*>(s) + p * 2 * C, C, c, a);
        else st4(s + p * C + c, a);
      }
    }
  }
}
// DecU blocks (networks.py:312-316): y = nearest-exact x2 of x, s = mp_silu(y) (fp32, or pairs with rows of Cp) in one pass
__global__ void k_up2_silu_f32(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ s, int H, int W,
                               int C4, long n4, int Cp) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long p = i / C4;
    const long pix = p;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const long b = p / H;
    f32x4 v = ld4(x + (((b * (H / 2) + h / 2) * (W / 2) + w / 2) * C4 + c4) * 4);
    st4(y + i * 4, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = mp_silu_f(v[e]);
    if (Cp) st4_pairs(reinterpret_cast<bf16*>(s) + pix * 2 * Cp, Cp, c4 * 4, v);
    else st4(s + i * 4, v);
  }
}
// the skip half of the decoder's concatenated operands, pairs form (the input half was written by the producer of `input`:
// edm_split_conv_o): cat[p, Ci + c] = skip[p, c] * gate[b, c], sil[p, Ci + c] = mp_silu of it; rows [hi(Ct) | lo(Ct)]
__global__ void k_skip_half_f32(const float* __restrict__ skip, const float* __restrict__ gate, bf16* __restrict__ catp,
                                bf16* __restrict__ silp, int HW, int Ci, int Cs, long n4) {
  const int CLs = Cs >> 2, Ct = Ci + Cs;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int cs = (int)(i % CLs) * 4;
    const long pix = i / CLs;
    f32x4 v = ld4(skip + pix * Cs + cs);
    const f32x4 gp = ld4(gate + (pix / HW) * Cs + cs);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= gp[e];
    st4_pairs(catp + pix * 2 * Ct, Ct, Ci + cs, v);
    if (silp) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mp_silu_f(v[e]);
      st4_pairs(silp + pix * 2 * Ct, Ct, Ci + cs, v);
    }
  }
}
// per-sample mean over H*W (fixed order) + the ScaleLong gate MLP, one workgroup per sample (as k_skip_gate_fwd)
__global__ __launch_bounds__(1024) void k_skip_gate_f32(const float* __restrict__ skip, const float* __restrict__ W1,
                                                          const float* __restrict__ W2, float* __restrict__ gate, int HW,
                                                          int C, int R) {
  extern __shared__ __attribute__((aligned(16))) float smg[];   // red[rpp*C] | m[C+1] | h[R]
  const int lpr = C >> 2, rpp = blockDim.x / lpr;
  float* red = smg;
  float* m = smg + rpp * C;
  float* h = m + C + 1;
  const int b = blockIdx.x;
  const int cl = threadIdx.x % lpr, rg = threadIdx.x / lpr;
  if (rg < rpp) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int p = rg; p < HW; p += rpp) {
      const f32x4 v = ld4(skip + ((long)b * HW + p) * C + cl * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rg * C + cl * 4 + e] = a[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < rpp; ++q) s += red[q * C + c];
    m[c] = s / (float)HW;
  }
  if (threadIdx.x == 0) m[C] = 1.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < R; r += nw) {
    float s = 0.f;
    for (int c = lane; c <= C; c += 64) s += W1[(long)r * (C + 1) + c] * m[c];
    s = wave_sum(s);
    if (lane == 0) h[r] = mp_silu_f(s);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += W2[(long)c * R + r] * h[r];
    gate[(long)b * C + c] = sigmoidf_(s);
  }
}
// cat = [inp, skip*gate[b]] ; sil = mp_silu(cat) (optional)
__global__ void k_concat_gate_f32(const float* __restrict__ inp, const float* __restrict__ skip,
                                  const float* __restrict__ gate, float* __restrict__ cat, float* __restrict__ sil, int HW,
                                  int Ci, int Cs, long n4, int pairs) {
  const int CLt = (Ci + Cs) >> 2, CLi = Ci >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % CLt);
    const long pix = i / CLt;
    f32x4 v;
    if (c4 < CLi) {
      v = ld4(inp + pix * Ci + c4 * 4);
    } else {
      const int cs = (c4 - CLi) * 4;
      v = ld4(skip + pix * Cs + cs);
      const f32x4 gp = ld4(gate + (pix / HW) * Cs + cs);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= gp[e];
    }
    const int Ct = Ci + Cs;
    if (pairs) st4_pairs(reinterpret_cast<bf16*>(cat) + pix * 2 * Ct, Ct, c4 * 4, v);
    else st4(cat + i * 4, v);
    if (sil) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mp_silu_f(v[e]);
      if (pairs) st4_pairs(reinterpret_cast<bf16*>(sil) + pix * 2 * Ct, Ct, c4 * 4, v);
      else st4(sil + i * 4, v);
    }
  }
}
// out[b,h,w,:] = [c_in(b) * noisy[b,:,h,w], 1, 0 ...]   (CP channels)
__global__ void k_precond_in_f32(const float* __restrict__ noisy, const float* __restrict__ sigma, int sstride, float sd,
                                 float* __restrict__ out, int Cimg, int HW, int CP, long npix) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long b = p / HW;
    const int hw = (int)(p % HW);
    const float s = sigma[b * sstride];
    const float cin = 1.0f / sqrtf(sd * sd + s * s);
    for (int c = 0; c < CP; ++c)
      out[p * CP + c] = c < Cimg ? cin * noisy[(b * Cimg + c) * HW + hw] : (c == Cimg ? 1.0f : 0.0f);
  }
}
// D[b,o,hw] = (sum_c x[p,c] wh[o,c]) * gain_out * c_out(b) + noisy * c_skip(b)       one wave per pixel, Co <= 8
__global__ __launch_bounds__(256) void k_conv_out_f32(const float* __restrict__ x, const float* __restrict__ wh,
                                                        const float* __restrict__ gain_out,
                                                        const float* __restrict__ noisy, const float* __restrict__ sigma,
                                                        int sstride, float sd, float* __restrict__ Dn, int HW, int C,
                                                        int Co, long npix) {
  // (round 6: 16 lanes per pixel, four pixels per wave trip -- one wave per pixel spent 48 shuffles on 1 KB of input and ran at
  // a third of the HBM rate: 336 us per evaluation at batch 512)
  constexpr int LPP = 16, GPW = 64 / LPP;
  const int lane = threadIdx.x & 63, lig = lane % LPP, grp = lane / LPP;
  const float go = *gain_out;
  const long stride = (long)gridDim.x * 4 * GPW;
  for (long p0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW; p0 < npix; p0 += stride) {
    const long p = p0 + grp;
    const bool pv = p < npix;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pv) {
      for (int c = lig * 4; c < C; c += 4 * LPP) {
        const f32x4 v = ld4(x + p * C + c);
#pragma unroll
        for (int o = 0; o < 8; ++o)
          if (o < Co) {
            const f32x4 w = ld4(wh + (long)o * C + c);
            acc[o] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
          }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < Co) acc[o] = group_sum<LPP>(acc[o]);
    if (pv && lig == 0) {
      const long b = p / HW;
      const int hw = (int)(p % HW);
      const float s = sigma[b * sstride];
      const float den = s * s + sd * sd;
      const float cskip = sd * sd / den, cout = s * sd / sqrtf(den);
      for (int o = 0; o < Co; ++o) {
        const long idx = (b * Co + o) * HW + hw;
        Dn[idx] = acc[o] * go * cout + noisy[idx] * cskip;
      }
    }
  }
}
// NHWC f32 <-> NCHW f32 (module-boundary calls of the reference API only)
__global__ void k_nchw_to_nhwc_f32(const float* __restrict__ x, float* __restrict__ y, int C, int HW, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long p = i / C;
    y[i] = x[((p / HW) * C + c) * HW + p % HW];
  }
}
__global__ void k_nhwc_to_nchw_f32(const float* __restrict__ x, float* __restrict__ y, int C, int HW, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int hw = (int)(i % HW);
    const long bc = i / HW;
    y[i] = x[((bc / C) * HW + hw) * C + bc % C];
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ split-bf16 operands
// An fp32 value as a PAIR of bf16: hi = bf16(x) (round to nearest even), lo = bf16(x - hi): x = hi + lo up to 2^-18 |x|.
// Activations: [rows][C] floats -> [rows][2 C] bf16 = [hi | lo] (the same bytes).  Weights: w_hat [O][I * taps] (master
// order) -> [taps][O][3 Ip] = [w_hi | w_lo | w_hi].  edm_split_conv (conv_dispatch.hip) multiplies them in three bf16 passes.
namespace {
__global__ void k_f32_to_pairs(const float* __restrict__ x, bf16* __restrict__ p, int C, long n8) {
  const int CL = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / CL;
    const int c8 = (int)(i % CL) * 8;
    const f32x4 a = ld4(x + i * 8), b = ld4(x + i * 8 + 4);
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = j < 4 ? a[j] : b[j - 4];
      const bf16 h = (bf16)v;
      hi[j] = h;
      lo[j] = (bf16)(v - (float)h);
    }
    *reinterpret_cast<bf16x8*>(p + row * 2 * C + c8) = hi;
    *reinterpret_cast<bf16x8*>(p + row * 2 * C + C + c8) = lo;
  }
}
__global__ void k_split_pack(const float* __restrict__ wh, bf16* __restrict__ pk, int O, int I, int taps, int Ip, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int i = (int)(e % Ip);
    const long to = e / Ip;
    const int o = (int)(to % O), t = (int)(to / O);
    const float v = i < I ? wh[((long)o * I + i) * taps + t] : 0.f;
    const bf16 h = (bf16)v, l = (bf16)(v - (float)h);
    bf16* row = pk + to * 3 * Ip;
    row[i] = h;
    row[Ip + i] = l;
    row[2 * Ip + i] = h;
  }
}
}  // namespace
extern "C" int edm_f32_to_pairs(const float* x, void* pairs, long rows, int C, hipStream_t st) {
  EDM_REQUIRE(x && pairs && rows > 0 && C > 0 && C % 8 == 0, "f32_to_pairs: bad args (C %% 8)");
  const long n8 = rows * C / 8;
  hipLaunchKernelGGL(k_f32_to_pairs, dim3(gridf(n8, 256)), dim3(256), 0, st, x, (bf16*)pairs, C, n8);
  EDM_CHECK_LAUNCH("f32_to_pairs");
  return EDM_OK;
}
// w_hat [O][I * taps] fp32 -> pack [taps][O][3 Ip] bf16, Ip >= I (channels I .. Ip-1 zero)
extern "C" int edm_split_pack(const float* w_hat, void* pack, int O, int I, int taps, int Ip, hipStream_t st) {
  EDM_REQUIRE(w_hat && pack && O > 0 && I > 0 && taps > 0 && Ip >= I, "split_pack: bad args");
  const long n = (long)taps * O * Ip;
  hipLaunchKernelGGL(k_split_pack, dim3(gridf(n, 256)), dim3(256), 0, st, w_hat, (bf16*)pack, O, I, taps, Ip, n);
  EDM_CHECK_LAUNCH("split_pack");
  return EDM_OK;
}

// X [B*H*W][Cin] fp32, w_hat [Cout][I*taps] fp32 (master OIHW order; I <= Cin: X may be zero-padded), Y / R [B*H*W][Cout].
// lin != NULL: Y = mp_silu(conv * (lin[b,:]*gain + 1)) (alpha/beta/R ignored except alpha).  taps in {1, 9}; W <= 64.
extern "C" int edm_f32_conv(const float* X, const float* w_hat, float* Y, const float* R, float alpha, float beta,
                            const float* lin, long lin_stride, const float* gain, int B, int H, int W, int Cin, int I,
                            int Cout, int taps, hipStream_t st) {
  EDM_REQUIRE(X && w_hat && Y, "f32_conv: null pointer");
  EDM_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "f32_conv: bad B/H/W");
  EDM_REQUIRE(taps == 1 || taps == 9, "f32_conv: taps must be 1 or 9");
  EDM_REQUIRE(Cout > 0 && I > 0 && I <= Cin, "f32_conv: bad channel counts");
  EDM_REQUIRE(!lin || (gain && lin_stride >= Cout), "f32_conv: modulation epilogue needs gain and lin_stride >= Cout");
  EDM_REQUIRE(taps == 1 || W <= 64, "f32_conv: W > 64 unsupported for 3x3");
  // the 256-pixel tile when it gives (nearly) every CU a workgroup, else the 128-pixel tile
  const long big = (((long)B * H * W + 255) / 256) * ((Cout + BN - 1) / BN);
  if (taps == 9) {
    EDM_REQUIRE(Cin % 8 == 0, "f32_conv: Cin %% 8 required (3x3)");
    if (big >= 200) launch_conv<9, 8, 4>(X, w_hat, Y, R, alpha, beta, lin, lin_stride, gain, B, H, W, Cin, I, Cout, st);
    else launch_conv<9, 8, 2>(X, w_hat, Y, R, alpha, beta, lin, lin_stride, gain, B, H, W, Cin, I, Cout, st);
  } else {
    EDM_REQUIRE(Cin % 32 == 0, "f32_conv: Cin %% 32 required (1x1)");
    if (big >= 200) launch_conv<1, 32, 4>(X, w_hat, Y, R, alpha, beta, lin, lin_stride, gain, B, H, W, Cin, I, Cout, st);
    else launch_conv<1, 32, 2>(X, w_hat, Y, R, alpha, beta, lin, lin_stride, gain, B, H, W, Cin, I, Cout, st);
  }
  EDM_CHECK_LAUNCH("f32_conv");
  return EDM_OK;
}

// qkv [B*N][3C] fp32 in the qkv conv's own channel order (head*3d + 3*dd + {q,k,v}) -> y [B*N][C] fp32 (head*d + dd).
// head_dim in {32, 64, 128, 144, 192} (every configuration of the reference); any number of tokens.
extern "C" int edm_f32_attention(const float* qkv, float* y, int B, int N, int C, int heads, hipStream_t st) {
  EDM_REQUIRE(qkv && y && B > 0 && N > 0 && heads > 0 && C > 0 && C % heads == 0, "f32_attention: bad args");
  const int D = C / heads;
#define ATTG(DV) \
  hipLaunchKernelGGL(k_attn_f32_g<DV>, dim3(B * heads, (N + 63) / 64), dim3(256), 0, st, qkv, y, N, C, heads)
  if (D == 32) ATTG(32);
  else if (D == 64) ATTG(64);
  else if (D == 128) ATTG(128);
  else if (D == 144) ATTG(144);
  else if (D == 192) ATTG(192);
  else {
    edm_set_error("f32_attention: head_dim %d is not built (32, 64, 128, 144, 192 are)", D);
    return EDM_ERR_UNSUPPORTED;
  }
#undef ATTG
  EDM_CHECK_LAUNCH("f32_attention");
  return EDM_OK;
}

extern "C" int edm_f32_pixelnorm_silu(const float* x, float* xn, float* s, long P, int C, int s_pairs, hipStream_t st) {
  EDM_REQUIRE(x && xn && s && P > 0 && C > 0 && C % 4 == 0, "f32_pixelnorm_silu: bad args");
  hipLaunchKernelGGL(k_pnorm_silu_f32, dim3(gridf(P, 4)), dim3(256), 0, st, x, xn, s, P, C, s_pairs);
  EDM_CHECK_LAUNCH("f32_pixelnorm_silu");
  return EDM_OK;
}
extern "C" int edm_f32_silu(const float* x, float* s, long n, int pairs_row, hipStream_t st) {
  EDM_REQUIRE(x && s && n > 0 && n % 4 == 0 && (pairs_row == 0 || (pairs_row % 4 == 0 && n % pairs_row == 0)), "f32_silu: bad args");
  hipLaunchKernelGGL(k_silu_f32, dim3(gridf(n / 4, 256)), dim3(256), 0, st, x, s, n / 4, pairs_row);
  EDM_CHECK_LAUNCH("f32_silu");
  return EDM_OK;
}
extern "C" int edm_f32_pool2(const float* x, float* y, int B, int Hout, int Wout, int C, hipStream_t st) {
  EDM_REQUIRE(x && y && B > 0 && Hout > 0 && Wout > 0 && C > 0 && C % 4 == 0, "f32_pool2: bad args");
  const long n4 = (long)B * Hout * Wout * C / 4;
  hipLaunchKernelGGL(k_pool2_f32, dim3(gridf(n4, 256)), dim3(256), 0, st, x, y, Hout, Wout, C / 4, n4);
  EDM_CHECK_LAUNCH("f32_pool2");
  return EDM_OK;
}
extern "C" int edm_f32_up2(const float* x, float* y, int B, int Hout, int Wout, int C, hipStream_t st) {
  EDM_REQUIRE(x && y && B > 0 && Hout > 0 && Wout > 0 && Hout % 2 == 0 && Wout % 2 == 0 && C > 0 && C % 4 == 0,
              "f32_up2: bad args");
  const long n4 = (long)B * Hout * Wout * C / 4;
  hipLaunchKernelGGL(k_up2_f32, dim3(gridf(n4, 256)), dim3(256), 0, st, x, y, Hout, Wout, C / 4, n4);
  EDM_CHECK_LAUNCH("f32_up2");
  return EDM_OK;
}
extern "C" int edm_f32_pool_pixelnorm_silu(const float* x, float* xn, float* s, int B, int Hout, int Wout, int C,
                                           int s_pairs, hipStream_t st) {
  EDM_REQUIRE(x && xn && s && B > 0 && Hout > 0 && Wout > 0 && C > 0 && C % 4 == 0 && C <= 1024,
              "f32_pool_pixelnorm_silu: bad args (C %% 4, C <= 1024)");
  const long P = (long)B * Hout * Wout;
  hipLaunchKernelGGL(k_pool_pnorm_silu_f32, dim3(gridf(P, 4)), dim3(256), 0, st, x, xn, s, P, Hout, Wout, C, s_pairs);
  EDM_CHECK_LAUNCH("f32_pool_pixelnorm_silu");
  return EDM_OK;
}
extern "C" int edm_f32_up2_silu(const float* x, float* y, float* s, int B, int Hout, int Wout, int C, int s_pairs,
                                hipStream_t st) {
  EDM_REQUIRE(x && y && s && B > 0 && Hout > 0 && Wout > 0 && Hout % 2 == 0 && Wout % 2 == 0 && C > 0 && C % 4 == 0,
              "f32_up2_silu: bad args");
  const long n4 = (long)B * Hout * Wout * C / 4;
  hipLaunchKernelGGL(k_up2_silu_f32, dim3(gridf(n4, 256)), dim3(256), 0, st, x, y, s, Hout, Wout, C / 4, n4, s_pairs ? C : 0);
  EDM_CHECK_LAUNCH("f32_up2_silu");
  return EDM_OK;
}
extern "C" int edm_f32_skip_half(const float* skip, const float* gate, void* cat_pairs, void* silu_pairs, int B, int HW,
                                 int Ci, int Cs, hipStream_t st) {
  EDM_REQUIRE(skip && gate && cat_pairs && B > 0 && HW > 0 && Ci > 0 && Cs > 0 && Ci % 4 == 0 && Cs % 4 == 0,
              "f32_skip_half: bad args");
  const long n4 = (long)B * HW * Cs / 4;
  hipLaunchKernelGGL(k_skip_half_f32, dim3(gridf(n4, 256)), dim3(256), 0, st, skip, gate, (bf16*)cat_pairs,
                     (bf16*)silu_pairs, HW, Ci, Cs, n4);
  EDM_CHECK_LAUNCH("f32_skip_half");
  return EDM_OK;
}
extern "C" int edm_f32_skip_gate(const float* skip, const float* W1h, const float* W2h, float* gate, int B, int HW, int C,
                                 int R, hipStream_t st) {
  EDM_REQUIRE(skip && W1h && W2h && gate && B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 4096 && R > 0 && R <= 1024,
              "f32_skip_gate: bad args");
  const int lpr = C / 4;
  const int threads = 1024 / lpr * lpr;
  EDM_REQUIRE(threads >= 64, "f32_skip_gate: C too large");
  const size_t lds = ((size_t)(threads / lpr) * C + C + 1 + R) * sizeof(float);
  hipLaunchKernelGGL(k_skip_gate_f32, dim3(B), dim3(threads), lds, st, skip, W1h, W2h, gate, HW, C, R);
  EDM_CHECK_LAUNCH("f32_skip_gate");
  return EDM_OK;
}
extern "C" int edm_f32_concat_gate(const float* inp, const float* skip, const float* gate, float* cat, float* silu_out,
                                   int B, int HW, int Ci, int Cs, int pairs, hipStream_t st) {
  EDM_REQUIRE(inp && skip && gate && cat && B > 0 && HW > 0 && Ci > 0 && Cs > 0 && Ci % 4 == 0 && Cs % 4 == 0,
              "f32_concat_gate: bad args");
  const long n4 = (long)B * HW * (Ci + Cs) / 4;
  hipLaunchKernelGGL(k_concat_gate_f32, dim3(gridf(n4, 256)), dim3(256), 0, st, inp, skip, gate, cat, silu_out, HW, Ci,
                     Cs, n4, pairs);
  EDM_CHECK_LAUNCH("f32_concat_gate");
  return EDM_OK;
}
extern "C" int edm_f32_precond_in(const float* noisy, const float* sigma, int sigma_stride, float sigma_data, float* out,
                                  int B, int Cimg, int HW, int CP, hipStream_t st) {
  EDM_REQUIRE(noisy && sigma && out && B > 0 && Cimg > 0 && HW > 0 && CP > Cimg && (sigma_stride == 0 || sigma_stride == 1),
              "f32_precond_in: bad args");
  const long npix = (long)B * HW;
  hipLaunchKernelGGL(k_precond_in_f32, dim3(gridf(npix, 256)), dim3(256), 0, st, noisy, sigma, sigma_stride, sigma_data,
                     out, Cimg, HW, CP, npix);
  EDM_CHECK_LAUNCH("f32_precond_in");
  return EDM_OK;
}
extern "C" int edm_f32_conv_out(const float* x, const float* w_hat, const float* gain_out, const float* noisy,
                                const float* sigma, int sigma_stride, float sigma_data, float* D, int B, int HW, int C,
                                int Co, hipStream_t st) {
  EDM_REQUIRE(x && w_hat && gain_out && noisy && sigma && D && B > 0 && HW > 0 && C % 4 == 0 && Co >= 1 && Co <= 8 &&
                  (sigma_stride == 0 || sigma_stride == 1),
              "f32_conv_out: bad args (Co <= 8 required)");
  const long npix = (long)B * HW;
  hipLaunchKernelGGL(k_conv_out_f32, dim3(gridf(npix, 16)), dim3(256), 0, st, x, w_hat, gain_out, noisy, sigma,
                     sigma_stride, sigma_data, D, HW, C, Co, npix);
  EDM_CHECK_LAUNCH("f32_conv_out");
  return EDM_OK;
}
extern "C" int edm_f32_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, hipStream_t st) {
  EDM_REQUIRE(x && y && B > 0 && C > 0 && HW > 0, "f32_nchw_to_nhwc: bad args");
  const long n = (long)B * C * HW;
  hipLaunchKernelGGL(k_nchw_to_nhwc_f32, dim3(gridf(n, 256)), dim3(256), 0, st, x, y, C, HW, n);
  EDM_CHECK_LAUNCH("f32_nchw_to_nhwc");
  return EDM_OK;
}
extern "C" int edm_f32_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, hipStream_t st) {
  EDM_REQUIRE(x && y && B > 0 && C > 0 && HW > 0, "f32_nhwc_to_nchw: bad args");
  const long n = (long)B * C * HW;
  hipLaunchKernelGGL(k_nhwc_to_nchw_f32, dim3(gridf(n, 256)), dim3(256), 0, st, x, y, C, HW, n);
  EDM_CHECK_LAUNCH("f32_nhwc_to_nchw");
  return EDM_OK;
}

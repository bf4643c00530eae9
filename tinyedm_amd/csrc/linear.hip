// fp32 side path: the weight-normalised Linear layers (networks.py:58-60) and the noise / class
// embedding (networks.py:121-178).  The reference keeps these in fp32 under autocast
// (networks.py:164, 255, 319); they are tiny ((B,64..1000)->(B,256..768)): one LDS-tiled fp32 GEMM on the
// f32-input matrix instruction, with arbitrary operand strides, covers forward, dgrad and wgrad.
#include "common.h"

namespace {

// C[m,n] (=|+=) alpha * sum_k A[m*asm + k*ask] * B[k*bsk + n*bsn]
// Round 6: the same GEMM on the f32-input matrix instruction (v_mfma_f32_32x32x2_f32: every product and sum a plain fp32
// operation, as in csrc/eval_f32.hip).  k_sgemm_smallk staged the whole K extent of BOTH operands per 32x32 tile -- the
// batched embed Linear of a sampler evaluation (512 x 256 -> 5 376: networks._EmbedAllFn) re-read 176 MB through L2 and took
// 121 us; this kernel gives a wave a 32x32 output block, a workgroup WM x WN of them, and walks K in chunks of 32 through LDS
// (operands of any stride: forward, dgrad and wgrad of the Linears share it).  Split-K over gridDim.z, partial sums by atomics.
template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_sgemm_mfma(const float* __restrict__ A, long asm_, long ask,
                                                               const float* __restrict__ Bm, long bsk, long bsn,
                                                               float* __restrict__ C, long csm, long csn, int M, int N, int K,
                                                               float alpha, int accumulate) {
  constexpr int TM = 32 * WM, TN = 32 * WN, TK = 32, NT = 64 * WM * WN;
  constexpr int LDA = TM + 1, LDB = TN + 1;        // odd row pitch: the k-contiguous staging writes hit distinct banks
  constexpr int NA_ = TM * TK / NT, NB_ = TN * TK / NT;   // elements of a chunk per thread (compile-time: 8 / 8 or 8 / 16)
  __shared__ float As[TK * LDA];
  __shared__ float Bs[TK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int kper = ((K + gridDim.z - 1) / gridDim.z + TK - 1) / TK * TK;
  const int kbeg = blockIdx.z * kper;
  const int kend = min(K, kbeg + kper);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // A chunk's elements go global -> registers -> LDS, ALL loads of a chunk issued before the first is stored (the first form
  // of this kernel stored each element behind its own load: 24 serial round trips per chunk, 51 us per call whatever the
  // problem size), and the NEXT chunk's loads are in flight while this chunk is multiplied.
  float ra[NA_], rb[NB_];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA_; ++i) {
      const int e = tid + i * NT;
      const int kk = ask == 1 ? (e & 31) : e / TM, mm = ask == 1 ? (e >> 5) : e % TM;
      const int m = m0 + mm, k = k0 + kk;
      ra[i] = (m < M && k < kend) ? A[m * asm_ + k * ask] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NB_; ++i) {
      const int e = tid + i * NT;
      const int kk = bsk == 1 ? (e & 31) : e / TN, nn = bsk == 1 ? (e >> 5) : e % TN;
      const int n = n0 + nn, k = k0 + kk;
      rb[i] = (n < N && k < kend) ? Bm[k * bsk + n * bsn] : 0.f;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NA_; ++i) {
      const int e = tid + i * NT;
      const int kk = ask == 1 ? (e & 31) : e / TM, mm = ask == 1 ? (e >> 5) : e % TM;
      As[kk * LDA + mm] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB_; ++i) {
      const int e = tid + i * NT;
      const int kk = bsk == 1 ? (e & 31) : e / TN, nn = bsk == 1 ? (e >> 5) : e % TN;
      Bs[kk * LDB + nn] = rb[i];
    }
  };
  if (kbeg < kend) load_chunk(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += TK) {
    store_chunk();
    __syncthreads();
    if (k0 + TK < kend) load_chunk(k0 + TK);
#pragma unroll
    for (int kk = 0; kk < TK; kk += 2) {
      const float a = As[(kk + lhi) * LDA + wm * 32 + l31];
      const float b = Bs[(kk + lhi) * LDB + wn * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // accumulator register r of a lane: row (of A) 8 (r / 4) + 4 lhi + (r % 4), column (of B) l31
  const int n = n0 + wn * 32 + l31;
  if (n >= N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
    if (m < M) {
      float* c = C + m * csm + n * csn;
      if (gridDim.z > 1) atomicAdd(c, alpha * acc[r]);
      else *c = accumulate ? *c + alpha * acc[r] : alpha * acc[r];
    }
  }
}

int sgemm(const float* A, long asm_, long ask, const float* B, long bsk, long bsn, float* C, long csm, long csn, int M,
          int N, int K, float alpha, int accumulate, hipStream_t st) {
  // tile: 64x64 (four waves) when that still gives every CU a workgroup, 32x64 (two waves) for the small problems
  const long t64 = (long)((N + 63) / 64) * ((M + 63) / 64);
  const bool big = t64 >= 256;
  const long tiles = big ? t64 : (long)((N + 63) / 64) * ((M + 31) / 32);
  int splits = 1;
  if (K >= 1024 && tiles < 512 && csn == 1 && csm == N) {  // long-K, few tiles: split K to fill the chip
    long sp = K / 256, cap = (1024 + tiles - 1) / tiles;
    if (sp > cap) sp = cap;
    if (sp > 32) sp = 32;
    splits = (int)(sp < 1 ? 1 : sp);
    if (splits > 1 && !accumulate) (void)hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), st);
  }
  if (big)
    hipLaunchKernelGGL((k_sgemm_mfma<2, 2>), dim3((N + 63) / 64, (M + 63) / 64, splits), dim3(256), 0, st, A, asm_, ask, B,
                       bsk, bsn, C, csm, csn, M, N, K, alpha, accumulate);
  else
    hipLaunchKernelGGL((k_sgemm_mfma<1, 2>), dim3((N + 63) / 64, (M + 31) / 32, splits), dim3(128), 0, st, A, asm_, ask, B,
                       bsk, bsn, C, csm, csn, M, N, K, alpha, accumulate);
  return 0;
}

// fourier[b,j] = cos(ln(sigma_b)/4 * freqs[j] + phases[j]) * sqrt(2)      (networks.py:138-141,165)
__global__ void k_fourier(const float* __restrict__ sigma, int sstride, const float* __restrict__ freqs,
                          const float* __restrict__ phases, float* __restrict__ out, int B, int Fd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Fd) return;
  const int b = i / Fd, j = i % Fd;
  const float c = logf(sigma[b * sstride]) * 0.25f;
  out[i] = cosf(c * freqs[j] + phases[j]) * 1.41421356237309515f;
}

// pre = labels ? mp_add(emb_sigma, Wcls_hat[:,label]*sqrt(K), t) : emb_sigma ; out = mp_silu(pre)   (networks.py:169-177)
__global__ void k_embed_combine_fwd(const float* __restrict__ es, const float* __restrict__ wcls,
                                    const long long* __restrict__ labels, float t, int K, float* __restrict__ pre,
                                    float* __restrict__ out, int B, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * E) return;
  const int b = i / E, e = i % E;
  float v = es[i];
  if (labels) {
    const float c = rsqrtf((1.f - t) * (1.f - t) + t * t);
    const long long lab = labels[b];
    const float cls = (lab >= 0 && lab < K) ? wcls[(long)e * K + lab] * sqrtf((float)K) : 0.f;
    v = ((1.f - t) * v + t * cls) * c;
  }
  pre[i] = v;
  out[i] = mp_silu_f(v);
}
__global__ void k_embed_combine_bwd(const float* __restrict__ gout, const float* __restrict__ pre,
                                    const long long* __restrict__ labels, float t, int K, float* __restrict__ ges,
                                    float* __restrict__ gwcls, int B, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * E) return;
  const int b = i / E, e = i % E;
  float g = gout[i] * mp_silu_grad_f(pre[i]);
  if (labels) {
    const float c = rsqrtf((1.f - t) * (1.f - t) + t * t);
    const long long lab = labels[b];
    if (lab >= 0 && lab < K) atomicAdd(gwcls + (long)e * K + lab, g * t * c * sqrtf((float)K));
    g *= (1.f - t) * c;
  }
  ges[i] = g;
}

}  // namespace

// Y[M,N] = X[M,K] W[N,K]^T   (weight-normalised Linear with the effective fp32 weight)
extern "C" int edm_linear_fwd(const float* X, const float* W, float* Y, int M, int N, int K, hipStream_t st) {
  EDM_REQUIRE(X && W && Y && M > 0 && N > 0 && K > 0, "linear_fwd: bad args");
  sgemm(X, K, 1, W, 1, K, Y, N, 1, M, N, K, 1.f, 0, st);
  EDM_CHECK_LAUNCH("linear_fwd");
  return EDM_OK;
}
// dX[M,K] = dY[M,N] W[N,K]
extern "C" int edm_linear_dgrad(const float* dY, const float* W, float* dX, int M, int N, int K, int accumulate,
                                hipStream_t st) {
  EDM_REQUIRE(dY && W && dX && M > 0 && N > 0 && K > 0, "linear_dgrad: bad args");
  sgemm(dY, N, 1, W, K, 1, dX, K, 1, M, K, N, 1.f, accumulate, st);
  EDM_CHECK_LAUNCH("linear_dgrad");
  return EDM_OK;
}
// dW[N,K] = dY[M,N]^T X[M,K]
extern "C" int edm_linear_wgrad(const float* dY, const float* X, float* dW, int M, int N, int K, int accumulate,
                                hipStream_t st) {
  EDM_REQUIRE(dY && X && dW && M > 0 && N > 0 && K > 0, "linear_wgrad: bad args");
  sgemm(dY, 1, N, X, K, 1, dW, K, 1, N, K, M, 1.f, accumulate, st);
  EDM_CHECK_LAUNCH("linear_wgrad");
  return EDM_OK;
}

extern "C" int edm_fourier_fwd(const float* sigma, int sigma_stride, const float* freqs, const float* phases,
                               float* out, int B, int Fd, hipStream_t st) {
  EDM_REQUIRE(sigma && freqs && phases && out && B > 0 && Fd > 0 && (sigma_stride == 0 || sigma_stride == 1),
              "fourier_fwd: bad args");
  hipLaunchKernelGGL(k_fourier, dim3((B * Fd + 255) / 256), dim3(256), 0, st, sigma, sigma_stride, freqs, phases, out,
                     B, Fd);
  EDM_CHECK_LAUNCH("fourier_fwd");
  return EDM_OK;
}
extern "C" int edm_embed_combine_fwd(const float* emb_sigma, const float* wcls_hat, const long long* labels,
                                     float add_factor, int K, float* pre, float* out, int B, int E, hipStream_t st) {
  EDM_REQUIRE(emb_sigma && pre && out && B > 0 && E > 0 && (!labels || (wcls_hat && K > 0)), "embed_combine_fwd: bad args");
  hipLaunchKernelGGL(k_embed_combine_fwd, dim3((B * E + 255) / 256), dim3(256), 0, st, emb_sigma, wcls_hat, labels,
                     add_factor, K, pre, out, B, E);
  EDM_CHECK_LAUNCH("embed_combine_fwd");
  return EDM_OK;
}
// gwcls_hat [E,K] is accumulated (+=): caller zero-fills.
extern "C" int edm_embed_combine_bwd(const float* gout, const float* pre, const long long* labels, float add_factor,
                                     int K, float* gemb_sigma, float* gwcls_hat, int B, int E, hipStream_t st) {
  EDM_REQUIRE(gout && pre && gemb_sigma && B > 0 && E > 0 && (!labels || (gwcls_hat && K > 0)), "embed_combine_bwd: bad args");
  hipLaunchKernelGGL(k_embed_combine_bwd, dim3((B * E + 255) / 256), dim3(256), 0, st, gout, pre, labels, add_factor, K,
                     gemb_sigma, gwcls_hat, B, E);
  EDM_CHECK_LAUNCH("embed_combine_bwd");
  return EDM_OK;
}

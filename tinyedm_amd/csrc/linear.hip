// fp32 side path: the weight-normalised Linear layers (networks.py:58-60) and the noise / class
// embedding (networks.py:121-178).  The reference keeps these in fp32 under autocast
// (networks.py:164, 255, 319); they are tiny ((B,64..1000)->(B,256..768)), so a plain LDS-tiled
// fp32 FMA GEMM with arbitrary strides covers forward, dgrad and wgrad.
#include "common.h"

namespace {

// C[m,n] (=|+=) alpha * sum_k A[m*asm + k*ask] * B[k*bsk + n*bsn]
__global__ __launch_bounds__(256) void k_sgemm(const float* __restrict__ A, long asm_, long ask,
                                                 const float* __restrict__ Bm, long bsk, long bsn,
                                                 float* __restrict__ C, long csm, long csn, int M, int N, int K,
                                                 float alpha, int accumulate) {
  constexpr int TM = 32, TN = 32, TK = 32;
  __shared__ float As[TK][TM + 1];
  __shared__ float Bs[TK][TN + 1];
  // 256 threads: 16x16 threads x (2x2 outputs) = 32x32 tile (small tiles: these GEMMs are tiny, parallelism first)
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  float acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = 0.f;
  // split-K over gridDim.z (partial sums are combined with float atomics; C pre-zeroed by the launcher)
  const int kper = ((K + gridDim.z - 1) / gridDim.z + TK - 1) / TK * TK;
  const int kbeg = blockIdx.z * kper;
  const int kend = min(K, kbeg + kper);
  K = kend;
  for (int k0 = kbeg; k0 < K; k0 += TK) {
    for (int e = threadIdx.x; e < TM * TK; e += 256) {
      const int kk = e % TK, mm = e / TK;
      const int m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < M && k < K) ? A[m * asm_ + k * ask] : 0.f;
    }
    for (int e = threadIdx.x; e < TN * TK; e += 256) {
      const int kk = e % TK, nn = e / TK;
      const int n = n0 + nn, k = k0 + kk;
      Bs[kk][nn] = (n < N && k < K) ? Bm[k * bsk + n * bsn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk][ty * 2 + i];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kk][tx * 2 + j];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + ty * 2 + i, n = n0 + tx * 2 + j;
      if (m < M && n < N) {
        float* c = C + m * csm + n * csn;
        if (gridDim.z > 1) atomicAdd(c, alpha * acc[i][j]);
        else *c = accumulate ? *c + alpha * acc[i][j] : alpha * acc[i][j];
      }
    }
}

// Fast path for K <= 512: the whole K extent of a 32x32 output tile is staged at once (one barrier, no K loop),
// because these GEMMs are latency-bound (a 128x256x256 problem is 17 MFLOP).  Round 4: the staging loops carry no integer
// division (the thread -> (k, row) map is fixed per operand orientation; the old `e % K`, `e / K` per element were ~1 300 of
// the kernel's ~3 500 instructions per thread), and a thread reads its two A rows / two B columns of a k with ONE 8-byte
// LDS read each (rows padded to 34 floats: even, so the pairs are aligned).  The sum over k runs in the same order with
// the same fma per element as before: results are bit-identical.
__global__ __launch_bounds__(256) void k_sgemm_smallk(const float* __restrict__ A, long asm_, long ask,
                                                        const float* __restrict__ Bm, long bsk, long bsn,
                                                        float* __restrict__ C, long csm, long csn, int M, int N, int K,
                                                        float alpha, int accumulate) {
  constexpr int TM = 32, TN = 32, LD = 34;
  extern __shared__ __attribute__((aligned(16))) float sm_[];
  float (*As)[LD] = reinterpret_cast<float (*)[LD]>(sm_);
  float (*Bs)[LD] = reinterpret_cast<float (*)[LD]>(sm_ + (size_t)K * LD);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  // the unit-stride dimension of each operand is the fast index of the thread map
  if (ask == 1) {            // k contiguous: 64 k per pass, 4 rows per pass
    const int tk = threadIdx.x & 63, tr = threadIdx.x >> 6;
    for (int mm = tr; mm < TM; mm += 4) {
      const int m = m0 + mm;
      for (int kk = tk; kk < K; kk += 64) As[kk][mm] = (m < M) ? A[m * asm_ + kk] : 0.f;
    }
  } else {                   // rows contiguous (or general strides): 32 rows per pass, 8 k per pass
    const int tr = threadIdx.x & 31, tk = threadIdx.x >> 5;
    const int m = m0 + tr;
    for (int kk = tk; kk < K; kk += 8) As[kk][tr] = (m < M) ? A[m * asm_ + kk * ask] : 0.f;
  }
  if (bsk == 1) {
    const int tk = threadIdx.x & 63, tr = threadIdx.x >> 6;
    for (int nn = tr; nn < TN; nn += 4) {
      const int n = n0 + nn;
      for (int kk = tk; kk < K; kk += 64) Bs[kk][nn] = (n < N) ? Bm[kk + n * bsn] : 0.f;
    }
  } else {
    const int tr = threadIdx.x & 31, tk = threadIdx.x >> 5;
    const int n = n0 + tr;
    for (int kk = tk; kk < K; kk += 8) Bs[kk][tr] = (n < N) ? Bm[kk * bsk + n * bsn] : 0.f;
  }
  __syncthreads();
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll 8
  for (int kk = 0; kk < K; ++kk) {
    const float2 a = *reinterpret_cast<const float2*>(&As[kk][ty * 2]);
    const float2 b = *reinterpret_cast<const float2*>(&Bs[kk][tx * 2]);
    acc[0][0] += a.x * b.x; acc[0][1] += a.x * b.y; acc[1][0] += a.y * b.x; acc[1][1] += a.y * b.y;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + ty * 2 + i, n = n0 + tx * 2 + j;
      if (m < M && n < N) {
        float* c = C + m * csm + n * csn;
        *c = accumulate ? *c + alpha * acc[i][j] : alpha * acc[i][j];
      }
    }
}

int sgemm(const float* A, long asm_, long ask, const float* B, long bsk, long bsn, float* C, long csm, long csn, int M,
          int N, int K, float alpha, int accumulate, hipStream_t st) {
  if (K <= 512) {
    EDM_MAX_LDS(k_sgemm_smallk, 160 * 1024);
    hipLaunchKernelGGL(k_sgemm_smallk, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), (size_t)2 * K * 34 * sizeof(float),
                       st, A, asm_, ask, B, bsk, bsn, C, csm, csn, M, N, K, alpha, accumulate);
    return 0;
  }
  int splits = 1;
  const long tiles = (long)((N + 31) / 32) * ((M + 31) / 32);
  if (K >= 1024 && tiles < 512 && csn == 1 && csm == N) {  // long-K, few tiles: split K to fill the chip
    long sp = K / 256, cap = (1024 + tiles - 1) / tiles;
    if (sp > cap) sp = cap;
    if (sp > 32) sp = 32;
    splits = (int)(sp < 1 ? 1 : sp);
    if (splits > 1 && !accumulate) (void)hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), st);
  }
  hipLaunchKernelGGL(k_sgemm, dim3((N + 31) / 32, (M + 31) / 32, splits), dim3(256), 0, st, A, asm_, ask, B, bsk, bsn, C,
                     csm, csn, M, N, K, alpha, accumulate);
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

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
  constexpr int TM = 64, TN = 64, TK = 16;
  __shared__ float As[TK][TM + 1];
  __shared__ float Bs[TK][TN + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += TK) {
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
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < M && n < N) {
        float* c = C + m * csm + n * csn;
        *c = accumulate ? *c + alpha * acc[i][j] : alpha * acc[i][j];
      }
    }
}

int sgemm(const float* A, long asm_, long ask, const float* B, long bsk, long bsn, float* C, long csm, long csn, int M,
          int N, int K, float alpha, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(k_sgemm, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, A, asm_, ask, B, bsk, bsn, C, csm,
                     csn, M, N, K, alpha, accumulate);
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

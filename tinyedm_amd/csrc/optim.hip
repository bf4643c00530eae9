// Step-level fp32 kernels around the network: Diffuser (edm.py:84-93), sigma-weighted MSE
// (edm.py:212, metric.py:8-18), fused Adam + power-function EMA over the flat parameter arena
// (edm.py:251-253, ema.py:137-140, 273), and the Heun updates of the sampler (solvers.py:49-57).
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * __logf(u1));
  float s, c;
  __sincosf(6.28318530717958648f * u2, &s, &c);
  n0 = r * c;
  n1 = r * s;
}

// sigma_b = exp(P_mean + P_std*eps_b);  noisy = clean + sigma_b * n      (4 elements per thread)
__global__ void k_diffuse(const float* __restrict__ clean, float* __restrict__ noisy, float* __restrict__ sigma,
                          float P_mean, float P_std, int B, long CHW, uint32_t seed_lo, uint32_t seed_hi,
                          uint32_t step, const StepParams* __restrict__ dyn) {
  if (dyn) { step = dyn->step; seed_lo = dyn->seed_lo ^ 0xD1FF05E5u; seed_hi = dyn->seed_hi; }
  const long n4 = ((long)B * CHW + 3) / 4;
  const long total = (long)B * CHW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    Philox4 r = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), 0xD1FFu, step, seed_lo, seed_hi);
    float nn[4];
    box_muller(r.x, r.y, nn[0], nn[1]);
    box_muller(r.z, r.w, nn[2], nn[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long e = i * 4 + j;
      if (e < total) {
        const int b = (int)(e / CHW);
        Philox4 rb = philox4x32_10((uint32_t)b, 0u, 0x5167u, step, seed_lo, seed_hi);
        float e0, e1;
        box_muller(rb.x, rb.y, e0, e1);
        const float s = __expf(P_mean + P_std * e0);
        noisy[e] = clean[e] + s * nn[j];
        if (e % CHW == 0) sigma[b] = s;
      }
    }
  }
}

// same with the two normal draws supplied (deterministic tests / external RNG)
__global__ void k_diffuse_given(const float* __restrict__ clean, const float* __restrict__ eps,
                                const float* __restrict__ noise, float* __restrict__ noisy, float* __restrict__ sigma,
                                float P_mean, float P_std, int B, long CHW) {
  const long total = (long)B * CHW;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e / CHW);
    const float s = expf(P_mean + P_std * eps[b]);
    noisy[e] = clean[e] + s * noise[e];
    if (e % CHW == 0) sigma[b] = s;
  }
}

// loss += sum_b mean_j w_b (D-x)^2 / B ;  dD = 2 w_b (D-x) / (CHW*B)   [w_b optional override]
__global__ void k_loss(const float* __restrict__ Dn, const float* __restrict__ clean, const float* __restrict__ sigma,
                       const float* __restrict__ wext, float sd, float* __restrict__ loss, float* __restrict__ dD, int B,
                       long CHW, float* __restrict__ acc_sum, long long* __restrict__ acc_total) {
  const long total = (long)B * CHW;
  const float inv = 1.0f / ((float)CHW * (float)B);
  float part = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e / CHW);
    float w;
    if (wext) {
      w = wext[b];
    } else {
      const float s = sigma[b];
      w = (s * s + sd * sd) / ((s * sd) * (s * sd));
    }
    const float d = Dn[e] - clean[e];
    part += w * d * d;
    if (dD) dD[e] = 2.0f * w * d * inv;
  }
  // one atomic per workgroup (one per wave serialised ~6000 adds on a single address: 81 us for 0.4 M elements)
  __shared__ float red[4];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float p = red[0] + red[1] + red[2] + red[3];
    atomicAdd(loss, p * inv);
    // epoch state of the metric (metric.py:38-49): sum_i mean_j w_i d_ij^2 and the sample count
    if (acc_sum) atomicAdd(acc_sum, p / (float)CHW);
    if (acc_total && blockIdx.x == 0) *acc_total += B;
  }
}

struct AdamArgs {
  float lr, b1, b2, eps, bc1, bc2sqrt, ema_beta, grad_scale;
};
// fused multi-tensor Adam + EMA over the flat arenas: 5 streams read, 4 written, one pass
// zero_grad: the gradient arena is cleared in the same pass (saves the separate fill of optimizer.zero_grad()).
__global__ void k_adam_ema(float* __restrict__ theta, float* __restrict__ grad, float* __restrict__ m,
                           float* __restrict__ v, float* __restrict__ ema, long n4, long n, AdamArgs a,
                           const StepParams* __restrict__ dyn, int zero_grad, unsigned* __restrict__ health) {
  if (dyn) { a.lr = dyn->lr; a.ema_beta = dyn->ema_beta; a.grad_scale = dyn->grad_scale; a.bc1 = dyn->bc1; a.bc2sqrt = dyn->bc2sqrt; }
  const float step_size = a.lr / a.bc1;
  bool bad = false;     // health word (nullable): bit 0 = a non-finite gradient or weight passed through this step
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    if (i * 4 + 3 < n) {
      f32x4 t = *reinterpret_cast<f32x4*>(theta + i * 4);
      f32x4 g = *reinterpret_cast<const f32x4*>(grad + i * 4);
      if (zero_grad) *reinterpret_cast<f32x4*>(grad + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 mm = *reinterpret_cast<f32x4*>(m + i * 4);
      f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
      f32x4 ee;
      if (ema) ee = *reinterpret_cast<f32x4*>(ema + i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = g[j] * a.grad_scale;
        mm[j] = a.b1 * mm[j] + (1.f - a.b1) * gj;
        vv[j] = a.b2 * vv[j] + (1.f - a.b2) * gj * gj;
        t[j] -= step_size * mm[j] / (sqrtf(vv[j]) / a.bc2sqrt + a.eps);
        if (ema) ee[j] = a.ema_beta * ee[j] + (1.f - a.ema_beta) * t[j];
        bad |= !(fabsf(gj) <= 3.0e38f) || !(fabsf(t[j]) <= 3.0e38f);
      }
      *reinterpret_cast<f32x4*>(theta + i * 4) = t;
      *reinterpret_cast<f32x4*>(m + i * 4) = mm;
      *reinterpret_cast<f32x4*>(v + i * 4) = vv;
      if (ema) *reinterpret_cast<f32x4*>(ema + i * 4) = ee;
    } else {
      for (long e = i * 4; e < n; ++e) {
        const float gj = grad[e] * a.grad_scale;
        const float mj = a.b1 * m[e] + (1.f - a.b1) * gj;
        const float vj = a.b2 * v[e] + (1.f - a.b2) * gj * gj;
        const float tj = theta[e] - step_size * mj / (sqrtf(vj) / a.bc2sqrt + a.eps);
        m[e] = mj;
        v[e] = vj;
        theta[e] = tj;
        if (zero_grad) grad[e] = 0.f;
        if (ema) ema[e] = a.ema_beta * ema[e] + (1.f - a.ema_beta) * tj;
        bad |= !(fabsf(gj) <= 3.0e38f) || !(fabsf(tj) <= 3.0e38f);
      }
    }
  }
  if (health && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(health, 1u);
}

// dx = (x-D)/t0 ; x1 = x + (t1-t0)*dx
__global__ void k_heun_euler(const float* __restrict__ x, const float* __restrict__ Dn, float t0, float t1,
                             float* __restrict__ dx, float* __restrict__ x1, long n, unsigned* __restrict__ health) {
  bool bad = false;     // health word (nullable): bit 1 = a non-finite sampler state
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = (x[i] - Dn[i]) / t0;
    dx[i] = d;
    const float o = x[i] + (t1 - t0) * d;
    x1[i] = o;
    bad |= !(fabsf(o) <= 3.0e38f);
  }
  if (health && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(health, 2u);
}
// dxp = (x1-D1)/t1 ; out = x + (t1-t0)*(0.5*dx + 0.5*dxp)
__global__ void k_heun_correct(const float* __restrict__ x, const float* __restrict__ dx, const float* __restrict__ x1,
                               const float* __restrict__ D1, float t0, float t1, float* __restrict__ out, long n,
                               unsigned* __restrict__ health) {
  bool bad = false;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float dp = (x1[i] - D1[i]) / t1;
    const float o = x[i] + (t1 - t0) * (0.5f * dx[i] + 0.5f * dp);
    out[i] = o;
    bad |= !(fabsf(o) <= 3.0e38f);
  }
  if (health && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(health, 2u);
}
__global__ void k_scale_f32(const float* __restrict__ x, float s, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = x[i] * s;
}

inline int grid_for(long work, int block) {
  long g = (work + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

// dyn (nullable, device edm_step_params): step and seed come from it; the Diffuser's stream is the network seed with
// its low word XORed by 0xD1FF05E5 (what tinyedm_amd.Diffuser passes by value otherwise).
extern "C" int edm_diffuse(const float* clean, float* noisy, float* sigma, float P_mean, float P_std, int B, long CHW,
                           unsigned long long seed, unsigned step, const void* dyn, hipStream_t st) {
  EDM_REQUIRE(clean && noisy && sigma && B > 0 && CHW > 0, "diffuse: bad args");
  hipLaunchKernelGGL(k_diffuse, dim3(grid_for((long)B * CHW / 4 + 1, 256)), dim3(256), 0, st, clean, noisy, sigma,
                     P_mean, P_std, B, CHW, (uint32_t)seed, (uint32_t)(seed >> 32), step, (const StepParams*)dyn);
  EDM_CHECK_LAUNCH("diffuse");
  return EDM_OK;
}
extern "C" int edm_diffuse_given(const float* clean, const float* eps, const float* noise, float* noisy, float* sigma,
                                 float P_mean, float P_std, int B, long CHW, hipStream_t st) {
  EDM_REQUIRE(clean && eps && noise && noisy && sigma && B > 0 && CHW > 0, "diffuse_given: bad args");
  hipLaunchKernelGGL(k_diffuse_given, dim3(grid_for((long)B * CHW, 256)), dim3(256), 0, st, clean, eps, noise, noisy,
                     sigma, P_mean, P_std, B, CHW);
  EDM_CHECK_LAUNCH("diffuse_given");
  return EDM_OK;
}
// loss (device scalar) is accumulated (+=): caller zero-fills.  dD may be null (validation).  acc_sum / acc_total
// (nullable device scalars): the metric's epoch state, sum_i mean_j w_i d_ij^2 and the number of samples, += in place.
extern "C" int edm_weighted_mse(const float* D, const float* clean, const float* sigma, const float* weight_override,
                                float sigma_data, float* loss, float* dD, int B, long CHW, float* acc_sum,
                                long long* acc_total, hipStream_t st) {
  EDM_REQUIRE(D && clean && (sigma || weight_override) && loss && B > 0 && CHW > 0, "weighted_mse: bad args");
  const long blocks = ((long)B * CHW + 1023) / 1024;  // ~4 elements per thread, at most 256 workgroups
  hipLaunchKernelGGL(k_loss, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, st, D, clean, sigma,
                     weight_override, sigma_data, loss, dD, B, CHW, acc_sum, acc_total);
  EDM_CHECK_LAUNCH("weighted_mse");
  return EDM_OK;
}
// step is 1-based (bias corrections use it); ema may be null.  dyn (nullable, device edm_step_params) overrides lr,
// ema_beta, grad_scale and the two bias corrections; zero_grad != 0 clears `grad` in the same pass.  health (nullable
// device word): bit 0 is OR-ed in when a non-finite gradient or updated weight was seen -- the in-graph sentinel a
// replayed step leaves behind for the host to read at its log interval.
extern "C" int edm_adam_ema(float* theta, float* grad, float* m, float* v, float* ema, long n, float lr, float b1,
                            float b2, float eps, int step, float ema_beta, float grad_scale, const void* dyn,
                            int zero_grad, unsigned* health, hipStream_t st) {
  EDM_REQUIRE(theta && grad && m && v && n > 0 && step >= 1, "adam_ema: bad args");
  EDM_REQUIRE(((uintptr_t)theta % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                  ((uintptr_t)v % 16 == 0) && (!ema || (uintptr_t)ema % 16 == 0),
              "adam_ema: arenas must be 16-byte aligned");
  AdamArgs a;
  a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)b1, (double)step));
  a.bc2sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
  a.ema_beta = ema_beta;
  a.grad_scale = grad_scale;
  const long n4 = (n + 3) / 4;
  hipLaunchKernelGGL(k_adam_ema, dim3(grid_for(n4, 256)), dim3(256), 0, st, theta, grad, m, v, ema, n4, n, a,
                     (const StepParams*)dyn, zero_grad, health);
  EDM_CHECK_LAUNCH("adam_ema");
  return EDM_OK;
}
// health (nullable device word): bit 1 is OR-ed in when the new state holds a non-finite value
extern "C" int edm_heun_euler(const float* x, const float* D, float t0, float t1, float* dx, float* x1, long n,
                              unsigned* health, hipStream_t st) {
  EDM_REQUIRE(x && D && dx && x1 && n > 0 && t0 != 0.f, "heun_euler: bad args");
  hipLaunchKernelGGL(k_heun_euler, dim3(grid_for(n, 256)), dim3(256), 0, st, x, D, t0, t1, dx, x1, n, health);
  EDM_CHECK_LAUNCH("heun_euler");
  return EDM_OK;
}
extern "C" int edm_heun_correct(const float* x, const float* dx, const float* x1, const float* D1, float t0, float t1,
                                float* out, long n, unsigned* health, hipStream_t st) {
  EDM_REQUIRE(x && dx && x1 && D1 && out && n > 0 && t1 != 0.f, "heun_correct: bad args");
  hipLaunchKernelGGL(k_heun_correct, dim3(grid_for(n, 256)), dim3(256), 0, st, x, dx, x1, D1, t0, t1, out, n, health);
  EDM_CHECK_LAUNCH("heun_correct");
  return EDM_OK;
}
extern "C" int edm_scale_f32(const float* x, float s, float* y, long n, hipStream_t st) {
  EDM_REQUIRE(x && y && n > 0, "scale_f32: bad args");
  hipLaunchKernelGGL(k_scale_f32, dim3(grid_for(n, 256)), dim3(256), 0, st, x, s, y, n);
  EDM_CHECK_LAUNCH("scale_f32");
  return EDM_OK;
}

// HBM-bound elementwise / per-pixel / per-(sample,channel) kernels of the EDM2 U-Net.
// Activations are NHWC bf16 ("pixels x channels", channel contiguous); each lane moves
// 8 channels = 16 B per access (coalesced dwordx4), reductions are wavefront shuffles.
// Reference semantics cited per kernel (file:line relative to /root/reference/src/tinyedm).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";
extern "C" void edm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* edm_last_error() { return g_err; }
extern "C" int edm_version() { return 1; }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int grid_for(long work, int block, int cap = 256 * 16) {
  long g = (work + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------ pixel_norm + mp_silu
// networks.py:9-14 (pixel_norm over C), :83-84 (mp_silu), used at :249-252.
// xn = x / (eps + ||x||/sqrt(C));  a = mp_silu(xn);  dsave = eps + ||x||/sqrt(C)
// POOL (round 6, EncD blocks without a 1x1 conv, networks.py:246-252): x is the tensor BEFORE the 2x2 average pool ((B, 2 Hp,
// 2 Wp, C); P = B Hp Wp pooled pixels): a pixel's row is the bf16-rounded mean of its four source rows -- k_pool2's values
// -- and the pooled tensor is never written (one launch, one HBM round trip less per EncD block; bit-identical)
template <int LPP, bool POOL = false>
__global__ __launch_bounds__(256) void k_pnorm_silu_fwd(const bf16* __restrict__ x, bf16* __restrict__ xn,
                                                          bf16* __restrict__ a, float* __restrict__ dsave, int P,
                                                          int C, int Hp = 0, int Wp = 0) {
  constexpr int GPW = 64 / LPP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lig = lane % LPP, grp = lane / LPP;
  const int CL = C >> 3;
  const float rsC = rsqrtf((float)C);
  const long stride = (long)gridDim.x * 4 * GPW;
  for (long p0 = ((long)blockIdx.x * 4 + wave) * GPW; p0 < P; p0 += stride) {
    const long p = p0 + grp;
    const bool pv = p < P;
    float v[2][8];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c8 = lig + it * LPP;
      if (pv && c8 < CL) {
        if constexpr (POOL) {
          const int w = (int)(p % Wp);
          const long t = p / Wp;
          const int h = (int)(t % Hp);
          const long b = t / Hp;
          const bf16* src = x + ((b * 2 * Hp + 2 * h) * 2 * Wp + 2 * w) * C + c8 * 8;
          const long row = (long)2 * Wp * C;
          float a1[8], a2[8], a3[8];
          load8(src, v[it]);
          load8(src + C, a1);
          load8(src + row, a2);
          load8(src + row + C, a3);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[it][i] = (float)(bf16)(0.25f * (v[it][i] + a1[i] + a2[i] + a3[i]));
        } else {
          load8(x + p * C + c8 * 8, v[it]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += v[it][i] * v[it][i];
      }
    }
    ss = group_sum<LPP>(ss);
    const float d = NORM_EPS + sqrtf(ss) * rsC;
    const float inv = 1.0f / d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c8 = lig + it * LPP;
      if (pv && c8 < CL) {
        float o[8], s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          o[i] = (float)(bf16)(v[it][i] * inv);
          s[i] = mp_silu_b(o[i]);
        }
        store8(xn + p * C + c8 * 8, o);
        store8(a + p * C + c8 * 8, s);
      }
    }
    if (pv && lig == 0) dsave[p] = d;
  }
}

// backward of the pair above.  g = gs*gxn + mp_silu'(xn)*ga ;  dx = (g - xn*<g,xn>*d/(C*(d-eps)))/d  (+ gadd: the
// gradient that reaches the same tensor along another path -- the U-Net skip -- summed here instead of by autograd)
// UP (the backward of the POOL forward): the gradient leaves at the resolution BEFORE the pool -- gx (B, 2 Hp, 2 Wp, C), each
// of a pooled pixel's four source pixels receives 0.25 * (the bf16-rounded pooled-resolution gradient) + gadd (gadd: at
// that resolution too, the U-Net skip's gradient): k_pnorm_silu_bwd followed by k_up2(0.25, add), bit for bit
template <int LPP, bool UP = false>
__global__ __launch_bounds__(256) void k_pnorm_silu_bwd(const bf16* __restrict__ xn, const float* __restrict__ dsave,
                                                          const bf16* __restrict__ gxn, float gs,
                                                          const bf16* __restrict__ ga, const bf16* __restrict__ gadd,
                                                          bf16* __restrict__ gx, int P, int C, int Hp = 0, int Wp = 0) {
  constexpr int GPW = 64 / LPP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lig = lane % LPP, grp = lane / LPP;
  const int CL = C >> 3;
  const long stride = (long)gridDim.x * 4 * GPW;
  for (long p0 = ((long)blockIdx.x * 4 + wave) * GPW; p0 < P; p0 += stride) {
    const long p = p0 + grp;
    const bool pv = p < P;
    float y[2][8], g[2][8];
    float dot = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c8 = lig + it * LPP;
      if (pv && c8 < CL) {
        load8(xn + p * C + c8 * 8, y[it]);
        float t[8];
        if (gxn) {
          load8(gxn + p * C + c8 * 8, t);
#pragma unroll
          for (int i = 0; i < 8; ++i) g[it][i] = gs * t[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) g[it][i] = 0.f;
        }
        if (ga) {
          load8(ga + p * C + c8 * 8, t);
#pragma unroll
          for (int i = 0; i < 8; ++i) g[it][i] += mp_silu_grad_b(y[it][i]) * t[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) dot += g[it][i] * y[it][i];
      }
    }
    dot = group_sum<LPP>(dot);
    float d = pv ? dsave[p] : 1.f;
    float s = d - NORM_EPS;
    float coef = s > 0.f ? dot * d / ((float)C * s) : 0.f;
    float inv = 1.0f / d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c8 = lig + it * LPP;
      if (pv && c8 < CL) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (g[it][i] - y[it][i] * coef) * inv;
        if constexpr (UP) {
          const int w = (int)(p % Wp);
          const long t2 = p / Wp;
          const int h = (int)(t2 % Hp);
          const long b = t2 / Hp;
          const long e0 = ((b * 2 * Hp + 2 * h) * 2 * Wp + 2 * w) * C + c8 * 8;
          const long row = (long)2 * Wp * C;
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = 0.25f * (float)(bf16)o[i];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const long e = e0 + (q >> 1) * row + (q & 1) * C;
            float r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = o[i];
            if (gadd) {
              float t[8];
              load8(gadd + e, t);
#pragma unroll
              for (int i = 0; i < 8; ++i) r[i] += t[i];
            }
            store8(gx + e, r);
          }
          continue;
        }
        if (gadd) {
          float t[8];
          load8(gadd + p * C + c8 * 8, t);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += t[i];
        }
        store8(gx + p * C + c8 * 8, o);
      }
    }
  }
}

static int pick_lpp(int C) {
  int cl = C / 8;
  return cl <= 16 ? 16 : (cl <= 32 ? 32 : 64);
}

extern "C" int edm_pixelnorm_silu_fwd(const void* x, void* xn, void* a, float* dsave, long P, int C,
                                      hipStream_t st) {
  EDM_REQUIRE(P > 0 && C > 0 && C % 8 == 0 && C <= 1024, "pixelnorm_silu_fwd: bad P=%ld C=%d", P, C);
  int lpp = pick_lpp(C);
  int gpw = 64 / lpp;
  int grid = grid_for(P, 4 * gpw);
#define L(N) hipLaunchKernelGGL(k_pnorm_silu_fwd<N>, dim3(grid), dim3(256), 0, st, (const bf16*)x, (bf16*)xn, (bf16*)a, dsave, (int)P, C)
  if (lpp == 16) L(16); else if (lpp == 32) L(32); else L(64);
#undef L
  EDM_CHECK_LAUNCH("pixelnorm_silu_fwd");
  return EDM_OK;
}

// pooled forms (see the POOL / UP template parameters): x / gx and gadd live at (B, 2 Hp, 2 Wp, C), xn / a / dsave / gxn / ga
// at (B, Hp, Wp, C)
extern "C" int edm_pool_pixelnorm_silu_fwd(const void* x, void* xn, void* a, float* dsave, int B, int Hp, int Wp, int C,
                                           hipStream_t st) {
  EDM_REQUIRE(x && xn && a && dsave && B > 0 && Hp > 0 && Wp > 0 && C > 0 && C % 8 == 0 && C <= 1024 &&
                  (long)B * Hp * Wp < (1L << 31), "pool_pixelnorm_silu_fwd: bad args");
  const long P = (long)B * Hp * Wp;
  int lpp = pick_lpp(C);
  int gpw = 64 / lpp;
  int grid = grid_for(P, 4 * gpw);
#define L(N) hipLaunchKernelGGL((k_pnorm_silu_fwd<N, true>), dim3(grid), dim3(256), 0, st, (const bf16*)x, (bf16*)xn, (bf16*)a, dsave, (int)P, C, Hp, Wp)
  if (lpp == 16) L(16); else if (lpp == 32) L(32); else L(64);
#undef L
  EDM_CHECK_LAUNCH("pool_pixelnorm_silu_fwd");
  return EDM_OK;
}
extern "C" int edm_pool_pixelnorm_silu_bwd(const void* xn, const float* dsave, const void* gxn, float gxn_scale,
                                           const void* ga, const void* gadd, void* gx, int B, int Hp, int Wp, int C,
                                           hipStream_t st) {
  EDM_REQUIRE(xn && dsave && gx && B > 0 && Hp > 0 && Wp > 0 && C > 0 && C % 8 == 0 && C <= 1024 &&
                  (long)B * Hp * Wp < (1L << 31), "pool_pixelnorm_silu_bwd: bad args");
  const long P = (long)B * Hp * Wp;
  int lpp = pick_lpp(C);
  int gpw = 64 / lpp;
  int grid = grid_for(P, 4 * gpw);
#define L(N) hipLaunchKernelGGL((k_pnorm_silu_bwd<N, true>), dim3(grid), dim3(256), 0, st, (const bf16*)xn, dsave, (const bf16*)gxn, gxn_scale, (const bf16*)ga, (const bf16*)gadd, (bf16*)gx, (int)P, C, Hp, Wp)
  if (lpp == 16) L(16); else if (lpp == 32) L(32); else L(64);
#undef L
  EDM_CHECK_LAUNCH("pool_pixelnorm_silu_bwd");
  return EDM_OK;
}

extern "C" int edm_pixelnorm_silu_bwd(const void* xn, const float* dsave, const void* gxn, float gxn_scale,
                                      const void* ga, const void* gadd, void* gx, long P, int C, hipStream_t st) {
  EDM_REQUIRE(P > 0 && C > 0 && C % 8 == 0 && C <= 1024, "pixelnorm_silu_bwd: bad P=%ld C=%d", P, C);
  int lpp = pick_lpp(C);
  int gpw = 64 / lpp;
  int grid = grid_for(P, 4 * gpw);
#define L(N) hipLaunchKernelGGL(k_pnorm_silu_bwd<N>, dim3(grid), dim3(256), 0, st, (const bf16*)xn, dsave, (const bf16*)gxn, gxn_scale, (const bf16*)ga, (const bf16*)gadd, (bf16*)gx, (int)P, C)
  if (lpp == 16) L(16); else if (lpp == 32) L(32); else L(64);
#undef L
  EDM_CHECK_LAUNCH("pixelnorm_silu_bwd");
  return EDM_OK;
}

// ------------------------------------------------------------------ plain mp_silu (decoder residual branch, networks.py:316)
__global__ void k_silu_fwd(const bf16* __restrict__ x, bf16* __restrict__ a, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = mp_silu_b(v[j]);
    store8(a + i * 8, v);
  }
}
// gx = mp_silu'(x)*ga + s*ge   (ge optional)
__global__ void k_silu_bwd(const bf16* __restrict__ x, const bf16* __restrict__ ga, const bf16* __restrict__ ge,
                           float s, bf16* __restrict__ gx, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8], g[8], e[8];
    load8(x + i * 8, v);
    load8(ga + i * 8, g);
    if (ge) load8(ge + i * 8, e);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = mp_silu_grad_b(v[j]) * g[j] + (ge ? s * e[j] : 0.f);
    store8(gx + i * 8, g);
  }
}
extern "C" int edm_silu_fwd(const void* x, void* a, long n, hipStream_t st) {
  EDM_REQUIRE(n > 0 && n % 8 == 0, "silu_fwd: n=%ld must be a positive multiple of 8", n);
  hipLaunchKernelGGL(k_silu_fwd, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16*)x, (bf16*)a, n / 8);
  EDM_CHECK_LAUNCH("silu_fwd");
  return EDM_OK;
}
extern "C" int edm_silu_bwd(const void* x, const void* ga, const void* gextra, float extra_scale, void* gx, long n,
                            hipStream_t st) {
  EDM_REQUIRE(n > 0 && n % 8 == 0, "silu_bwd: n=%ld must be a positive multiple of 8", n);
  hipLaunchKernelGGL(k_silu_bwd, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16*)x, (const bf16*)ga,
                     (const bf16*)gextra, extra_scale, (bf16*)gx, n / 8);
  EDM_CHECK_LAUNCH("silu_bwd");
  return EDM_OK;
}

// ------------------------------------------------------------------ out = alpha*a + beta*b  (mp_add, networks.py:87-88)
__global__ void k_axpby(const bf16* __restrict__ a, float alpha, const bf16* __restrict__ b, float beta,
                        bf16* __restrict__ o, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float u[8], v[8];
    load8(a + i * 8, u);
    if (b) load8(b + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = alpha * u[j] + (b ? beta * v[j] : 0.f);
    store8(o + i * 8, u);
  }
}
extern "C" int edm_axpby(const void* a, float alpha, const void* b, float beta, void* out, long n, hipStream_t st) {
  EDM_REQUIRE(n > 0 && n % 8 == 0, "axpby: n=%ld must be a positive multiple of 8", n);
  hipLaunchKernelGGL(k_axpby, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16*)a, alpha, (const bf16*)b,
                     beta, (bf16*)out, n / 8);
  EDM_CHECK_LAUNCH("axpby");
  return EDM_OK;
}

// ------------------------------------------------------------------ embedding modulation + mp_silu + dropout
// networks.py:255-260 / 319-324:  a = dropout(mp_silu(r * (lin*gain + 1)))
// lin is the per-block embed Linear output (B,C) fp32, gain a device scalar.

__global__ void k_mod_silu_drop_fwd(const bf16* __restrict__ r, const float* __restrict__ lin,
                                    const float* __restrict__ gain, bf16* __restrict__ a, int HW, int C, long n8,
                                    long lin_stride, float pdrop, uint32_t seed_lo, uint32_t seed_hi, uint32_t sub, uint32_t step,
                                    const StepParams* __restrict__ dyn) {
  if (dyn) { step = dyn->step; seed_lo = dyn->seed_lo; seed_hi = dyn->seed_hi; }
  const int CL = C >> 3;
  const float g = *gain;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % CL);
    const long pix = i / CL;
    const int b = (int)(pix / HW);
    float v[8];
    load8(r + i * 8, v);
    const float* lp = lin + (long)b * lin_stride + c8 * 8;
    const Keep8 keep = dropout_keep8(i, pdrop, sub, step, seed_lo, seed_hi);
    const float keep_scale = keep.scale;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float m = lp[j] * g + 1.0f;
      float o = mp_silu_b(v[j] * m);
      if (pdrop > 0.f) o = keep[j] ? o * keep_scale : 0.f;
      v[j] = o;
    }
    store8(a + i * 8, v);
  }
}

// backward: gr = ga*keep*mp_silu'(u)*m ;  gm[b,c] += sum_hw ga*keep*mp_silu'(u)*r   (u = r*m)
// block = CL*PS threads (thread -> fixed channel chunk), grid = (B, ceil(HW/PIXW))
__global__ void k_mod_silu_drop_bwd(const bf16* __restrict__ r, const float* __restrict__ lin,
                                    const float* __restrict__ gain, const bf16* __restrict__ ga,
                                    bf16* __restrict__ gr, float* __restrict__ gm, long gm_stride, int HW, int C, int PIXW,
                                    long lin_stride, float pdrop, uint32_t seed_lo, uint32_t seed_hi, uint32_t sub, uint32_t step,
                                    const StepParams* __restrict__ dyn) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  if (dyn) { step = dyn->step; seed_lo = dyn->seed_lo; seed_hi = dyn->seed_hi; }
  const int CL = C >> 3;
  const int PS = blockDim.x / CL;
  const int c8 = threadIdx.x % CL, ps = threadIdx.x / CL;
  const int b = blockIdx.x;
  const int p_begin = blockIdx.y * PIXW;
  const int p_end = min(HW, p_begin + PIXW);
  const float g = *gain;
  float m[8], acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    m[j] = lin[(long)b * lin_stride + c8 * 8 + j] * g + 1.0f;
    acc[j] = 0.f;
  }
  for (int p = p_begin + ps; p < p_end; p += PS) {
    const long i = ((long)b * HW + p) * CL + c8;
    float v[8], gg[8];
    load8(r + i * 8, v);
    load8(ga + i * 8, gg);
    const Keep8 keep = dropout_keep8(i, pdrop, sub, step, seed_lo, seed_hi);
    const float keep_scale = keep.scale;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float u = (pdrop > 0.f && !keep[j]) ? 0.f : v[j];   // (a dropped element may hold the NaN mark of conv3x3_mod)
      float gu = gg[j] * mp_silu_grad_b(u * m[j]);
      if (pdrop > 0.f) gu = keep[j] ? gu * keep_scale : 0.f;
      acc[j] += gu * u;
      gg[j] = gu * m[j];
    }
    store8(gr + i * 8, gg);
  }
  // reduce the PS partial sums per channel through LDS, one atomic per (b,c)
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ps * C + c8 * 8 + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < PS; ++q) s += red[q * C + c];
    atomicAdd(gm + (long)b * gm_stride + c, s);
  }
}
// glin = gm*gain ; ggain += sum gm*lin
__global__ void k_mod_finish(const float* __restrict__ gm, const float* __restrict__ lin,
                             const float* __restrict__ gain, float* __restrict__ glin, float* __restrict__ ggain,
                             long n, int C, long lin_stride, long glin_stride) {
  const float g = *gain;
  float part = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / C;
    const int c = (int)(i - b * C);
    float v = gm[i];
    glin[b * glin_stride + c] = v * g;
    part += v * lin[b * lin_stride + c];
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) atomicAdd(ggain, part);
}

static int block_for_chunks(int CL) { return (256 / CL) * CL; }

// lin: row b starts at lin + b*lin_stride (lin_stride >= C: a column slice of the batched embed-linear output)
extern "C" int edm_mod_silu_drop_fwd(const void* r, const float* lin, long lin_stride, const float* gain, void* a, int B,
                                     int HW, int C, float pdrop, unsigned long long seed, unsigned sub, unsigned step,
                                     const void* dyn, hipStream_t st) {
  EDM_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && C > 0 && pdrop >= 0.f && pdrop < 1.f && lin_stride >= C,
              "mod_silu_drop_fwd: bad args");
  long n8 = (long)B * HW * C / 8;
  hipLaunchKernelGGL(k_mod_silu_drop_fwd, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const bf16*)r, lin, gain,
                     (bf16*)a, HW, C, n8, lin_stride, pdrop, (uint32_t)seed, (uint32_t)(seed >> 32), sub, step,
                     (const StepParams*)dyn);
  EDM_CHECK_LAUNCH("mod_silu_drop_fwd");
  return EDM_OK;
}

// gm must be zero-filled [B,C] fp32 scratch; glin rows at glin + b*glin_stride; ggain device scalar accumulated (+=).
extern "C" int edm_mod_silu_drop_bwd(const void* r, const float* lin, long lin_stride, const float* gain,
                                     const void* ga, void* gr, float* gm, float* glin, long glin_stride, float* ggain,
                                     int B, int HW, int C, float pdrop, unsigned long long seed, unsigned sub,
                                     unsigned step, const void* dyn, hipStream_t st) {
  EDM_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && C > 0 && C <= 1024 && lin_stride >= C && glin_stride >= C,
              "mod_silu_drop_bwd: bad args");
  int CL = C / 8, block = block_for_chunks(CL), PS = block / CL;
  int PIXW = HW >= 256 ? 128 : HW;
  hipLaunchKernelGGL(k_mod_silu_drop_bwd, dim3(B, cdiv(HW, PIXW)), dim3(block), PS * C * sizeof(float), st,
                     (const bf16*)r, lin, gain, (const bf16*)ga, (bf16*)gr, gm, (long)C, HW, C, PIXW, lin_stride, pdrop,
                     (uint32_t)seed, (uint32_t)(seed >> 32), sub, step, (const StepParams*)dyn);
  EDM_CHECK_LAUNCH("mod_silu_drop_bwd");
  hipLaunchKernelGGL(k_mod_finish, dim3(grid_for((long)B * C, 256, 64)), dim3(256), 0, st, gm, lin, gain, glin,
                     ggain, (long)B * C, C, lin_stride, glin_stride);
  EDM_CHECK_LAUNCH("mod_finish");
  return EDM_OK;
}

// The first half of edm_mod_silu_drop_bwd alone: gr, and the RAW modulation gradient accumulated into gm (zero-filled fp32 rows
// of gm_stride floats -- a column slice of the buffer all blocks share); ONE edm_mod_finish_multi at the end of the backward
// pass finishes every block (as after edm_conv3x3_modbwd with a shared buffer, which refuses maps with H*W % 32 != 0: MNIST's
// 28x28 / 14x14 / 7x7 levels took 27 separate finish launches per step before this).
extern "C" int edm_mod_silu_drop_bwd_raw(const void* r, const float* lin, long lin_stride, const float* gain,
                                         const void* ga, void* gr, float* gm, long gm_stride, int B, int HW, int C,
                                         float pdrop, unsigned long long seed, unsigned sub, unsigned step,
                                         const void* dyn, hipStream_t st) {
  EDM_REQUIRE(r && lin && gain && ga && gr && gm, "mod_silu_drop_bwd_raw: null pointer");
  EDM_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && C > 0 && C <= 1024 && lin_stride >= C && gm_stride >= C,
              "mod_silu_drop_bwd_raw: bad args");
  int CL = C / 8, block = block_for_chunks(CL), PS = block / CL;
  int PIXW = HW >= 256 ? 128 : HW;
  hipLaunchKernelGGL(k_mod_silu_drop_bwd, dim3(B, cdiv(HW, PIXW)), dim3(block), PS * C * sizeof(float), st,
                     (const bf16*)r, lin, gain, (const bf16*)ga, (bf16*)gr, gm, gm_stride, HW, C, PIXW, lin_stride, pdrop,
                     (uint32_t)seed, (uint32_t)(seed >> 32), sub, step, (const StepParams*)dyn);
  EDM_CHECK_LAUNCH("mod_silu_drop_bwd_raw");
  return EDM_OK;
}

// second half of edm_mod_silu_drop_bwd on its own (the first half can ride in a conv epilogue: edm_conv3x3_modbwd)
extern "C" int edm_mod_finish(const float* gm, const float* lin, long lin_stride, const float* gain, float* glin,
                              long glin_stride, float* ggain, int B, int C, hipStream_t st) {
  EDM_REQUIRE(gm && lin && gain && glin && ggain && B > 0 && C > 0 && lin_stride >= C && glin_stride >= C,
              "mod_finish: bad args");
  hipLaunchKernelGGL(k_mod_finish, dim3(grid_for((long)B * C, 256, 64)), dim3(256), 0, st, gm, lin, gain, glin, ggain,
                     (long)B * C, C, lin_stride, glin_stride);
  EDM_CHECK_LAUNCH("mod_finish");
  return EDM_OK;
}

// The finish of EVERY block of a network in one launch (round 3: 21 launches of 6 us -> 1).  gm_all / lin_all / glin_all
// are [B][stride] buffers whose column ranges [col0, col0 + C) belong to one block each (the batched embed Linear's
// layout); items is a DEVICE array.  glin += gm * gain (+=: a block that took the unfused path has written its glin
// already and left its gm columns zero), ggain += sum gm * lin.
struct ModFinItem {
  const float* gain;
  float* ggain;
  int col0, C;
};
__global__ void k_mod_finish_multi(const float* __restrict__ gm, const float* __restrict__ lin, float* __restrict__ glin,
                                   long stride, const ModFinItem* __restrict__ items, int B) {
  const ModFinItem it = items[blockIdx.x];
  const float g = *it.gain;
  const long n = (long)B * it.C;
  float part = 0.f;
  for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.y * blockDim.x) {
    const long b = i / it.C;
    const long e = b * stride + it.col0 + (i - b * it.C);
    const float v = gm[e];
    glin[e] += v * g;
    part += v * lin[e];
  }
  __shared__ float red[4];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(it.ggain, red[0] + red[1] + red[2] + red[3]);
}
extern "C" int edm_mod_finish_multi(const float* gm_all, const float* lin_all, float* glin_all, long stride,
                                    const void* items_dev, int n_items, int B, hipStream_t st) {
  EDM_REQUIRE(gm_all && lin_all && glin_all && items_dev && n_items > 0 && n_items <= 4096 && B > 0 && stride > 0,
              "mod_finish_multi: bad args");
  hipLaunchKernelGGL(k_mod_finish_multi, dim3(n_items, 8), dim3(256), 0, st, gm_all, lin_all, glin_all, stride,
                     (const ModFinItem*)items_dev, B);
  EDM_CHECK_LAUNCH("mod_finish_multi");
  return EDM_OK;
}

// exported for tests: the keep-mask the two kernels above derive from (seed, sub, step)
__global__ void k_dropout_mask(uint8_t* __restrict__ mask, long n8, float pdrop, uint32_t seed_lo, uint32_t seed_hi,
                               uint32_t sub, uint32_t step) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const Keep8 keep = dropout_keep8(i, pdrop, sub, step, seed_lo, seed_hi);
#pragma unroll
    for (int j = 0; j < 8; ++j) mask[i * 8 + j] = keep[j] ? 1 : 0;
  }
}
extern "C" int edm_dropout_mask(unsigned char* mask, long n, float pdrop, unsigned long long seed, unsigned sub,
                                unsigned step, hipStream_t st) {
  EDM_REQUIRE(n > 0 && n % 8 == 0, "dropout_mask: n must be a multiple of 8");
  hipLaunchKernelGGL(k_dropout_mask, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, mask, n / 8, pdrop,
                     (uint32_t)seed, (uint32_t)(seed >> 32), sub, step);
  EDM_CHECK_LAUNCH("dropout_mask");
  return EDM_OK;
}

// ------------------------------------------------------------------ 2x resampling (networks.py:72, :80)
// avgpool:  y[b,h,w,c] = s * sum_{i,j<2} x[b,2h+i,2w+j,c]     (H,W = OUTPUT dims)
__global__ void k_pool2(const bf16* __restrict__ x, bf16* __restrict__ y, int H, int W, int CL, long n8, float s) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    int c8 = (int)(i % CL);
    long pix = i / CL;
    int w = (int)(pix % W);
    long t = pix / W;
    int h = (int)(t % H);
    long b = t / H;
    const long row = (long)2 * W * CL;
    const bf16* src = x + (((b * 2 * H + 2 * h) * 2 * W + 2 * w) * CL + c8) * 8;
    float a0[8], a1[8], a2[8], a3[8];
    load8(src, a0);
    load8(src + CL * 8, a1);
    load8(src + row * 8, a2);
    load8(src + row * 8 + CL * 8, a3);
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = s * (a0[j] + a1[j] + a2[j] + a3[j]);
    store8(y + i * 8, a0);
  }
}
// nearest-exact x2:  y[b,h,w,c] = s * x[b,h/2,w/2,c] (+ add[b,h,w,c])   (H,W = OUTPUT dims; add: see k_pnorm_silu_bwd)
__global__ void k_up2(const bf16* __restrict__ x, const bf16* __restrict__ add, bf16* __restrict__ y, int H, int W, int CL,
                      long n8, float s) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    int c8 = (int)(i % CL);
    long pix = i / CL;
    int w = (int)(pix % W);
    long t = pix / W;
    int h = (int)(t % H);
    long b = t / H;
    const bf16* src = x + (((b * (H / 2) + h / 2) * (W / 2) + w / 2) * CL + c8) * 8;
    float a0[8];
    load8(src, a0);
    if (s != 1.0f) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a0[j] *= s;
    }
    if (add) {
      float t[8];
      load8(add + i * 8, t);
#pragma unroll
      for (int j = 0; j < 8; ++j) a0[j] += t[j];
    }
    store8(y + i * 8, a0);
  }
}
// DecU blocks (networks.py:312-316): y = nearest-exact x2 of x and a = mp_silu(y), one pass (k_up2 then k_silu_fwd, bit for bit)
__global__ void k_up2_silu(const bf16* __restrict__ x, bf16* __restrict__ y, bf16* __restrict__ a, int H, int W, int CL,
                           long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    int c8 = (int)(i % CL);
    long pix = i / CL;
    int w = (int)(pix % W);
    long t = pix / W;
    int h = (int)(t % H);
    long b = t / H;
    const u32x4 raw = *reinterpret_cast<const u32x4*>(x + (((b * (H / 2) + h / 2) * (W / 2) + w / 2) * CL + c8) * 8);
    *reinterpret_cast<u32x4*>(y + i * 8) = raw;
    const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = mp_silu_b((float)v[j]);
    store8(a + i * 8, f);
  }
}
extern "C" int edm_up2_silu(const void* x, void* y, void* a, int B, int Hout, int Wout, int C, hipStream_t st) {
  EDM_REQUIRE(x && y && a && B > 0 && Hout > 0 && Wout > 0 && Hout % 2 == 0 && Wout % 2 == 0 && C % 8 == 0, "up2_silu: bad args");
  long n8 = (long)B * Hout * Wout * C / 8;
  hipLaunchKernelGGL(k_up2_silu, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, (bf16*)a, Hout, Wout,
                     C / 8, n8);
  EDM_CHECK_LAUNCH("up2_silu");
  return EDM_OK;
}
extern "C" int edm_pool2(const void* x, void* y, int B, int Hout, int Wout, int C, float scale, hipStream_t st) {
  EDM_REQUIRE(B > 0 && Hout > 0 && Wout > 0 && C % 8 == 0, "pool2: bad args");
  long n8 = (long)B * Hout * Wout * C / 8;
  hipLaunchKernelGGL(k_pool2, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, Hout, Wout, C / 8,
                     n8, scale);
  EDM_CHECK_LAUNCH("pool2");
  return EDM_OK;
}
extern "C" int edm_up2(const void* x, const void* add, void* y, int B, int Hout, int Wout, int C, float scale,
                       hipStream_t st) {
  EDM_REQUIRE(B > 0 && Hout > 0 && Wout > 0 && Hout % 2 == 0 && Wout % 2 == 0 && C % 8 == 0, "up2: bad args");
  long n8 = (long)B * Hout * Wout * C / 8;
  hipLaunchKernelGGL(k_up2, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const bf16*)x, (const bf16*)add, (bf16*)y, Hout,
                     Wout, C / 8, n8, scale);
  EDM_CHECK_LAUNCH("up2");
  return EDM_OK;
}

// ------------------------------------------------------------------ ScaleLong skip gate (networks.py:106-118) + concat (:311)
// Deterministic form (the default): one workgroup owns a 64-channel slice of ONE sample and walks all of its pixels,
// 32 pixel rows in flight (8 lanes x 16 B = the 128 contiguous bytes of a row's slice), four loads per thread issued
// together; the 32 per-lane partial sums meet in LDS in a fixed order.  out[b,c] = scale * sum (plain store: no
// atomics, no zero-fill needed, bit-reproducible -- a mean that differs in its last bit from run to run flips bf16
// roundings downstream and makes two evaluations of the same network differ by ~1e-3).
__global__ __launch_bounds__(256) void k_reduce_hw_det(const bf16* __restrict__ x, long xs, const bf16* __restrict__ y,
                                                         long ys, float* __restrict__ out, int HW, int C, float scale) {
  __shared__ float red[32][65];
  const int b = blockIdx.x, c0 = blockIdx.y * 64;
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = c0 + cl * 8;
  const bool cok = c < C;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long base = (long)b * HW;
  constexpr int UN = 16;  // 16-byte loads in flight per thread and operand: the pass is latency-bound (one sample slice
                          // per workgroup, two workgroups per CU), so the 32x32 slice (1024 rows) is two round trips
  for (int p0 = pl; p0 < HW; p0 += 32 * UN) {
    u32x4 v[UN], u[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int p = p0 + 32 * k;
      const bool ok = cok && p < HW;
      v[k] = ok ? *reinterpret_cast<const u32x4*>(x + (base + p) * xs + c) : u32x4{0u, 0u, 0u, 0u};
      if (y) u[k] = ok ? *reinterpret_cast<const u32x4*>(y + (base + p) * ys + c) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, v[k]);
      if (y) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, u[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)a[j] * (float)b[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)a[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[pl][cl * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += red[q][threadIdx.x];
    out[(long)b * C + c0 + threadIdx.x] = s * scale;
  }
}
// x: rows of xs elements (uses first C), y optional rows of ys elements; out[b,c] = scale*sum_hw x*(y)
// (written, not accumulated; deterministic summation order).
extern "C" int edm_reduce_hw(const void* x, long x_stride, const void* y, long y_stride, float* out, int B, int HW,
                             int C, float scale, hipStream_t st) {
  EDM_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && C > 0 && C <= 4096 && x_stride % 8 == 0 && y_stride % 8 == 0,
              "reduce_hw: bad args");
  hipLaunchKernelGGL(k_reduce_hw_det, dim3(B, cdiv(C, 64)), dim3(256), 0, st, (const bf16*)x, x_stride, (const bf16*)y,
                     y_stride, out, HW, C, scale);
  EDM_CHECK_LAUNCH("reduce_hw");
  return EDM_OK;
}

// s = sum_r w[r * stride] * v[r], the loads issued 16 at a time (the plain loop waits for one L2 round trip per term)
__device__ __forceinline__ float strided_dot(const float* __restrict__ w, long stride, const float* v, int n) {
  float s = 0.f;
  int r = 0;
  for (; r + 16 <= n; r += 16) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = w[(long)(r + k) * stride];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += t[k] * v[r + k];
  }
  for (; r < n; ++r) s += w[(long)r * stride] * v[r];
  return s;
}

// per-sample gate MLP, fp32:  m=[mean;1] -> z1=W1 m -> h=mp_silu(z1) -> z2=W2 h -> gate=sigmoid(z2)
// W1h [R][C+1], W2h [C][R] are effective (normalised, /sqrt(fan_in)) fp32 weights.
__global__ void k_scalelong_fwd(const float* __restrict__ mean, const float* __restrict__ W1, const float* __restrict__ W2,
                                float* __restrict__ gate, float* __restrict__ z1save, int C, int R) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // m[C+1], h[R]
  float* m = sm;
  float* h = sm + C + 1;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) m[c] = mean[(long)b * C + c];
  if (threadIdx.x == 0) m[C] = 1.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < R; r += nw) {
    float s = 0.f;
    for (int c = lane; c <= C; c += 64) s += W1[(long)r * (C + 1) + c] * m[c];
    s = wave_sum(s);
    if (lane == 0) {
      z1save[(long)b * R + r] = s;
      h[r] = mp_silu_f(s);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) gate[(long)b * C + c] = sigmoidf_(strided_dot(W2 + (long)c * R, 1, h, R));
}
// backward: ggate[b,C] -> gmean[b,C], gW1 += , gW2 += (fp32 atomics; tiny)
__global__ void k_scalelong_bwd(const float* __restrict__ mean, const float* __restrict__ W1, const float* __restrict__ W2,
                                const float* __restrict__ gate, const float* __restrict__ z1save,
                                const float* __restrict__ ggate, float* __restrict__ gmean, float* __restrict__ gW1,
                                float* __restrict__ gW2, int C, int R) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // m[C+1], h[R], gz2[C], gz1[R]
  float* m = sm;
  float* h = m + C + 1;
  float* gz2 = h + R;
  float* gz1 = gz2 + C;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    m[c] = mean[(long)b * C + c];
    float g = gate[(long)b * C + c];
    gz2[c] = ggate[(long)b * C + c] * g * (1.0f - g);
  }
  if (threadIdx.x == 0) m[C] = 1.0f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) h[r] = mp_silu_f(z1save[(long)b * R + r]);
  __syncthreads();
  // gW2[c][r] += gz2[c]*h[r]
  for (int i = threadIdx.x; i < C * R; i += blockDim.x) atomicAdd(gW2 + i, gz2[i / R] * h[i % R]);
  // gh[r] = sum_c W2[c][r] gz2[c] ; gz1 = gh * mp_silu'(z1)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < R; r += nw) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += W2[(long)c * R + r] * gz2[c];
    s = wave_sum(s);
    if (lane == 0) gz1[r] = s * mp_silu_grad_f(z1save[(long)b * R + r]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R * (C + 1); i += blockDim.x) atomicAdd(gW1 + i, gz1[i / (C + 1)] * m[i % (C + 1)]);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += W1[(long)r * (C + 1) + c] * gz1[r];
    gmean[(long)b * C + c] = s;
  }
}
extern "C" int edm_scalelong_fwd(const float* mean, const float* W1h, const float* W2h, float* gate, float* z1save,
                                 int B, int C, int R, hipStream_t st) {
  EDM_REQUIRE(B > 0 && C > 0 && R > 0 && C <= 4096, "scalelong_fwd: bad args");
  hipLaunchKernelGGL(k_scalelong_fwd, dim3(B), dim3(256), (C + 1 + R) * sizeof(float), st, mean, W1h, W2h, gate,
                     z1save, C, R);
  EDM_CHECK_LAUNCH("scalelong_fwd");
  return EDM_OK;
}
extern "C" int edm_scalelong_bwd(const float* mean, const float* W1h, const float* W2h, const float* gate,
                                 const float* z1save, const float* ggate, float* gmean, float* gW1h, float* gW2h,
                                 int B, int C, int R, hipStream_t st) {
  EDM_REQUIRE(B > 0 && C > 0 && R > 0 && C <= 4096, "scalelong_bwd: bad args");
  hipLaunchKernelGGL(k_scalelong_bwd, dim3(B), dim3(256), (2 * C + 1 + 2 * R) * sizeof(float), st, mean, W1h, W2h,
                     gate, z1save, ggate, gmean, gW1h, gW2h, C, R);
  EDM_CHECK_LAUNCH("scalelong_bwd");
  return EDM_OK;
}

// ---- the skip gate in ONE launch per direction (round 3).  The mean over H*W and the gate MLP were two launches each way
// (k_reduce_hw_det + k_scalelong_*: 36 launches, 0.48 ms of the CIFAR-10 step, every one latency-bound: 512 small
// workgroups, then 128 tiny ones).  Here ONE workgroup of 1024 threads owns ONE sample: C/8 lanes cover a pixel row's
// channels 16 B each, blockDim / (C/8) rows are in flight per pass (x 8 unrolled loads per thread: 128 KiB of loads in
// flight per CU), the row groups meet in LDS in a fixed order (bit-reproducible, as k_reduce_hw_det), and the same
// workgroup then runs the sample's gate MLP from LDS -- no second launch, no cross-workgroup hand-off.
template <bool MUL>
__device__ __forceinline__ void sample_reduce(const bf16* __restrict__ x, long xs, const bf16* __restrict__ y, long ys,
                                              int HW, int C, float* red, float* out_lds, float scale) {
  const int lpr = C >> 3;                       // lanes per pixel row
  const int rpp = blockDim.x / lpr;             // rows per pass
  const int cl = threadIdx.x % lpr, rg = threadIdx.x / lpr;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rg < rpp) {
    constexpr int UN = 8;
    for (int p0 = rg; p0 < HW; p0 += rpp * UN) {
      u32x4 v[UN], u[UN];
#pragma unroll
      for (int k = 0; k < UN; ++k) {
        const int p = p0 + rpp * k;
        const bool ok = p < HW;
        v[k] = ok ? *reinterpret_cast<const u32x4*>(x + (long)p * xs + cl * 8) : u32x4{0u, 0u, 0u, 0u};
        if (MUL) u[k] = ok ? *reinterpret_cast<const u32x4*>(y + (long)p * ys + cl * 8) : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int k = 0; k < UN; ++k) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, v[k]);
        if (MUL) {
          const bf16x8 b = __builtin_bit_cast(bf16x8, u[k]);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += (float)a[j] * (float)b[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += (float)a[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rg * C + cl * 8 + j] = acc[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int q = 0; q < rpp; ++q) s += red[q * C + c];
    out_lds[c] = s * scale;
  }
  __syncthreads();
}

// forward: mean[b,:] = mean_hw skip[b];  gate = sigmoid(W2 mp_silu(W1 [mean;1]))   (networks.py:112-118)
template <bool KEEP = false>
__device__ __forceinline__ void skip_gate_fwd_body(const bf16* __restrict__ skip, const float* __restrict__ W1,
                                                   const float* __restrict__ W2, float* __restrict__ mean,
                                                   float* __restrict__ gate, float* __restrict__ z1save, int HW,
                                                   int C, int R, int b) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // red[rpp*C] | m[C+1] | h[R]
  const int rpp = blockDim.x / (C >> 3);
  float* red = sm;
  float* m = sm + rpp * C;
  float* h = m + C + 1;
  sample_reduce<false>(skip + (long)b * HW * C, C, nullptr, 0, HW, C, red, m, 1.0f / (float)HW);
  for (int c = threadIdx.x; c < C; c += blockDim.x) mean[(long)b * C + c] = m[c];
  if (threadIdx.x == 0) m[C] = 1.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < R; r += nw) {
    float s = 0.f;
    for (int c = lane; c <= C; c += 64) s += W1[(long)r * (C + 1) + c] * m[c];
    s = wave_sum(s);
    if (lane == 0) {
      z1save[(long)b * R + r] = s;
      h[r] = mp_silu_f(s);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float gv = sigmoidf_(strided_dot(W2 + (long)c * R, 1, h, R));
    gate[(long)b * C + c] = gv;
    if (KEEP) red[c] = gv;      // (the reduction scratch is free again: the caller's second pass reads the gate from LDS)
  }
}
__global__ __launch_bounds__(1024) void k_skip_gate_fwd(const bf16* __restrict__ skip, const float* __restrict__ W1,
                                                          const float* __restrict__ W2, float* __restrict__ mean,
                                                          float* __restrict__ gate, float* __restrict__ z1save, int HW,
                                                          int C, int R) {
  skip_gate_fwd_body(skip, W1, W2, mean, gate, z1save, HW, C, R, blockIdx.x);
}
// The gates of SEVERAL skip tensors in one launch (round 6).  A gate depends on its skip tensor and two small weights only, and
// the U-Net's skips all exist when the encoder ends -- but each decoder block launched its own k_skip_gate_fwd: one workgroup
// per sample, 128 workgroups on 256 CUs at the training batch, nine launches per step at 2.2-3.1 TB/s.  Here the (tensor, sample)
// workgroups of every gate with the same channel count are laid end to end (blk0 = first workgroup of a gate): one launch
// that fills the chip.  Same body, same values.
constexpr int MAXSGF = 32;
struct SgfItem {
  const bf16* skip;
  const float* W1;
  const float* W2;
  float* mean;
  float* gate;
  float* z1;
  bf16* cat;     // optional: the (cat, mp_silu(cat)) buffers of the decoder block that concatenates this skip, rows of Ci + C
  bf16* sil;     // elements -- the workgroup then also writes the sample's gated skip into their right halves (k_skip_half_fwd's
  int Ci, pad;   // work) while the sample is still in the cache it was just reduced from
  int B, HW, R, blk0;
};
struct SgfGroup {
  SgfItem it[MAXSGF];
  int n, C, pad[2];
};
__global__ __launch_bounds__(1024) void k_skip_gate_fwd_multi(const SgfGroup* __restrict__ g) {
  int k = 0;
  const int n = g->n;
  while (k + 1 < n && (int)blockIdx.x >= g->it[k + 1].blk0) ++k;
  const SgfItem it = g->it[k];
  const int C = g->C, b = (int)blockIdx.x - it.blk0;
  skip_gate_fwd_body<true>(it.skip, it.W1, it.W2, it.mean, it.gate, it.z1, it.HW, C, it.R, b);
  if (it.cat == nullptr) return;      // (workgroup-uniform)
  __syncthreads();
  extern __shared__ __attribute__((aligned(16))) float sm[];   // sm[0 .. C): this sample's gate (skip_gate_fwd_body<true>)
  const int CLs = C >> 3;
  const long ld = it.Ci + C;
  const bf16* __restrict__ sk = it.skip + (long)b * it.HW * C;
  bf16* __restrict__ cat = it.cat + (long)b * it.HW * ld + it.Ci;
  bf16* __restrict__ sil = it.sil ? it.sil + (long)b * it.HW * ld + it.Ci : nullptr;
  const int n8 = it.HW * CLs;
  for (int i = threadIdx.x; i < n8; i += blockDim.x) {     // (as k_skip_half_fwd: same arithmetic, same roundings)
    const int cs = (i % CLs) * 8;
    const int pix = i / CLs;
    float v[8];
    load8(sk + (long)pix * C + cs, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)(bf16)(v[j] * sm[cs + j]);
    store8(cat + (long)pix * ld + cs, v);
    if (sil) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = mp_silu_b(v[j]);
      store8(sil + (long)pix * ld + cs, v);
    }
  }
}
// backward: ggate[b,c] = sum_hw gcat[b,hw,Ci+c] * skip[b,hw,c], then the MLP backward of k_scalelong_bwd per sample.
// The weight gradients are sums over the batch of per-sample outer products; adding them with atomics from every
// workgroup (the first form of this kernel) meant B-way contention on each of the 2 C R addresses -- 20 us per launch
// whatever the image size.  Each workgroup now leaves its vectors (d z2, d z1, h) in ws[b][C + 2R] and
// k_skip_gate_wgrad sums the outer products over b in a fixed order.
__device__ __forceinline__ void skip_gate_bwd_body(const bf16* __restrict__ gcat, long gs, const bf16* __restrict__ skip,
                                                   const float* __restrict__ W1, const float* __restrict__ W2,
                                                   const float* __restrict__ gate, const float* __restrict__ z1save,
                                                   float* __restrict__ gmean, float* __restrict__ ws, int HW, int C,
                                                   int R, int b) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // red[rpp*C] | gz2[C] | gz1[R]
  const int rpp = blockDim.x / (C >> 3);
  float* red = sm;
  float* gz2 = sm + rpp * C;
  float* gz1 = gz2 + C;
  float* wsb = ws + (long)b * (C + 2 * R);
  sample_reduce<true>(gcat + (long)b * HW * gs, gs, skip + (long)b * HW * C, C, HW, C, red, gz2, 1.0f);   // gz2 <- ggate
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float g = gate[(long)b * C + c];
    gz2[c] *= g * (1.0f - g);
    wsb[c] = gz2[c];
  }
  for (int r = threadIdx.x; r < R; r += blockDim.x) wsb[C + R + r] = mp_silu_f(z1save[(long)b * R + r]);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < R; r += nw) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += W2[(long)c * R + r] * gz2[c];
    s = wave_sum(s);
    if (lane == 0) {
      gz1[r] = s * mp_silu_grad_f(z1save[(long)b * R + r]);
      wsb[C + r] = gz1[r];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) gmean[(long)b * C + c] = strided_dot(W1 + c, C + 1, gz1, R);
}
__global__ __launch_bounds__(1024) void k_skip_gate_bwd(const bf16* __restrict__ gcat, long gs, const bf16* __restrict__ skip,
                                                          const float* __restrict__ W1, const float* __restrict__ W2,
                                                          const float* __restrict__ gate, const float* __restrict__ z1save,
                                                          float* __restrict__ gmean, float* __restrict__ ws, int HW, int C,
                                                          int R) {
  skip_gate_bwd_body(gcat, gs, skip, W1, W2, gate, z1save, gmean, ws, HW, C, R, blockIdx.x);
}
// The per-sample backward passes of SEVERAL gates in one launch (round 6; the forward's k_skip_gate_fwd_multi has the why: one
// workgroup per sample is 128 workgroups on 256 CUs).  Nothing on the backward's critical chain needs a gate's gmean -- it only
// feeds the gradient of the U-Net skip, which the ENCODER's backward consumes -- so the decoder blocks queue their gate
// backward and the last of them launches all (networks.py _flush_sgb).  Same body, same values.
struct SgbItem {
  const bf16* gcat;
  long gs;
  const bf16* skip;
  const float* W1;
  const float* W2;
  const float* gate;
  const float* z1;
  float* gmean;
  float* ws;
  bf16* gskip;   // optional: the gradient of the U-Net skip, gcat * gate + gmean / HW (edm_skip_half_bwd's work), written by
                 // the workgroup that just reduced the sample (its gcat rows are still in the caches)
  int B, HW, R, blk0;
};
struct SgbGroup {
  SgbItem it[MAXSGF];
  int n, C, pad[2];
};
__global__ __launch_bounds__(1024) void k_skip_gate_bwd_multi(const SgbGroup* __restrict__ g) {
  int k = 0;
  const int n = g->n;
  while (k + 1 < n && (int)blockIdx.x >= g->it[k + 1].blk0) ++k;
  const SgbItem it = g->it[k];
  const int C = g->C, b = (int)blockIdx.x - it.blk0;
  skip_gate_bwd_body(it.gcat, it.gs, it.skip, it.W1, it.W2, it.gate, it.z1, it.gmean, it.ws, it.HW, C, it.R, b);
  if (it.gskip == nullptr) return;    // (workgroup-uniform)
  __syncthreads();                    // this sample's gmean row (global, written above by this workgroup) is complete
  const int CLs = C >> 3;
  const float inv_hw = 1.0f / (float)it.HW;
  const bf16* __restrict__ gc = it.gcat + (long)b * it.HW * it.gs;
  bf16* __restrict__ out = it.gskip + (long)b * it.HW * C;
  const float* gp0 = it.gate + (long)b * C;
  const float* mp0 = it.gmean + (long)b * C;
  const int n8 = it.HW * CLs;
  for (int i = threadIdx.x; i < n8; i += blockDim.x) {     // (as skip_half_bwd_body: same expression, same rounding)
    const int cs = (i % CLs) * 8;
    const int pix = i / CLs;
    float v[8];
    load8(gc + (long)pix * it.gs + cs, v);
    const float* gp = gp0 + cs;
    const float* mp = mp0 + cs;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * gp[j] + mp[j] * inv_hw;
    store8(out + (long)pix * C + cs, v);
  }
}
// gW2[c][r] = sum_b dz2[b][c] h[b][r];  gW1[r][c'] = sum_b dz1[b][r] [mean[b]; 1][c']   (b ascending: reproducible).
// A workgroup owns four columns of the C (resp. C+1) dimension and all R rows: it stages A[b][4] and Bm[b][R] in LDS with
// every load in flight at once, then each thread sums its (column, row) pairs over b.  blockIdx < CT: gW2, else gW1.
// The batch is walked in chunks of BC samples (what 64 KiB of LDS holds; one chunk up to B * (R + 4) = 16384), the
// partial sums stay in registers across chunks, so any batch size works and the order of the sum (b ascending) does not
// depend on the chunking.
__device__ __forceinline__ void skip_gate_wgrad_body(const float* __restrict__ ws, const float* __restrict__ mean,
                                                     float* __restrict__ gW1, float* __restrict__ gW2, int B, int C, int R,
                                                     int CT, int BC, int blk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // A[BC][4] | Bm[BC][R]
  float* A = sm;
  float* Bm = sm + 4 * BC;
  const int W = C + 2 * R;
  const bool second = blk >= CT;
  const int c0 = (second ? blk - CT : blk) * 4;
  const int ncol = second ? C + 1 : C;
  const int boff = second ? C : C + R;     // Bm = dz1 (gW1) or h (gW2)
  constexpr int MAXO = 16;                 // 4 * R / 256 outputs per thread, R <= 1024
  float acc[MAXO];
#pragma unroll
  for (int k = 0; k < MAXO; ++k) acc[k] = 0.f;
  for (int b0 = 0; b0 < B; b0 += BC) {
    const int nb = min(BC, B - b0);
    if (b0) __syncthreads();
    for (int i = threadIdx.x; i < 4 * nb; i += blockDim.x) {
      const int b = b0 + (i >> 2), c = c0 + (i & 3);
      float v = 0.f;
      if (c < ncol) v = second ? (c < C ? mean[(long)b * C + c] : 1.0f) : ws[(long)b * W + c];
      A[i] = v;
    }
    for (int i = threadIdx.x; i < nb * R; i += blockDim.x) Bm[i] = ws[(long)(b0 + i / R) * W + boff + i % R];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXO; ++k) {
      const int o = threadIdx.x + k * 256;
      if (o < 4 * R) {
        const int cc = o / R, r = o % R;
        float s = acc[k];
        for (int b = 0; b < nb; ++b) s += A[b * 4 + cc] * Bm[b * R + r];
        acc[k] = s;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXO; ++k) {
    const int o = threadIdx.x + k * 256;
    if (o >= 4 * R) continue;
    const int cc = o / R, r = o % R;
    if (c0 + cc >= ncol) continue;
    if (second) gW1[(long)r * (C + 1) + c0 + cc] = acc[k];
    else gW2[(long)(c0 + cc) * R + r] = acc[k];
  }
}
__global__ __launch_bounds__(256) void k_skip_gate_wgrad(const float* __restrict__ ws, const float* __restrict__ mean,
                                                           float* __restrict__ gW1, float* __restrict__ gW2, int B, int C,
                                                           int R, int CT, int BC) {
  skip_gate_wgrad_body(ws, mean, gW1, gW2, B, C, R, CT, BC, blockIdx.x);
}
// Round 6: the batch sums of ALL the ScaleLong gates of a backward pass in ONE launch behind it (nine launches of 129
// workgroups each in the CIFAR-10 step, every one a kernel boundary on the backward's critical chain).  The table lives in
// device memory (common.h "launch tables"); blk0 = first workgroup of a layer.
constexpr int MAXSG = 32;
struct SgItem {
  const float* ws;
  const float* mean;
  float* gW1;
  float* gW2;
  int B, C, R, blk0;
};
struct SgGroup {
  SgItem it[MAXSG];
  int n, pad[3];
};
__global__ __launch_bounds__(256) void k_skip_gate_wgrad_multi(const SgGroup* __restrict__ g) {
  int k = 0;
  const int n = g->n;
  while (k + 1 < n && (int)blockIdx.x >= g->it[k + 1].blk0) ++k;
  const SgItem it = g->it[k];
  const int CT = (it.C + 3) / 4;
  const int BC = it.B < 16384 / (it.R + 4) ? it.B : 16384 / (it.R + 4);
  skip_gate_wgrad_body(it.ws, it.mean, it.gW1, it.gW2, it.B, it.C, it.R, CT, BC, (int)blockIdx.x - it.blk0);
}
static inline int skip_gate_threads(int C) {   // a multiple of the C/8 lanes of a pixel row, <= 1024
  const int lpr = C / 8;
  int t = 1024 / lpr * lpr;
  return t < 64 ? 0 : t;
}
// skip [B*HW][C] bf16, W1h [R][C+1], W2h [C][R] fp32 -> mean, gate [B][C], z1save [B][R]   (C % 8 == 0, C <= 4096)
extern "C" int edm_skip_gate_fwd(const void* skip, const float* W1h, const float* W2h, float* mean, float* gate,
                                 float* z1save, int B, int HW, int C, int R, hipStream_t st) {
  EDM_REQUIRE(skip && W1h && W2h && mean && gate && z1save, "skip_gate_fwd: null pointer");
  EDM_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 8 == 0 && C <= 4096 && R > 0 && R <= 1024, "skip_gate_fwd: bad args");
  const int threads = skip_gate_threads(C);
  EDM_REQUIRE(threads > 0, "skip_gate_fwd: C too large");
  const size_t lds = ((size_t)(threads / (C / 8)) * C + C + 1 + R) * sizeof(float);
  hipLaunchKernelGGL(k_skip_gate_fwd, dim3(B), dim3(threads), lds, st, (const bf16*)skip, W1h, W2h, mean, gate, z1save,
                     HW, C, R);
  EDM_CHECK_LAUNCH("skip_gate_fwd");
  return EDM_OK;
}
struct edm_skip_gate_fwd_item_ {   // = edm_skip_gate_fwd_item (include/tinyedm_hip.h)
  const void* skip;
  const float* W1h;
  const float* W2h;
  float* mean;
  float* gate;
  float* z1save;
  void* cat;
  void* silu_out;
  int B, HW, C, R, Ci, pad;
};
extern "C" long edm_skip_gate_fwd_multi_table_bytes(void) { return (long)sizeof(SgfGroup); }
// edm_skip_gate_fwd for up to 32 skip tensors of ONE channel count in one launch; `items` is host memory read during the
// call, the table goes to device memory (launch tables, include/tinyedm_hip.h)
extern "C" int edm_skip_gate_fwd_multi(const void* items_, int n, void* table_host, void* table_dev, int defer_upload,
                                       hipStream_t st) {
  const edm_skip_gate_fwd_item_* items = (const edm_skip_gate_fwd_item_*)items_;
  EDM_REQUIRE(items && n > 0 && n <= MAXSGF, "skip_gate_fwd_multi: need 1..%d gates, got %d", MAXSGF, n);
  SgfGroup g;
  g.n = n;
  g.C = items[0].C;
  g.pad[0] = g.pad[1] = 0;
  const int C = g.C;
  EDM_REQUIRE(C > 0 && C % 8 == 0 && C <= 4096, "skip_gate_fwd_multi: bad channel count %d", C);
  const int threads = skip_gate_threads(C);
  EDM_REQUIRE(threads > 0, "skip_gate_fwd_multi: C too large");
  long blk = 0;
  int rmax = 0;
  for (int k = 0; k < n; ++k) {
    const edm_skip_gate_fwd_item_& a = items[k];
    EDM_REQUIRE(a.skip && a.W1h && a.W2h && a.mean && a.gate && a.z1save, "skip_gate_fwd_multi: null pointer (gate %d)", k);
    EDM_REQUIRE(a.C == C, "skip_gate_fwd_multi: the gates of one launch share a channel count (%d vs %d)", a.C, C);
    EDM_REQUIRE(a.B > 0 && a.HW > 0 && a.R > 0 && a.R <= 1024, "skip_gate_fwd_multi: bad item %d", k);
    EDM_REQUIRE(a.cat ? (a.Ci > 0 && a.Ci % 8 == 0) : (a.silu_out == nullptr),
                "skip_gate_fwd_multi: cat needs Ci %% 8 == 0 (gate %d); silu_out needs cat", k);
    g.it[k] = SgfItem{(const bf16*)a.skip, a.W1h, a.W2h, a.mean, a.gate, a.z1save, (bf16*)a.cat, (bf16*)a.silu_out, a.Ci, 0,
                      a.B, a.HW, a.R, (int)blk};
    blk += a.B;
    EDM_REQUIRE(blk < (1L << 30), "skip_gate_fwd_multi: grid too large");
    if (a.R > rmax) rmax = a.R;
  }
  for (int k = n; k < MAXSGF; ++k)
    g.it[k] = SgfItem{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, (int)blk};
  const size_t lds = ((size_t)(threads / (C / 8)) * C + C + 1 + rmax) * sizeof(float);
  EDM_UPLOAD_TABLE(table_dev, table_host, &g, sizeof(SgfGroup), st, "skip_gate_fwd_multi", defer_upload);
  hipLaunchKernelGGL(k_skip_gate_fwd_multi, dim3((unsigned)blk), dim3(threads), lds, st, (const SgfGroup*)table_dev);
  EDM_CHECK_LAUNCH("skip_gate_fwd_multi");
  return EDM_OK;
}
// gcat rows of gcat_stride elements whose channels [c_off, c_off + C) are the gradient of skip * gate; gmean [B][C],
// gW1h [R][C+1] and gW2h [C][R] are WRITTEN; ws: [B][C + 2R] floats of scratch (two launches: per-sample pass, then the
// batch sums of the weight gradients)
extern "C" int edm_skip_gate_bwd(const void* gcat, long gcat_stride, int c_off, const void* skip, const float* mean,
                                 const float* W1h, const float* W2h, const float* gate, const float* z1save, float* gmean,
                                 float* gW1h, float* gW2h, float* ws, int B, int HW, int C, int R, hipStream_t st) {
  EDM_REQUIRE(gcat && skip && mean && W1h && W2h && gate && z1save && gmean && ws && (!gW1h == !gW2h), "skip_gate_bwd: null pointer");
  EDM_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 8 == 0 && C <= 4096 && R > 0 && R <= 1024 && c_off >= 0 && c_off % 8 == 0 &&
                  gcat_stride >= c_off + C && gcat_stride % 8 == 0, "skip_gate_bwd: bad args");
  const int threads = skip_gate_threads(C);
  EDM_REQUIRE(threads > 0, "skip_gate_bwd: C too large");
  const size_t lds = ((size_t)(threads / (C / 8)) * C + C + R) * sizeof(float);
  hipLaunchKernelGGL(k_skip_gate_bwd, dim3(B), dim3(threads), lds, st, (const bf16*)gcat + c_off, gcat_stride,
                     (const bf16*)skip, W1h, W2h, gate, z1save, gmean, ws, HW, C, R);
  if (gW1h) {   // (NULL, NULL: the caller sums the weight gradients of several gates later, edm_skip_gate_wgrad_multi)
    const int CT = (C + 3) / 4, CT1 = (C + 4) / 4;
    const int BC = B < 16384 / (R + 4) ? B : 16384 / (R + 4);      // samples per LDS chunk (64 KiB); R <= 1024 -> BC >= 15
    const size_t lds2 = ((size_t)4 * BC + (size_t)BC * R) * sizeof(float);
    hipLaunchKernelGGL(k_skip_gate_wgrad, dim3(CT + CT1), dim3(256), lds2, st, ws, mean, gW1h, gW2h, B, C, R, CT, BC);
  }
  EDM_CHECK_LAUNCH("skip_gate_bwd");
  return EDM_OK;
}

struct edm_skip_gate_bwd_item_ {   // = edm_skip_gate_bwd_item (include/tinyedm_hip.h)
  const void* gcat;
  long gcat_stride;
  const void* skip;
  const float* W1h;
  const float* W2h;
  const float* gate;
  const float* z1save;
  float* gmean;
  float* ws;
  void* gskip;
  int c_off, B, HW, C, R, pad;
};
// the first launch of edm_skip_gate_bwd (gW1h == gW2h == NULL form: gmean and ws written) for up to 32 gates of ONE channel
// count at once
extern "C" int edm_skip_gate_bwd_multi(const void* items_, int n, void* table_host, void* table_dev, int defer_upload,
                                       hipStream_t st) {
  const edm_skip_gate_bwd_item_* items = (const edm_skip_gate_bwd_item_*)items_;
  EDM_REQUIRE(items && n > 0 && n <= MAXSGF, "skip_gate_bwd_multi: need 1..%d gates, got %d", MAXSGF, n);
  SgbGroup g;
  g.n = n;
  g.C = items[0].C;
  g.pad[0] = g.pad[1] = 0;
  const int C = g.C;
  EDM_REQUIRE(C > 0 && C % 8 == 0 && C <= 4096, "skip_gate_bwd_multi: bad channel count %d", C);
  const int threads = skip_gate_threads(C);
  EDM_REQUIRE(threads > 0, "skip_gate_bwd_multi: C too large");
  long blk = 0;
  int rmax = 0;
  for (int k = 0; k < n; ++k) {
    const edm_skip_gate_bwd_item_& a = items[k];
    EDM_REQUIRE(a.gcat && a.skip && a.W1h && a.W2h && a.gate && a.z1save && a.gmean && a.ws,
                "skip_gate_bwd_multi: null pointer (gate %d)", k);
    EDM_REQUIRE(a.C == C, "skip_gate_bwd_multi: the gates of one launch share a channel count (%d vs %d)", a.C, C);
    EDM_REQUIRE(a.B > 0 && a.HW > 0 && a.R > 0 && a.R <= 1024 && a.c_off >= 0 && a.c_off % 8 == 0 &&
                    a.gcat_stride >= a.c_off + C && a.gcat_stride % 8 == 0, "skip_gate_bwd_multi: bad item %d", k);
    g.it[k] = SgbItem{(const bf16*)a.gcat + a.c_off, a.gcat_stride, (const bf16*)a.skip, a.W1h, a.W2h, a.gate, a.z1save,
                      a.gmean, a.ws, (bf16*)a.gskip, a.B, a.HW, a.R, (int)blk};
    blk += a.B;
    EDM_REQUIRE(blk < (1L << 30), "skip_gate_bwd_multi: grid too large");
    if (a.R > rmax) rmax = a.R;
  }
  for (int k = n; k < MAXSGF; ++k)
    g.it[k] = SgbItem{nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, (int)blk};
  const size_t lds = ((size_t)(threads / (C / 8)) * C + C + rmax) * sizeof(float);
  EDM_UPLOAD_TABLE(table_dev, table_host, &g, sizeof(SgbGroup), st, "skip_gate_bwd_multi", defer_upload);
  hipLaunchKernelGGL(k_skip_gate_bwd_multi, dim3((unsigned)blk), dim3(threads), lds, st, (const SgbGroup*)table_dev);
  EDM_CHECK_LAUNCH("skip_gate_bwd_multi");
  return EDM_OK;
}
struct edm_skip_gate_wgrad_item_ {   // = edm_skip_gate_wgrad_item (include/tinyedm_hip.h)
  const float* ws;
  const float* mean;
  float* gW1h;
  float* gW2h;
  int B, C, R, pad;
};
extern "C" long edm_skip_gate_wgrad_multi_table_bytes(void) { return (long)sizeof(SgGroup); }
// the second launch of edm_skip_gate_bwd for up to 32 gates at once: items[k] = {ws, mean of gate k's forward, gW1h [R][C+1],
// gW2h [C][R] (WRITTEN), B, C, R}; `items` is host memory read during the call, the table goes to device memory
extern "C" int edm_skip_gate_wgrad_multi(const void* items_, int n, void* table_host, void* table_dev, int defer_upload,
                                         hipStream_t st) {
  const edm_skip_gate_wgrad_item_* items = (const edm_skip_gate_wgrad_item_*)items_;
  EDM_REQUIRE(items && n > 0 && n <= MAXSG, "skip_gate_wgrad_multi: need 1..%d gates, got %d", MAXSG, n);
  SgGroup g;
  g.n = n;
  int blk = 0;
  size_t lds = 0;
  for (int k = 0; k < n; ++k) {
    const edm_skip_gate_wgrad_item_& a = items[k];
    EDM_REQUIRE(a.ws && a.mean && a.gW1h && a.gW2h && a.B > 0 && a.C > 0 && a.C % 8 == 0 && a.C <= 4096 && a.R > 0 && a.R <= 1024,
                "skip_gate_wgrad_multi: bad item %d", k);
    g.it[k] = SgItem{a.ws, a.mean, a.gW1h, a.gW2h, a.B, a.C, a.R, blk};
    blk += (a.C + 3) / 4 + (a.C + 4) / 4;
    const int BC = a.B < 16384 / (a.R + 4) ? a.B : 16384 / (a.R + 4);
    const size_t need = ((size_t)4 * BC + (size_t)BC * a.R) * sizeof(float);
    if (need > lds) lds = need;
  }
  EDM_MAX_LDS(k_skip_gate_wgrad_multi, 128 * 1024);
  EDM_UPLOAD_TABLE(table_dev, table_host, &g, sizeof(SgGroup), st, "skip_gate_wgrad_multi", defer_upload);
  hipLaunchKernelGGL(k_skip_gate_wgrad_multi, dim3(blk), dim3(256), lds, st, (const SgGroup*)table_dev);
  EDM_CHECK_LAUNCH("skip_gate_wgrad_multi");
  return EDM_OK;
}

// cat[b,p,:Ci] = inp ; cat[b,p,Ci:] = skip*gate[b,:]     (and optionally s = mp_silu(cat))
__global__ void k_concat_gate_fwd(const bf16* __restrict__ inp, const bf16* __restrict__ skip,
                                  const float* __restrict__ gate, bf16* __restrict__ cat, bf16* __restrict__ sil,
                                  int HW, int Ci, int Cs, long n8) {
  const int CLt = (Ci + Cs) >> 3, CLi = Ci >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    int c8 = (int)(i % CLt);
    long pix = i / CLt;
    float v[8];
    if (c8 < CLi) {
      load8(inp + pix * Ci + c8 * 8, v);
    } else {
      int cs = (c8 - CLi) * 8;
      int b = (int)(pix / HW);
      load8(skip + pix * Cs + cs, v);
      const float* gp = gate + (long)b * Cs + cs;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)(bf16)(v[j] * gp[j]);
    }
    store8(cat + i * 8, v);
    if (sil) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = mp_silu_b(v[j]);
      store8(sil + i * 8, v);
    }
  }
}
// ginp = gcat[..., :Ci] ; gskip = gcat[..., Ci:]*gate + gmean/HW
__global__ void k_concat_gate_bwd(const bf16* __restrict__ gcat, const float* __restrict__ gate,
                                  const float* __restrict__ gmean, bf16* __restrict__ ginp, bf16* __restrict__ gskip,
                                  int HW, int Ci, int Cs, long n8, float inv_hw) {
  const int CLt = (Ci + Cs) >> 3, CLi = Ci >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    int c8 = (int)(i % CLt);
    long pix = i / CLt;
    float v[8];
    load8(gcat + i * 8, v);
    if (c8 < CLi) {
      store8(ginp + pix * Ci + c8 * 8, v);
    } else {
      int cs = (c8 - CLi) * 8;
      int b = (int)(pix / HW);
      const float* gp = gate + (long)b * Cs + cs;
      const float* mp = gmean + (long)b * Cs + cs;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * gp[j] + mp[j] * inv_hw;
      store8(gskip + pix * Cs + cs, v);
    }
  }
}
// The skip half alone (round 4): cat[b,p,Ci:] = skip*gate[b,:] and sil[b,p,Ci:] = mp_silu(of it), rows of Ci + Cs elements.
// The input half of both buffers is written by the kernel that PRODUCES `input` (edm_conv_igemm_o: strided output +
// mp_silu output), so torch.cat((input, skip * gate)) (networks.py:311) costs one read of the skip and two half-row writes.
__global__ void k_skip_half_fwd(const bf16* __restrict__ skip, const float* __restrict__ gate, bf16* __restrict__ cat,
                                bf16* __restrict__ sil, int HW, int Ci, int Cs, long n8) {
  const int CLs = Cs >> 3;
  const long ld = Ci + Cs;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int cs = (int)(i % CLs) * 8;
    const long pix = i / CLs;
    const int b = (int)(pix / HW);
    float v[8];
    load8(skip + pix * Cs + cs, v);
    const float* gp = gate + (long)b * Cs + cs;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)(bf16)(v[j] * gp[j]);
    store8(cat + pix * ld + Ci + cs, v);
    if (sil) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = mp_silu_b(v[j]);
      store8(sil + pix * ld + Ci + cs, v);
    }
  }
}
// gskip = gcs*gate + gmean/HW  (gcs: the skip half of d loss / d cat, [B*HW][Cs], written by the split-output 1x1 dgrad)
__device__ __forceinline__ void skip_half_bwd_body(const bf16* __restrict__ gcs, const float* __restrict__ gate,
                                                   const float* __restrict__ gmean, bf16* __restrict__ gskip, int HW, int Cs,
                                                   long n8, float inv_hw, long first, long stride) {
  const int CLs = Cs >> 3;
  for (long i = first; i < n8; i += stride) {
    const int cs = (int)(i % CLs) * 8;
    const int b = (int)((i / CLs) / HW);
    float v[8];
    load8(gcs + i * 8, v);
    const float* gp = gate + (long)b * Cs + cs;
    const float* mp = gmean + (long)b * Cs + cs;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * gp[j] + mp[j] * inv_hw;
    store8(gskip + i * 8, v);
  }
}
__global__ void k_skip_half_bwd(const bf16* __restrict__ gcs, const float* __restrict__ gate,
                                const float* __restrict__ gmean, bf16* __restrict__ gskip, int HW, int Cs, long n8,
                                float inv_hw) {
  skip_half_bwd_body(gcs, gate, gmean, gskip, HW, Cs, n8, inv_hw, (long)blockIdx.x * blockDim.x + threadIdx.x,
                     (long)gridDim.x * blockDim.x);
}
// ... and the gradients of the gated skips of those gates, one launch: item k owns the workgroups [blk0, blk0 + nblk)
struct ShbItem {
  const bf16* gcs;
  const float* gate;
  const float* gmean;
  bf16* gskip;
  long n8;
  float inv_hw;
  int HW, Cs, blk0, nblk, pad;
};
struct ShbGroup {
  ShbItem it[MAXSGF];
  int n, pad[3];
};
__global__ void k_skip_half_bwd_multi(const ShbGroup* __restrict__ g) {
  int k = 0;
  const int n = g->n;
  while (k + 1 < n && (int)blockIdx.x >= g->it[k + 1].blk0) ++k;
  const ShbItem it = g->it[k];
  skip_half_bwd_body(it.gcs, it.gate, it.gmean, it.gskip, it.HW, it.Cs, it.n8, it.inv_hw,
                     (long)((int)blockIdx.x - it.blk0) * blockDim.x + threadIdx.x, (long)it.nblk * blockDim.x);
}
extern "C" int edm_skip_half_fwd(const void* skip, const float* gate, void* cat, void* silu_out, int B, int HW, int Ci,
                                 int Cs, hipStream_t st) {
  EDM_REQUIRE(skip && gate && cat, "skip_half_fwd: null pointer");
  EDM_REQUIRE(B > 0 && HW > 0 && Ci % 8 == 0 && Cs % 8 == 0 && Ci > 0 && Cs > 0, "skip_half_fwd: bad args");
  long n8 = (long)B * HW * Cs / 8;
  hipLaunchKernelGGL(k_skip_half_fwd, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const bf16*)skip, gate, (bf16*)cat,
                     (bf16*)silu_out, HW, Ci, Cs, n8);
  EDM_CHECK_LAUNCH("skip_half_fwd");
  return EDM_OK;
}
extern "C" int edm_skip_half_bwd(const void* gcs, const float* gate, const float* gmean, void* gskip, int B, int HW, int Cs,
                                 hipStream_t st) {
  EDM_REQUIRE(gcs && gate && gmean && gskip, "skip_half_bwd: null pointer");
  EDM_REQUIRE(B > 0 && HW > 0 && Cs % 8 == 0 && Cs > 0, "skip_half_bwd: bad args");
  long n8 = (long)B * HW * Cs / 8;
  hipLaunchKernelGGL(k_skip_half_bwd, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const bf16*)gcs, gate, gmean,
                     (bf16*)gskip, HW, Cs, n8, 1.0f / (float)HW);
  EDM_CHECK_LAUNCH("skip_half_bwd");
  return EDM_OK;
}
extern "C" long edm_skip_gate_bwd_multi_table_bytes(void) {
  return (long)(sizeof(SgbGroup) > sizeof(ShbGroup) ? sizeof(SgbGroup) : sizeof(ShbGroup));
}
struct edm_skip_half_bwd_item_ {   // = edm_skip_half_bwd_item (include/tinyedm_hip.h)
  const void* gcs;
  const float* gate;
  const float* gmean;
  void* gskip;
  int B, HW, Cs, pad;
};
// edm_skip_half_bwd for up to 32 tensors at once (table of edm_skip_gate_bwd_multi_table_bytes() bytes)
extern "C" int edm_skip_half_bwd_multi(const void* items_, int n, void* table_host, void* table_dev, int defer_upload,
                                       hipStream_t st) {
  const edm_skip_half_bwd_item_* items = (const edm_skip_half_bwd_item_*)items_;
  EDM_REQUIRE(items && n > 0 && n <= MAXSGF, "skip_half_bwd_multi: need 1..%d tensors, got %d", MAXSGF, n);
  ShbGroup g;
  g.n = n;
  g.pad[0] = g.pad[1] = g.pad[2] = 0;
  long blk = 0;
  for (int k = 0; k < n; ++k) {
    const edm_skip_half_bwd_item_& a = items[k];
    EDM_REQUIRE(a.gcs && a.gate && a.gmean && a.gskip, "skip_half_bwd_multi: null pointer (tensor %d)", k);
    EDM_REQUIRE(a.B > 0 && a.HW > 0 && a.Cs % 8 == 0 && a.Cs > 0, "skip_half_bwd_multi: bad item %d", k);
    const long n8 = (long)a.B * a.HW * a.Cs / 8;
    const int nblk = grid_for(n8, 256);
    g.it[k] = ShbItem{(const bf16*)a.gcs, a.gate, a.gmean, (bf16*)a.gskip, n8, 1.0f / (float)a.HW, a.HW, a.Cs, (int)blk, nblk, 0};
    blk += nblk;
    EDM_REQUIRE(blk < (1L << 30), "skip_half_bwd_multi: grid too large");
  }
  for (int k = n; k < MAXSGF; ++k) g.it[k] = ShbItem{nullptr, nullptr, nullptr, nullptr, 0, 0.f, 1, 8, (int)blk, 0, 0};
  EDM_UPLOAD_TABLE(table_dev, table_host, &g, sizeof(ShbGroup), st, "skip_half_bwd_multi", defer_upload);
  hipLaunchKernelGGL(k_skip_half_bwd_multi, dim3((unsigned)blk), dim3(256), 0, st, (const ShbGroup*)table_dev);
  EDM_CHECK_LAUNCH("skip_half_bwd_multi");
  return EDM_OK;
}

extern "C" int edm_concat_gate_fwd(const void* inp, const void* skip, const float* gate, void* cat, void* silu_out,
                                   int B, int HW, int Ci, int Cs, hipStream_t st) {
  EDM_REQUIRE(B > 0 && HW > 0 && Ci % 8 == 0 && Cs % 8 == 0 && Ci > 0 && Cs > 0, "concat_gate_fwd: bad args");
  long n8 = (long)B * HW * (Ci + Cs) / 8;
  hipLaunchKernelGGL(k_concat_gate_fwd, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const bf16*)inp,
                     (const bf16*)skip, gate, (bf16*)cat, (bf16*)silu_out, HW, Ci, Cs, n8);
  EDM_CHECK_LAUNCH("concat_gate_fwd");
  return EDM_OK;
}
extern "C" int edm_concat_gate_bwd(const void* gcat, const float* gate, const float* gmean, void* ginp, void* gskip,
                                   int B, int HW, int Ci, int Cs, hipStream_t st) {
  EDM_REQUIRE(B > 0 && HW > 0 && Ci % 8 == 0 && Cs % 8 == 0 && Ci > 0 && Cs > 0, "concat_gate_bwd: bad args");
  long n8 = (long)B * HW * (Ci + Cs) / 8;
  hipLaunchKernelGGL(k_concat_gate_bwd, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const bf16*)gcat, gate, gmean,
                     (bf16*)ginp, (bf16*)gskip, HW, Ci, Cs, n8, 1.0f / (float)HW);
  EDM_CHECK_LAUNCH("concat_gate_bwd");
  return EDM_OK;
}

// ------------------------------------------------------------------ preconditioning (networks.py:578-587, 602-603)
// out[b,h,w,:] = [ c_in(b)*noisy[b,:,h,w] , 1 , 0... ]  bf16, CP channels
__global__ void k_precond_in(const float* __restrict__ noisy, const float* __restrict__ sigma, int sstride, float sd,
                             bf16* __restrict__ out, int Cimg, int HW, int CP, long npix) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    long b = p / HW;
    int hw = (int)(p % HW);
    float s = sigma[b * sstride];
    float cin = rsqrtf(sd * sd + s * s);
    bf16* o = out + p * CP;
    for (int c0 = 0; c0 < CP; c0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int c = c0 + j;
        v[j] = c < Cimg ? cin * noisy[(b * Cimg + c) * HW + hw] : (c == Cimg ? 1.0f : 0.0f);
      }
      store8(o + c0, v);
    }
  }
}
extern "C" int edm_precond_in(const float* noisy, const float* sigma, int sigma_stride, float sigma_data, void* out,
                              int B, int Cimg, int HW, int CP, hipStream_t st) {
  EDM_REQUIRE(B > 0 && Cimg > 0 && HW > 0 && CP % 8 == 0 && CP > Cimg && (sigma_stride == 0 || sigma_stride == 1),
              "precond_in: bad args");
  long npix = (long)B * HW;
  hipLaunchKernelGGL(k_precond_in, dim3(grid_for(npix, 256)), dim3(256), 0, st, noisy, sigma, sigma_stride,
                     sigma_data, (bf16*)out, Cimg, HW, CP, npix);
  EDM_CHECK_LAUNCH("precond_in");
  return EDM_OK;
}

// conv_out (1x1, C -> Co<=8) fused with the output preconditioning:
//   F[b,o,hw] = sum_c x[p,c]*wh[o,c] ;  D = F*gain_out*c_out(b) + noisy*c_skip(b)     (D, F fp32 NCHW)
template <int LPP>
__global__ __launch_bounds__(256) void k_conv_out_fwd(const bf16* __restrict__ x, const float* __restrict__ wh,
                                                        const float* __restrict__ gain_out,
                                                        const float* __restrict__ noisy,
                                                        const float* __restrict__ sigma, int sstride, float sd,
                                                        float* __restrict__ D, float* __restrict__ Fraw, int HW,
                                                        int C, int Co, long npix) {
  constexpr int GPW = 64 / LPP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lig = lane % LPP, grp = lane / LPP;
  const int CL = C >> 3;
  const float go = *gain_out;
  const long stride = (long)gridDim.x * 4 * GPW;
  for (long p0 = ((long)blockIdx.x * 4 + wave) * GPW; p0 < npix; p0 += stride) {
    const long p = p0 + grp;
    const bool pv = p < npix;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pv) {
      for (int c8 = lig; c8 < CL; c8 += LPP) {
        float v[8];
        load8(x + p * C + c8 * 8, v);
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          if (o < Co) {
            const float* wp = wh + (long)o * C + c8 * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[o] += v[j] * wp[j];
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < Co) acc[o] = group_sum<LPP>(acc[o]);   // (Co is 3 or 4: the unconditional form spent 25 of its 40 shuffles on zeros)
    if (pv && lig == 0) {
      const int b = (int)p / HW, hw = (int)p - b * HW;     // (npix < 2^31: host-checked; the 64-bit division cost 100+ instructions)
      float s = sigma[b * sstride];
      float den = s * s + sd * sd;
      float cskip = sd * sd / den, cout = s * sd * rsqrtf(den);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        if (o < Co) {
          long idx = ((long)b * Co + o) * HW + hw;
          if (Fraw) Fraw[idx] = acc[o];
          D[idx] = acc[o] * go * cout + noisy[idx] * cskip;
        }
      }
    }
  }
}
extern "C" int edm_conv_out_fwd(const void* x, const float* w_hat, const float* gain_out, const float* noisy,
                                const float* sigma, int sigma_stride, float sigma_data, float* D, float* Fraw, int B,
                                int HW, int C, int Co, hipStream_t st) {
  EDM_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && Co >= 1 && Co <= 8 && (sigma_stride == 0 || sigma_stride == 1),
              "conv_out_fwd: bad args (Co<=8 required)");
  long npix = (long)B * HW;
  EDM_REQUIRE(npix < (1L << 31), "conv_out_fwd: too many pixels");
  // 8 lanes per pixel (8 pixels per wave trip, four 16-byte loads per lane in flight, 9 shuffles per 8 pixels): 26.5 us at the
  // CIFAR-10 size against 44.8 with 32 lanes per pixel (one load per lane, 15 shuffles per 2 pixels), 32.1 with 16, 29.2 with 4
  hipLaunchKernelGGL(k_conv_out_fwd<8>, dim3(grid_for(npix, 32)), dim3(256), 0, st, (const bf16*)x, w_hat, gain_out,
                     noisy, sigma, sigma_stride, sigma_data, D, Fraw, HW, C, Co, npix);
  EDM_CHECK_LAUNCH("conv_out_fwd");
  return EDM_OK;
}

// backward of conv_out + preconditioning.  dD fp32 NCHW.
//   dF = dD*c_out*gain_out ; gx[p,c] = sum_o dF[p,o]*wh[o,c]
__global__ void k_conv_out_bwd_x(const float* __restrict__ dD, const float* __restrict__ wh,
                                 const float* __restrict__ gain_out, const float* __restrict__ sigma, int sstride,
                                 float sd, bf16* __restrict__ gx, int HW, int C, int Co, long n8) {
  // (32-bit index arithmetic: n8 < 2^31 is host-checked; the 64-bit i / CL, p / HW, p % HW of the first form were ~300
  // instructions per 16 output bytes)
  const unsigned CL = (unsigned)C >> 3, uHW = (unsigned)HW;
  const float go = *gain_out;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)n8; i += gridDim.x * blockDim.x) {
    const unsigned p = i / CL, c8 = i - p * CL;
    const unsigned b = p / uHW, hw = p - b * uHW;
    float s = sigma[b * sstride];
    float cout = s * sd * rsqrtf(s * s + sd * sd) * go;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int o = 0; o < Co; ++o) {
      float df = dD[((long)b * Co + o) * HW + hw] * cout;
      const float* wp = wh + (long)o * C + c8 * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += df * wp[j];
    }
    store8(gx + (long)i * 8, v);
  }
}
//   gwh[o,c] += sum_p dF[p,o]*x[p,c] ;  ggain += sum dD*c_out*F
// block = CL*PS threads (thread -> fixed 8-channel chunk c8, pixel phase ps), grid = ceil(npix/PIXW);
// partial sums are reduced across the PS phases in LDS, then ONE atomic per (o,c) per workgroup.
__global__ void k_conv_out_bwd_w(const bf16* __restrict__ x, const float* __restrict__ dD,
                                 const float* __restrict__ Fraw, const float* __restrict__ gain_out,
                                 const float* __restrict__ sigma, int sstride, float sd, float* __restrict__ gwh,
                                 float* __restrict__ ggain, int HW, int C, int Co, long npix, int PIXW) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [PS][Co*C]
  const int CL = C >> 3;
  const int PS = blockDim.x / CL;
  const int c8 = threadIdx.x % CL, ps = threadIdx.x / CL;
  const long p0 = (long)blockIdx.x * PIXW, p1 = min(npix, p0 + PIXW);
  const float go = *gain_out;
  float acc[8][8];
#pragma unroll
  for (int o = 0; o < 8; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
  float gg = 0.f;
  // (sample index, pixel-in-sample and the sample's c_out advance incrementally -- npix < 2^31 host-checked -- instead of a 64-bit division and a dependent sigma load per pixel; the next pixel's operands are loaded
  // before this one's are consumed: the trip was one exposed load -> use chain per pixel)
  int p = (int)p0 + ps;
  const int pend = (int)p1;
  int b = p / HW, hw = p - b * HW;
  float cout = 0.f;
  {
    const float s = sigma[(p < pend ? b : 0) * sstride];
    cout = s * sd * rsqrtf(s * s + sd * sd);
  }
  float vn[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dn[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto fetch = [&](int pp, int bb, int hh) {
    load8(x + (long)pp * C + c8 * 8, vn);
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < Co) {
        const long idx = ((long)bb * Co + o) * HW + hh;
        dn[o] = dD[idx];
        if (c8 == 0) fn[o] = Fraw[idx];
      }
  };
  if (p < pend) fetch(p, b, hw);
  while (p < pend) {
    float v[8], dd[8], ff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = vn[j]; dd[j] = dn[j]; ff[j] = fn[j]; }
    const float cur = cout;
    p += PS;
    hw += PS;
    if (hw >= HW) {
      do {
        hw -= HW;
        ++b;
      } while (hw >= HW);
      if (p < pend) {
        const float s = sigma[b * sstride];
        cout = s * sd * rsqrtf(s * s + sd * sd);
      }
    }
    if (p < pend) fetch(p, b, hw);
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      if (o < Co) {
        float d = dd[o] * cur;
        if (c8 == 0) gg += d * ff[o];
        d *= go;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[o][j] += d * v[j];
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 8; ++o)
    if (o < Co) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(ps * Co + o) * C + c8 * 8 + j] = acc[o][j];
    }
  __syncthreads();
  for (int e = threadIdx.x; e < Co * C; e += blockDim.x) {
    float sacc = 0.f;
    for (int q = 0; q < PS; ++q) sacc += red[q * Co * C + e];
    atomicAdd(gwh + e, sacc);
  }
  if (c8 == 0) atomicAdd(ggain, gg);
}
// gw_hat [Co,C] and ggain are accumulated (+=): caller zero-fills.
extern "C" int edm_conv_out_bwd(const void* x, const float* w_hat, const float* gain_out, const float* Fraw,
                                const float* dD, const float* sigma, int sigma_stride, float sigma_data, void* gx,
                                float* gw_hat, float* ggain, int B, int HW, int C, int Co, hipStream_t st) {
  EDM_REQUIRE(B > 0 && HW > 0 && C % 8 == 0 && Co >= 1 && Co <= 8 && (sigma_stride == 0 || sigma_stride == 1),
              "conv_out_bwd: bad args");
  long npix = (long)B * HW, n8 = npix * C / 8;
  EDM_REQUIRE(n8 < (1L << 31), "conv_out_bwd: too many elements (32-bit index arithmetic)");
  hipLaunchKernelGGL(k_conv_out_bwd_x, dim3(grid_for(n8, 256)), dim3(256), 0, st, dD, w_hat, gain_out, sigma,
                     sigma_stride, sigma_data, (bf16*)gx, HW, C, Co, n8);
  EDM_CHECK_LAUNCH("conv_out_bwd_x");
  {
    const int CL = C / 8;
    EDM_REQUIRE(CL <= 256, "conv_out_bwd: C=%d too large", C);
    int block = (256 / CL) * CL, PS = block / CL;
    while (PS > 1 && (size_t)PS * Co * C * sizeof(float) > 48 * 1024) { --PS; block = PS * CL; }
    const int PIXW = 256;   // sweep at the CIFAR-10 size, x + w kernels (us), round 4: 128 -> 89, 256 -> 73, 512 -> 80, 1024 -> 115
                            // (512 / 1024 threads per workgroup: no better)
    hipLaunchKernelGGL(k_conv_out_bwd_w, dim3(cdiv(npix, PIXW)), dim3(block), (size_t)PS * Co * C * sizeof(float), st,
                       (const bf16*)x, dD, Fraw, gain_out, sigma, sigma_stride, sigma_data, gw_hat, ggain, HW, C, Co,
                       npix, PIXW);
  }
  EDM_CHECK_LAUNCH("conv_out_bwd_w");
  return EDM_OK;
}

// ------------------------------------------------------------------ layout helpers (boundary <-> kernel layout)
// NCHW fp32 -> NHWC bf16 and back (tests / boundary plumbing; C multiple of 8)
__global__ void k_nchw_f32_to_nhwc_bf16(const float* __restrict__ x, bf16* __restrict__ y, int C, int HW, long n8) {
  const int CL = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    int c8 = (int)(i % CL);
    long p = i / CL;
    long b = p / HW;
    int hw = (int)(p % HW);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[(b * C + c8 * 8 + j) * HW + hw];
    store8(y + i * 8, v);
  }
}
__global__ void k_nhwc_bf16_to_nchw_f32(const bf16* __restrict__ x, float* __restrict__ y, int C, int HW, long n8) {
  const int CL = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    int c8 = (int)(i % CL);
    long p = i / CL;
    long b = p / HW;
    int hw = (int)(p % HW);
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[(b * C + c8 * 8 + j) * HW + hw] = v[j];
  }
}
extern "C" int edm_nchw_to_nhwc_bf16(const float* x, void* y, int B, int C, int HW, hipStream_t st) {
  EDM_REQUIRE(B > 0 && C % 8 == 0 && HW > 0, "nchw_to_nhwc_bf16: bad args");
  long n8 = (long)B * HW * C / 8;
  hipLaunchKernelGGL(k_nchw_f32_to_nhwc_bf16, dim3(grid_for(n8, 256)), dim3(256), 0, st, x, (bf16*)y, C, HW, n8);
  EDM_CHECK_LAUNCH("nchw_to_nhwc_bf16");
  return EDM_OK;
}
extern "C" int edm_nhwc_bf16_to_nchw(const void* x, float* y, int B, int C, int HW, hipStream_t st) {
  EDM_REQUIRE(B > 0 && C % 8 == 0 && HW > 0, "nhwc_bf16_to_nchw: bad args");
  long n8 = (long)B * HW * C / 8;
  hipLaunchKernelGGL(k_nhwc_bf16_to_nchw_f32, dim3(grid_for(n8, 256)), dim3(256), 0, st, (const bf16*)x, y, C, HW, n8);
  EDM_CHECK_LAUNCH("nhwc_bf16_to_nchw");
  return EDM_OK;
}

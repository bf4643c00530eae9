// Shared device helpers for the tinyedm_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define EDM_OK 0
#define EDM_ERR_ARG (-1)
#define EDM_ERR_LAUNCH (-2)
#define EDM_ERR_UNSUPPORTED (-3)

extern "C" void edm_set_error(const char* fmt, ...);
// per-device page of zeros (runtime.hip): allocated by edm_init(device), never inside a launch function
extern "C" const void* edm_zero_page(void);
#define EDM_ZERO_PAGE(var, name)          \
  const void* var = edm_zero_page();      \
  EDM_REQUIRE(var, name ": edm_init(device) has not been called for the current device")

#define EDM_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      edm_set_error(__VA_ARGS__);         \
      return EDM_ERR_ARG;                 \
    }                                     \
  } while (0)

#define EDM_CHECK_LAUNCH(name)                                        \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      edm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return EDM_ERR_LAUNCH;                                          \
    }                                                                 \
  } while (0)

// Launch TABLES live in device memory (round 4).  The grouped weight-gradient / finish kernels used to take their layer
// tables (1.3-2 KB) as by-value kernel arguments; under hipGraph capture those sat in the one suspect of the round-2
// corruption that could not be ruled out.  Now the host builds the table in caller-provided PINNED memory, one
// stream-ordered copy puts it into caller-provided device memory, and the kernels take a pointer: no kernel of a captured
// step carries more than ~200 bytes of arguments.  Caller contract: `host_stage` stays valid and unmodified until the copy
// has executed (for a captured stream: for the life of the graph), `dev` until the kernels have.
// `defer` != 0: the table is only written to host_stage (any host memory) and the CALLER copies it to `dev` before the launch
// can execute -- for a stream capture: once, after the capture has ended, instead of a memcpy node that every replay would
// run again (the table of a captured launch never changes between replays); ops._LaunchTables / ops.capture_end do that.
#define EDM_UPLOAD_TABLE(dev, host_stage, src, bytes, st, name, defer)                                                 \
  do {                                                                                                                 \
    EDM_REQUIRE((dev) && (host_stage), name ": the launch table needs a host staging buffer and a device buffer");     \
    memcpy((host_stage), (src), (bytes));                                                                              \
    if (!(defer) && hipMemcpyAsync((dev), (host_stage), (bytes), hipMemcpyHostToDevice, (st)) != hipSuccess) {         \
      edm_set_error("%s: uploading the launch table failed: %s", name, hipGetErrorString(hipGetLastError()));         \
      return EDM_ERR_LAUNCH;                                                                                           \
    }                                                                                                                  \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of (kernel, DEVICE): done once per device a process
// drives (a bit per device id), not once per process (round 3 guarded it with one bool: fine for one process per GPU, wrong
// for one process that launches on two).  Idempotent: a race only repeats the call.
inline void edm_max_lds_once(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
  }
}
#define EDM_MAX_LDS(kern, bytes)                                                   \
  do {                                                                             \
    static std::atomic<unsigned long long> done_{0};                               \
    edm_max_lds_once(reinterpret_cast<const void*>(kern), (bytes), done_);         \
  } while (0)

#define SILU_DIV 0.596f
#define NORM_EPS 1e-4f

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// magnitude-preserving SiLU  (reference networks.py:83-84)
__device__ __forceinline__ float mp_silu_f(float x) { return x * sigmoidf_(x) * (1.0f / SILU_DIV); }
// d/dx of mp_silu
__device__ __forceinline__ float mp_silu_grad_f(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s)) * (1.0f / SILU_DIV);
}

// The same two functions for results that are rounded to bf16 right away (the conv epilogues and the bf16 elementwise
// kernels): v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE division -- the epilogues run with no MFMA work to hide
// behind and are bound by the vector ALU.  The fp32 evaluation path, the Linears and the gate MLPs keep the exact forms.
__device__ __forceinline__ float sigmoid_b(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float mp_silu_b(float x) { return x * sigmoid_b(x) * (1.0f / SILU_DIV); }
__device__ __forceinline__ float mp_silu_grad_b(float x) {
  float s = sigmoid_b(x);
  return s * (1.0f + x * (1.0f - s)) * (1.0f / SILU_DIV);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void load8(const bf16* p, float (&f)[8]) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ void store8(bf16* p, const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
  *reinterpret_cast<bf16x8*>(p) = v;
}

// ---------------- Philox4x32-10 (counter-based RNG; stateless, replayable) -----------
struct Philox4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // the 64-bit products as ONE v_mad_u64_u32 each: hipcc emits v_mul_hi_u32 + v_mul_lo_u32 for __umulhi(a, b) and a * b,
    // two quarter-rate instructions where one does (tools/valu/philox_mul.hip: 546 vs 441 G calls/s, same bits)
    unsigned long long p0, p1, cy0, cy1;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p0), "=s"(cy0) : "v"(c0), "v"(0xD2511F53u));
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p1), "=s"(cy1) : "v"(c2), "v"(0xCD9E8D57u));
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u32_to_unit(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }  // [0,1)

// Dropout keep mask of the 8 elements of 16-byte chunk i8 (bit j = element j kept), shared by the forward and backward
// kernels and edm_dropout_mask: ONE Philox4x32-10 call per chunk, 16 random bits per element compared with
// round(p * 65536) (p = 0.13 -> 8520/65536; the rescale uses the same quantised p).  Round 1 drew 32 bits per
// element, two Philox calls per chunk: 25 us of a 170-us fused forward conv at 32x32.
struct Keep8 {
  uint32_t bits;
  float scale;
  __device__ __forceinline__ bool operator[](int j) const { return (bits >> j) & 1u; }
};
__device__ __forceinline__ Keep8 dropout_keep8(long i8, float p, uint32_t sub, uint32_t step, uint32_t seed_lo,
                                               uint32_t seed_hi) {
  Keep8 k{0xFFu, 1.0f};
  if (p > 0.f) {
    const Philox4 r = philox4x32_10((uint32_t)i8, (uint32_t)(i8 >> 32), sub, step, seed_lo, seed_hi);
    const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) m |= (((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) >= thr ? 1u : 0u) << j;
    k.bits = m;
    k.scale = 65536.0f / (float)(65536u - thr);
  }
  return k;
}

// Optional second epilogue output of the implicit-GEMM kernels: the embedding modulation + mp_silu + dropout that
// follows the first 3x3 conv of every block (networks.py:255-260 / 319-324), a2 = dropout(mp_silu(u*(lin*gain+1))),
// computed from the bf16-rounded conv output u exactly like k_mod_silu_drop_fwd (same Philox counters), so the
// backward kernel regenerates the same mask.  Y2 == nullptr: plain conv.
//
// Backward form (U != nullptr): the kernel is the dgrad of the block's SECOND conv, its bf16 result ga is consumed in
// the epilogue instead of being written: Y2 = gr = ga*keep*mp_silu'(u*m)*m and gm[b,c] += sum_px ga*keep*silu'(u*m)*u
// (same arithmetic as k_mod_silu_drop_bwd; needs HW % 32 == 0 so that a 32-pixel block never straddles images).
// Per-step scalars of a captured training step.  A by-value kernel argument is frozen into a hipGraph at capture
// time; the values below change on every step, so the host rewrites this 48-byte DEVICE record before each (re)play
// and kernels given a non-null `dyn` pointer read step/seed (Philox streams of dropout and the Diffuser) or
// lr / ema_beta / grad_scale / bias corrections (fused Adam) from it instead of from their by-value arguments.
struct StepParams {  // mirrored by include/tinyedm_hip.h (edm_step_params) and tinyedm_amd/graph.py
  uint32_t step, seed_lo, seed_hi, reserved;
  float lr, ema_beta, grad_scale, bc1, bc2sqrt, pad[3];
};
struct ModEpilogue {
  const float* lin;   // [B][lin_stride] fp32 embed-linear output
  const float* gain;  // device scalar
  bf16* Y2;           // second output, same shape as Y (forward: a2; backward: gr)
  long lin_stride;
  int HW;
  float pdrop;
  uint32_t seed_lo, seed_hi, sub, step;
  const bf16* U;      // backward forms: pre-activation tensor saved by the forward
  float* gm;          // modulation backward: [B][Cout] fp32, zero-filled by the caller, accumulated with atomics
  const bf16* ADD;    // silu backward: optional extra gradient, Y2 = mp_silu'(U)*g + add_scale*ADD
  float add_scale;
  int mode;           // 0: none / forward modulation (Y2 set), 1: modulation backward, 2: mp_silu backward,
                      // 3: plain + Y2 = mp_silu(Y) (see ldY below)
  const StepParams* dyn;  // non-null: step / seed of the Philox stream come from device memory (captured steps)
  long gm_stride;         // row stride of gm in floats; 0 = Cout (a private contiguous [B][Cout] buffer)
  int u_marks;            // forward: write a NaN into Y (= U) where the element was dropped; backward: U carries those
                          // marks (dropped <=> NaN), no Philox stream is regenerated
  // ---- output descriptor of the plain epilogue (round 4): where the result rows go.  The decoder's
  // torch.cat((input, skip * gate)) (networks.py:311) is never a copy: the kernel that PRODUCES `input` writes its rows
  // straight into the left half of the cat buffer (ldY = Ci + Cs) and mp_silu of them into the same half of the buffer the
  // block's first conv reads (mode 3); the kernel that produces d loss / d cat writes the two halves to the two tensors
  // their consumers read (split).
  long ldY;               // row stride of Y (and of a mode-3 Y2) in elements; 0 = Cout.  R / U / ADD stay contiguous.
  bf16* Yb;               // non-null: output channels >= split go to Yb[row * ldYb + (channel - split)]
  long ldYb;
  int split;              // multiple of 8
  // mode 3 (plain epilogue, EPI 0): Y2 = mp_silu(Y), same addressing as Y
  int wfrag;              // (not an epilogue matter, but it travels with the launch): the weight pack is FRAGMENT-MAJOR
                          // (weights.hip; only k_conv3x3_s reads that layout -- every other kernel refuses it)
  // ---- split-bf16 evaluation (mode 4, EPI 4): X rows are [hi | lo] bf16 pairs of an fp32 activation (row stride ldX
  // elements = 2 C), the pack is [tap][co][3 C] = [w_hi | w_lo | w_hi], and K chunk c reads X channels of chunk
  // (c < kwrap ? c : c - kwrap): hi.w_hi, hi.w_lo, lo.w_hi -- three bf16 passes, an fp32-accurate sum; Y and R are FLOATS
  int ldX;                // 0 = Cin
  int kwrap;              // 0 = off; chunks of 32 channels
  // ---- folded 1x1 projection (round 6; k_conv3x3_v6<..., FOLD = true> only): behind the nine taps the accumulators are
  // multiplied by fold_scale and a SECOND reduction runs over X2 [pixels][ldX2] -- centre tap only -- with the pack
  // W2 [Cout][C2] (C2 % 64 == 0): Y = alpha * (fold_scale * conv3x3(X, Wp) + conv1x1(X2, W2)).  The decoder block's skip
  // projection conv_1x1(cat) (networks.py:313) rides in the block's second 3x3 conv and its result is never written, rounded
  // to bf16 and read back as a residual.  kwrap2: as kwrap, for X2 (split-bf16 form).
  const bf16* X2;
  const bf16* W2;
  int C2;
  int ldX2;
  float fold_scale;
  int kwrap2;
};
constexpr uint32_t U_DROPPED = 0x7FFFu;   // the bf16 pattern of a dropped element in a marked U
__device__ __forceinline__ void apply_dyn(ModEpilogue& m) {
  if (m.dyn) {
    m.step = m.dyn->step;
    m.seed_lo = m.dyn->seed_lo;
    m.seed_hi = m.dyn->seed_hi;
  }
}
// mode 2 on 8 channels: gx = mp_silu'(x)*g + s*ge  (k_silu_bwd's arithmetic; g = the conv result rounded to bf16)
__device__ __forceinline__ u32x4 silu_bwd8(const u32x4& graw, const u32x4& xraw, const u32x4& eraw, bool add, float s) {
  const bf16x8 gv = __builtin_bit_cast(bf16x8, graw), xv = __builtin_bit_cast(bf16x8, xraw);
  const bf16x8 ev = __builtin_bit_cast(bf16x8, eraw);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)(mp_silu_grad_b((float)xv[j]) * (float)gv[j] + (add ? s * (float)ev[j] : 0.f));
  return __builtin_bit_cast(u32x4, o);
}
// backward form on 8 channels: returns gr, accumulates ga*keep*silu'(u*m)*u into part[]
__device__ __forceinline__ u32x4 mod_silu_drop_bwd8(const u32x4& garaw, const u32x4& uraw, long i8,
                                                    const float (&mv)[8], const ModEpilogue& m, float (&part)[8]) {
  const bf16x8 gv = __builtin_bit_cast(bf16x8, garaw), uv = __builtin_bit_cast(bf16x8, uraw);
  Keep8 keep{0xFFu, 1.0f};
  if (m.u_marks) {
    if (m.pdrop > 0.f) {
      const uint32_t thr = (uint32_t)(m.pdrop * 65536.0f + 0.5f);
      keep.scale = 65536.0f / (float)(65536u - thr);
    }
  } else {
    keep = dropout_keep8(i8, m.pdrop, m.sub, m.step, m.seed_lo, m.seed_hi);
  }
  const float keep_scale = keep.scale;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float mm = mv[j];
    float u = (float)uv[j];
    const bool kept = m.u_marks ? u == u : keep[j];
    u = kept ? u : 0.f;
    float gu = (float)gv[j] * mp_silu_grad_b(u * mm);
    if (m.pdrop > 0.f) gu = kept ? gu * keep_scale : 0.f;
    part[j] += gu * u;
    o[j] = (bf16)(gu * mm);
  }
  return __builtin_bit_cast(u32x4, o);
}
// (U with the dropped elements of `bits` marked)
__device__ __forceinline__ uint32_t mark2(uint32_t bits2) {   // two keep bits -> the OR mask of a bf16 pair
  return ((bits2 & 1u) ? 0u : U_DROPPED) | ((bits2 & 2u) ? 0u : U_DROPPED << 16);
}
__device__ __forceinline__ u32x4 mark_dropped8(const u32x4& uraw, uint32_t bits) {
  return u32x4{uraw[0] | mark2(bits), uraw[1] | mark2(bits >> 2), uraw[2] | mark2(bits >> 4), uraw[3] | mark2(bits >> 6)};
}
__device__ __forceinline__ u32x4 mod_silu_drop8(const u32x4& uraw, long i8, const float (&mv)[8], const ModEpilogue& m,
                                                uint32_t& keepbits) {
  const bf16x8 uv = __builtin_bit_cast(bf16x8, uraw);
  const Keep8 keep = dropout_keep8(i8, m.pdrop, m.sub, m.step, m.seed_lo, m.seed_hi);
  keepbits = keep.bits;
  const float keep_scale = keep.scale;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float mm = mv[j];
    float v = mp_silu_b((float)uv[j] * mm);
    if (m.pdrop > 0.f) v = keep[j] ? v * keep_scale : 0.f;
    o[j] = (bf16)v;
  }
  return __builtin_bit_cast(u32x4, o);
}

// ---- wave-level transposed tile store for the implicit-GEMM kernels.
// acc[NI][NJ] are 32x32 MFMA accumulator blocks whose ROWS are output channels cw0 + 32 i + (8 g + 4 lhi + r) and
// whose COLUMNS are pixels mb0 + 32 j + l31.  Storing them straight from that layout puts 8-byte pieces of 64
// different rows into each store instruction (measured 1.7-2x the algorithmic HBM write bytes).  Instead each wave
// transposes 32 pixels x NI*32 channels at a time through `stage` (wave-private LDS, 32 * (NI*64 + 16) bytes,
// padded rows): the residual is read and the result written 16 B per lane, whole rows per instruction.
// y = alpha * acc + beta * R is formed in fp32 and rounded to bf16 once.  The caller must have passed a workgroup
// barrier after its last read of the LDS bytes that `stage` overlays.
// `put(j, stage)` writes alpha * acc (+ beta * the residual already staged) of the 32-pixel block j into `stage` as
// bf16 [32 px][NI*32 co] rows of EROW bytes: the part that depends on the MFMA shape (see the two wrappers below).
// Epilogue forms that read a second tensor (EPI 1 / 2: the saved pre-activation U) issue its loads one 32-pixel block
// AHEAD of their use -- block 0's before anything else, block j+1's right after the accumulators of block j have gone
// to LDS and freed their registers (EPI 2's optional extra gradient ADD: before put()) -- and the modulation factors of a block
// (one sample per block: HW % 32 == 0) are formed once: the epilogue runs with no other workgroup on the CU, so a
// load -> use -> store chain per 4-8 pixel rows (the first form of this code) was 16 exposed memory latencies per wave.
template <int NI, int NJ, int EPI, class Put>
__device__ __forceinline__ void store_tile_core(Put&& put, char* stage, bf16* __restrict__ Y, const bf16* __restrict__ R,
                                                long mb0, long Npix, int cw0, int Cout, const ModEpilogue& mod,
                                                float* gmred = nullptr) {
  constexpr int EROW = NI * 64 + 16, CPR = NI * 4, RPI = 64 / CPR;  // 16-byte chunks per row, rows per instruction
  constexpr int IT = 32 / RPI;
  const int lane = threadIdx.x & 63;
  const int c16 = lane % CPR, prow = lane / CPR;
  const int co_c = cw0 + c16 * 8;
  u32x4 ub[EPI ? NJ : 1][EPI ? IT : 1], ab[EPI == 2 ? IT : 1];
  const bool has_add = EPI == 2 && mod.ADD != nullptr;
  auto prefetch = [&](int j) {
    const long mb = mb0 + j * 32;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int px = it * RPI + prow;
      const bool ok = mb + px < Npix && co_c < Cout;
      const long e = (mb + px) * Cout + co_c;
      const u32x4 z = {0u, 0u, 0u, 0u};
      ub[j][it] = ok ? *reinterpret_cast<const u32x4*>(mod.U + e) : z;
    }
  };
  if constexpr (EPI != 0) prefetch(0);
  // plain epilogue with a residual: R runs a block ahead in the same way
  u32x4 rb[EPI == 0 ? NJ : 1][EPI == 0 ? IT : 1];
  auto prefetch_r = [&](int j) {
    const long mb = mb0 + j * 32;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int px = it * RPI + prow;
      rb[j][it] = u32x4{0u, 0u, 0u, 0u};
      if (mb + px < Npix && co_c < Cout) rb[j][it] = *reinterpret_cast<const u32x4*>(R + (mb + px) * Cout + co_c);
    }
  };
  if constexpr (EPI == 0) {
    if (R) prefetch_r(0);
  }
  // forward modulation / modulation backward with one sample per 32-pixel block: the factors m = lin*gain + 1 once per block
  const bool silu_out = EPI == 0 && mod.mode == 3;   // Y2 = mp_silu(Y): no modulation operands
  const bool block_mod = (EPI == 1) || (EPI == 0 && mod.Y2 && !silu_out && mod.HW % 32 == 0);
  float gain = 0.f;
  if (EPI != 2 && mod.Y2 && !silu_out) gain = *mod.gain;
  float part[8];  // backward form: per-lane sums over a block's (gmred: the wave's) pixels
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const long mb = mb0 + j * 32;
    float mm[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (EPI != 2 && block_mod && mb < Npix && co_c < Cout) {
      const float* lp = mod.lin + ((int)mb / mod.HW) * mod.lin_stride + co_c;
#pragma unroll
      for (int k = 0; k < 8; ++k) mm[k] = lp[k] * gain + 1.0f;
    }
    if (R) {
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int px = it * RPI + prow;
        u32x4 rv = {0u, 0u, 0u, 0u};
        if constexpr (EPI == 0) rv = rb[j][it];
        else if (mb + px < Npix && co_c < Cout) rv = *reinterpret_cast<const u32x4*>(R + (mb + px) * Cout + co_c);
        *reinterpret_cast<u32x4*>(stage + px * EROW + c16 * 16) = rv;
      }
    }
    if constexpr (EPI == 2) {  // (U runs a block ahead; U and ADD both would not fit beside the accumulators)
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int px = it * RPI + prow;
        ab[it] = u32x4{0u, 0u, 0u, 0u};
        if (has_add && mb + px < Npix && co_c < Cout) ab[it] = *reinterpret_cast<const u32x4*>(mod.ADD + (mb + px) * Cout + co_c);
      }
    }
    put(j, stage);
    if constexpr (EPI == 0) {
      if (R && j + 1 < NJ) {
        __builtin_amdgcn_sched_barrier(0);
        prefetch_r(j + 1);
      }
    }
    if constexpr (EPI != 0) {
      __builtin_amdgcn_sched_barrier(0);  // (hoisted above put(), the next block's loads would not find free registers)
      if (j + 1 < NJ) prefetch(j + 1);
    }
    if (!gmred || j == 0) {
#pragma unroll
      for (int j8 = 0; j8 < 8; ++j8) part[j8] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int px = it * RPI + prow;
      const u32x4 ov = *reinterpret_cast<const u32x4*>(stage + px * EROW + c16 * 16);
      if (mb + px < Npix && co_c < Cout) {
        const long e = (mb + px) * Cout + co_c;
        if constexpr (EPI == 0) {
          if (mod.ldY || mod.Yb) {      // strided / split destination (plain and mode-3 forms only)
            const bool second = mod.Yb && co_c >= mod.split;
            const long ey = second ? (mb + px) * mod.ldYb + (co_c - mod.split) : (mb + px) * (mod.ldY ? mod.ldY : (long)Cout) + co_c;
            *reinterpret_cast<u32x4*>((second ? mod.Yb : Y) + ey) = ov;
            if (silu_out) {
              const bf16x8 yv = __builtin_bit_cast(bf16x8, ov);
              bf16x8 sv;
#pragma unroll
              for (int k = 0; k < 8; ++k) sv[k] = (bf16)mp_silu_b((float)yv[k]);
              *reinterpret_cast<bf16x8*>(mod.Y2 + ey) = sv;
            }
            continue;
          }
        }
        if (Y && !(EPI == 0 && mod.Y2 && mod.u_marks)) *reinterpret_cast<u32x4*>(Y + e) = ov;
        if (mod.Y2) {
          if constexpr (EPI == 1) {
            *reinterpret_cast<u32x4*>(mod.Y2 + e) = mod_silu_drop_bwd8(ov, ub[j][it], e >> 3, mm, mod, part);
          } else if constexpr (EPI == 2) {
            *reinterpret_cast<u32x4*>(mod.Y2 + e) = silu_bwd8(ov, ub[j][it], ab[it], has_add, mod.add_scale);
          } else {
            float mv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) mv[k] = mm[k];
            if (!block_mod) {   // a 32-pixel block may straddle samples: the factors of this pixel's own sample
              const float* lp = mod.lin + ((int)(mb + px) / mod.HW) * mod.lin_stride + co_c;
#pragma unroll
              for (int k = 0; k < 8; ++k) mv[k] = lp[k] * gain + 1.0f;
            }
            uint32_t kb;
            *reinterpret_cast<u32x4*>(mod.Y2 + e) = mod_silu_drop8(ov, e >> 3, mv, mod, kb);
            if (Y && mod.u_marks) *reinterpret_cast<u32x4*>(Y + e) = mark_dropped8(ov, kb);
          }
        }
      }
    }
    if (EPI == 1 && (!gmred || j == NJ - 1)) {  // lanes with equal c16 hold partial sums of the same 8 channels: fold the RPI pixel rows
#pragma unroll
      for (int j8 = 0; j8 < 8; ++j8) {
#pragma unroll
        for (int off = CPR; off < 64; off <<= 1) part[j8] += __shfl_xor(part[j8], off, 64);
      }
      if (gmred) {
        // all NJ blocks of this wave lie in one sample: their sums go to the wave's row of the workgroup's LDS table
        // (plain stores; the kernel adds the rows of a tile up and issues ONE global atomic per sample and channel)
        if (prow == 0) {
          *reinterpret_cast<f32x4*>(gmred + c16 * 8) = f32x4{part[0], part[1], part[2], part[3]};
          *reinterpret_cast<f32x4*>(gmred + c16 * 8 + 4) = f32x4{part[4], part[5], part[6], part[7]};
        }
      } else if (prow == 0 && mb < Npix && co_c < Cout) {
        float* gp = mod.gm + ((int)mb / mod.HW) * (mod.gm_stride ? mod.gm_stride : (long)Cout) + co_c;
#pragma unroll
        for (int j8 = 0; j8 < 8; ++j8) atomicAdd(gp + j8, part[j8]);
      }
    }
    if constexpr (EPI != 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// 4 accumulator values (consecutive channels of ONE pixel) -> alpha * v (+ beta * staged residual) -> bf16x4 at sp
__device__ __forceinline__ void stage4(char* sp, float v0, float v1, float v2, float v3, float alpha, float beta, bool hasR) {
  float v[4] = {alpha * v0, alpha * v1, alpha * v2, alpha * v3};
  if (hasR) {
    bf16x4 rv = *reinterpret_cast<const bf16x4*>(sp);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += beta * (float)rv[r];
  }
  bf16x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
  *reinterpret_cast<bf16x4*>(sp) = o;
}

// v_mfma_f32_32x32x16 accumulators: acc[i][j] rows = channels 32 i + 8 g + 4 lhi + r, columns = pixels 32 j + l31
template <int NI, int NJ, int EPI = 0>
__device__ __forceinline__ void store_tile_transposed(const f32x16 (&acc)[NI][NJ], char* stage, bf16* __restrict__ Y,
                                                      const bf16* __restrict__ R, float alpha, float beta, long mb0,
                                                      long Npix, int cw0, int Cout, const ModEpilogue& mod = ModEpilogue{}) {
  constexpr int EROW = NI * 64 + 16;
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  store_tile_core<NI, NJ, EPI>(
      [&](int j, char* st) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            stage4(st + l31 * EROW + (i * 32 + 8 * g + 4 * lhi) * 2, acc[i][j][4 * g], acc[i][j][4 * g + 1],
                   acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], alpha, beta, R != nullptr);
      },
      stage, Y, R, mb0, Npix, cw0, Cout, mod);
}

// v_mfma_f32_16x16x32 accumulators: acc[i][j] (i over 2 NI channel blocks of 16, j over 2 NJ pixel blocks of 16): rows =
// channels 16 i + 4 (lane >> 4) + r, columns = pixels 16 j + (lane & 15).  Same staging, same coalesced second phase.
template <int NI, int NJ, int EPI = 0>
__device__ __forceinline__ void store_tile_transposed16(const f32x4 (&acc)[2 * NI][2 * NJ], char* stage,
                                                        bf16* __restrict__ Y, const bf16* __restrict__ R, float alpha,
                                                        float beta, long mb0, long Npix, int cw0, int Cout,
                                                        const ModEpilogue& mod = ModEpilogue{}, float* gmred = nullptr) {
  constexpr int EROW = NI * 64 + 16;
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  store_tile_core<NI, NJ, EPI>(
      [&](int j, char* st) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 2 * NI; ++i)
            stage4(st + (h * 16 + l15) * EROW + (i * 16 + 4 * lq) * 2, acc[i][2 * j + h][0], acc[i][2 * j + h][1],
                   acc[i][2 * j + h][2], acc[i][2 * j + h][3], alpha, beta, R != nullptr);
      },
      stage, Y, R, mb0, Npix, cw0, Cout, mod, gmred);
}

// ---- fp32-out epilogues of the split-bf16 evaluation path (EPI 4; round 4, csrc/eval_f32.hip "split"): the accumulators
// hold an fp32-accurate sum (three bf16 MFMA passes over hi/lo operand pairs), so the result leaves as FLOATS:
//   Y = alpha * acc + beta * R          (R fp32, contiguous [pixels][Cout]; mp_add of networks.py:87-88), or
//   Y = mp_silu(acc * (lin[b,:] * gain + 1))   when mod.lin is set (the block's modulation, networks.py:253-260; eval: no
//                                               dropout) -- the exact functions, as k_conv_f32's epilogue.
// Straight from the accumulator layout: a lane holds 4 consecutive channels of one pixel -> one 16-byte load / store per
// block (64-byte row segments per instruction: an evaluation-path epilogue, not tuned).
__device__ __forceinline__ void f32_out4v(float* __restrict__ Y, const float* __restrict__ R, float beta, long px, long Npix,
                                          int co, int Cout, const ModEpilogue& mod, float gain, f32x4 v);
__device__ __forceinline__ void f32_out4(float* __restrict__ Y, const float* __restrict__ R, float alpha, float beta,
                                         long px, long Npix, int co, int Cout, const ModEpilogue& mod, float gain,
                                         float v0, float v1, float v2, float v3) {
  f32_out4v(Y, R, beta, px, Npix, co, Cout, mod, gain, f32x4{alpha * v0, alpha * v1, alpha * v2, alpha * v3});
}
// (v = alpha * acc already)
__device__ __forceinline__ void f32_out4v(float* __restrict__ Y, const float* __restrict__ R, float beta, long px, long Npix,
                                          int co, int Cout, const ModEpilogue& mod, float gain, f32x4 v) {
  if (px >= Npix || co >= Cout) return;
  const long e = px * Cout + co;
  if (R) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(R + e);
    v += beta * r;
  }
  if (mod.lin) {
    const f32x4 l = *reinterpret_cast<const f32x4*>(mod.lin + (px / mod.HW) * mod.lin_stride + co);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = mp_silu_f(v[k] * (l[k] * gain + 1.0f));
  }
  if (Y) *reinterpret_cast<f32x4*>(Y + e) = v;
  if (mod.Y2 || mod.Yb) {
    // the same values as (hi, lo) bf16 pairs: the next conv's operand format.  Rows of ldY elements (0 = 2 Cout), the lo
    // halves ldYb elements behind the hi halves (0 = Cout): plain [hi(Cout) | lo(Cout)] rows, or -- round 6, the decoder's
    // torch.cat((input, skip * gate)) of the split evaluation without a copy -- the left column blocks of the next block's
    // concatenated operand [hi(Ci + Cs) | lo(Ci + Cs)].  mod.Yb: mp_silu of the result as pairs, same addressing (the
    // operand of the consumer's first 3x3 conv, networks.py:316)
    const long ld = mod.ldY ? mod.ldY : 2L * Cout, lo_off = mod.ldYb ? mod.ldYb : (long)Cout;
    bf16x4 hi, lo;
    if (mod.Y2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = (bf16)v[k];
        lo[k] = (bf16)(v[k] - (float)hi[k]);
      }
      *reinterpret_cast<bf16x4*>(mod.Y2 + px * ld + co) = hi;
      *reinterpret_cast<bf16x4*>(mod.Y2 + px * ld + lo_off + co) = lo;
    }
    if (mod.Yb) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float sv = mp_silu_f(v[k]);
        hi[k] = (bf16)sv;
        lo[k] = (bf16)(sv - (float)hi[k]);
      }
      *reinterpret_cast<bf16x4*>(mod.Yb + px * ld + co) = hi;
      *reinterpret_cast<bf16x4*>(mod.Yb + px * ld + lo_off + co) = lo;
    }
  }
}
// staged form for the v_mfma_f32_32x32x16 layout (k_conv_igemm's 1x1 / small-map split convs; see store_tile_f32_16_staged):
// a 32-pixel block of the wave's NI * 32 channels through LDS as floats, out in rows.  stage: 32 * (NI * 128 + 16) bytes per wave.
template <int NI, int NJ>
__device__ __forceinline__ void store_tile_f32_staged(const f32x16 (&acc)[NI][NJ], char* stage, float* __restrict__ Y,
                                                      const float* __restrict__ R, float alpha, float beta, long mb0, long Npix,
                                                      int cw0, int Cout, const ModEpilogue& mod) {
  constexpr int SROW = NI * 128 + 16, LPR = NI * 8, RPP = 64 / LPR, NPASS = 32 / RPP;
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  const int c4 = lane % LPR, prow = lane / LPR;
  const float gain = mod.lin ? *mod.gain : 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(stage + l31 * SROW + (32 * i + 8 * g + 4 * lhi) * 4) =
            f32x4{alpha * acc[i][j][4 * g], alpha * acc[i][j][4 * g + 1], alpha * acc[i][j][4 * g + 2], alpha * acc[i][j][4 * g + 3]};
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      const int px = it * RPP + prow;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SROW + c4 * 16);
      f32_out4v(Y, R, beta, mb0 + 32 * j + px, Npix, cw0 + c4 * 4, Cout, mod, gain, v);
    }
  }
}
// v_mfma_f32_32x32x16 accumulators (k_conv_igemm): rows = channels 32 i + 8 g + 4 lhi + r, columns = pixels 32 j + l31
template <int NI, int NJ>
__device__ __forceinline__ void store_tile_f32(const f32x16 (&acc)[NI][NJ], float* __restrict__ Y, const float* __restrict__ R,
                                               float alpha, float beta, long mb0, long Npix, int cw0, int Cout,
                                               const ModEpilogue& mod) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  const float gain = mod.lin ? *mod.gain : 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        f32_out4(Y, R, alpha, beta, mb0 + 32 * j + l31, Npix, cw0 + 32 * i + 8 * g + 4 * lhi, Cout, mod, gain,
                 acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
}
// The same epilogue STAGED through wave-private LDS (round 6): straight from the accumulator layout a store instruction
// covers 16 pixels x 64 bytes (fp32) or x 32 bytes (a pairs half) -- and the `dest` form of the split evaluation's copy-free
// concat issues four such 8-byte-per-lane stores per 16-byte fp32 store: +252 us per 32x32 launch at batch 512 where the
// extra bytes alone are +107 (tools/microbench_split_epi.py).  Here a 16-pixel block goes to LDS as floats
// [16 px][NI * 32 ch] (row pitch + 16 bytes) and leaves row-contiguously: a lane owns four consecutive channels of a pixel,
// NI * 8 lanes cover a row -- 512-byte (256-byte: pairs) segments per row and instruction, R read the same way.
// `stage`: 16 * (NI * 128 + 16) bytes per wave, behind a workgroup barrier after the main loop's last LDS read.
template <int NI, int NJ>
__device__ __forceinline__ void store_tile_f32_16_staged(const f32x4 (&acc)[2 * NI][2 * NJ], char* stage,
                                                         float* __restrict__ Y, const float* __restrict__ R, float alpha,
                                                         float beta, long mb0, long Npix, int cw0, int Cout,
                                                         const ModEpilogue& mod) {
  constexpr int SROW = NI * 128 + 16, LPR = NI * 8, RPP = 64 / LPR, NPASS = 16 / RPP;
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  const int c4 = lane % LPR, prow = lane / LPR;
  const float gain = mod.lin ? *mod.gain : 0.f;
#pragma unroll
  for (int j = 0; j < 2 * NJ; ++j) {
#pragma unroll
    for (int i = 0; i < 2 * NI; ++i)
      *reinterpret_cast<f32x4*>(stage + l15 * SROW + (16 * i + 4 * lq) * 4) =
          f32x4{alpha * acc[i][j][0], alpha * acc[i][j][1], alpha * acc[i][j][2], alpha * acc[i][j][3]};
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      const int px = it * RPP + prow;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SROW + c4 * 16);
      f32_out4v(Y, R, beta, mb0 + 16 * j + px, Npix, cw0 + c4 * 4, Cout, mod, gain, v);
    }
  }
}
// v_mfma_f32_16x16x32 accumulators (k_conv3x3_v6): rows = channels 16 i + 4 (lane >> 4) + r, columns = pixels 16 j + (lane & 15)
template <int NI, int NJ>
__device__ __forceinline__ void store_tile_f32_16(const f32x4 (&acc)[2 * NI][2 * NJ], float* __restrict__ Y,
                                                  const float* __restrict__ R, float alpha, float beta, long mb0, long Npix,
                                                  int cw0, int Cout, const ModEpilogue& mod) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  const float gain = mod.lin ? *mod.gain : 0.f;
#pragma unroll
  for (int j = 0; j < 2 * NJ; ++j)
#pragma unroll
    for (int i = 0; i < 2 * NI; ++i)
      f32_out4(Y, R, alpha, beta, mb0 + 16 * j + l15, Npix, cw0 + 16 * i + 4 * lq, Cout, mod, gain, acc[i][j][0], acc[i][j][1],
               acc[i][j][2], acc[i][j][3]);
}

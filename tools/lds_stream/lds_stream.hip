// Diagnostic: how fast can a kernel STREAM through LDS?  Every LDS-staged streaming kernel of this library (1x1 convs,
// their weight gradients) plateaus near 3.6 TB/s of HBM traffic while the elementwise kernels reach 5+.  Three copy kernels
// over the same 1-GiB buffer (read 1 GiB, write 1 GiB), rotating over 3 buffer pairs so that the Infinity Cache cannot
// serve the reads:
//   direct   global_load_dwordx4 -> global_store_dwordx4, grid-stride (what an elementwise kernel does)
//   dma<R>   tiles of TILE bytes per workgroup through an R-deep LDS ring filled by LDS-DMA (global_load_lds, counted vmcnt
//            waits + one barrier per tile), read back with ds_read_b128, stored (what the LDS-DMA conv kernels do)
//   reg<R>   the same ring filled by global_load_dwordx4 -> ds_write_b128 (register-staged)
//   hipcc --offload-arch=gfx950 -O3 -o lds_stream lds_stream.hip && ./lds_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256) void k_direct(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = i + u * 256 < n16 ? src[i + u * 256] : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n16) dst[i + u * 256] = v[u];
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// THREADS threads; a tile is THREADS * 16 * PER bytes (PER 16-byte pieces per thread); ring of R tiles
template <int THREADS, int PER, int R, bool DMA>
__global__ __launch_bounds__(THREADS) void k_ring(const char* __restrict__ src, char* __restrict__ dst, long ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = THREADS * 16 * PER;
  const int tid = threadIdx.x;
  const long first = blockIdx.x, step = gridDim.x;
  const long mine = first < ntiles ? (ntiles - first + step - 1) / step : 0;
  u32x4 hold[DMA ? 1 : R][PER];
  auto issue = [&](long k, int slot) {   // tile k of this workgroup into ring slot `slot`
    const char* s = src + (first + k * step) * TILE;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      if constexpr (DMA)
        __builtin_amdgcn_global_load_lds((gptr_t)(s + (p * THREADS + tid) * 16), (lptr_t)(smem + slot * TILE + p * THREADS * 16), 16, 0, 0);
    }
  };
  if constexpr (DMA) {
#pragma unroll
    for (int r = 0; r < R - 1; ++r)
      if (r < mine) issue(r, r);
    for (long k = 0; k < mine; ++k) {
      const int slot = (int)(k % R);
      if (k + R - 1 < mine) {
        issue(k + R - 1, (int)((k + R - 1) % R));
        wait_vmcnt<(R - 1) * PER>();
      } else {
        wait_vmcnt<0>();
      }
      __syncthreads();
      char* d = dst + (first + k * step) * TILE;
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        // read a DIFFERENT thread's piece (as a GEMM's fragment reads would): rotate by one wave
        const int t2 = (tid + 64) % THREADS;
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + slot * TILE + (p * THREADS + t2) * 16);
        *reinterpret_cast<u32x4*>(d + (p * THREADS + t2) * 16) = v;
      }
      __syncthreads();   // the slot is free for tile k + R
    }
  } else {
    // register-staged: loads of tile k + R - 1 in flight while tile k goes registers -> LDS -> registers -> global
#pragma unroll
    for (int r = 0; r < R - 1; ++r)
      if (r < mine) {
        const char* s = src + (first + r * step) * TILE;
#pragma unroll
        for (int p = 0; p < PER; ++p) hold[r][p] = *reinterpret_cast<const u32x4*>(s + (p * THREADS + tid) * 16);
      }
    for (long k0 = 0; k0 < mine; k0 += R) {
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const long k = k0 + rr;
        if (k >= mine) break;
        constexpr int dummy = 0;
        (void)dummy;
        const int nslot = (rr + R - 1) % R;
        if (k + R - 1 < mine) {
          const char* s = src + (first + (k + R - 1) * step) * TILE;
#pragma unroll
          for (int p = 0; p < PER; ++p) hold[nslot][p] = *reinterpret_cast<const u32x4*>(s + (p * THREADS + tid) * 16);
        }
#pragma unroll
        for (int p = 0; p < PER; ++p) *reinterpret_cast<u32x4*>(smem + (p * THREADS + tid) * 16) = hold[rr][p];
        __syncthreads();
        char* d = dst + (first + k * step) * TILE;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          const int t2 = (tid + 64) % THREADS;
          const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (p * THREADS + t2) * 16);
          *reinterpret_cast<u32x4*>(d + (p * THREADS + t2) * 16) = v;
        }
        __syncthreads();
      }
    }
  }
}

// The access shape of the 1x1 kernels: the source is [rows][ROWB bytes] (ROWB = 2 * channels); a stage is 64 rows; one
// LDS-DMA instruction fetches SEG bytes of each of 1024 / SEG rows (SEG = 64: the [64 rows][32 ch] sub-images of
// k_wgrad1x1_group / the conv kernels' slabs; 128, 256: wider segments) -- every byte is still loaded exactly once.
template <int ROWB, int SEG, int R>
__global__ __launch_bounds__(512) void k_strided(const char* __restrict__ src, char* __restrict__ dst, long nstages) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 64 * ROWB;                 // bytes per stage
  constexpr int NINSTR = STAGE / 1024;             // DMA instructions per stage (all waves together)
  constexpr int PER = NINSTR / 8;                  // per wave
  constexpr int RPI = 1024 / SEG;                  // rows per instruction
  constexpr int SEGS = ROWB / SEG;                 // segments per row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long first = blockIdx.x, step = gridDim.x;
  const long mine = first < nstages ? (nstages - first + step - 1) / step : 0;
  auto issue = [&](long k, int slot) {
    const char* s = src + (first + k * step) * STAGE;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int j = wave + 8 * i;                  // instruction j: segment column j % SEGS, row group j / SEGS
      const int seg = j % SEGS, rg = j / SEGS;
      const int row = rg * RPI + lane / (SEG / 16), piece = lane % (SEG / 16);
      __builtin_amdgcn_global_load_lds((gptr_t)(s + (long)row * ROWB + seg * SEG + piece * 16), (lptr_t)(smem + slot * STAGE + j * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int r = 0; r < R - 1; ++r)
    if (r < mine) issue(r, r);
  for (long k = 0; k < mine; ++k) {
    const int slot = (int)(k % R);
    if (k + R - 1 < mine) {
      issue(k + R - 1, (int)((k + R - 1) % R));
      wait_vmcnt<(R - 1) * PER>();
    } else {
      wait_vmcnt<0>();
    }
    __syncthreads();
    char* d = dst + (first + k * step) * STAGE;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int t2 = (tid + 64) % 512;
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + slot * STAGE + (p * 512 + t2) * 16);
      *reinterpret_cast<u32x4*>(d + (p * 512 + t2) * 16) = v;
    }
    __syncthreads();
  }
}

// The PHASE shape of a 1x1 conv workgroup: read RD stages of 16 KB through the ring (the K loop), THEN write WR KB (the
// epilogue) -- one tile per workgroup (grid = tiles), or persistent workgroups that keep the next tile's loads in flight
// under the stores (PERSIST).  Reads 2/3, writes 1/3 of the bytes, as the 512->256 layers.
template <int RD, int WRK, int R, bool PERSIST>
__global__ __launch_bounds__(512) void k_phased(const char* __restrict__ src, char* __restrict__ dst, long ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 16 * 1024, PER = 2;
  const int tid = threadIdx.x;
  const long first = blockIdx.x, step = PERSIST ? gridDim.x : ntiles;
  const long mine = PERSIST ? (first < ntiles ? (ntiles - first + step - 1) / step : 0) : 1;
  const long nst = mine * RD;                       // stages of this workgroup, tile after tile
  auto issue = [&](long q) {
    const long tile = first + (q / RD) * step;
    const char* s = src + (tile * RD + q % RD) * STAGE;
#pragma unroll
    for (int p = 0; p < PER; ++p)
      __builtin_amdgcn_global_load_lds((gptr_t)(s + (p * 512 + tid) * 16), (lptr_t)(smem + (q % R) * STAGE + p * 8192), 16, 0, 0);
  };
  u32x4 keep = {0, 0, 0, 0};
  for (int r = 0; r < R - 1; ++r)
    if (r < nst) issue(r);
  for (long q = 0; q < nst; ++q) {
    if (q + R - 1 < nst) {
      issue(q + R - 1);
      wait_vmcnt<(R - 1) * PER>();
    } else {
      wait_vmcnt<0>();
    }
    __syncthreads();
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (q % R) * STAGE + ((tid + 64) % 512) * 16);
    keep[0] ^= v[0]; keep[1] ^= v[1]; keep[2] ^= v[2]; keep[3] ^= v[3];
    __syncthreads();
    if (q % RD == RD - 1) {                         // the tile's K loop is over: its outputs
      const long tile = first + (q / RD) * step;
      char* d = dst + tile * (WRK * 1024);
#pragma unroll
      for (int p = 0; p < WRK * 1024 / (512 * 16); ++p) *reinterpret_cast<u32x4*>(d + (p * 512 + tid) * 16) = keep;
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class F>
double timeit(F&& f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f(i);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const long bytes = 1L << 30;
  char *src[3], *dst[3];
  for (int i = 0; i < 3; ++i) {
    CK(hipMalloc(&src[i], bytes)); CK(hipMalloc(&dst[i], bytes));
    CK(hipMemset(src[i], i + 1, bytes)); CK(hipMemset(dst[i], 0, bytes));
  }
  const int reps = 12;
  auto report = [&](const char* name, double ms) { printf("%-44s %8.3f ms  %6.2f TB/s (read + write)\n", name, ms, 2.0 * bytes / ms / 1e9); fflush(stdout); };
  for (int wg : {256 * 8, 256 * 16, 256 * 32})
    { char nm[64]; snprintf(nm, 64, "direct, %d workgroups", wg);
      report(nm, timeit([&](int i) { hipLaunchKernelGGL(k_direct, dim3(wg), dim3(256), 0, 0, (const u32x4*)src[i % 3], (u32x4*)dst[i % 3], bytes / 16); }, reps)); }
#define RING(THREADS, PER, R, DMA, WGPC)                                                                              \
  {                                                                                                                   \
    constexpr int TILE = THREADS * 16 * PER;                                                                          \
    const size_t lds = DMA ? (size_t)R * TILE : (size_t)TILE;                                                         \
    auto kern = k_ring<THREADS, PER, R, DMA>;                                                                         \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
    char nm[96];                                                                                                      \
    snprintf(nm, 96, "%s, %d thr, tile %d KB, ring %d, %d WG/CU", DMA ? "lds-dma" : "reg-staged", THREADS, TILE / 1024, R, WGPC); \
    report(nm, timeit([&](int i) { hipLaunchKernelGGL(kern, dim3(256 * WGPC), dim3(THREADS), lds, 0, src[i % 3], dst[i % 3], bytes / TILE); }, reps)); \
  }
  RING(256, 2, 3, true, 2) RING(256, 2, 3, true, 4) RING(256, 2, 6, true, 2) RING(256, 2, 6, true, 4)
  RING(256, 4, 3, true, 2) RING(256, 4, 4, true, 2) RING(512, 2, 3, true, 1) RING(512, 2, 6, true, 1) RING(512, 4, 4, true, 1)
  RING(256, 2, 8, true, 4) RING(256, 1, 8, true, 8)
  RING(256, 2, 3, false, 2) RING(256, 2, 3, false, 4) RING(256, 4, 3, false, 2) RING(256, 4, 4, false, 4) RING(512, 2, 3, false, 2)
  RING(256, 2, 4, false, 8)
#define STR(ROWB, SEG, R)                                                                                              \
  {                                                                                                                    \
    constexpr int STAGE = 64 * ROWB;                                                                                   \
    auto kern = k_strided<ROWB, SEG, R>;                                                                               \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                \
    char nm[96];                                                                                                       \
    snprintf(nm, 96, "lds-dma rows of %d B, %d-B segments, ring %d", ROWB, SEG, R);                                    \
    report(nm, timeit([&](int i) { hipLaunchKernelGGL(kern, dim3(256), dim3(512), (size_t)R * STAGE, 0, src[i % 3], dst[i % 3], bytes / STAGE); }, reps)); \
  }
  STR(512, 64, 3) STR(512, 128, 3) STR(512, 256, 3) STR(512, 512, 3)
  STR(768, 64, 3) STR(768, 128, 3) STR(768, 256, 3)
  STR(1024, 64, 2) STR(1024, 128, 2) STR(1024, 256, 2) STR(1024, 1024, 2)
#define PHS(RD, WRK, R, PERSIST, GRIDMUL)                                                                               \
  {                                                                                                                    \
    const long ntiles = bytes / (RD * 16 * 1024);                                                                      \
    auto kern = k_phased<RD, WRK, R, PERSIST>;                                                                         \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                \
    char nm[128];                                                                                                      \
    snprintf(nm, 128, "phased: read %d KB then write %d KB per tile, ring %d, %s", RD * 16, WRK, R,                    \
             PERSIST ? "persistent (loads run under the stores)" : "one tile per workgroup");                          \
    const double ms = timeit([&](int i) { hipLaunchKernelGGL(kern, dim3(PERSIST ? 256 * GRIDMUL : (unsigned)ntiles), dim3(512), (size_t)R * 16 * 1024, 0, src[i % 3], dst[i % 3], ntiles); }, reps); \
    printf("%-100s %8.3f ms  %6.2f TB/s (read + write)\n", nm, ms, (bytes * (1.0 + (double)WRK / (RD * 16))) / ms / 1e9); fflush(stdout); \
  }
  PHS(16, 128, 3, false, 1) PHS(16, 128, 4, false, 1) PHS(16, 128, 3, true, 1) PHS(16, 128, 4, true, 1) PHS(16, 128, 4, true, 2)
  PHS(8, 128, 3, false, 1) PHS(8, 128, 4, true, 1)
  return 0;
}

// Data formats on either side of the hot path: the image datasets stay RESIDENT in HBM
// as uint8 (CIFAR-10 train = 150 MB of 288 GB) and a batch is produced by one gather kernel -- no host dataloader,
// no pinned staging, no H2D copy in the training loop -- and sampled images leave the GPU already as bytes.
//
// Byte/float conversions follow the reference's arithmetic operation by operation (separately rounded fp32
// multiply / add / divide, no FMA contraction) so that the uint8 outputs are bit-exact:
//   * load  : torchvision v2.ToDtype(float32, scale=True) -> x/255, RandomHorizontalFlip, Normalize(0.5, 0.5) ->
//             (x/255 - 0.5)/0.5                      (datamodules/cifar10datamodule.py:18-32, mnistdatamodule.py:18-30)
//   * store : (x*127.5 + 128).clip(0, 255).to(uint8)                    (cifar10datamodule.py:34-35)
//   * PNG   : clamp(pred*std*2 + mean, 0, 1).permute(0,2,3,1)*255 -> uint8          (callbacks.py:126-156)
#include "common.h"

// bit-exact parity with the reference's op-by-op fp32 arithmetic: no mul+add fusion anywhere in this file
#pragma clang fp contract(off)

namespace {

// dataset u8 [N][C][H][W] (planar, the on-disk order of CIFAR-10 / MNIST); out fp32 NCHW [B][C][H][W];
// sample b reads image index[b]; flip decided per sample by Philox(seed, (epoch, b)) bit 0 when flip != 0.
__global__ __launch_bounds__(256) void k_u8_gather_normalize(const unsigned char* __restrict__ data,
                                                               const long* __restrict__ index, float* __restrict__ out,
                                                               int C, int H, int W, long n_images, float mean, float stdv,
                                                               int flip, unsigned long long seed, unsigned epoch) {
  const int b = blockIdx.y;
  const long img = index[b];
  if (img < 0 || img >= n_images) return;  // host validates; never read out of bounds
  bool do_flip = false;
  if (flip) {
    const Philox4 r = philox4x32_10((uint32_t)b, 0u, 0x0da7u, epoch, (uint32_t)seed, (uint32_t)(seed >> 32));
    do_flip = (r.x & 1u) != 0u;
  }
  const int chw = C * H * W;
  const unsigned char* src = data + img * chw;
  float* dst = out + (long)b * chw;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < chw; e += gridDim.x * blockDim.x) {
    const int w = e % W;
    const int se = do_flip ? e - w + (W - 1 - w) : e;
    const float x = (float)src[se] / 255.0f;  // IEEE division (hipcc default: correctly rounded fp32 divide)
    dst[e] = (x - mean) / stdv;
  }
}

// x fp32 NCHW -> u8 NCHW: (x*scale + offset).clip(0,255) truncated
__global__ __launch_bounds__(256) void k_denormalize_u8(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                          long n, float scale, float offset) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float m = x[e] * scale;  // separately rounded multiply and add (contraction is off in this file)
    float v = m + offset;
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    out[e] = (unsigned char)(int)v;
  }
}

// pred fp32 NCHW -> u8 NHWC: clamp(pred*std[c]*2 + mean[c], 0, 1)*255 truncated
__global__ __launch_bounds__(256) void k_prediction_to_u8_nhwc(const float* __restrict__ pred,
                                                                 unsigned char* __restrict__ out, int C, int HW, long n,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ stdv) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const long p = e / C;  // b*HW + pixel
    const long b = p / HW, px = p - b * HW;
    const float v0 = pred[(b * C + c) * HW + px];
    const float m = v0 * stdv[c] * 2.0f;
    float v = m + mean[c];
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    out[e] = (unsigned char)(int)(v * 255.0f);
  }
}

}  // namespace

extern "C" int edm_u8_gather_normalize(const void* data, const long* index, float* out, int B, int C, int H, int W,
                                       long n_images, float mean, float stdv, int flip, unsigned long long seed,
                                       unsigned epoch, hipStream_t st) {
  EDM_REQUIRE(data && index && out, "u8_gather_normalize: null pointer");
  EDM_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && n_images > 0 && stdv != 0.0f,
              "u8_gather_normalize: bad args B=%d C=%d H=%d W=%d", B, C, H, W);
  const int chw = C * H * W;
  const int gx = (chw + 255) / 256 < 64 ? (chw + 255) / 256 : 64;
  hipLaunchKernelGGL(k_u8_gather_normalize, dim3(gx, B), dim3(256), 0, st, (const unsigned char*)data, index, out, C, H,
                     W, n_images, mean, stdv, flip, seed, epoch);
  EDM_CHECK_LAUNCH("u8_gather_normalize");
  return EDM_OK;
}

extern "C" int edm_denormalize_u8(const float* x, void* out, long n, float scale, float offset, hipStream_t st) {
  EDM_REQUIRE(x && out && n > 0, "denormalize_u8: bad args");
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_denormalize_u8, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, x,
                     (unsigned char*)out, n, scale, offset);
  EDM_CHECK_LAUNCH("denormalize_u8");
  return EDM_OK;
}

extern "C" int edm_prediction_to_u8_nhwc(const float* pred, void* out, int B, int C, int H, int W, const float* mean,
                                         const float* stdv, hipStream_t st) {
  EDM_REQUIRE(pred && out && mean && stdv && B > 0 && C > 0 && H > 0 && W > 0, "prediction_to_u8_nhwc: bad args");
  const long n = (long)B * C * H * W;
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_prediction_to_u8_nhwc, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, pred,
                     (unsigned char*)out, C, H * W, n, mean, stdv);
  EDM_CHECK_LAUNCH("prediction_to_u8_nhwc");
  return EDM_OK;
}

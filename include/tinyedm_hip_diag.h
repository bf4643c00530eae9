/* tinyedm_hip_diag.h -- diagnostic entry points of libtinyedm_hip.so used by tools/ only (in-kernel clocks, stamps,
 * timing-only ablations of the implicit-GEMM kernels).  Not part of the product ABI (include/tinyedm_hip.h): their
 * outputs are wrong or meaningless by construction and nothing in tinyedm_amd/ calls them. */
#ifndef TINYEDM_HIP_DIAG_H
#define TINYEDM_HIP_DIAG_H
#include "tinyedm_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* diagnostic : s_memtime stamps per loop segment of the v2 kernel, summed over waves into dbg[0..5] */
int edm_conv_igemm_v2_stamp(const void* X, const void* Wp, void* Y, int B, int H, int W, int Cin, int Cout,
                            unsigned long long* dbg, edm_stream_t stream);
/* diagnostic : timing-only ablations of the v2 kernel (bit0 no MFMA, bit1 no DMA, bit2 no reads, bit3 no barrier) */
int edm_conv_igemm_v2_ablate(const void* X, const void* Wp, void* Y, int B, int H, int W, int Cin, int Cout, int mode,
                             edm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TINYEDM_HIP_DIAG_H */

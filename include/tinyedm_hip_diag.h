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

/* measurement : the next edm_wgrad3_group call of this thread records the two HIP events (hipEvent_t handles) around its
 * k_wgrad3 launch ONLY -- not around the launch-table upload before it and k_wgrad3_finish behind it.  One-shot. */
int edm_wgrad3_probe(void* ev_start, void* ev_end);

/* diagnostics : K shares per tile that edm_wgrad3_group's plan gives each of the n layers of `items` (host memory, as for
 * edm_wgrad3_group; 1 = the layer's K range is not split) -- lets the parity test assert that the split form really ran. */
int edm_wgrad3_plan_ksplit(const edm_wgrad3_item* items, int n, int* ksplit_out);

/* measurement : launches of k_conv3x3_v6's persistent form (EDM_V6_PERSIST=1) by this process so far -- lets the parity test
 * assert that the form it compares really ran. */
long edm_v6_persistent_launches(void);

#ifdef __cplusplus
}
#endif
#endif /* TINYEDM_HIP_DIAG_H */

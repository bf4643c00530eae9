/* tinyedm_hip.h -- C ABI of libtinyedm_hip.so, the MI355X (gfx950) kernels behind the tinyedm hot path.
 *
 * The reference (YichengDWu/tinyedm) has no native code: every entry point below replaces an ATen op
 * call site of its Python hot path (cited as file:line relative to /root/reference/src/tinyedm) or the
 * autograd backward of one.  Conventions:
 *   - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller (the one exception,
 *     edm_wgrad3_group's descriptor array, is HOST memory read during the call); compute entry points do no
 *     allocation and no host sync: work is enqueued on `stream` and is graph-capturable from the first call.
 *     The only allocating call is edm_init(device), made once per device before anything else.
 *   - activations: NHWC bf16, i.e. a row-major [pixels = B*H*W][channels] matrix of 16-bit brain floats.
 *   - per-(sample,channel) / parameter-side quantities: fp32.
 *   - return 0 on success, <0 on error (edm_last_error() holds the message, thread-local).
 */
#ifndef TINYEDM_HIP_H
#define TINYEDM_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* edm_stream_t; /* hipStream_t */

int edm_version(void);
const char* edm_last_error(void);
/* per-device constants (a 4 KiB page of zeros the LDS-DMA kernels point padding rows at): allocate once per device,
 * outside any stream capture; idempotent and thread-safe.  Every kernel that needs the page fails with -1 until then. */
int edm_init(int device);
/* hipGraph capture of these entry points: on ROCm 7.2 the runtime's default AQL-packet-capture path runs the first
 * replay that follows a hipStreamSynchronize / hipDeviceSynchronize with clobbered kernel arguments.  The library's
 * load-time constructor therefore sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 when the variable is unset (the runtime reads
 * it at its first HIP call: load this library before that).  Returns 1 when the environment holds the safe value. */
int edm_graph_replay_safe(void);
/* Per-step scalars of a hipGraph-captured training step.  By-value arguments are frozen into a graph at capture time;
 * entry points with a `dyn` parameter read these fields from DEVICE memory instead when dyn != NULL, so the host can
 * rewrite the 48-byte record (one small async copy) before every replay: step/seed feed the Philox streams of dropout
 * and the Diffuser (reference: torch RNG state advancing between steps), lr comes from the LambdaLR schedule
 * (edm.py:305-320), ema_beta from ema.py:137-140, bc1 = 1-b1^t and bc2sqrt = sqrt(1-b2^t) are Adam's bias corrections. */
typedef struct {
  unsigned step, seed_lo, seed_hi, reserved;
  float lr, ema_beta, grad_scale, bc1, bc2sqrt, pad[3];
} edm_step_params;

/* ---------------------------------------------------------------- convolution (networks.py:35-37: F.conv2d) */
/* Y[p,co] = alpha * sum_{tap,ci} X[p+off(tap),ci] * Wp[tap,co,ci] + beta * R[p,co].  taps in {1,9}; "same" padding.
 * Forward conv with the forward pack; input-gradient (dgrad) with the flipped/transposed pack.  R may be NULL. */
int edm_conv_igemm(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                   int W, int Cin, int Cout, int taps, edm_stream_t stream);
/* second-generation kernel, same contract (LDS-DMA staging, 3-deep weight ring, 256x128 tile); returns -3 for shapes
 * it does not cover. */
int edm_conv_igemm_v2(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                      int W, int Cin, int Cout, int taps, edm_stream_t stream);
/* the 3x3 kernel of the 32x32 / 16x16 layers ("static schedule": 512 pixels x 128 or 64 channels per workgroup, the
 * (tap, 2 channel-chunk) loop unrolled 18x, per-lane fragment addresses computed once, counted waits) on
 * v_mfma_f32_16x16x32_bf16; -3 for shapes it does not cover (taps != 9, Cin % 64 != 0, W > 64).  (Its 32x32x16
 * predecessor edm_conv_igemm_v4 and the tall-tile edm_conv_igemm_v3 were retired in round 4: no dispatch reached the
 * first, and the second ran one launch per step -- conv_in -- 10 us slower than edm_conv_igemm.) */
int edm_conv_igemm_v6(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                      int W, int Cin, int Cout, int taps, edm_stream_t stream);
/* small feature maps (8x8 layers; W <= 16, Cin % 256 == 0), 3x3 only: 128x64 tile whose reduction dimension is split
 * over the four waves of the workgroup (private LDS regions, no barrier in the main loop, fixed-order LDS reduction);
 * -3 for shapes it does not cover */
int edm_conv_igemm_s(const void* X, const void* Wp, void* Y, const void* R, float alpha, float beta, int B, int H,
                     int W, int Cin, int Cout, int taps, edm_stream_t stream);
/* Same operation with an OUTPUT DESCRIPTOR -- how torch.cat((input, skip * gate)) (networks.py:311) and its backward stop
 * being copies: Y rows have stride ldY elements (0 = Cout: Y may be the left column block of the next block's
 * concatenated operand); Ysilu (optional) receives mp_silu(Y) at the same offsets of a second buffer with the same stride
 * (the operand of that block's first 3x3 conv, networks.py:315); Yb (optional): output channels >= split go to
 * Yb[pixel * ldYb + channel - split] (the 1x1 dgrad that produces d loss / d cat writes d loss / d input and the raw
 * gradient of the gated skip to the two tensors their consumers read).  kernel: which generation runs (1 edm_conv_igemm,
 * 2 _v2, 5 _s, 6 _v6; -3 = shape not covered by it).  R stays contiguous [pixels][Cout].
 * wfrag != 0: Wp is a FRAGMENT-MAJOR pack (edm_weight_prep_multi, bits 8 / 9 of a record's taps field) -- the layout the
 * 8x8 layers' kernel loads with coalesced 1-KiB instructions straight into MFMA operand registers; kernel 5 only.  The
 * three fused 3x3 entry points below take the same flag (then they run kernel 5 whatever the shape heuristics say). */
int edm_conv_igemm_o(const void* X, const void* Wp, void* Y, long ldY, void* Ysilu, void* Yb, long ldYb, int split,
                     const void* R, float alpha, float beta, int B, int H, int W, int Cin, int Cout, int taps, int kernel,
                     int wfrag, edm_stream_t stream);
/* Folded skip projection (round 6): Y = alpha3 * conv3x3(X, Wp) + alpha1 * conv1x1(X2, W2p) in ONE launch of the
 * static-schedule 3x3 kernel -- the decoder block's conv_1x1(cat) (networks.py:313) as a second reduction behind the nine taps
 * of the block's second 3x3 conv (networks.py:325-327), whose residual it was.  X2 [pixels][ldX2] (first C2 channels read),
 * W2p [Cout][C2] = the 1x1 conv's forward pack; Y / ldY / Ysilu as edm_conv_igemm_o.  edm_conv3x3_fold_supported() -> 1 / 0;
 * the entry point returns -3 for shapes it does not cover (the caller keeps the two launches). */
int edm_conv3x3_fold_supported(int B, int H, int W, int Cin, int Cout, int C2);
int edm_conv3x3_fold(const void* X, const void* Wp, const void* X2, long ldX2, const void* W2p, int C2, void* Y, long ldY,
                     void* Ysilu, float alpha3, float alpha1, int B, int H, int W, int Cin, int Cout, edm_stream_t stream);
/* first 3x3 conv of a block with the embedding modulation fused into its epilogue (networks.py:253-260 / 317-324):
 * Y = conv3x3(X) (bf16, may be NULL in eval), Y2 = dropout(mp_silu(Y * (lin[b,:]*gain + 1))) -- bit-identical to
 * edm_mod_silu_drop_fwd applied to Y (same Philox counters), so edm_mod_silu_drop_bwd serves as its backward. */
int edm_conv3x3_mod(const void* X, const void* Wp, void* Y, void* Y2, const float* lin, long lin_stride,
                    const float* gain, float pdrop, unsigned long long seed, unsigned sub, unsigned step, int mark_dropped,
                    int B, int H, int W, int Cin, int Cout, const void* dyn, int wfrag, edm_stream_t stream);
/* backward counterpart: dgrad of the block's second 3x3 conv (ga = alpha*conv3x3(dY, Wd), never written) with the
 * modulation backward in the epilogue: GR = ga*keep*mp_silu'(u*m)*m, gm[b,c] += sum_px ga*keep*mp_silu'(u*m)*u (gm
 * zero-filled fp32, rows of gm_stride floats, 0 = Cout); finish with edm_mod_finish, or -- when gm is a column slice of a
 * buffer shared by all blocks -- with ONE edm_mod_finish_multi at the end of the backward pass.  -3 when H*W % 32 != 0
 * (use the separate kernels).  mark_dropped (forward) writes NaN into the elements of Y (= U) the dropout removed; with
 * u_marked (backward) the mask is read back from those NaNs instead of regenerating the Philox stream (a third of the
 * backward epilogue's vector-ALU work). */
int edm_conv3x3_modbwd(const void* dY, const void* Wd, float alpha, const void* U, const float* lin, long lin_stride,
                       const float* gain, void* GR, float* gm, long gm_stride, float pdrop, unsigned long long seed,
                       unsigned sub, unsigned step, int u_marked, int B, int H, int W, int Cin, int Cout, const void* dyn,
                       int wfrag, edm_stream_t stream);
/* dgrad of a block's first 3x3 conv with the mp_silu backward of the block input in its epilogue
 * (g = conv3x3(dY, Wd) never written): GX = mp_silu'(Xpre)*g + add_scale*ADD (ADD may be NULL). */
int edm_conv3x3_silubwd(const void* dY, const void* Wd, const void* Xpre, const void* ADD, float add_scale, void* GX,
                        int B, int H, int W, int Cin, int Cout, int wfrag, edm_stream_t stream);
int edm_mod_finish(const float* gm, const float* lin, long lin_stride, const float* gain, float* glin, long glin_stride,
                   float* ggain, int B, int C, edm_stream_t stream);
/* every block's finish in one launch: gm_all / lin_all / glin_all are [B][stride] fp32, items a DEVICE array of
 * {const float* gain; float* ggain; int col0, C;}: glin[:, col0:col0+C] += gm * *gain, *ggain += sum gm * lin */
int edm_mod_finish_multi(const float* gm_all, const float* lin_all, float* glin_all, long stride, const void* items_dev,
                         int n_items, int B, edm_stream_t stream);
/* weight gradient: slabs[s,tap,co,ci] (fp32, nsplit = edm_conv_wgrad_nsplit(...)) partial sums over pixels; LDS-DMA staging,
 * rolling X window; Cin, Cout % 32 == 0; -3 = not covered (3x3 with W > 126).  (The first-generation edm_conv_wgrad was retired
 * in round 6: no default dispatch reached it.) */
int edm_conv_wgrad_nsplit(int B, int H, int W, int Cin, int Cout, int taps);
int edm_conv_wgrad_v2(const void* X, const void* dY, float* slabs, int B, int H, int W, int Cin, int Cout, int taps,
                      int nsplit, edm_stream_t stream);
/* 1x1 layers (networks.py:21-38 with kernel_size 1: skip projections, attention qkv/out): 256x128-output tiles,
 * slabs fp32 [nsplit][Cout][Cin] with nsplit = edm_conv_wgrad_1x1_nsplit(npix, Cin, Cout); Cin, Cout % 32 == 0. */
int edm_conv_wgrad_1x1_nsplit(long npix, int Cin, int Cout);
int edm_conv_wgrad_1x1(const void* X, const void* dY, float* slabs, long npix, int Cin, int Cout, int nsplit,
                       edm_stream_t stream);
/* up to 16 1x1 layers in ONE launch (their (tile, split) workgroups laid end to end in one grid); `items` is a HOST array
 * read during the call */
typedef struct {
  const void* X;   /* bf16 [npix][Cin] */
  const void* dY;  /* bf16 [npix][Cout] */
  float* slabs;    /* fp32 [nsplit][Cout][Cin], nsplit = edm_conv_wgrad_1x1_nsplit_grouped(npix, Cin, Cout) */
  long npix;
  int Cin, Cout, nsplit, pad;
} edm_wgrad1_item;
int edm_conv_wgrad_1x1_nsplit_grouped(long npix, int Cin, int Cout);
/* LAUNCH TABLES (round 4): the three grouped entry points below keep their per-layer table in DEVICE memory -- no kernel
 * takes a multi-KB by-value argument (a suspect of the hipGraph corruption of round 2).  The caller passes
 * table_host (PINNED host memory) and table_dev (device memory), each of at least edm_*_table_bytes() bytes: the call
 * fills table_host, issues ONE stream-ordered copy to table_dev (a memcpy node when the stream is being captured) and
 * launches with the device pointer.  table_host must stay valid and unmodified until that copy has executed (for a
 * captured stream: for the life of the graph), table_dev until the kernels have.  defer_upload != 0: the call only fills
 * table_host (any host memory) and launches; the CALLER copies table_host to table_dev before the launch can execute -- for
 * a stream capture once, after the capture has ended, instead of a copy node that every replay would run again. */
long edm_conv_wgrad_1x1_group_table_bytes(void);
int edm_conv_wgrad_1x1_group(const edm_wgrad1_item* items, int n, void* table_host, void* table_dev, int defer_upload,
                             edm_stream_t stream);
/* third generation, 3x3 layers, a GROUP of layers per call (autograd wgrad of networks.py:35-37 + the projection of
 * networks.py:32-36's normalisation): the reduction dimension of all layers is laid end to end and cut into equal
 * ranges, one per workgroup (128x64x9 tile, one wave per SIMD); partial tiles go to `workspace`; a second launch sums
 * each weight row's partials, projects through w_hat = w/(eps + |w|/sqrt(n))/sqrt(n) and writes (accumulate=0) or
 * accumulates (1) grad.  `items` is a HOST array read during the call; <= edm_wgrad3_max_layers() (48) layers, all with W <= 62 or all with
 * 62 < W <= 126; Cin, Cout % 32 == 0.  workspace >= edm_wgrad3_workspace(items, n) bytes (-1 on bad arguments). */
typedef struct {
  const void* X;    /* bf16 [B*H*W][Cin]  layer input */
  const void* dY;   /* bf16 [B*H*W][Cout] gradient of the layer output */
  const float* w;   /* fp32 master weight [Cout][I][3][3] */
  float* grad;      /* fp32 gradient of w, same layout */
  const int* perm;  /* packed output row -> master row, or NULL */
  int B, H, W, Cin, Cout, I; /* I <= Cin: input channels of w (X may be zero-padded to Cin) */
  float scale;      /* gradient scale folded in before the projection */
  int accumulate;
} edm_wgrad3_item;
long edm_wgrad3_workspace(const edm_wgrad3_item* items, int n);
long edm_wgrad3_table_bytes(void);
int edm_wgrad3_max_layers(void);
int edm_wgrad3_group(const edm_wgrad3_item* items, int n, void* workspace, long workspace_bytes, void* table_host,
                     void* table_dev, int defer_upload, edm_stream_t stream);

/* ---------------------------------------------------------------- weights (networks.py:17-19, 32-36, 55-59) */
/* forced weight normalisation (in place when normalize_inplace) + effective weight w/(eps+|w|/sqrt(n))/sqrt(n),
 * written as bf16 forward pack [taps,O,Ipad], bf16 dgrad pack [taps,I,O] (taps flipped) and/or fp32 [O,I*taps]. */
int edm_weight_prep(float* w, int O, int I, int taps, int Ipad, void* wp_fwd, void* wp_dgrad, float* w_hat,
                    const int* perm, int normalize_inplace, edm_stream_t stream);
/* multi-tensor form: one launch for every weight of a network.  descs = device array of 64-byte records
 * {float* w; bf16* fwd; bf16* dgrad; float* hat; const int* perm; int O, I, taps, Ipad, row0, rb;}; groups = device
 * int32 [n_groups][2] = (record index, first packed row): one workgroup prepares <= rb consecutive rows of a record
 * through an LDS tile of rb*I*taps bf16 (lds_bytes = the largest such tile, <= 128 KiB). */
int edm_weight_prep_multi(const void* descs, const int* groups, int n_groups, int lds_bytes, int normalize_inplace,
                          edm_stream_t stream);
/* reduce split-K slabs and project through the normalisation -> gradient of the fp32 master weight [O,I,taps]. */
int edm_wgrad_finish(const float* slabs, int S, const float* w, float* grad, const int* perm, int O, int I, int Ipad,
                     int taps, float scale, int accumulate, edm_stream_t stream);

/* the same for up to 40 tensors in ONE launch (`items` is a HOST array read during the call) */
typedef struct {
  const float* slabs; /* fp32 [S][taps][O][Ipad] */
  const float* w;     /* fp32 master weight [O][I][taps] */
  float* grad;        /* fp32 gradient, same layout */
  const int* perm;    /* or NULL */
  int S, O, I, Ipad, taps;
  float scale;
  int accumulate;
} edm_finish_item;
long edm_wgrad_finish_multi_table_bytes(void);
int edm_wgrad_finish_multi(const edm_finish_item* items, int n, void* table_host, void* table_dev, int defer_upload,
                           edm_stream_t stream);

/* ---------------------------------------------------------------- attention (networks.py:194-202) */
/* qkv [B*N,3C] channel order [head][q|k|v][d]; q,k,v pixel-normalised over d; softmax(qk^T/sqrt(d)) v. d = 64, N<=256 */
int edm_attention_fwd(const void* qkv, void* y, int B, int N, int C, int heads, edm_stream_t stream);
int edm_attention_bwd(const void* qkv, const void* y, const void* gy, void* gqkv, int B, int N, int C, int heads,
                      edm_stream_t stream);

/* CosineAttention with the qkv projection inside the attention kernel (csrc/attention_fused.hip, round 5; replaces
 * networks.py:193-202 = qkv_conv -> view -> pixel_norm -> scaled_dot_product_attention as ONE launch; the qkv tensor never
 * exists in HBM).  x [B*N, C] bf16 tokens, Wqkv [3C, C] bf16 = the qkv conv's forward pack (rows in [head][q|k|v][d] order),
 * y [B*N, C] bf16, stat [B, heads, N] fp32 (softmax normaliser, kept for the backward; may be NULL in evaluation).
 * Covered: C = 256, 4 heads (head_dim 64), 33..256 tokens (edm_attention_qkv_supported); heads_per_wg 0 = default. */
int edm_attention_qkv_supported(int N, int C, int heads);
int edm_attention_qkv_fwd(const void* x, const void* Wqkv, void* y, void* stat, int B, int N, int C, int heads,
                          int heads_per_wg, edm_stream_t stream);
/* its backward (autograd of networks.py:193-205 up to the projections' own GEMMs): gqkv [B*N, 3C] bf16, packed channel order,
 * = d loss / d qkv_conv(x) from gout = d loss / d out_conv(y) [B*N, C].  dO = alpha * gout . W_out is formed inside from
 * Wd_out = the out conv's dgrad pack [C, C] (alpha = the mp_add coefficient of the attention branch); q, k, v are recomputed
 * from x and Wqkv; y, stat = the forward's outputs.  The callers' remaining launches: the qkv conv's dgrad (gx) and the two
 * weight gradients.  heads_per_wg: 0 or 1 (the backward runs one head per workgroup; anything else is refused). */
int edm_attention_qkv_bwd(const void* x, const void* y, const void* gout, const void* stat, const void* Wqkv,
                          const void* Wd_out, void* gqkv, float alpha, int B, int N, int C, int heads, int heads_per_wg,
                          edm_stream_t stream);

/* ---------------------------------------------------------------- per-pixel / elementwise */
/* pixel_norm over C + mp_silu (networks.py:9-14, 83-84, 249-252); dsave[p] = eps + |x_p|/sqrt(C) */
int edm_pixelnorm_silu_fwd(const void* x, void* xn, void* a, float* dsave, long P, int C, edm_stream_t stream);
/* gadd (optional, bf16 like gx): added to the result -- the gradient reaching the same tensor along the U-Net skip, which
 * autograd would otherwise sum with a separate add (networks.py:592-600: the encoder output feeds the next block AND a
 * decoder block) */
int edm_pixelnorm_silu_bwd(const void* xn, const float* dsave, const void* gxn, float gxn_scale, const void* ga,
                           const void* gadd, void* gx, long P, int C, edm_stream_t stream);
/* round 6: the same pair with the 2x2 average pool of an EncD block folded in (networks.py:246-252, blocks without a 1x1
 * conv): x / gx / gadd are (B, 2 Hp, 2 Wp, C) -- the tensor BEFORE the pool --, xn / a / dsave / gxn / ga (B, Hp, Wp, C); the
 * pooled tensor is never written.  Bit-identical to edm_pool2(0.25) + edm_pixelnorm_silu_fwd and to edm_pixelnorm_silu_bwd +
 * edm_up2(0.25, add = gadd).  C <= 1024. */
int edm_pool_pixelnorm_silu_fwd(const void* x, void* xn, void* a, float* dsave, int B, int Hp, int Wp, int C,
                                edm_stream_t stream);
int edm_pool_pixelnorm_silu_bwd(const void* xn, const float* dsave, const void* gxn, float gxn_scale, const void* ga,
                                const void* gadd, void* gx, int B, int Hp, int Wp, int C, edm_stream_t stream);
/* mp_silu (networks.py:316) and its backward: gx = mp_silu'(x)*ga + extra_scale*gextra */
int edm_silu_fwd(const void* x, void* a, long n, edm_stream_t stream);
int edm_silu_bwd(const void* x, const void* ga, const void* gextra, float extra_scale, void* gx, long n,
                 edm_stream_t stream);
/* out = alpha*a + beta*b (mp_add, networks.py:87-88) */
int edm_axpby(const void* a, float alpha, const void* b, float beta, void* out, long n, edm_stream_t stream);
/* a = dropout(mp_silu(r * (lin*gain + 1)))  (networks.py:255-260 / 319-324); Philox mask from (seed, sub, step) */
int edm_mod_silu_drop_fwd(const void* r, const float* lin, long lin_stride, const float* gain, void* a, int B, int HW,
                          int C, float pdrop, unsigned long long seed, unsigned sub, unsigned step, const void* dyn,
                          edm_stream_t stream);
int edm_mod_silu_drop_bwd(const void* r, const float* lin, long lin_stride, const float* gain, const void* ga, void* gr,
                          float* gm, float* glin, long glin_stride, float* ggain, int B, int HW, int C, float pdrop,
                          unsigned long long seed, unsigned sub, unsigned step, const void* dyn, edm_stream_t stream);
/* its first half alone: gr, and the raw modulation gradient accumulated into gm (zero-filled fp32, rows of gm_stride floats:
 * a column slice of the buffer all blocks share) -- finished for every block by ONE edm_mod_finish_multi (the form
 * edm_conv3x3_modbwd takes with a shared buffer, for maps whose H*W is no multiple of 32, which that entry refuses) */
int edm_mod_silu_drop_bwd_raw(const void* r, const float* lin, long lin_stride, const float* gain, const void* ga, void* gr,
                              float* gm, long gm_stride, int B, int HW, int C, float pdrop, unsigned long long seed,
                              unsigned sub, unsigned step, const void* dyn, edm_stream_t stream);
int edm_dropout_mask(unsigned char* mask, long n, float pdrop, unsigned long long seed, unsigned sub, unsigned step,
                     edm_stream_t stream);
/* 2x2 average pool / nearest-exact x2 (networks.py:80, 72); H,W = OUTPUT dims; scale folds the backward factors */
int edm_pool2(const void* x, void* y, int B, int Hout, int Wout, int C, float scale, edm_stream_t stream);
/* up2: y = scale * up(x) + add (add optional, same shape as y: see edm_pixelnorm_silu_bwd's gadd) */
int edm_up2(const void* x, const void* add, void* y, int B, int Hout, int Wout, int C, float scale, edm_stream_t stream);
/* DecU blocks (networks.py:312-316): y = nearest-exact x2 of x and a = mp_silu(y) in one pass (== edm_up2 + edm_silu_fwd) */
int edm_up2_silu(const void* x, void* y, void* a, int B, int Hout, int Wout, int C, edm_stream_t stream);
/* out[b,c] = scale * sum_hw x[b,hw,c] (* y[b,hw,c])  -- ScaleLong mean (networks.py:116) and its gate gradient;
 * written, not accumulated, in a fixed summation order (bit-reproducible) */
int edm_reduce_hw(const void* x, long x_stride, const void* y, long y_stride, float* out, int B, int HW, int C,
                  float scale, edm_stream_t stream);
/* ScaleLong gate MLP (networks.py:112-118), fp32, per sample */
int edm_scalelong_fwd(const float* mean, const float* W1h, const float* W2h, float* gate, float* z1save, int B, int C,
                      int R, edm_stream_t stream);
int edm_scalelong_bwd(const float* mean, const float* W1h, const float* W2h, const float* gate, const float* z1save,
                      const float* ggate, float* gmean, float* gW1h, float* gW2h, int B, int C, int R,
                      edm_stream_t stream);
/* The two steps above in ONE launch per direction (one workgroup per sample: mean over H*W in a fixed order, then the
 * sample's gate MLP from LDS).  fwd: skip [B*HW][C] bf16 -> mean, gate [B][C], z1save [B][R].  bwd: channels
 * [c_off, c_off+C) of gcat (rows of gcat_stride elements) are d loss / d (skip*gate); ggate = sum_hw gcat*skip is formed
 * and consumed in place; gmean, gW1h [R][C+1] and gW2h [C][R] are written (the weight gradients by a second small launch
 * that sums the per-sample outer products over the batch in a fixed order); ws: [B][C + 2R] floats of scratch. */
int edm_skip_gate_fwd(const void* skip, const float* W1h, const float* W2h, float* mean, float* gate, float* z1save,
                      int B, int HW, int C, int R, edm_stream_t stream);
int edm_skip_gate_bwd(const void* gcat, long gcat_stride, int c_off, const void* skip, const float* mean,
                      const float* W1h, const float* W2h, const float* gate, const float* z1save, float* gmean,
                      float* gW1h, float* gW2h, float* ws, int B, int HW, int C, int R, edm_stream_t stream);
/* round 6: gW1h == gW2h == NULL in edm_skip_gate_bwd skips its second launch (the batch sums of the weight gradients); the
 * caller then runs it ONCE for all the gates of a backward pass (<= 32): items[k] = {ws, mean of gate k, gW1h, gW2h (written),
 * B, C, R}.  Launch-table contract as edm_wgrad_finish_multi (table_host / table_dev / defer_upload). */
typedef struct {
  const float* ws;
  const float* mean;
  float* gW1h;
  float* gW2h;
  int B, C, R, pad;
} edm_skip_gate_wgrad_item;
long edm_skip_gate_wgrad_multi_table_bytes(void);
int edm_skip_gate_wgrad_multi(const edm_skip_gate_wgrad_item* items, int n, void* table_host, void* table_dev,
                              int defer_upload, edm_stream_t stream);
/* round 6: edm_skip_gate_fwd for up to 32 skip tensors of ONE channel count in one launch (the U-Net's skips all exist when
 * the encoder ends; one workgroup per (tensor, sample): a launch that fills the chip instead of nine half-empty ones on the
 * decoder's critical chain).  Same values as the per-tensor entry point.  Launch-table contract as above. */
typedef struct {
  const void* skip;     /* bf16 [B*HW][C] */
  const float* W1h;     /* [R][C+1] */
  const float* W2h;     /* [C][R] */
  float* mean;          /* [B][C]  written */
  float* gate;          /* [B][C]  written */
  float* z1save;        /* [B][R]  written */
  void* cat;            /* optional: bf16 rows of Ci + C elements; columns [Ci, Ci + C) receive skip * gate (edm_skip_half_fwd's */
  void* silu_out;       /* work, done by the workgroup that just reduced the sample) and, if given, mp_silu of it here */
  int B, HW, C, R, Ci, pad;
} edm_skip_gate_fwd_item;
long edm_skip_gate_fwd_multi_table_bytes(void);
int edm_skip_gate_fwd_multi(const edm_skip_gate_fwd_item* items, int n, void* table_host, void* table_dev, int defer_upload,
                            edm_stream_t stream);
/* ... and the backward of those gates, deferred: nothing on the backward's critical chain needs a gate's gmean (it only feeds
 * the gradient of the U-Net skip, which the encoder's backward consumes), so the decoder blocks queue and the last one
 * launches (a) edm_skip_gate_bwd_multi = the first launch of edm_skip_gate_bwd (gmean, ws written; weight gradients by
 * edm_skip_gate_wgrad_multi as before) for all gates of one channel count, (b) edm_skip_half_bwd_multi = edm_skip_half_bwd for
 * all their tensors.  Tables of edm_skip_gate_bwd_multi_table_bytes() bytes. */
typedef struct {
  const void* gcat;     /* bf16 rows of gcat_stride elements; channels [c_off, c_off + C) = d loss / d (skip * gate) */
  long gcat_stride;
  const void* skip;     /* bf16 [B*HW][C] */
  const float* W1h;
  const float* W2h;
  const float* gate;
  const float* z1save;
  float* gmean;         /* [B][C]       written */
  float* ws;            /* [B][C + 2R]  written */
  void* gskip;          /* optional: bf16 [B*HW][C] written = gcat[:, c_off:c_off+C] * gate + gmean / HW (edm_skip_half_bwd's work) */
  int c_off, B, HW, C, R, pad;
} edm_skip_gate_bwd_item;
typedef struct {
  const void* gcs;      /* bf16 [B*HW][Cs] */
  const float* gate;
  const float* gmean;
  void* gskip;          /* bf16 [B*HW][Cs]  written: gcs * gate + gmean / HW */
  int B, HW, Cs, pad;
} edm_skip_half_bwd_item;
long edm_skip_gate_bwd_multi_table_bytes(void);
int edm_skip_gate_bwd_multi(const edm_skip_gate_bwd_item* items, int n, void* table_host, void* table_dev, int defer_upload,
                            edm_stream_t stream);
int edm_skip_half_bwd_multi(const edm_skip_half_bwd_item* items, int n, void* table_host, void* table_dev, int defer_upload,
                            edm_stream_t stream);
/* cat = [inp, skip*gate] (networks.py:311) and backward */
int edm_concat_gate_fwd(const void* inp, const void* skip, const float* gate, void* cat, void* silu_out, int B, int HW,
                        int Ci, int Cs, edm_stream_t stream);
int edm_concat_gate_bwd(const void* gcat, const float* gate, const float* gmean, void* ginp, void* gskip, int B,
                        int HW, int Ci, int Cs, edm_stream_t stream);
/* The skip half alone (the input half is written by the producer of `input` through edm_conv_igemm_o):
 * cat[b,p,Ci:] = skip*gate, silu_out[b,p,Ci:] = mp_silu(of it) (rows of Ci + Cs elements; silu_out may be NULL); and
 * gskip = gcs*gate + gmean/HW from the skip half gcs [B*HW][Cs] of d loss / d cat. */
int edm_skip_half_fwd(const void* skip, const float* gate, void* cat, void* silu_out, int B, int HW, int Ci, int Cs,
                      edm_stream_t stream);
int edm_skip_half_bwd(const void* gcs, const float* gate, const float* gmean, void* gskip, int B, int HW, int Cs,
                      edm_stream_t stream);

/* ---------------------------------------------------------------- EDM preconditioning (networks.py:578-587, 602-603) */
/* out[b,h,w,:] = [c_in(b)*noisy[b,:,h,w], 1, 0...] (bf16, CP channels); sigma_stride 0 = one scalar sigma */
int edm_precond_in(const float* noisy, const float* sigma, int sigma_stride, float sigma_data, void* out, int B,
                   int Cimg, int HW, int CP, edm_stream_t stream);
/* D = conv_out(x)*gain_out*c_out + noisy*c_skip (fp32 NCHW); Fraw = conv_out(x) saved for the backward */
int edm_conv_out_fwd(const void* x, const float* w_hat, const float* gain_out, const float* noisy, const float* sigma,
                     int sigma_stride, float sigma_data, float* D, float* Fraw, int B, int HW, int C, int Co,
                     edm_stream_t stream);
int edm_conv_out_bwd(const void* x, const float* w_hat, const float* gain_out, const float* Fraw, const float* dD,
                     const float* sigma, int sigma_stride, float sigma_data, void* gx, float* gw_hat, float* ggain,
                     int B, int HW, int C, int Co, edm_stream_t stream);
int edm_nchw_to_nhwc_bf16(const float* x, void* y, int B, int C, int HW, edm_stream_t stream);
int edm_nhwc_bf16_to_nchw(const void* x, float* y, int B, int C, int HW, edm_stream_t stream);

/* ---------------------------------------------------------------- fp32 linears + embedding (networks.py:58-60, 121-178) */
int edm_linear_fwd(const float* X, const float* W, float* Y, int M, int N, int K, edm_stream_t stream);
int edm_linear_dgrad(const float* dY, const float* W, float* dX, int M, int N, int K, int accumulate,
                     edm_stream_t stream);
int edm_linear_wgrad(const float* dY, const float* X, float* dW, int M, int N, int K, int accumulate,
                     edm_stream_t stream);
int edm_fourier_fwd(const float* sigma, int sigma_stride, const float* freqs, const float* phases, float* out, int B,
                    int Fd, edm_stream_t stream);
int edm_embed_combine_fwd(const float* emb_sigma, const float* wcls_hat, const long long* labels, float add_factor,
                          int K, float* pre, float* out, int B, int E, edm_stream_t stream);
int edm_embed_combine_bwd(const float* gout, const float* pre, const long long* labels, float add_factor, int K,
                          float* gemb_sigma, float* gwcls_hat, int B, int E, edm_stream_t stream);

/* ---------------------------------------------------------------- step level (edm.py:84-93, 212; metric.py:8-18; edm.py:251; ema.py:137-140; solvers.py:49-57) */
int edm_diffuse(const float* clean, float* noisy, float* sigma, float P_mean, float P_std, int B, long CHW,
                unsigned long long seed, unsigned step, const void* dyn, edm_stream_t stream);
int edm_diffuse_given(const float* clean, const float* eps, const float* noise, float* noisy, float* sigma,
                      float P_mean, float P_std, int B, long CHW, edm_stream_t stream);
/* weight = (sigma^2+sd^2)/(sigma*sd)^2 (edm.py:212) unless weight_override; acc_sum/acc_total (nullable): the metric's
 * epoch state (metric.py:38-49) accumulated in the same pass */
int edm_weighted_mse(const float* D, const float* clean, const float* sigma, const float* weight_override,
                     float sigma_data, float* loss, float* dD, int B, long CHW, float* acc_sum, long long* acc_total,
                     edm_stream_t stream);
/* zero_grad != 0: the gradient arena is cleared in the same pass (optimizer.zero_grad()).
 * health (nullable device word, sticky): bit 0 is OR-ed in when a non-finite gradient / updated weight passed through
 * the step, bit 1 (Heun updates) when the sampler state went non-finite -- the sentinel a replayed hipGraph leaves for
 * the host (Trainer.fit reads it at its log interval, the solver after a solve) so a corrupted replay fails loudly. */
int edm_adam_ema(float* theta, float* grad, float* m, float* v, float* ema, long n, float lr, float b1, float b2,
                 float eps, int step, float ema_beta, float grad_scale, const void* dyn, int zero_grad,
                 unsigned* health, edm_stream_t stream);
int edm_heun_euler(const float* x, const float* D, float t0, float t1, float* dx, float* x1, long n, unsigned* health,
                   edm_stream_t stream);
int edm_heun_correct(const float* x, const float* dx, const float* x1, const float* D1, float t0, float t1, float* out,
                     long n, unsigned* health, edm_stream_t stream);
int edm_scale_f32(const float* x, float s, float* y, long n, edm_stream_t stream);

/* ---------------------------------------------------------------- reference-precision evaluation (eval_f32.hip)
 * The reference samples / validates in fp32 (generate.py:39-44, callbacks.py:41-49, solvers.py:43-59).  These entry
 * points are the exact-fp32 forward: NHWC fp32 activations [pixels][channels], fp32 effective weights (the w_hat arrays
 * of edm_weight_prep: [Cout][I*taps], master OIHW order), every convolution on v_mfma_f32_32x32x2_f32 (fp32 products,
 * fp32 sums).  Forward only, eval semantics (no dropout, no in-place weight normalisation). */
/* Y = alpha*conv(X, w_hat) + beta*R, or (lin != NULL) the block's modulation epilogue Y = mp_silu(alpha*conv *
 * (lin[b,:]*gain + 1)) (networks.py:253-260).  taps in {1,9}; Cin % 8 (3x3) / % 32 (1x1); I <= Cin (X zero-padded). */
int edm_f32_conv(const float* X, const float* w_hat, float* Y, const float* R, float alpha, float beta, const float* lin,
                 long lin_stride, const float* gain, int B, int H, int W, int Cin, int I, int Cout, int taps,
                 edm_stream_t stream);
/* Split-bf16 form of the same evaluation (round 4): an fp32-accurate convolution at a third of the bf16 kernels' rate
 * instead of the f32-MFMA rate (1/16).  An fp32 value travels as a PAIR of bf16, hi = bf16(x), lo = bf16(x - hi):
 * edm_f32_to_pairs turns [rows][C] floats into [rows][2 C] bf16 = [hi | lo]; edm_split_pack turns w_hat [O][I*taps] into
 * [taps][O][3 Ip] bf16 = [w_hi | w_lo | w_hi]; edm_split_conv accumulates hi.w_hi + hi.w_lo + lo.w_hi in fp32 (what is
 * dropped, lo.w_lo, is 2^-18 of a product) and writes FLOATS: Y = alpha*conv + beta*R (R floats), or, with lin,
 * Y = mp_silu(conv * (lin[b,:]*gain + 1)); Ypairs (optional; Y may then be NULL) receives the same result as pairs, the
 * next conv's Xp.  C % 32 == 0, Cout % 8 == 0, taps in {1, 9}. */
int edm_f32_to_pairs(const float* x, void* pairs, long rows, int C, edm_stream_t stream);
int edm_split_pack(const float* w_hat, void* pack, int O, int I, int taps, int Ip, edm_stream_t stream);
int edm_split_conv(const void* Xp, const void* Wp3, float* Y, void* Ypairs, const float* R, float alpha, float beta,
                   const float* lin, long lin_stride, const float* gain, int B, int H, int W, int C, int Cout, int taps,
                   edm_stream_t stream);
/* the same with an output descriptor for the pairs (round 6; how the split evaluation's torch.cat((input, skip * gate)),
 * networks.py:311, stops being a copy): Ypairs rows have ld_pairs elements (0 = 2 Cout) and the lo halves sit lo_off elements
 * behind the hi halves (0 = Cout) -- i.e. Ypairs may be the left column blocks of the NEXT decoder block's concatenated
 * operand [hi(Ci + Cs) | lo(Ci + Cs)]; Ysilu_pairs (optional) receives mp_silu of the result as pairs at the same offsets of
 * a second buffer (that block's first 3x3 conv reads it, networks.py:316).  Any of Y / Ypairs / Ysilu_pairs may be NULL
 * (not all three).  edm_f32_skip_half fills the right column blocks (the gated skip and mp_silu of it). */
int edm_split_conv_o(const void* Xp, const void* Wp3, float* Y, void* Ypairs, long ld_pairs, long lo_off, void* Ysilu_pairs,
                     const float* R, float alpha, float beta, const float* lin, long lin_stride, const float* gain, int B,
                     int H, int W, int C, int Cout, int taps, edm_stream_t stream);
/* edm_conv3x3_fold for the split evaluation: Y = alpha3 * conv3x3(Xp, Wp3) + alpha1 * conv1x1(X2p, W2p3); X2p [pixels][2 C2]
 * pairs, W2p3 [Cout][3 C2]; outputs as edm_split_conv_o; -3 where the folded form is not built (the caller keeps two launches) */
int edm_split_conv_fold(const void* Xp, const void* Wp3, const void* X2p, const void* W2p3, int C2, float* Y, void* Ypairs,
                        long ld_pairs, long lo_off, void* Ysilu_pairs, float alpha3, float alpha1, int B, int H, int W, int C,
                        int Cout, edm_stream_t stream);
int edm_f32_skip_half(const float* skip, const float* gate, void* cat_pairs, void* silu_pairs, int B, int HW, int Ci, int Cs,
                      edm_stream_t stream);
/* cosine attention (networks.py:194-202) on the qkv conv's own output order: channel head*3d + 3*dd + {q,k,v};
 * y [B*N][C] with channel head*d + dd.  head_dim in {32, 64, 128, 144, 192}; any number of tokens (key tiles of 64). */
int edm_f32_attention(const float* qkv, float* y, int B, int N, int C, int heads, edm_stream_t stream);
/* the same attention for the split-bf16 ("f32x3") evaluation path: every matrix product in three bf16 MFMA passes over
 * (hi, lo) operand pairs (fp32-accurate to 2^-17 per operand, as edm_split_conv), normalisation / softmax / accumulation in
 * fp32.  y (fp32 [B*N][C]) and / or ypairs (bf16 [B*N][2C] = [hi | lo], the out conv's operand format) -- either may be
 * NULL.  head_dim 64 and N <= 256 tokens; -3 otherwise (callers fall back to edm_f32_attention). */
int edm_split_attention(const float* qkv, float* y, void* ypairs, int B, int N, int C, int heads, edm_stream_t stream);
/* (s_pairs / pairs_row / pairs != 0 in the three entry points below: the mp_silu / concat outputs are written as split-bf16
 * PAIRS -- rows [hi(C) | lo(C)] of bf16 in the same bytes -- ready to be edm_split_conv's Xp) */
int edm_f32_pixelnorm_silu(const float* x, float* xn, float* s, long P, int C, int s_pairs, edm_stream_t stream);
int edm_f32_silu(const float* x, float* s, long n, int pairs_row, edm_stream_t stream);
int edm_f32_pool2(const float* x, float* y, int B, int Hout, int Wout, int C, edm_stream_t stream);
int edm_f32_up2(const float* x, float* y, int B, int Hout, int Wout, int C, edm_stream_t stream);
/* round 6, one pass each: EncD blocks without a 1x1 conv (networks.py:246-252): 2x2 average pool -> pixel norm -> mp_silu,
 * the pooled tensor never written (bit-identical to edm_f32_pool2 + edm_f32_pixelnorm_silu; C <= 1024); DecU blocks
 * (networks.py:312-316): y = nearest-exact x2 of x and s = mp_silu(y) (floats, or pairs when s_pairs != 0) */
int edm_f32_pool_pixelnorm_silu(const float* x, float* xn, float* s, int B, int Hout, int Wout, int C, int s_pairs,
                                edm_stream_t stream);
int edm_f32_up2_silu(const float* x, float* y, float* s, int B, int Hout, int Wout, int C, int s_pairs, edm_stream_t stream);
/* ScaleLong gate of a skip tensor (networks.py:112-118): mean over H*W in a fixed order + the gate MLP, per sample */
int edm_f32_skip_gate(const float* skip, const float* W1h, const float* W2h, float* gate, int B, int HW, int C, int R,
                      edm_stream_t stream);
int edm_f32_concat_gate(const float* inp, const float* skip, const float* gate, float* cat, float* silu_out, int B,
                        int HW, int Ci, int Cs, int pairs, edm_stream_t stream);
int edm_f32_precond_in(const float* noisy, const float* sigma, int sigma_stride, float sigma_data, float* out, int B,
                       int Cimg, int HW, int CP, edm_stream_t stream);
int edm_f32_conv_out(const float* x, const float* w_hat, const float* gain_out, const float* noisy, const float* sigma,
                     int sigma_stride, float sigma_data, float* D, int B, int HW, int C, int Co, edm_stream_t stream);
int edm_f32_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, edm_stream_t stream);
int edm_f32_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, edm_stream_t stream);

/* ---------------------------------------------------------------- data formats either side of the path */
/* resident uint8 dataset [N][C][H][W] -> fp32 NCHW batch: sample b = image index[b]; (x/255 - mean)/std with the
 * optional per-sample horizontal flip (datamodules/cifar10datamodule.py:18-32, mnistdatamodule.py:18-30). */
int edm_u8_gather_normalize(const void* data, const long* index, float* out, int B, int C, int H, int W, long n_images,
                            float mean, float stdv, int flip, unsigned long long seed, unsigned epoch,
                            edm_stream_t stream);
/* (x*scale + offset).clip(0,255) -> uint8, layout preserved (cifar10datamodule.py:34-35: scale 127.5, offset 128) */
int edm_denormalize_u8(const float* x, void* out, long n, float scale, float offset, edm_stream_t stream);
/* clamp(pred*std[c]*2 + mean[c], 0, 1)*255 -> uint8 NHWC (callbacks.py:126-156, PreditionWriter) */
int edm_prediction_to_u8_nhwc(const float* pred, void* out, int B, int C, int H, int W, const float* mean,
                              const float* stdv, edm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TINYEDM_HIP_H */

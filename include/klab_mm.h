/* klab_mm.h -- C ABI of the MI355X-native Swin-V2 -> T5 caption-training hot path.
 *
 * The reference (Da-Tsuchi/KLab_MultiModalModel) has no FFI of its own: its hot path is the
 * Python call `MyModel.forward` + autograd backward (ref/models/model.py:19-26, ref/train.py:58-62),
 * whose arithmetic runs inside `transformers` (HF/swinv2, HF/t5; see SURVEY.md for the prefixes).
 * This header is the boundary a maintainer would bind instead (last row): plain
 * pointers and sizes, no torch types.  Conventions for EVERY entry point:
 *   - all buffers are caller-owned device allocations; nothing is allocated, freed or retained
 *     (the engine object is the one exception: it owns only host-side plans);
 *   - every call is stream-ordered on the `hipStream_t` passed as `void* stream`, never
 *     synchronises and never throws; it returns 0, a positive `hipError_t`, or a negative KLAB_ERR_*;
 *   - `dtype` is the storage type of GEMM operands / activations: KLAB_F32 (parity mode, exact f32
 *     MFMA) or KLAB_BF16 (bf16 operands, fp32 accumulation, fp32 residual stream and statistics);
 *   - dropout masks are a pure function of (*seed_dev, tag, element index) so that backward
 *     regenerates them; `seed_dev` points to ONE uint32 in device memory (graph-replay friendly).
 */
#ifndef KLAB_MM_H
#define KLAB_MM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KLAB_OK 0
#define KLAB_ERR_UNSUPPORTED (-2)
#define KLAB_ERR_BADARG (-3)

#define KLAB_F32 0
#define KLAB_BF16 1
#define KLAB_FP8 2 /* engine mode only (klab_model_cfg.dtype): bf16 storage and backward, forward Linear GEMMs on fp8 MFMA */

#define KLAB_ACT_NONE 0
#define KLAB_ACT_RELU 1 /* T5 DenseReluDense, HF/t5:83-94 */
#define KLAB_ACT_GELU 2 /* erf GELU, Swin-V2 MLP, HF/swinv2:539-548 */

#define KLAB_AUX_NONE 0
#define KLAB_AUX_NONZERO 1 /* x = aux!=0 ? x*aux_scale : 0   (backward of relu+dropout) */
#define KLAB_AUX_DGELU 2   /* x *= gelu'(aux)                (backward of erf GELU)    */

int klab_version(void);

/* ---- dense contraction (replaces every nn.Linear / Conv2d-as-GEMM of HF/t5 and HF/swinv2 and
 *      their autograd dgrad/wgrad; SURVEY §2.4 K1,K5,K6,K11-K14) -------------------------------
 * C[M,N] = epilogue(alpha * sum_k A(m,k) * B(n,k)).
 * a_kmajor=1: A(m,k) = A[m*lda + k]   (x[M,K] of a Linear forward)
 * a_kmajor=0: A(m,k) = A[k*lda + m]   (contraction over the slow dimension: wgrad)
 * same for B (b_kmajor=1 is a torch Linear weight W[N,K]).                                      */
typedef struct klab_gemm_args {
  int M, N, K;
  int dtype;   /* operand dtype */
  const void* A; long lda; int a_kmajor;
  const void* B; long ldb; int b_kmajor;
  void* C; long ldc; int c_dtype; /* KLAB_F32 or == dtype */
  int accumulate;                 /* C += result */
  float alpha; const float* alpha_dev; /* effective alpha = alpha * (*alpha_dev if non-null) */
  const float* bias;              /* [N] f32 or NULL */
  int act;
  const void* aux; long ldaux; int aux_mode; float aux_scale; /* aux: [M,N] in `dtype` */
  const void* residual; long ldr; int r_dtype;                /* added last */
  float drop_p; const uint32_t* seed_dev; uint32_t drop_tag;  /* dropout before the residual */
  int name_tag; /* 1: launch under the symbol klab_lmhead_gemm (128x128 NT tile) so profiles can single it out;
                   2: take the 256x256 eight-wave kernel (mm8p) whenever the shape is legal for it, 3: never (A/B, tests) */
  int atomic_ok; /* with accumulate=1 and a plain f32 product: the kernel may split K and add partial sums with
                    float atomics (summation order, hence the last bits, then vary run to run) */
} klab_gemm_args;
int klab_gemm(const klab_gemm_args* args, void* stream);
/* measurement only: HIP events around every klab_gemm launch (on the stream it is launched on) while enabled; read returns the launch
 * count, the summed duration and the summed algorithmic FLOPs (2 M N K) since it was enabled.  Call read after a synchronize.          */
int klab_gemm_probe_enable(int on);
int klab_gemm_probe_read(int* launches, float* total_ms, double* flops_total);
/* fp8 forward GEMM (BASELINE configs[4]): C = epilogue(alpha * sa[m] * sb[n * b_scale_stride] * sum_k A8(m,k) B8(n,k)).
 * A, B: OCP e4m3 bytes, both K-major (lda / ldb in bytes = elements); a_row_scale [M] and b_row_scale are the per-row
 * dequantisation scales klab_quant_fp8_rows / klab_quant_fp8_arena produce (amax / 448).  Every other field of klab_gemm_args
 * as for klab_gemm with dtype = KLAB_BF16 (bias, act, dropout, residual, bf16 / f32 output); accumulate is not supported.
 * K % 16 == 0, 16-byte aligned operands.                                                                             */
int klab_gemm_fp8(const klab_gemm_args* a, const float* a_row_scale, const float* b_row_scale, long b_scale_stride, void* stream);
/* x [M, K] bf16 rows -> x8 [M, K] e4m3 + row_scale[m] = amax(row) / 448 (1 for an all-zero row); K <= 4096, K % 8 == 0 */
int klab_quant_fp8_rows(const void* x, long ldx, int M, int K, void* x8, long ld8, float* row_scale, void* stream);
/* all GEMM weights of a bf16 arena in one launch: desc_dev[i] = {long arena_off, rows, K, first_row}; row r of tensor i is
 * quantised to arena_fp8 + off + r*K with its scale at scales[(off + r*K) / 8]                                         */
int klab_quant_fp8_arena(const void* desc_dev, int ndesc, long total_rows, const void* arena_bf16, void* arena_fp8, float* scales,
                         void* stream);
/* n independent GEMMs.  Split-K weight-gradient members (bf16, both operands m-major, f32 C with accumulate + atomic_ok, no
 * epilogue extras) are batched into single launches of up to 8; every other member is run through klab_gemm. */
int klab_gemm_grouped(const klab_gemm_args* list, int n, void* stream);

/* ---- T5 RMS-norm (T5LayerNorm, HF/t5:59-72) ------------------------------------------------
 * y = drop(x * rsqrt(mean(x^2)+eps) * w); x is the f32 residual stream [rows,d]; y (dtype y_dtype)
 * and/or y_f32 are written at row (row/grp)*grp_stride + row%grp + off when grp>0 (this is how the
 * frozen language encoder writes its output straight into the concatenated encoder input,
 * ref/models/model.py:23).  rstd[rows] is saved for backward (may be NULL).                     */
int klab_rmsnorm_fwd(const float* x, const float* w, void* y, int y_dtype, float* y_f32, float* rstd, int rows, int d,
                     float eps, int grp, int grp_stride, int off, float drop_p, const uint32_t* seed_dev,
                     uint32_t tag, void* stream);
/* backward: dx = dres + rmsnorm'(dy * dropmask_y); dw += ...; dxt = dtype(dx * dropmask_prev)
 * (dxt is the gradient of the previous sub-layer's GEMM output: residual add + dropout of
 * HF/t5:400,141).  dy is indexed through the same row remap as y.                              */
/* fp8 mode: the norm's bf16 output AND the same rows in OCP e4m3 (y8 [rows, d]) with one dequantisation scale per row --
 * klab_quant_fp8_rows folded into the kernel that owns the row; bit-identical to norm followed by that pass.  d <= 1024.        */
int klab_rmsnorm_fwd_q8(const float* x, const float* w, void* y_bf16, float* rstd, void* y8, float* yscale, int rows, int d, float eps,
                        float drop_p, const uint32_t* seed_dev, uint32_t tag, void* stream);
int klab_layernorm_fwd_q8(const void* y_bf16, const float* gamma, const float* beta, const float* shortcut, float* out, void* outt_bf16,
                          float* mean, float* rstd, void* o8, float* oscale, int rows, int C, float eps, void* stream);
int klab_gelu_fwd_q8(const void* x_bf16, void* y_bf16, void* y8, float* yscale, int rows, int F, void* stream); /* F <= 4096, F % 8 == 0 */
int klab_rmsnorm_bwd(const float* dy, const float* x, const float* w, const float* rstd, const float* dres, float* dx,
                     void* dxt, int dxt_dtype, float* dw, int rows, int d, int grp, int grp_stride, int off,
                     float p_y, uint32_t tag_y, float p_prev, uint32_t tag_prev, const uint32_t* seed_dev,
                     void* stream);
/* Deferred weight gradient: same as klab_rmsnorm_bwd, but dw is left as per-workgroup partial sums
 * dw_part[klab_rmsnorm_part_rows(rows), d]; several calls' partials (call c at part + c*call_stride) are then folded into
 * their dw vectors by ONE klab_colpart_reduce (fixed summation order: bit-reproducible, and no same-address atomics). */
int klab_rmsnorm_part_rows(int rows);
int klab_rmsnorm_bwd_part(const float* dy, const float* x, const float* w, const float* rstd, const float* dres, float* dx, void* dxt,
                          int dxt_dtype, float* dw_part, int rows, int d, int grp, int grp_stride, int off, float p_y, uint32_t tag_y,
                          float p_prev, uint32_t tag_prev, const uint32_t* seed_dev, void* stream);
int klab_colpart_reduce(const float* part, long call_stride, int nparts, int d, float* const* dst_dev, int ncalls, void* stream);

/* ---- Swin-V2 LayerNorm, res-post-norm form (HF/swinv2:697-702; also :242, :354, :953) --------
 * out = drop(shortcut + LN(y)*gamma + beta)  (f32, remapped rows) and/or outt (dtype copy).     */
int klab_layernorm_fwd(const void* y, int y_dtype, const float* gamma, const float* beta, const float* shortcut,
                       float* out, void* outt, int outt_dtype, float* mean, float* rstd, int rows, int C, float eps,
                       int grp, int grp_stride, int off, float drop_p, const uint32_t* seed_dev, uint32_t tag,
                       void* stream);
int klab_layernorm_bwd(const float* dout, const void* y, int y_dtype, const float* gamma, const float* mean,
                       const float* rstd, void* dy, float* dgamma, float* dbeta, int rows, int C, int grp,
                       int grp_stride, int off, float drop_p, const uint32_t* seed_dev, uint32_t tag, void* stream);
/* the same, and dprev_bias[c] += sum_rows dy[:, c] -- the bias gradient of the Linear whose output the norm consumed (its input
 * gradient is dy) -- in the same pass (C <= 1024: one fused kernel for dy, dgamma, dbeta and dprev_bias)                      */
int klab_layernorm_bwd_bias(const float* dout, const void* y, int y_dtype, const float* gamma, const float* mean,
                            const float* rstd, void* dy, float* dgamma, float* dbeta, float* dprev_bias, int rows, int C, int grp,
                            int grp_stride, int off, float drop_p, const uint32_t* seed_dev, uint32_t tag, void* stream);

/* ---- T5 attention core (HF/t5:144-173 as called from :281-369) ------------------------------
 * S = Q K^T (unscaled, HF/t5:196-197) + bias[h,Lq,Lk] (+ causal); P = softmax(S); ctx = drop(P) V.
 * q/k/v/ctx/dq/dk/dv are [B*L, ld] matrices in `dtype`, head h at column h*dk (no transposes).
 * lse[B,H,Lq] (f32) is written by fwd and read by bwd; dbias[h,Lq,Lk] (f32) is ACCUMULATED by
 * bwd (position-bias gradient, shared by all layers: HF/t5:739-742) and may be NULL.            */
typedef struct klab_attn_args {
  int dtype;
  const void* q; long ldq;
  const void* k; long ldk;
  const void* v; long ldv;
  const float* bias; int causal;
  void* ctx; long ldo;
  float* lse;
  int B, H, Lq, Lk, dk;
  float drop_p; const uint32_t* seed_dev; uint32_t drop_tag;
  /* backward only */
  const void* dctx; long lddo;
  void* dq; long lddq;
  void* dk_out; long lddk;
  void* dv; long lddv;
  float* dbias;
  void* ds_ws; /* optional scratch [B,H,Lq,roundup(Lk,32)] in `dtype`: with it the position-bias gradient is a
                  deterministic batch reduction of stored dS instead of 64-way contended float atomics */
  int ds_defer; /* 1: only store dS; the caller reduces several layers' scratch at once with klab_dbias_reduce */
  /* backward, bf16 MFMA path only (used for the Swin-V2 window attention on window-ordered copies): */
  const float* score_scale; /* [H] device: S = score_scale[h] * Q K^T + bias (fp32), NULL = 1 */
  int bias_mod;             /* > 0: bias is [bias_mod, H, Lq, Lk] and batch element b uses slab b % bias_mod */
} klab_attn_args;
int klab_t5_attn_fwd(const klab_attn_args* a, void* stream);
int klab_t5_attn_bwd(const klab_attn_args* a, void* stream);
/* Front half of a T5 attention sub-layer in ONE launch (forward): T5LayerNorm (HF/t5:59-72) -> q|k|v projection (self, HF/t5:206-209)
 * or q projection (cross) -> attention core (HF/t5:144-173), one workgroup per (sample, head).  Replaces klab_rmsnorm_fwd + the
 * projection klab_gemm + klab_t5_attn_fwd of the serial chain; everything the backward pass reads is still written: xn (the
 * normalised rows, bf16) and rstd, the projected q|k|v (self: [B*Lq, 3*inner] column blocks q | k | v; cross: q [B*Lq, inner]),
 * attn.lse and attn.ctx.  attn.q / .k / .v are ignored for self attention; cross attention reads attn.k / attn.v (the projected
 * encoder output).  Envelope: bf16, d_model = 512, head dim 64, Lq <= 64, Lk <= 64 (self: Lk == Lq); otherwise
 * KLAB_ERR_UNSUPPORTED and the caller issues the three launches.                                                              */
typedef struct klab_attn_fused_args {
  const float* x; const float* gamma; float eps; int d_model;
  const void* w;            /* bf16 projection rows: self q|k|v [3*inner, d_model]; cross q [inner, d_model] */
  void* xn; float* rstd;    /* [B*Lq, d_model] bf16, [B*Lq] */
  void* proj; long ldproj;
  int cross;
  klab_attn_args attn;
} klab_attn_fused_args;
int klab_t5_attn_fused_fwd(const klab_attn_fused_args* a, void* stream);
/* dbias[H,Lq,Lk] += sum over nbatch slabs of ds_ws [nbatch, H, Lq, roundup(Lk,32)] (bf16), in a fixed order */
int klab_dbias_reduce(const void* ds_ws, int dtype, float* dbias, int nbatch, int H, int Lq, int Lk, void* stream);

/* ---- Swin-V2 shifted-window cosine attention (HF/swinv2:389-455 + roll/partition/mask/reverse of
 * :652-705, :146-166, :620-643) -- one call per block.  qkv [B*R*R, 3C], ctx [B*R*R, C] (`dtype`),
 * bias [H, w*w, w*w] f32 (= 16*sigmoid(CPB), klab_swin_cpb_bias), logit_scale [H] f32 (raw param),
 * lse [B*nW*H*w*w] f32 (fwd out, optional; bwd in).  bwd writes dqkv and accumulates dbias /
 * dlogit_scale.  R % w must be 0 (the padded-window path is out of scope).
 * bwd_ws / bwd_ws_bytes (optional): scratch of at least klab_swin_attn_bwd_ws_bytes(...) bytes; with it (bf16, head dim 32,
 * w*w <= 64) the backward runs on the matrix cores -- window-ordered unit q|k|v copies, the T5 attention backward kernel with a
 * per-head score scale and the per-window bias+mask table, then the L2-normalisation Jacobian on the way back to token
 * order.  Without it (or outside that envelope) the vector-ALU kernel runs.                       */
typedef struct klab_swin_attn_args {
  int dtype;
  const void* qkv; void* ctx; const float* bias; const float* logit_scale; float* lse;
  int B, R, w, shift, H, C;
  const void* dctx; void* dqkv; float* dbias; float* dlogit_scale;
  void* bwd_ws; size_t bwd_ws_bytes;
  /* Windows of more than 64 tokens (384 px / window 24: n = 576), or any window when bias == NULL: the position bias is
   * looked up per score in bias_table [(2w-1)^2, H] = 16*sigmoid(CPB MLP) (klab_swin_cpb_table) instead of a dense
   * [H, n, n] tensor (21 MB per block at n = 576, H = 16); backward accumulates d(bias_table) into dbias_table
   * [(2w-1)^2, H] (zeroed by the caller).  These shapes run tiled kernels (keys streamed in blocks of 64, LDS use
   * independent of n); lse / dlogit_scale / dqkv as above.                                                         */
  const float* bias_table; float* dbias_table;
  /* Window padding (R % w != 0, HF/swinv2:645-650, 688-690): the token grid is padded to ceil(R/w)*w with zero input rows that
   * still act as keys -- k = 0, v = v_bias [C] (the value Linear's bias, NULL = 0) -- and are cropped from the output; their
   * d v is added to dv_bias [C] (optional).  Padded shapes always run the tiled / streaming kernels; lse then has
   * B * ceil(R/w)^2 * H * w*w entries.                                                                              */
  const float* v_bias; float* dv_bias;
} klab_swin_attn_args;
int klab_swin_attn_fwd(const klab_swin_attn_args* a, void* stream);
/* scratch bytes the matrix-core backward needs for this shape (0: shape outside its envelope) */
size_t klab_swin_attn_bwd_ws_bytes(int dtype, int B, int R, int w, int H, int C);
/* Frozen tower (forward only): the q|k|v projection fused into the window attention, HF/swinv2:389-455 in one kernel.
 * x [B*R*R, C] (the block's LN'd input), wqkv [3C, C] rows q|k|v, bqkv [3C] f32 (k part zero) or NULL, ctx [B*R*R, C].
 * bf16, head dim 32, C in {64, 128, 256}, w*w <= 64; otherwise KLAB_ERR_UNSUPPORTED (caller: klab_gemm + klab_swin_attn_fwd). */
int klab_swin_qkv_attn_fused(const void* x, const void* wqkv, const float* bqkv, void* ctx, const float* bias, const float* logit_scale,
                             int dtype, int B, int R, int w, int shift, int H, int C, void* stream);
int klab_swin_attn_bwd(const klab_swin_attn_args* a, void* stream);
/* continuous position bias (HF/swinv2:376-378,418-428): coords [(2w-1)^2,2], index [n*n] are the
 * input-independent buffers of HF/swinv2:457-492; table [(2w-1)^2,H] and hidden [(2w-1)^2,512]
 * (optional, for backward) are scratch/outputs.                                                 */
int klab_swin_cpb_bias(const float* coords, const int* index, const float* w0, const float* b0, const float* w2,
                       float* table, float* hidden, float* bias, int ntab, int n, int heads, int nhidden, void* stream);
/* table form for large windows: table [(2w-1)^2, H] (raw MLP output), hidden (optional), bias_table = 16*sigmoid(table);
 * backward: d(bias_table) -> MLP weight gradients (accumulated: dw0 [512,2], db0 [512], dw2 [H,512]); dtable is scratch
 * [(2w-1)^2, H]; any table size, H <= 64.                                                                         */
int klab_swin_cpb_table(const float* coords, const float* w0, const float* b0, const float* w2, float* table, float* hidden,
                        float* bias_table, int ntab, int heads, int nhidden, void* stream);
int klab_swin_cpb_table_bwd(const float* dbias_table, const float* bias_table, const float* coords, const float* hidden, const float* w2,
                            float* dtable, float* dw0, float* db0, float* dw2, int ntab, int heads, int nhidden, void* stream);

/* ---- input pipeline ---------------------------------------------------------
 * Replaces, for a batch of decoded RGB images, `Image.resize((256,256))` + `ToTensor()` (ref/modules/loader.py:15-16; Pillow
 * ImagingResample, default BICUBIC) and `image_processor(images, return_tensors="pt")` (ref/train.py:55; ViTImageProcessor,
 * HF/vitproc:20-27: Pillow BILINEAR to 224x224, rescale, normalise).  Every intermediate uint8 image is bit-identical to
 * Pillow's (22-bit fixed-point weights from the same double-precision evaluation).
 *   src: device bytes holding the images as HWC uint8 RGB; desc_dev[i] = {byte offset into src, height, width} (device).
 *   filter_a / filter_b: PIL resampling ids of the two resizes (2 = BILINEAR, 3 = BICUBIC); filter_a = 0: src already is
 *   the loader's [n, mid, mid, 3] uint8 batch (desc_dev / ws unused) and only the processor's part runs.
 *   pixel_values [n, 3, out, out] f32 = (u8 * rescale - mean[c]) / std[c]; the reference's effective rescale is 1/255/255
 *   (its processor divides the ToTensor output by 255 a second time).  mean3 / std3 are host pointers to 3 floats.
 *   KLAB_ERR_UNSUPPORTED: other filters, or a size ratio whose weight table exceeds the LDS budget.                        */
typedef struct klab_image_desc { long long offset; int height, width; } klab_image_desc;
size_t klab_image_preprocess_ws_bytes(int n_images, int max_h, int mid);
int klab_image_preprocess(const unsigned char* src, const klab_image_desc* desc_dev, int n_images, int max_h, int max_w, int mid,
                          int out_size, int filter_a, int filter_b, double rescale, const float* mean3, const float* std3,
                          float* pixel_values, void* ws, size_t ws_bytes, void* stream);

/* ---- JPEG decoding (row f-1: `Image.open(path).convert('RGB')`, ref/modules/loader.py:15) -------------------------------
 * Hybrid decoder for baseline / extended-sequential / progressive 8-bit Huffman JPEG (SOF0 / SOF1 / SOF2; grey, or three
 * components with 4:4:4, 4:2:2 or 4:2:0 sampling; restart intervals; sequential files: one interleaved scan).  The serial part -- marker parsing and Huffman
 * decoding -- runs on the host (`klab_jpeg_entropy_decode*`, threaded over the images of a batch) and yields quantised DCT
 * coefficients; dequantisation, the inverse DCT, chroma upsampling and the colour transform run on the GPU
 * (`klab_jpeg_decode_device`) and write HWC uint8 RGB at the offsets `klab_image_preprocess` reads.  Output is bit-identical to
 * Pillow's decode (libjpeg-turbo defaults: integer "islow" IDCT, triangle-filter upsampling, 16-bit colour tables).
 * Arithmetic-coded / lossless / 12-bit / CMYK / multi-scan sequential files: KLAB_ERR_UNSUPPORTED (`supported` = 0 in klab_jpeg_info).
 *   coefs:  int16, 64 per block in natural (row-major) order; per image the blocks of component 0, then 1, then 2, each as
 *           [bh][bw] over the MCU-padded block grid; qt: uint16 [3][64] per image (natural order, one table per component).   */
#define KLAB_JPEG_GRAY 0
#define KLAB_JPEG_YCC 1
#define KLAB_JPEG_RGB 2
typedef struct klab_jpeg_info {
  int width, height, ncomp, precision, progressive, supported, colour;
  int hmax, vmax, mcus_x, mcus_y;
  int hs[3], vs[3], bw[3], bh[3], tq[3];
  long long coef_blocks; /* sum of bw*bh over the components */
} klab_jpeg_info;
typedef struct klab_jpeg_item { klab_jpeg_info info; long long coef_block0; long long rgb_off; } klab_jpeg_item;
int klab_jpeg_read_info(const unsigned char* data, size_t n, klab_jpeg_info* info);                       /* host, header only */
int klab_jpeg_entropy_decode(const unsigned char* data, size_t n, short* coefs, unsigned short* qt, klab_jpeg_info* info); /* host */
/* host, n_threads workers; coefs[i] -> room for infos[i].coef_blocks*64 shorts; qt: [n][3][64]; rcs[i]: per-image status      */
int klab_jpeg_entropy_decode_batch(const unsigned char* const* data, const size_t* sizes, int n, short* const* coefs,
                                   unsigned short* qt, klab_jpeg_info* infos, int* rcs, int n_threads);
/* device: items (host copy, validated and used to size the grids) / items_dev (the same bytes on the device): per image the
 * header info, the first coefficient block in coefs_dev and the byte offset of its RGB image in rgb_dev                        */
size_t klab_jpeg_decode_ws_bytes(const klab_jpeg_item* items, int n);
int klab_jpeg_decode_device(const short* coefs_dev, const unsigned short* qt_dev, const klab_jpeg_item* items,
                            const klab_jpeg_item* items_dev, int n, unsigned char* rgb_dev, void* ws, size_t ws_bytes, void* stream);

/* ---- glue --------------------------------------------------------------------------------- */
/* multi-tensor f32 -> dtype cast into one arena; desc_dev: device array of
 * {const float* src; long dst_off; long n4_prefix} (prefix sums of element counts / 4)           */
int klab_cast_pack(const void* desc_dev, int ndesc, long total4, void* dst, int dtype, void* stream);
/* embedding gather with optional T5 _shift_right (HF/t5:618-637) and input dropout (HF/t5:725)  */
int klab_embed_fwd(const long long* ids, int shift_right, int L, int start_id, int pad_id, const float* table, int vocab,
                   float* out, int rows, int d, float drop_p, const uint32_t* seed_dev, uint32_t tag, int* err_flag,
                   void* stream);
int klab_embed_bwd(const long long* ids, int shift_right, int L, int start_id, int pad_id, const float* dh, float* dtable,
                   int vocab, int rows, int d, float drop_p, const uint32_t* seed_dev, uint32_t tag, void* stream);
/* relative position bias gather / scatter (HF/t5:264-279); bucket[Lq*Lk] int32 from the host     */
int klab_relbias_fwd(const float* table, const int* bucket, float* bias, int heads, int Lq, int Lk, void* stream);
int klab_relbias_bwd(const float* dbias, const int* bucket, float* dtable, int heads, int Lq, int Lk, int nbuckets,
                     void* stream);
/* cross-entropy over logits [rows, V] (HF/t5:1050-1054); with write_grad (bit 0) the logits are replaced in
 * place by d loss / d logits = (softmax - onehot) / n_valid.  loss_row [rows], inv_n [1], loss [1].
 * write_grad bit 1: inv_n was already produced by klab_ce_count for the same labels (saves the counting launch). */
int klab_ce_count(const long long* labels, int rows, float* inv_n, void* stream);
int klab_ce_fwd(void* logits, long ld, int dtype, const long long* labels, int rows, int V, float* inv_n, float* loss_row,
                float* loss, int write_grad, void* stream);
int klab_im2col_patch(const float* pixels, void* out, int dtype, int B, int Cin, int Himg, int P, void* stream);
/* same, rows of `ldo` >= Cin*P*P elements with the tail columns zero-filled (lets the patch-embedding GEMM run with K
 * padded to a multiple of 32); P == 4 takes a one-thread-per-patch vector path */
int klab_im2col_patch_ld(const float* pixels, void* out, int dtype, int B, int Cin, int Himg, int P, int ldo, void* stream);
int klab_merge_gather(const float* x, void* out, int dtype, int B, int R, int C, void* stream);
int klab_merge_scatter(const float* dmerged, float* dx, int B, int R, int C, void* stream);
int klab_colsum(const void* dy, long ld, int dtype, int M, int N, float* out, void* stream);
int klab_convert(const float* x, void* y, int dtype, long n, float scale, void* stream);
int klab_add_f32(float* y, const float* x, long n, void* stream);

int klab_gelu_fwd(const void* x, void* y, int dtype, long n, void* stream);
/* Frozen-tower (forward-only) fusion of the MLP half of a Swin-V2 block, HF/swinv2:539-563 + 697-702:
 *   out[M,C] = shortcut + LayerNorm(fc2(GELU(fc1(x) + b1)) + b2) * gamma + beta,  outt = bf16(out) (optional)
 * x [M,C], w1 [4C,C], w2 [C,4C] in `dtype` (bf16 only), everything else f32.  C in {64, 128}; other widths return
 * KLAB_ERR_UNSUPPORTED and the caller keeps the three-kernel path (klab_gemm x2 + klab_layernorm_fwd). */
/* ... and of the attention half's tail, HF/swinv2:496-506 + 697-700: out = shortcut + LayerNorm(x Wp^T + bp) * gamma + beta */
int klab_swin_proj_ln_fused(const void* x, const float* shortcut, const void* w, const float* b, const float* gamma, const float* beta,
                            float* out, void* outt, int dtype, int M, int C, float eps, void* stream);
int klab_swin_mlp_fused(const void* x, const float* shortcut, const void* w1, const float* b1, const void* w2, const float* b2,
                        const float* gamma, const float* beta, float* out, void* outt, int dtype, int M, int C, float eps,
                        void* stream);
/* Frozen-tower fusion of the patch embedding, HF/swinv2:234-259, 281, 293-302: out = LayerNorm(Conv2d(3 -> C, k 4, s 4)(pixels)) in one
 * launch (no column matrix, no stored GEMM output).  w: bf16 [C, ldw >= 64], columns 48..63 zero (the Conv2d weight [C,3,4,4] flattened
 * and zero-padded); out [B*(image_size/4)^2, C] f32, outt the same in bf16 (optional).  bf16, patch 4, 3 channels, C in {64, 96, 128};
 * otherwise KLAB_ERR_UNSUPPORTED (caller: klab_im2col_patch_ld + klab_gemm + klab_layernorm_fwd).                                 */
int klab_swin_patch_embed_fused(const float* pixels, const void* w, int ldw, const float* bias, const float* gamma, const float* beta,
                                float* out, void* outt, int dtype, int B, int in_ch, int image_size, int patch, int C, float eps, void* stream);
/* Frozen-tower fusion for the WIDE stages, HF/swinv2:496-506 / 555-563 + 697-702: out = shortcut + LayerNorm(x W^T + b) * gamma + beta in
 * one launch, for the attention output projection (K = C) and the MLP's second Linear (K = 4C).  x [M, K], w [C, K] in `dtype` (bf16
 * only), the rest f32; out [M, C] f32, outt its bf16 copy (optional).  C == 256, K % 32 == 0, K >= 128; otherwise
 * KLAB_ERR_UNSUPPORTED (caller: klab_gemm + klab_layernorm_fwd).                                                                    */
int klab_swin_linear_ln_fused(const void* x, const float* shortcut, const void* w, const float* bias, const float* gamma,
                              const float* beta, float* out, void* outt, int dtype, int M, int K, int C, float eps, void* stream);
int klab_swin_cpb_bias_bwd(const float* dbias, const float* bias, const int* index, const float* coords, const float* hidden,
                           const float* w0, const float* w2, float* dtable, float* dw0, float* db0, float* dw2, int ntab, int n,
                           int heads, int nhidden, void* stream);
/* the same with `dtable_zeroed`: non-zero = the caller has cleared dtable[ntab * heads] (the engine clears every block's slice in one fill) */
int klab_swin_cpb_bias_bwd_pz(const float* dbias, const float* bias, const int* index, const float* coords, const float* hidden,
                              const float* w0, const float* w2, float* dtable, float* dw0, float* db0, float* dw2, int ntab, int n,
                              int heads, int nhidden, int dtable_zeroed, void* stream);

/* ==== the whole path: MyModel.forward + backward (ref/models/model.py:19-26, ref/train.py:58-62) ====
 * The engine is a host-side plan over the kernels above.  It owns no device memory: parameters are
 * the caller's fp32 tensors (named exactly like the HuggingFace state dict), activations
 * live in one caller-provided workspace, gradients are written into caller-provided flat f32 buffers
 * at the offsets published by klab_engine_param_info.  model index: 0 = Swin-V2 image_model,
 * 1 = frozen language_model (T5 encoder), 2 = transformer (T5ForConditionalGeneration).            */
typedef struct klab_swin_cfg {  /* HF/swinv2cfg:56-73 */
  int image_size, patch, in_ch, embed_dim, n_stages;
  int depths[8], heads[8];
  int window;
  int pretrained_window[8];
  int mlp_ratio, qkv_bias;
  float ln_eps;
} klab_swin_cfg;
typedef struct klab_t5_cfg {    /* HF/t5cfg:44-62,82-83 */
  int vocab, d_model, d_kv, n_heads, d_ff, n_layers, n_dec_layers, rel_buckets, rel_max_dist;
  float dropout, ln_eps;
  int start_id, pad_id, scale_decoder_outputs;
} klab_t5_cfg;
typedef struct klab_model_cfg {
  klab_swin_cfg swin;
  klab_t5_cfg lang, main;
  int dtype;      /* KLAB_F32 parity mode | KLAB_BF16 | KLAB_FP8 (bf16 + fp8 forward GEMMs with per-row scales) */
  int train_swin; /* args.image_model_train (ref/models/model.py:15) */
} klab_model_cfg;
typedef struct klab_engine klab_engine;

klab_engine* klab_engine_create(const klab_model_cfg* cfg); /* NULL on an invalid config (e.g. Swin width != d_model:
                                                               the reference raises at its torch.cat, model.py:23) */
void klab_engine_destroy(klab_engine* e);
int klab_engine_num_params(const klab_engine* e, int model);
int klab_engine_param_info(const klab_engine* e, int model, int i, char* name, int name_cap, long* shape4, int* ndim,
                           long* grad_off /* element offset in the model's flat grad buffer, -1 = frozen */);
long klab_engine_grad_elems(const klab_engine* e, int model);
/* backward segment -> (model, offset, length) of the flat-grad slice that is final when it returns */
/* Gradient buckets INSIDE a backward segment (data parallelism, ref/train.py:26,62: DDP reduces buckets as they become
 * ready).  Bucket i of a segment = the GEMM-weight gradients of one T5 layer (segments 0, 1) or one Swin block (segment 2),
 * a contiguous range [off, off+len) of that segment's flat gradient buffer; i counts in the order the backward finishes
 * them (last layer first).  klab_engine_bucket_wait makes `stream` wait until bucket i of the LAST klab_engine_backward of
 * that segment is final (an engine-owned event behind the layer's weight-gradient launch); KLAB_ERR_UNSUPPORTED under graph
 * replay or before any backward.  Everything of the segment outside its buckets is final when klab_engine_backward returns
 * (on the stream passed to it).                                                                                           */
int klab_engine_num_buckets(const klab_engine* e, int segment);
int klab_engine_bucket(const klab_engine* e, int segment, int i, long* off, long* len);
int klab_engine_bucket_wait(klab_engine* e, int segment, int i, void* stream);
/* the per-bucket events are recorded only after klab_engine_set_bucket_events(e, 1) (a data-parallel reducer is attached);
 * off by default: a single-GPU step pays nothing for them                                                            */
int klab_engine_set_bucket_events(klab_engine* e, int on);
int klab_engine_segment(const klab_engine* e, int seg, int* model, long* off, long* len);
size_t klab_engine_workspace_bytes(klab_engine* e, int B, int Ls, int Lt);
/* *_params: host arrays of device pointers in klab_engine_param_info order.  *_bucket: int32 device
 * arrays [L*L] of T5 relative-position buckets (HF/t5:216-262, computed by the host exactly as the
 * reference does).  swin_coords[s] / swin_index[s]: per-stage CPB tables (HF/swinv2:457-492).      */
int klab_engine_bind(klab_engine* e, int B, int Ls, int Lt, void* workspace, size_t ws_bytes, const void* const* swin_params,
                     const void* const* lang_params, const void* const* main_params, float* main_grads, float* swin_grads,
                     const int* lang_bucket, const int* enc_bucket, const int* dec_bucket, const void* const* swin_coords,
                     const void* const* swin_index, void* stream);
/* pixels [B,3,H,W] f32, src_ids [B,Ls] / tgt_ids [B,Lt] int64 (device).  training: T5 dropout on
 * (model.module.transformer.train(), ref/train.py:52).  `seed` (re)bases the device-side counter RNG whenever
 * it changes; every forward advances the counter itself.  The loss lands in *klab_engine_loss_ptr.      */
int klab_engine_forward(klab_engine* e, const float* pixels, const long long* src_ids, const long long* tgt_ids, int training,
                        uint32_t seed, int want_grad, void* stream);
/* hipGraph replay of the forward / backward launch sequences (first use eager, second captured, then replayed;
 * inputs are staged into engine-owned buffers so node addresses stay fixed).  Off by default.          */
int klab_engine_set_graph(klab_engine* e, int on);
/* dropout RNG state of the binding (base seed, forwards since seeding): saved / restored by a true resume so that the run
 * continues its mask stream.  get synchronises `stream`.                                                                  */
int klab_engine_get_rng(klab_engine* e, uint32_t* base, uint32_t* counter, void* stream);
int klab_engine_set_rng(klab_engine* e, uint32_t base, uint32_t counter, void* stream);
/* torch.optim.Adam's update (ref/train.py:28; no amsgrad, L2 weight decay) for ALL parameters of the
 * trainable T5 in one pass.  m / v: caller-owned f32 state laid out like the "main" flat gradient buffer.  bias_corr{1,2} =
 * 1 - beta^step.  Also refreshes the compute-dtype copies of the GEMM weights, so the next klab_engine_forward may be
 * told (training bit 2) that they are current. */
int klab_engine_adam_step(klab_engine* e, float* m, float* v, float lr, float beta1, float beta2, float eps, float weight_decay,
                          float bias_corr1, float bias_corr2, void* stream);
/* the kernel behind it: desc = device array of {float* p; long grad_off; long arena_off (<0: none); long n4_prefix} */
int klab_adam_step(const void* desc_dev, int ndesc, long total4, const float* grads, float* m, float* v, void* arena, int dtype, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, void* stream);
/* the same update restricted to the tensors of ONE backward segment (0: decoder + tied embedding, 1: encoder): under data
 * parallelism segment 0 can be updated while segment 1's gradient all-reduce is still in flight.  Both segments = one full step. */
int klab_engine_adam_step_segment(klab_engine* e, int segment, float* m, float* v, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, float bias_corr1, float bias_corr2, void* stream);
/* vec4 range [begin4, end4) of the descriptor table's prefix space */
int klab_adam_step_range(const void* desc_dev, int ndesc, long begin4, long end4, const float* grads, float* m, float* v, void* arena,
                         int dtype, float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                         void* stream);
/* Greedy decoding with a K/V cache (ref/models/model.py:27-28, HF/t5:308-332): the decoder over ONE new position t >= 1 per
 * sample.  Precondition: a klab_engine_forward in evaluation mode on this binding (prefill: encoder, cross K/V, position 0),
 * then steps 1, 2, ... in order; the per-layer q|k|v buffers of the binding are the cache.  prev_tokens [B] (device) = ids
 * generated at position t-1.  Logits of position t: klab_engine_buffer("logits_step") [B, vocab].                        */
int klab_engine_decode_step(klab_engine* e, int t, const long long* prev_tokens, void* stream);
/* one query row per (batch, head) against cached keys / values (element strides; bias_row [H, bias_ld] or NULL) */
int klab_t5_decode_attn(int dtype, const void* q, long q_bstride, const void* k, const void* v, long kv_bstride, long ldk,
                        const float* bias_row, long bias_ld, void* ctx, long ctx_bstride, int B, int H, int Lk, int dk, void* stream);
/* segment 0: LM head + decoder + tied embedding; 1: encoder; 2: Swin (no-op unless train_swin).
 * dloss_dev: device scalar d(objective)/d(loss) (NULL = 1).                                       */
int klab_engine_backward(klab_engine* e, int segment, const float* dloss_dev, void* stream);
/* timing probes: when enabled, selected launches are bracketed by two HIP events ON THE STREAM THEY ARE LAUNCHED ON; read back
 * after a synchronize.  channel 0: the LM-head logits GEMM of every forward (symbol klab_lmhead_gemm, compute stream);
 * channel 1: every grouped weight-gradient launch of the T5 backward (symbol gemm_glds_grouped_tn_kernel, the engine's side
 * stream).  probe_read returns the launch count, the summed duration and the summed algorithmic FLOPs (2*M*N*K) of a channel. */
int klab_engine_probe_enable(klab_engine* e, int on);
int klab_engine_probe_read(klab_engine* e, int channel, int* launches, float* total_ms, double* flops_total);
const float* klab_engine_loss_ptr(const klab_engine* e);
/* One-shot: the NEXT klab_engine_forward writes its mean loss to `out` (a device float owned by the caller) instead of the engine's
 * slot -- a caller that hands out a fresh tensor per step (MyModel.forward, ref/models/model.py:26) then needs no copy launch behind
 * the cross-entropy.  Returns 1 if it will be honoured, 0 under graph replay (the loss stays at klab_engine_loss_ptr), < 0 on error. */
int klab_engine_set_loss_out(klab_engine* e, float* out);
const int* klab_engine_err_ptr(const klab_engine* e);
/* device words {seed of the current step, base seed, forwards since seeding} (for stream-ordered snapshots; see klab_engine_get_rng) */
const uint32_t* klab_engine_rng_ptr(const klab_engine* e);
const void* klab_engine_buffer(const klab_engine* e, const char* name, long* rows, long* cols, int* dtype);

/* ---- crash diagnostics (opt-in; csrc/diag.cpp) -------------------------------------------------------------------------
 * klab_segv_trace_install: on SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL write the context string and the native backtrace
 * of the faulting thread to file descriptor fd (< 0: stderr), then re-raise with the default action.
 * klab_segv_set_context: the string to print (the test harness passes the running pytest node id).                       */
int klab_segv_trace_install(int fd);
int klab_segv_set_context(const char* text);

#ifdef __cplusplus
}
#endif
#endif /* KLAB_MM_H */

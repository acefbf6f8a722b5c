/*
 * mvp_hip.h — C ABI of libmvp_hip.so: the MI355X (gfx950) kernels behind the
 * midvision-probe feature-extraction + probe-training hot path.
 *
 * Conventions (SURVEY.md §8b "Lower" boundary):
 *   - every entry point is  extern "C" int mvp_<op>(const mvp_<op>_args*, void* hip_stream)
 *     returning 0 on success or a negative MVP_E* code; nothing throws, nothing exits;
 *   - plain pointers and sizes only (no torch types); all pointers are DEVICE pointers
 *     unless a field says "host";
 *   - the caller allocates every buffer, including workspaces (sizes documented per op);
 *   - kernels are enqueued asynchronously on the stream passed in; no hidden syncs, no
 *     allocation, no global mutable state  => safe under hipGraph capture, re-entrant
 *     across streams and devices;
 *   - "bf16 pair" = two uint16 bf16 arrays (hi, lo) with value = hi + lo.  With
 *     precision MVP_PREC_BF16 only hi is read/written (lo pointers may be NULL); with
 *     MVP_PREC_BF16X3 contractions run as hi*hi + hi*lo + lo*hi on the bf16 MFMA pipe
 *     with fp32 accumulation (~2^-16 relative operand error, needed for the reference's
 *     1e-3 feature-parity bar).
 *
 * Each op cites the reference interface (file:line under the upstream repo) it replaces.
 */
#ifndef MVP_HIP_H
#define MVP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVP_OK 0
#define MVP_EINVAL (-1)   /* bad argument (shape / alignment / NULL) */
#define MVP_ELAUNCH (-2)  /* hipLaunch failed */
#define MVP_ENODEV (-3)   /* no gfx950 device */

#define MVP_PREC_BF16 1   /* one bf16 MFMA pass                              */
#define MVP_PREC_BF16X3 3 /* three passes (split bf16), ~fp32 operand accuracy */
#define MVP_PREC_F16X2 2  /* (ABI 6, opt-in) TWO fp16 passes per contraction of mvp_gemm_bias_act_res / mvp_gemm_pp (plain linear GEMMs only), with the
                             weight's fp16 rounding error carried by the second pass ("compensated" pairs).  With s = 2^-6:
                               activation  a_hi = fp16(a),            a_lo = fp16(8 (a - a_hi) + a_hi / 8)       (LayerNorm / attention out_f16 = 1,
                                                                                                                  a GEMM epilogue's out_f16_col0 = -1)
                               weight      w_hi = fp16((1 - s) w),    w_lo = fp16((w + d / s) / 8),  d = (1 - s) w - w_hi    (built once, frozen)
                             and the product runs as  a_lo . w_lo + a_hi . w_hi  (two f16 MFMAs, fp32 accumulate).  Exactly:
                               a_hi w_hi + a_lo w_lo = a_hi ((1 - s) w - d) + ((a - a_hi) + s a_hi) (w + d / s) = a w + (a - a_hi) d / s,
                             i.e. the s a_hi w that the first pass leaves out and its rounding error d both ride in the second; what is left is
                             (a - a_hi) d / s <= 2^-12 * 2^-6 |a w| plus the fp16 roundings of a_lo and w_lo (2^-18 each) — the accuracy class of
                             BF16X3 (tests/test_gpu_kernels.py measures it against fp64, with the ViT goldens' feature error) at 2/3 of its
                             matrix-pipe work, on a chip whose clock is held down by exactly that work (the large-M kernel: -19 % time).
                             Range: |a| <= 65504 (hi and lo saturate there; beyond it the result is wrong, not NaN); values whose lo half falls
                             below fp16's normal range (|a| < 5e-4, |w| < 5e-4) keep absolute, not relative, precision.                     */

#define MVP_ACT_NONE 0
#define MVP_ACT_GELU 1 /* exact erf GELU (torch nn.GELU default)            */
#define MVP_ACT_RELU 2

typedef uint16_t mvp_bf16;

/* Library / device introspection. Returns MVP_OK and fills the fields. */
typedef struct {
  int abi_version;      /* = MVP_ABI_VERSION                                 */
  int device_count;     /* visible HIP devices (0 on a CPU-only box)         */
  int gfx950;           /* 1 if device 0 is gfx950                           */
  int cu_count;         /* multiProcessorCount of device 0                   */
  char arch[64];        /* gcnArchName of device 0                           */
} mvp_info_t;
#define MVP_ABI_VERSION 7 /* 7: mvp_bn_running_update_n (a new export); mvp_gemm_args.out_f16_col0 < -1 and MVP_ATT_V_F16_QK_F16 (new values of existing fields); every struct as in 6
                             6: mvp_gemm_args.out_f16_col0, mvp_attention_args.v_format (both structs grew by one int at the end; zero = the ABI 5 behaviour)
                             5: mvp_upconv3_fwd_gather, mvp_upconv3_grad_boxsum; mvp_gemm_pp accepts conv; depth-loss workspace grew (query mvp_depth_loss_workspace_bytes) */
int mvp_get_info(mvp_info_t* out);
const char* mvp_strerror(int code);
int mvp_sizeof(const char* struct_name); /* sizeof(<struct_name>) as compiled into the library, -1 if unknown: for bindings to self-check */

/* ------------------------------------------------------------------------------------
 * Elementwise split:  fp32 [n] -> bf16 pair.  Used once per frozen-weight tensor.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* src;
  mvp_bf16* hi;
  mvp_bf16* lo; /* may be NULL */
  int64_t n;
} mvp_split_bf16_args;
int mvp_split_bf16(const mvp_split_bf16_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Patch gather (im2col) for the 16x16/16 patch-embed conv.
 * Replaces the data movement inside nn.Conv2d(3,768,16,16) of
 * evals/models/ibot_transformers.py:216-222 (PatchEmbed.proj) and center_padding
 * (evals/models/utils.py:55-72): zero padding (pad_top/pad_left) is applied on the fly.
 * out row (b*gh*gw + py*gw + px), col (c*P*P + ky*P + kx)  — matches weight.view(768,-1).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* images; /* [B,C,H,W] fp32 NCHW                                  */
  mvp_bf16* out_hi;    /* [B*gh*gw, C*P*P]                                      */
  mvp_bf16* out_lo;    /* or NULL                                              */
  int B, C, H, W;      /* un-padded image dims                                 */
  int P;               /* patch size (16); must be a multiple of 4             */
  int gh, gw;          /* patch grid after padding                             */
  int pad_top, pad_left;
} mvp_patch_gather_args;
int mvp_patch_gather(const mvp_patch_gather_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * GEMM with fused epilogue:   Y = act(A · Wᵀ + bias) + residual
 *   A  [M, K]  bf16 pair, row stride lda;   W  [N, K]  bf16 pair (torch Linear layout),
 *   row stride ldw;  K % 64 == 0 (K % 32 == 0 in bf16x3 mode, without split-K);  fp32 accumulation on the bf16 MFMA pipe.
 * Replaces nn.Linear / 1x1 conv call sites: ibot_transformers.py:124,143 (qkv, proj),
 * :95-106 (fc1+GELU+fc2), :216-222 (patch-embed as GEMM), probes.py:352-355,420-432.
 * Output rows can be remapped (patch-embed writes token rows behind the CLS slot):
 *   out_row(m) = (m / row_group) * row_group_stride + row_group_off + (m % row_group)
 * when row_group > 0; residual row = m % res_row_mod when res_row_mod > 0 (pos-embed),
 * else the (remapped) output row.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const mvp_bf16* a_hi; const mvp_bf16* a_lo;
  const mvp_bf16* w_hi; const mvp_bf16* w_lo;
  const float* bias;      /* [N] or NULL                                        */
  const float* residual;  /* fp32, row stride ldr, or NULL                      */
  float* out_f32;         /* row stride ldo, or NULL                            */
  mvp_bf16* out_hi;       /* row stride ldob, or NULL                           */
  mvp_bf16* out_lo;       /* or NULL                                            */
  int M, N, K;
  int lda, ldw, ldr, ldo, ldob;
  int act;                /* MVP_ACT_*                                          */
  int precision;          /* MVP_PREC_*                                         */
  int row_group, row_group_stride, row_group_off, res_row_mod;
  /* --- implicit-GEMM convolution (conv != 0): A = channels-last activation [B, cH>>cup, cW>>cup, cC]
   * read as the im2col of its (nearest-upsampled by 2^cup) [B, cH, cW, cC] view; K = ckh*ckw*cC with
   * k = (ky*ckw + kx)*cC + c (weights laid out [N, ckh, ckw, cC]); M = B*cHo*cWo; cC % 32 == 0.
   * Replaces nn.Conv2d 3x3 / strided convs of probes.py:283-306,318-375 and the ResNet trunk
   * (dino_res50.py:38-51); padding taps are read as out-of-range buffer offsets (zeros); the activation must
   * stay below 2 GiB; zero_page (>= 256 zero bytes) is still validated for ABI stability but no longer read.  */
  int conv, cH, cW, cC, cHo, cWo, ckh, ckw, cstride, cpad, cup;
  const mvp_bf16* zero_page;
  /* --- ReLU bookkeeping (byte masks, row stride ldm):
   * out_mask  (forward): 1 where the activated value (before any residual add) is > 0;
   * relu_mask (backward): gate by a saved mask; mask_mode 2 gates the result before the residual
   * add, mask_mode 1 gates only the bf16-pair output (out_f32 stays un-gated: skip-path gradient). */
  const uint8_t* relu_mask; uint8_t* out_mask; int ldm; int mask_mode;
  const float* residual2; /* optional second fp32 addend, same row stride ldr (fusion-block "+ skip") */
  int act_after_res;      /* 1: apply MVP_ACT_RELU after the residual adds (ResNet bottleneck)          */
  /* --- split-K (splitk > 1, plain linear GEMMs only: ignored with conv / mask / residual2 / act_after_res):
   * the K range is cut into `splitk` parts that run as separate workgroups; the last part to finish sums
   * the fp32 partial tiles in a fixed order (bit-reproducible) and runs the fused epilogue.  For GEMMs
   * with too few output tiles to fill 256 CUs (N = 768 projections, the probe head at M ~ 3k rows).
   * splitk_ws: >= mvp_gemm_splitk_workspace_bytes(M, N, splitk) bytes whose leading tile counters are ZERO
   * at first use (they reset themselves); not shared by GEMMs running concurrently on other streams.
   * splitk == MVP_GEMM_STREAMK (-1): stream-K scheduling instead (csrc/gemm_sk.hip): one workgroup per CU, every CU gets the
   * same number of k-iterations of 128x128x64 tiles whatever the tile count; same bit-reproducible fixed-order reduction of
   * partial tiles; splitk_ws >= mvp_gemm_streamk_workspace_bytes(), 16-byte aligned, counters ZERO at first use.        */
  int splitk; void* splitk_ws; int64_t splitk_ws_bytes;
  /* --- optional residual given as a bf16 pair (row stride ldr, elements), added like `residual`:
   * lets a frozen trunk keep block outputs only as pairs (ResNet identities, dino_res50.py:83-101). */
  const mvp_bf16* residual_hi; const mvp_bf16* residual_lo;
  /* --- tile policy of the plain (non-conv, non-split-K) bf16x3 GEMMs:
   * MVP_TILES_ALONE (0): the launch has the chip to itself (one serial kernel chain): tiles small enough that every CU holds
   *   several workgroups (64x64 for the N = 768 projections at M ~ 3k: 600 tiles on 256 CUs);
   * MVP_TILES_SHARED (1): other kernel chains run beside it on other streams (mvp/pipeline.py keeps >= 3 frozen forwards in
   *   flight): 128x128 tiles everywhere.  They need 31 % less SIMD time per output (an LDS-DMA piece costs its SIMD ~5 MFMAs, and a
   *   128x128 tile issues 6 MFMAs per piece against 3), and the CUs their coarse grid leaves idle are filled by the other chains. */
  int tile_policy;
  /* --- layout of the bf16-pair operands A and W (MVP_PREC_BF16X3 only):
   * MVP_PAIR_SEPARATE (0): a_hi / a_lo and w_hi / w_lo are separate K-contiguous arrays (what every producer of this library writes);
   * MVP_PAIR_A_ILV32 (1) / MVP_PAIR_W_ILV32 (2), or-ed: that operand is ONE array, hi | lo interleaved per 32-deep k block — row =
   *   [k / 32][hi 32 | lo 32] bf16, so a 32-deep k-step of a row is one whole 128-byte line; a_hi (w_hi) points at the array, lda
   *   (ldw) is its row stride in elements (2 * K when dense), a_lo (w_lo) is ignored.  Only the large-M kernel (mvp_gemm_pp) reads
   *   this layout; MVP_PAIR_ILV32 (3) = both operands.                                                                           */
  int pair_layout;
  /* --- layout of the bf16-pair OUTPUT (out_hi / out_lo): MVP_PAIR_SEPARATE, or MVP_PAIR_A_ILV32 (1): out_hi is ONE array
   * [rows][N / 32][hi 32 | lo 32] with row stride ldob (2 * N when dense), out_lo is ignored, N % 32 == 0 — the A operand of a following
   * large-M GEMM (fc1 -> fc2).  Any kernel of mvp_gemm_bias_act_res writes it.                                                        */
  int out_pair_layout;
  /* --- 16-bit forms other than the bf16 pair inside the pair output (ABI 6; MVP_PREC_BF16X3 / F16X2, out_hi and a lo half required).
   * out_f16_col0 = -1: EVERY column is written as the activation pair of MVP_PREC_F16X2 above (fc1 -> fc2).
   * out_f16_col0 > 0 (a multiple of 64): columns >= out_f16_col0 are written as hi = fp16(v) (round to nearest even), lo = bf16(v - hi) instead of
   * hi = bf16(v), lo = bf16(v - hi) — the V third of the fused qkv projection (out_f16_col0 = 2 * H * 64), which the attention
   * kernel multiplies with probabilities held as ONE fp16 value (mvp_attention_args.v_format).  Same 2 + 2 bytes, same arrays and
   * layouts; |v - hi - lo| <= 2^-20 |v|.  Every kernel of mvp_gemm_bias_act_res / mvp_gemm_pp writes it (not stream-K).
   * out_f16_col0 = -V0 < -1 (ABI 7; V0 a multiple of 128 = the first column of the V third, 2 * H * 64): the fused qkv projection for
   * MVP_ATT_V_F16_QK_F16 — columns [0, V0 / 2) (Q) as the compensated activation pair, [V0 / 2, V0) (K) as the compensated WEIGHT-side pair
   * (hi = fp16((1 - 2^-6) v), lo = fp16((v + 64 d) / 8), d = (1 - 2^-6) v - hi, fp32 arithmetic; hi + lo / 8 = v to ~2^-17), [V0, N) (V) as above. */
  int out_f16_col0;
} mvp_gemm_args;
#define MVP_TILES_ALONE 0
#define MVP_TILES_SHARED 1
#define MVP_TILES_NO_PP 2 /* flag, or-ed in: never dispatch to the large-M kernel mvp_gemm_pp (A/B measurements, tests of the tile kernels) */
#define MVP_TILES_NO_UNI 4 /* flag, or-ed in: keep the row-guarded epilogue where the universal branch-free one (gemm_epilogue_uni) would serve (A/B, bit-identity tests) */
#define MVP_GEMM_STREAMK (-1)
#define MVP_PAIR_SEPARATE 0
#define MVP_PAIR_A_ILV32 1
#define MVP_PAIR_W_ILV32 2
#define MVP_PAIR_ILV32 3
int mvp_gemm_bias_act_res(const mvp_gemm_args*, void* stream);
/* The large-M bf16x3 kernel (csrc/gemm_pp.hip): 256x256 output tiles, one 8-wave workgroup per CU, LDS-DMA prefetch kept in flight
 * across raw barriers with counted vmcnt, the two wave groups of a SIMD alternating between MFMA clusters and loads.  Same contract,
 * same epilogue and — same accumulation order — the same bits as the tile kernels; plain linear GEMMs only (no conv, no split-K),
 * K % 32 == 0, K >= 64.  mvp_gemm_bias_act_res dispatches to it by itself when M is large enough (see the rule in gemm.hip);
 * this entry point forces it (benchmarks, tests).  Replaces the same nn.Linear call sites (ibot_transformers.py:95-106,124-145). */
int mvp_gemm_pp(const mvp_gemm_args*, void* stream);
int64_t mvp_gemm_splitk_workspace_bytes(int M, int N, int splits);
int64_t mvp_gemm_streamk_workspace_bytes(void);
int mvp_gemm_streamk(const mvp_gemm_args*, void* stream);  /* what mvp_gemm_bias_act_res dispatches to for splitk == MVP_GEMM_STREAMK */

/* ------------------------------------------------------------------------------------
 * LayerNorm forward: fp32 rows [M, C] -> bf16 pair [M, C] (the next GEMM's A operand).
 * Replaces nn.LayerNorm(eps=1e-6) at ibot_transformers.py:164,174,194,199 (eps 1e-12 for
 * the HF ViT-MAE path, evals/models/mae.py:33).  One wave64 per row, two-pass in registers.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* x; const float* gamma; const float* beta;
  mvp_bf16* out_hi; mvp_bf16* out_lo; /* lo may be NULL */
  float* out_f32;                     /* optional fp32 copy, or NULL */
  int M, C;                           /* C % 8 == 0, C <= 2048 */
  float eps;
  int out_layout;                     /* MVP_PAIR_SEPARATE, or MVP_PAIR_A_ILV32 (1): out_hi is ONE [M][C / 32][hi 32 | lo 32] array (row stride
                                         2 * C), out_lo ignored, C % 32 == 0 — the A operand of the large-M GEMM (mvp_gemm_args.pair_layout) */
  int out_f16;                        /* (ABI 6) 1: the pair is written as the activation operand of MVP_PREC_F16X2 (hi = fp16(y), lo = fp16(8 (y - hi) + hi / 8)) */
} mvp_layernorm_args;
int mvp_layernorm_fwd(const mvp_layernorm_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-head self-attention forward, flash-style (no N x N materialisation):
 *   O = softmax(Q Kᵀ * scale) V   per (batch, head), head_dim = 64.
 * qkv is the fused projection output [B*N, 3*H*64] (bf16 pair) laid out as the reference
 * reshapes it: col = which*H*64 + head*64 + d  (ibot_transformers.py:129-142).
 * out [B*N, H*64] bf16 pair = (attn @ v).transpose(1,2).reshape(B,N,C)  (:142).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const mvp_bf16* qkv_hi; const mvp_bf16* qkv_lo;
  mvp_bf16* out_hi; mvp_bf16* out_lo;
  int B, N, H;      /* tokens per image N (incl. CLS), heads H; head_dim fixed 64 */
  int ld_qkv, ld_out;
  float scale;      /* head_dim^-0.5 */
  int precision;
  int out_layout;   /* MVP_PAIR_SEPARATE, or MVP_PAIR_A_ILV32 (1): out_hi is ONE [B*N][H*2][hi 32 | lo 32] array with row stride ld_out
                       (>= 2 * H * 64), out_lo ignored — the A operand of the large-M proj GEMM (mvp_gemm_args.pair_layout)               */
  int v_format;     /* MVP_PREC_BF16X3 only.  MVP_ATT_V_BF16_PAIR (0): V is a bf16 pair like Q and K; the probabilities are split into a bf16
                       pair too and P.V runs three products (hi.hi + hi.lo + lo.hi).  MVP_ATT_V_F16 (1, ABI 6): the V third of qkv holds
                       hi = fp16(v), lo = bf16(v - hi) (mvp_gemm_args.out_f16_col0); the probabilities are held as ONE fp16 value
                       (+ its bf16 rounding for the lo product): P.V = v_hi.p on the f16 MFMA + v_lo.p on the bf16 MFMA — two products
                       instead of three and no hi / lo split of P on the vector pipe (the kernel is VALU-bound); the online softmax
                       rescales its running maximum only when it rises by more than 2^6.  Q.K^T keeps its three products either way.
                       Relative error of the output: ~2^-12 per probability (random, averaged over the keys) instead of 2^-17.        */
  int out_f16;      /* (ABI 6) 1: the output pair is written as the activation operand of a MVP_PREC_F16X2 proj GEMM (hi = fp16(o), lo = fp16(8 (o - hi) + hi / 8)) */
} mvp_attention_args;
#define MVP_ATT_V_BF16_PAIR 0
#define MVP_ATT_V_F16 1
#define MVP_ATT_V_F16_QK_F16 2 /* ABI 7.  V and the probabilities as MVP_ATT_V_F16; Q is the compensated ACTIVATION pair and K the compensated WEIGHT-side
                                  pair of MVP_PREC_F16X2 (mvp_gemm_args.out_f16_col0 = -(first column of the V third)): Q.K^T as two f16 products instead of
                                  three bf16 ones, same ~2^-18 relative error per term; Q and K must stay within fp16's range (saturating) */
int mvp_attention_fwd(const mvp_attention_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * CLS row writer: x[b, 0, :] = cls[:] + pos[0, :]   (ibot_transformers.py:347-352).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* cls; const float* pos0; float* x; int B, N, C;
} mvp_cls_rows_args;
int mvp_cls_rows(const mvp_cls_rows_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Tap BatchNorm over tokens (train-mode batch statistics, CLS included) fused with the
 * token -> NCHW dense-map transpose.  Replaces nn.BatchNorm1d on x.permute(0,2,1)
 * (evals/models/dino.py:185-191) + tokens_to_output("dense") (evals/models/utils.py:111-114).
 *   stats pass : per-channel mean / biased var over all B*N rows -> stats[2*C] (mean, var),
 *                running stats updated with momentum & unbiased var when running_* != NULL (unless defer_running);
 *   apply pass : y = (x - mean) * rsqrt(var + eps) * gamma + beta, spatial tokens only
 *                (the last hw tokens of each image), written as
 *                  nchw   [B, C, h, w] fp32          (the reference's return value)
 *                  tok_hi/lo [B*hw (ld_tok)] bf16 pair at column offset col_off
 *                           (token-major operand of the probe-head GEMM; optional)
 *                  tokT_hi/lo [C rows at row offset col_off][ldT] bf16 pair
 *                           (transposed copy, operand of the head's dW GEMM; optional)
 * mode: 0 = train (batch stats), 1 = eval (use running stats), 2 = no norm (identity).
 * workspace: ws_bytes = mvp_bn_tokens_workspace_bytes(M, C) (fp32 partials).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* x;       /* [B, N, C] fp32 token stream                           */
  const float* gamma; const float* beta; /* [C] or NULL (identity)               */
  float* running_mean; float* running_var; /* [C] or NULL                        */
  float* stats;         /* [2*C] out: batch mean, biased var                      */
  float* nchw;          /* [B, C, hw] or NULL                                     */
  mvp_bf16* tok_hi; mvp_bf16* tok_lo; int ld_tok; int col_off;
  mvp_bf16* tokT_hi; mvp_bf16* tokT_lo; int ldT;
  void* workspace; int64_t workspace_bytes;
  int B, N, C, hw;      /* hw = spatial tokens per image (N - hw leading tokens dropped) */
  float eps, momentum;
  int mode;
  float* cls_out;       /* [B, C] fp32 or NULL: the normalised FIRST token of each image (the CLS token that
                           tokens_to_output(output="cls" / "dense-cls") returns, utils.py:105-124); needs N > hw */
  int64_t* num_batches_tracked; /* or NULL: BatchNorm's step counter, incremented in train mode (mode 0) by the statistics kernel */
  int defer_running;    /* 1 (mode 0 only): leave running_mean / running_var / num_batches_tracked untouched and write the unbiased
                           variance to stats[2*C .. 3*C) (stats then holds 3*C floats); mvp_bn_running_update applies the update later.
                           For forwards that run concurrently on several streams (mvp/pipeline.py): the running statistics are the only
                           state a frozen forward mutates, and their momentum updates must happen in batch order.                     */
  /* --- several batches in one launch (groups = G > 1; mvp/pipeline.py stacks G equal batches into one frozen forward): x holds
   * G * B images; statistics, normalisation and outputs are those of each batch of B images ALONE (train-mode BatchNorm is per batch,
   * dino.py:185-191) — the same bits as G separate calls.  Group g reads x + g * B*N*C and writes stats + g * stats_gstride,
   * nchw + g * nchw_gstride, tok_* + g * tok_gstride, cls_out + g * cls_gstride (strides in elements).  Train mode needs
   * defer_running = 1 (the running statistics are then updated per batch by mvp_bn_running_update, in batch order); tokT_* is not
   * supported; workspace_bytes >= G * mvp_bn_tokens_workspace_bytes(B * N, C).  groups <= 1: one batch, the strides are ignored.     */
  int groups;
  int64_t stats_gstride, nchw_gstride, tok_gstride, cls_gstride;
} mvp_bn_tokens_args;
int64_t mvp_bn_tokens_workspace_bytes(int M, int C);
int mvp_bn_tokens_to_nchw_fwd(const mvp_bn_tokens_args*, void* stream);

/* The deferred half of nn.BatchNorm's train-mode bookkeeping (torch/nn/modules/batchnorm.py; dino.py:185-191 keeps the tap BNs in
 * train mode): running = (1 - momentum) * running + momentum * batch statistic (unbiased variance), num_batches_tracked += 1.
 * Same arithmetic, bit for bit, as the in-kernel update of mvp_bn_tokens_to_nchw_fwd with defer_running = 0. */
typedef struct {
  const float* stats;   /* [3*C]: mean, biased var, unbiased var — as written with defer_running = 1 */
  float* running_mean; float* running_var;  /* [C] */
  int64_t* num_batches_tracked;             /* or NULL */
  int C; float momentum;
} mvp_bn_running_update_args;
int mvp_bn_running_update(const mvp_bn_running_update_args*, void* stream);
/* The same update for n (1..MVP_BN_RUNNING_MAX) BatchNorm modules in ONE launch — the four tap BNs of a multilayer extract
 * (dino.py:185-191) are handed to the trainer together, and beside a frozen forward every launch of the probe step's stream costs
 * its queueing delay, not its 2 us of work.  Element-wise identical to n calls of mvp_bn_running_update. */
#define MVP_BN_RUNNING_MAX 8
int mvp_bn_running_update_n(const mvp_bn_running_update_args* items, int n, void* stream);

/* ------------------------------------------------------------------------------------
 * NCHW fp32 feature maps -> token-major bf16 pair (+ transposed copy).  Fallback packer
 * used when the probe is handed plain NCHW tensors (e.g. features loaded from disk).
 * Replaces torch.cat(feats, dim=1) of probes.py:427-429 as an operand-layout change.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* nchw;    /* [B, C, hw]                                             */
  mvp_bf16* tok_hi; mvp_bf16* tok_lo; int ld_tok; int col_off;
  mvp_bf16* tokT_hi; mvp_bf16* tokT_lo; int ldT;
  int B, C, hw;
} mvp_pack_nchw_args;
int mvp_pack_nchw_tokens(const mvp_pack_nchw_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Bilinear / bicubic / nearest resampling of NCHW-like planes, forward and adjoint.
 * Replaces F.interpolate(..., mode=..., align_corners=False|True) at
 * train_depth.py:114 (bilinear), train_snorm.py:110 (bicubic), probes.py:255-258,388,396-398.
 * Coordinates follow ATen's area_pixel_compute_source_index; bicubic uses A = -0.75 with
 * clamped taps.  planes = B*C.  bwd computes grad_in[planes, Hi, Wi] (gather form, no atomics).
 * channels_last = 1 treats the tensor as [B, H, W, C] (token-major logits), planes = B.
 * ---------------------------------------------------------------------------------- */
#define MVP_RESIZE_NEAREST 0
#define MVP_RESIZE_BILINEAR 1
#define MVP_RESIZE_BICUBIC 2
typedef struct {
  const float* src; float* dst;
  int planes, Hi, Wi, Ho, Wo;
  int mode; int align_corners;
  int channels_last; int C; /* C used only when channels_last */
  float scale_h, scale_w;   /* >0: the scale_factor given by the caller (recompute_scale_factor=False semantics); 0: derive from sizes */
} mvp_resize_args;
int mvp_resize_fwd(const mvp_resize_args*, void* stream);
int mvp_resize_bwd(const mvp_resize_args*, void* stream); /* src = grad_out [.,Ho,Wo], dst = grad_in [.,Hi,Wi] */
/* Antialiased bilinear, forward only, planar, align_corners = 0, sizes given (torchvision transforms.Resize on a tensor =
 * interpolate(bilinear, antialias=True), dino_res50.py:80,85 for inputs larger than fixed_size on an axis). */
int mvp_resize_aa_fwd(const mvp_resize_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Depth predictors on token-major logits [P, K] (P = B*H*W pixels, K channels contiguous).
 *   bins   : p = relu(l) + 0.1; p /= sum(p); depth = sum_k p_k * linspace(min,max,K)[k]
 *            (probes.py:176-200);  saves inv_sum[P] for the backward.
 *   sigmoid: depth = min + sigmoid(l) * (max - min)          (probes.py:209-212), K == 1.
 * bwd: grad_logits[P,K] from grad_depth[P].
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* logits; float* depth; float* inv_sum;
  const float* grad_depth; float* grad_logits;
  int64_t P; int K;
  float min_depth, max_depth;
  int kind; /* 0 = bins, 1 = sigmoid */
} mvp_depth_predict_args;
int mvp_depth_predict_fwd(const mvp_depth_predict_args*, void* stream);
int mvp_depth_predict_bwd(const mvp_depth_predict_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * DepthLoss = 10 * sig_loss + 0.5 * gradient_loss  (evals/utils/losses.py:97-154),
 * including the reference's quirks: target > max_depth is zeroed IN PLACE (Q2) and the
 * "gradient" term differences samples b and b+2 over batch strides {1,2,4,6} (Q1).
 * pred/target [B, HW] fp32.  Outputs: loss[0] (total), loss[1] sig, loss[2] grad term;
 * grad_pred [B, HW] = d loss / d pred (already scaled by the 10 / 0.5 weights).
 * workspace >= mvp_depth_loss_workspace_bytes(B, HW) bytes, zero-initialised by the op.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* pred; float* target; float* loss; float* grad_pred;
  void* workspace; int64_t workspace_bytes;
  int B; int64_t HW;
  float w_sig, w_grad, max_depth, eps, sigma;
} mvp_depth_loss_args;
int64_t mvp_depth_loss_workspace_bytes(int B, int64_t HW);
int mvp_depth_loss_fwd_bwd(const mvp_depth_loss_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * angular_loss (evals/utils/losses.py:157-182), optional uncertainty-aware (4-channel).
 * pred [B, Cp, HW], gt [B, 3, HW], mask [B, HW] (uint8, non-zero = valid).
 * loss[0] = masked mean; grad_pred = d loss / d pred.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* pred; const float* gt; const uint8_t* mask;
  float* loss; float* grad_pred;
  void* workspace; int64_t workspace_bytes; /* >= 4096 bytes */
  int B; int Cp; int64_t HW;
  float eps;
} mvp_angular_loss_args;
int mvp_angular_loss_fwd_bwd(const mvp_angular_loss_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Column sums of a fp32 matrix [M, N] -> out[N] (bias gradients), deterministic two-level sum.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* x; float* out; int M, N, ld;
  int accumulate;                       /* 1: out += column sums (gradient accumulation) */
  void* workspace; int64_t workspace_bytes; /* >= mvp_colsum_workspace_bytes(M, N)       */
} mvp_colsum_args;
int64_t mvp_colsum_workspace_bytes(int M, int N);
int mvp_colsum(const mvp_colsum_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused AdamW over a flat fp32 parameter buffer (torch.optim.AdamW defaults,
 * train_depth.py:624-627): decoupled weight decay, bias correction, eps outside sqrt.
 * Schedule state: hyper == NULL -> lr, bias_c1 = 1 - beta1^t, bias_c2 = 1 - beta2^t are taken BY VALUE from the
 * struct (no host->device staging that a host running ahead of the stream could overwrite);
 * hyper != NULL -> device scalars hyper[0..2] = {lr, bias_c1, bias_c2} (replay of a captured hipGraph).
 * grad_scale multiplies the gradient first (1/world_size after an all-reduce SUM).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
  const float* hyper; int64_t n;
  float beta1, beta2, eps, weight_decay, grad_scale;
  float lr, bias_c1, bias_c2;
} mvp_adamw_args;
int mvp_adamw_step(const mvp_adamw_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * SPair correspondence core (evaluate_spair_correspondence.py:59-83 + argmax_2d,
 * evals/utils/correspondence.py:179-190): L2-normalise features over C, bilinearly sample
 * the source map at K keypoints (grid_sample align_corners=True), cosine heat-map against
 * the target map, 2-D argmax.  out_xy [K,2] int64 = (col,row), out_val [K] fp32.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* src_feat; const float* tgt_feat; /* [C, h, w] fp32 each            */
  const float* kp_xy;                            /* [K,2] normalised coords in [-1,1] (x,y) */
  int64_t* out_xy; float* out_val;
  float* workspace; int64_t workspace_bytes;    /* >= mvp_corr_workspace_bytes(C, h, w, K), 8-byte aligned */
  int C, h, w, K;
  float* heat_out;                               /* optional [K, h, w]: the cosine heat-maps (compute_errors(return_heatmaps=True)) */
} mvp_corr_argmax_args;
int64_t mvp_corr_workspace_bytes(int C, int h, int w, int K);
int mvp_corr_argmax(const mvp_corr_argmax_args*, void* stream);
/* argmax_2d (evals/utils/correspondence.py:179-190) on materialised maps x [K, h, w] fp32: out_xy [K,2] int64 = (col,row) of
 * the flat argmax (max_value != 0) or argmin (max_value == 0); ties -> lowest flat index, as torch. */
typedef struct {
  const float* x; int64_t* out_xy; int K, h, w, max_value;
} mvp_argmax_2d_args;
int mvp_argmax_2d(const mvp_argmax_2d_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Convolution weight re-layout (per step; the probe's conv weights are trained):
 *   mode 0: [Cout,Cin,kh,kw] fp32 -> forward GEMM operand [Cout, kh*kw*Cin] bf16 pair (k = tap*Cin + c)
 *   mode 1: -> data-gradient operand [Cin, kh*kw*Cout] (k = tap'*Cout + n, taps flipped)
 * Replaces the implicit weight access of nn.Conv2d fwd / bwd-data (probes.py:283-288,371-375).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* w; mvp_bf16* out_hi; mvp_bf16* out_lo;
  int Cout, Cin, kh, kw, mode;
} mvp_conv_weight_pack_args;
int mvp_conv_weight_pack(const mvp_conv_weight_pack_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Nearest-neighbour upsample by an integer factor f on channels-last fp32 [B,H,W,C]
 * (F.interpolate(x, scale_factor=f), probes.py:388,396,398); backward = f x f block sums
 * (src is then the fine [B,H*f,W*f,C] gradient).  Outputs: fp32 and/or bf16 pair.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* src; float* dst_f32; mvp_bf16* dst_hi; mvp_bf16* dst_lo;
  int B, H, W, C, f; int backward; /* H,W = COARSE dims in both directions */
} mvp_upsample_cl_args;
int mvp_upsample_nearest_cl(const mvp_upsample_cl_args*, void* stream);

/* Backward of "nearest x f upsample, then 3x3 / pad 1 convolution" folded onto the COARSE grid (the DPT probe's out_conv,
 * probes.py:384-398: F.interpolate(scale_factor=4) then Conv2d(3x3)): box sums of the fine-grid output gradient g [B, H*f, W*f, C]
 * per coarse pixel and tap, G[(b,i,j), ky*3+kx, c] = sum of g[p, c] over fine pixels p with p + (ky-1, kx-1) in block (i, j), written
 * as a bf16 pair [B*H*W, 9*C] (out_lo may be null).  Then dW = Gᵀ·x (mvp_gemm_tn_conv, 1x1) and dx = G·Wᵀ (mvp_gemm_bias_act_res)
 * over the coarse pixels replace the autograd backward of that Conv2d + interpolate pair.  f in {2, 4}; C % 4 == 0. */
typedef struct {
  const float* g; mvp_bf16* out_hi; mvp_bf16* out_lo;
  int B, H, W, C, f; /* H, W = COARSE dims */
} mvp_upconv_boxsum_args;
int mvp_upconv3_grad_boxsum(const mvp_upconv_boxsum_args*, void* stream);

/* Forward of the same pair from COARSE tap products: t [B*H*W, 9*C] fp32 = x · W_tapᵀ for the 9 taps (ONE GEMM over the coarse pixels,
 * column tap*C + c); y[p] = act(bias + sum over the taps of t[cell(p + d_tap), tap]) per fine pixel p (taps outside the image = zero
 * padding), written as the convolution's epilogue would: fp32 and / or bf16 pair [B*H*f*W*f, C], gate mask (post-ReLU value > 0).
 * act in {MVP_ACT_NONE, MVP_ACT_RELU}; f in {2, 4}; C % 4 == 0. */
typedef struct {
  const float* t; const float* bias; float* out_f32; mvp_bf16* out_hi; mvp_bf16* out_lo; uint8_t* out_mask;
  int B, H, W, C, f, act; /* H, W = COARSE dims */
} mvp_upconv_gather_args;
int mvp_upconv3_fwd_gather(const mvp_upconv_gather_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Convolution weight gradient (TN GEMM over pixels, split-K):
 *   dW[n, c, ky, kx] (+)= sum_m G[m, n] * X[pix(m, ky, kx), c]
 * G = output gradient [M = B*Ho*Wo, ldg] channels-last bf16 pair (ldg >= roundup(Cout,128),
 * pad columns zero), X = layer input [B, H>>up, W>>up, ldx] channels-last bf16 pair seen through
 * a nearest upsample by 2^up; Cin % 128 == 0.  1x1 convs / linear layers use kh=kw=1, pad=0.
 * partial: fp32 workspace of mvp_gemm_tn_workspace_bytes(...) bytes.  Replaces the autograd
 * weight-gradient of nn.Conv2d (probes.py:283-288,352-355,371-375).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const mvp_bf16* g_hi; const mvp_bf16* g_lo; const mvp_bf16* x_hi; const mvp_bf16* x_lo;
  float* partial; float* dw; const mvp_bf16* zero_page; /* >= 512 zero bytes (ABI stability: no longer read, padding = out-of-range offsets) */
  int64_t M; int Cout, Cin, ldg, ldx;
  int H, W, Ho, Wo, kh, kw, stride, pad, up;
  int splits, accumulate, precision;
} mvp_gemm_tn_args;
/* ------------------------------------------------------------------------------------
 * Validation metrics, fused masked reductions per image:
 *   depth : evaluate_depth global metrics (evals/utils/metrics.py:106-178): out[b] =
 *           {d1,d2,d3,rmse,mean_pred,std_pred,variance_pred,mean_gt,std_gt,variance_gt,variance_ratio,n_valid};
 *           scale_invariant != 0 first solves match_scale_and_shift (metrics.py:742-780) per image and
 *           writes (scale, shift) to scale_shift[b].  pred/gt [B, HW].
 *   snorm : evaluate_surface_norm global metrics (metrics.py:397-440): out[b] = {d1,d2,d3,rmse_deg,n_valid};
 *           pred [B,Cp>=3,HW] (first 3 channels), gt [B,3,HW], valid = |gt|_1 > 0.
 * workspace >= mvp_metrics_workspace_bytes(B).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* pred; const float* gt; float* out; float* scale_shift;
  void* workspace; int64_t workspace_bytes; int B; int64_t HW; int scale_invariant;
} mvp_depth_metrics_args;
typedef struct {
  const float* pred; const float* gt; float* out;
  void* workspace; int64_t workspace_bytes; int B; int Cp; int64_t HW; float t1, t2, t3;
} mvp_snorm_metrics_args;
int64_t mvp_metrics_workspace_bytes(int B);
int mvp_depth_metrics(const mvp_depth_metrics_args*, void* stream);
int mvp_snorm_metrics(const mvp_snorm_metrics_args*, void* stream);
/* y = x * scale[b] + shift[b] per image (match_scale_and_shift, metrics.py:775-777), optionally clamped to [lo, hi]
 * (scale-invariant training, train_depth.py:116-118).  backward != 0: out = d/dx = grad_out * scale[b] where the clamp
 * is inactive, else 0 (scale/shift are detached in the reference).  x / out / grad_out [B, HW]; scale_shift [B, 2]. */
typedef struct {
  const float* x; const float* scale_shift; const float* grad_out; float* out;
  int B; int64_t HW; float lo, hi; int clamp; int backward;
} mvp_scale_shift_args;
int mvp_scale_shift(const mvp_scale_shift_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Per-level / per-segment breakdown of the validation metrics (evaluate_depth
 * evals/utils/metrics.py:179-358, evaluate_surface_norm metrics.py:441-577) as one segmented masked
 * reduction per batch.  Every pixel is binned by its "centroid level" (nested centre boxes whose
 * offset (H / L) * (L - level) / 2 is applied to both axes, metrics.py:249-262) and by its
 * integer segment id:
 *   level_sums [B, L, 5] fp64 = {n_valid, #(x<t1), #(x<t2), #(x<t3), sum err^2}
 *   seg_sums   [B, S, 6] fp64 = {n_all, n_valid, #(x<t1), #(x<t2), #(x<t3), sum err^2}   (ids outside [0,S) are skipped)
 * Cp == 0: depth — pred/gt [B,H,W], valid = gt > 0, x = max(gt/pred, pred/gt) with thresholds 1.25^k,
 *          err = gt - pred; scale_shift != NULL applies the per-image (scale, shift) of mvp_depth_metrics first.
 * Cp >= 3: surface normals — pred [B,Cp,H,W], gt [B,3,H,W], valid = |gt|_1 > 0, x = err = angular error in degrees,
 *          thresholds t1..t3.
 * seg == NULL: levels only (the reference's is_navi=True path).  The 1e-6 / clamp(1) normalisations, the
 * stuff/things grouping and the segment list are finished by the host from these bins.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* pred; const float* gt; const int32_t* seg; const float* scale_shift;
  double* level_sums; double* seg_sums;
  void* workspace; int64_t workspace_bytes;   /* >= mvp_metrics_breakdown_workspace_bytes(B, L, S), 8-byte aligned */
  int B, H, W, Cp, num_levels, num_ids;
  float t1, t2, t3;
} mvp_metrics_breakdown_args;
int64_t mvp_metrics_breakdown_workspace_bytes(int B, int num_levels, int num_ids);
int mvp_metrics_breakdown(const mvp_metrics_breakdown_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused tail of the linear depth-bin probe: bilinear x f (align_corners=False) of the
 * token-resolution logits L0 [B,h,w,K] fused with DepthBinPrediction (probes.py:431 + :176-200).
 * The upsampled logits are never materialised; gate holds 1 bit per (pixel, bin) = [logit > 0].
 *   fwd: l0 -> depth [B*hf*wf], inv_sum [B*hf*wf], gate [B*hf*wf, K/8]
 *   bwd: grad_depth, depth, inv_sum, gate -> grad_l0 [B,h,w,K]          (K % 8 == 0)
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* l0; float* depth; float* inv_sum; uint8_t* gate;
  const float* grad_depth; float* grad_l0;
  int B, h, w, K, f;
  float min_depth, max_depth;
} mvp_linear_bins_args;
int mvp_linear_bins_fwd(const mvp_linear_bins_args*, void* stream);
int mvp_linear_bins_bwd(const mvp_linear_bins_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * im2col of an NCHW fp32 image for convs whose Cin is not a multiple of 32 (the ResNet 7x7/2 RGB
 * stem, dino_res50.py:38-44): out[(b,yo,xo), (ky*kw+kx)*C + c], zero padded to ldk columns (ldk % 8 == 0).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* src; mvp_bf16* out_hi; mvp_bf16* out_lo;
  int B, C, H, W, Ho, Wo, kh, kw, stride, pad, ldk;
} mvp_im2col_args;
int mvp_im2col_nchw(const mvp_im2col_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Max-pool k x k / stride on channels-last fp32 [B,H,W,C] (-inf padding, torchvision
 * resnet50.maxpool 3x3/2 pad 1).  Outputs fp32 and/or bf16 pair.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* src; float* dst_f32; mvp_bf16* dst_hi; mvp_bf16* dst_lo;
  int B, H, W, C, Ho, Wo, k, stride, pad;
} mvp_maxpool_cl_args;
int mvp_maxpool_cl(const mvp_maxpool_cl_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * ResNet-50 stem fused: conv 7x7 / 2 / pad 3 (3 -> 64, BatchNorm folded into w / bias) + ReLU + max-pool 3x3 / 2 / pad 1,
 * NCHW fp32 image -> channels-last [B, Hp, Wp, 64] (torchvision resnet50.conv1/bn1/relu/maxpool, dino_res50.py:38-44,83-90).
 * w_hi/lo [64, 160] bf16: k = (ky*7 + kx)*3 + c, zero padded from 147.  Ho = (H-1)/2+1, Hp = (Ho-1)/2+1 (same for W).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* images; const mvp_bf16* w_hi; const mvp_bf16* w_lo; const float* bias;
  float* out_f32; mvp_bf16* out_hi; mvp_bf16* out_lo;
  int B, H, W, precision;
} mvp_stem_args;
int mvp_stem7x7_pool(const mvp_stem_args*, void* stream);

/* ------------------------------------------------------------------------------------
 * Gradient gate + split: dst = src * (mask != 0) as fp32 (may alias src) and as a bf16 pair
 * with row stride ldo >= N (pad columns zeroed) — the ReLU backward of probes.py:283-288 fused
 * with the operand conversion for the next MFMA GEMM.  mask may be NULL (plain split).
 * relu_mask_out != NULL: forward ReLU instead — dst = max(src, 0), relu_mask_out[r*ldm + c] = (src > 0)
 * (the ReLU that follows a bilinear upsample in MultiscaleHead, probes.py:449-457, fused with the split).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const float* src; const uint8_t* mask; float* dst_f32; mvp_bf16* dst_hi; mvp_bf16* dst_lo;
  int64_t M; int N, lds, ldm, ldo;
  uint8_t* relu_mask_out;
} mvp_mask_split_args;
int mvp_mask_split(const mvp_mask_split_args*, void* stream);

int64_t mvp_gemm_tn_workspace_bytes(int Cout, int Cin, int kh, int kw, int splits);
int mvp_gemm_tn_conv(const mvp_gemm_tn_args*, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVP_HIP_H */

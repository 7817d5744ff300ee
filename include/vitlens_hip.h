/* vitlens_hip.h — C ABI of libvitlens_hip.so (MI355X / gfx950).
 *
 * The reference (TencentARC/ViT-Lens) has NO native layer: its hot path is stock torch.nn
 * modules.  The drop-in boundary is therefore the Python module surface
 * (open_clip.TriCLIP.encode_*, open_clip.loss.*, mm_vit_lens.ViTLens.encode); this C ABI
 * sits directly under that surface and is what a maintainer's ctypes stub binds (see
 * INTEGRATION.md).  Each entry point names the reference call it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed for the duration of the call; nothing is
 *     retained or freed; workspaces are supplied by the caller.  The few HOST arrays are named as
 *     such at their declaration (stride tables of the attention entries, mean / stdv of the
 *     image resampler) and are read before the call returns
 *   - all matrices are row-major; "bf16" is a 16-bit brain-float bit pattern (uint16_t)
 *   - every function is stream-ordered and non-blocking on `stream` (a hipStream_t)
 *   - return value: 0 on success, non-zero on failure; vl_last_error() gives the message
 *   - no exceptions cross the ABI; no global mutable state except the thread-local last-error string; one-time
 *     initialisation (hipFuncSetAttribute, CU count) is done through C++11 function-local statics (thread-safe);
 *     entry points are re-entrant per stream
 */
#ifndef VITLENS_HIP_H
#define VITLENS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

/* epilogues of vl_gemm_bf16 */
#define VL_EPI_BF16 0     /* out bf16[M,N]   = act(alpha*acc + bias)                     */
#define VL_EPI_F32 1      /* out f32 [M,N]   = alpha*acc + bias                          */
#define VL_EPI_RES_F32 2  /* out f32 [M,N]   = res f32 + alpha*acc + bias (in place ok)  */
#define VL_EPI_RES_BF16 3 /* out bf16[M,N]   = res bf16 + alpha*acc + bias               */
#define VL_EPI_GEGLU 5    /* out bf16[M,N/2] = a*gelu(gate), W rows interleaved (a,gate) */
#define VL_EPI_DGELU 6    /* out bf16[M,N]   = alpha*acc * gelu'(res bf16[M,N])  (dX through GELU); with act =
                           * VL_ACT_GELU_DSAVE res IS gelu' (left by the forward): the epilogue is one multiplication */
#define VL_EPI_DGEGLU 7   /* acc = dy[M,N]; res = h bf16[M,2N] interleaved (a,g); out bf16[M,2N] = d h  (dX through GEGLU) */
#define VL_ACT_NONE 0
#define VL_ACT_GELU 1     /* exact-erf GELU (nn.GELU default)                            */
#define VL_ACT_RELU 2
#define VL_ACT_GELU_DSAVE 4 /* forward of a TRAINED MLP: out = gelu(pre), out2 (required) = gelu'(pre), both of the bf16-rounded
                             * pre-activation; pair it with VL_EPI_DGELU + the same act in the backward               */

/* dtype tags for mixed-dtype entry points */
#define VL_F32 0
#define VL_BF16 1
#define VL_F16 2          /* IEEE half: output of vl_layernorm_fwd (from f32 rows) feeding vl_gemm_f16 */

#define VL_GEMM_AUTO (-1)
#define VL_GEMM_PERSIST 8
#define VL_GEMM_PINGPONG 10

const char* vl_last_error(void);
/* ABI version of THIS header: bumped whenever an exported signature changes (round 5 changed vl_ln_row_stats and added the
 * VL_F16 tag and the fp16 / fp32 entries without bumping it; round 6 starts counting: 600 = round 6, first revision).
 * vl_version() returns the library's value; a client built against this header must find them equal before its first
 * call - the Python binding (vitlens_hip/_lib.py) and tests/native/abi_c_client.c both refuse to run otherwise. */
#define VL_ABI_VERSION 601
int vl_version(void);

/* C[M,N] = A[M,K] · W[N,K]^T with fused epilogue.  A, W bf16.  K % 64 == 0, N % 4 == 0.
 * cfg selects the kernel family - the ONE selector of this library: nothing else (no environment variable, no global
 * setter) changes which code runs.  VL_GEMM_AUTO (-1) is what every product path passes; the explicit values exist for the
 * parity tests (every family against fp32 and against each other) and the probes under tools/:
 *   -1 VL_GEMM_AUTO     whole rounds of 256-row tiles on the persistent 256x256 kernel (8; 10 where N % 256 == 128), the
 *                       leftover rows on 64x64 tiles or the split-K tail kernel, small problems on 128x128 tiles
 *    8 VL_GEMM_PERSIST  vl_gemm_park.hip: persistent, 256x256 tiles, LDS-DMA two k-steps ahead, 16x16x32 MFMA, burst of
 *                       non-temporal stores; M % 256 == N % 256 == 0, K >= 512
 *   10 VL_GEMM_PINGPONG vl_gemm_pp.hip: two 4-wave workgroups per CU on 256x128 tiles (widths that are multiples of 128 only)
 *    5 (4 = alias)      round-1 persistent kernel (any shape) | 6: the same on 256x128 tiles
 *    0 / 1              one 256x256 / 128x128 LDS-DMA tile per workgroup | 2 / 3: the same with register staging
 *    9                  split-K tail kernel (32x32 tile per workgroup) | 11 / 12: 64x64 / 128x64 tiles
 * Replaces nn.Linear / out_proj / mlp.c_fc(+GELU) / mlp.c_proj(+residual)
 * (open_clip/transformer.py:226-234,268-271), Perceiver to_q/to_kv/to_out/FeedForward
 * (open_clip/perceiver.py:85-123), pooled @ proj (transformer.py:786-787) and
 * logit_scale * x @ y.T (open_clip/loss.py:131-136). */
int vl_gemm_bf16(const void* A, const void* W, const float* bias, void* out, const void* res,
                 int M, int N, int K, int lda, int ldw, int ldo, float alpha, int epi, int act,
                 int cfg, hipStream_t stream);

/* Split-K accumulate for weight-gradient GEMMs (tiny M x N, K = token count): out[M,N] f32 (row stride ldo) +=
 * alpha * A[M,K] . W[N,K]^T.  The K range is cut into `splits` slices computed by separate workgroups into the
 * caller's workspace ws (splits*M*N floats) and summed in a fixed order (deterministic, no atomics).
 * Autograd counterpart of the dW of nn.Linear / 1x1 Conv1d in the trainable Lens. */
int vl_gemm_splitk_accum_f32(const void* A, const void* W, float* out, int M, int N, int K, int lda, int ldw, long ldo,
                             float alpha, int splits, float* ws, hipStream_t stream);
/* The same accumulate on TOKEN-MAJOR operands: out[M,N] f32 += alpha * At[K,M]^T . Bt[K,N], At = dY and Bt = X as the backward
 * holds them (row = token, row strides lda >= M, ldb >= N, multiples of 8) - no transposed copies.  M % 256 == N % 256 == 0,
 * K % 64 == 0; the K/64 steps are cut into slices of ceil(K/64/splits) (the last one may be shorter; every slice >= 4 steps);
 * ws = splits*M*N floats.  Fragments come from gfx950's LDS
 * transpose read (csrc/vl_gemm_tn.hip).  Replaces the dW of nn.Linear / in_proj / out_proj under loss.backward()
 * (open_clip/transformer.py:215,226-234,252-272; training/train.py:212-216). */
int vl_gemm_tn_splitk_accum_f32(const void* At, const void* Bt, float* out, int M, int N, int K, int lda, int ldb, long ldo,
                                float alpha, int splits, float* ws, hipStream_t stream);
/* vl_gemm_bf16 + `out2` (with VL_EPI_BF16/VL_ACT_GELU also stores the pre-activation, bf16, for the
 * backward) + `res_div` (VL_EPI_RES_BF16: residual row = m / res_div, i.e. one row broadcast over a group:
 * the PointNet concat([global, local]) conv of dvae.py:207-210 split into two GEMMs).
 * The packed in-projection of nn.MultiheadAttention (transformer.py:215,252) and the Perceiver's to_q / to_kv
 * (perceiver.py:115-116,124-126) are plain calls of this entry point: the attention kernels read q, k, v out of the
 * [tokens, 3*width] result in place (rounds 1-2a had a head-scatter epilogue and transposed copies instead). */
int vl_gemm_bf16_ex(const void* A, const void* W, const float* bias, void* out, const void* res, void* out2,
                    int M, int N, int K, int lda, int ldw, int ldo, float alpha, int epi, int act,
                    int res_div, int cfg, hipStream_t stream);

/* The FROZEN text tower on IEEE-half operands (round 5).  TriCLIP.encode_text (open_clip/model.py:528-540) is forward-only in
 * every recipe and its cosine-similarity MATRIX amplifies operand rounding (text features share a mutual cosine of ~0.6):
 * bf16 operands leave it 0.8-1.9e-3 from the fp32 CPU path, above BASELINE.json's 1e-3; rounds 4's two-term bf16 weights
 * brought it to 6-8e-4 at twice the products.  fp16 carries three more mantissa bits on EVERY operand (weights, GEMM inputs,
 * q / k / v, attention probabilities) at the bf16 MFMA rate: 1-2e-4 at one product per weight.  The reference itself converts
 * CLIP to fp16 (`convert_weights_to_fp16`, open_clip/model.py:393-419; precision "fp16", factory.py:260-295).  The residual
 * stream stays fp32, 16-bit stores saturate at +-65504.
 *   vl_gemm_f16      C = A . W^T with A [M,K], W [N,K] fp16; epi = VL_EPI_BF16 (here: out fp16 [M,N] = act(alpha*acc + bias),
 *                    act none / GELU) or VL_EPI_RES_F32 (out f32 = res f32 + alpha*acc + bias, in place allowed).  The
 *                    persistent 256x256 kernel only: M % 256 == N % 256 == 0, K % 64 == 0, K >= 512, 16-byte aligned
 *                    operands - the caller pads its rows to whole tiles (replaces in_proj / out_proj / c_fc / c_proj of
 *                    the text tower's ResidualAttentionBlocks, transformer.py:226-234,254-272)
 *   vl_attn_fwd_f16  vl_attn_fwd_bf16 on fp16 q, k, v -> fp16 out: head dim 64, <= 288 keys, causal or not
 *                    (F.multi_head_attention_forward with the additive causal mask, transformer.py:241-252,870-876)
 *   vl_layernorm_fwd with y_dtype = VL_F16 writes the GEMM inputs. */
int vl_gemm_f16(const void* A, const void* W, const float* bias, void* out, const void* res, int M, int N, int K,
                int lda, int ldw, int ldo, float alpha, int epi, int act, hipStream_t stream);
int vl_attn_fwd_f16(const void* q, const void* k, const void* v, const long* strides, void* out, float* lse,
                    int B, int H, int Lq, int Lk, int dh, float qscale, int causal, hipStream_t stream);

/* True fp32 arithmetic for INFERENCE (round 5): `precision="fp32"` of the reference's factory (open_clip/factory.py:260-295,
 * training/precision.py:5-12) means fp32 nn.Linear / attention products; rounds 1-4 ran bf16 operands there and said so in a
 * warning.  gfx950 has fp32-input MFMA at the fp32 vector rate (157 TFLOP/s, 1/16 of bf16), exact fmaf chains:
 *   vl_gemm_f32      out f32 [M,N] = act(alpha * A[M,K] W[N,K]^T + bias) (+ res f32 [M,N], in place allowed); any M, N;
 *                    K, lda, ldw multiples of 4; act none / GELU (erff) / ReLU        (csrc/vl_f32.hip, v_mfma_f32_32x32x2_f32)
 *   vl_attn_fwd_f32  softmax(scale * q k^T [+ causal mask]) v on strided f32 [B,H,L,dh] views (strides as vl_attn_fwd_bf16,
 *                    multiples of 4), dh = 32 or 64; out f32 [B,Lq,H*dh]; lse optional (natural log)
 *   vl_im2col_f32    vl_im2col_bf16 with f32 patches
 * LayerNorm, the token assembly and the text embedding already take f32 in and out.  Forward only: training under
 * precision="fp32" keeps bf16 operands with fp32 residual / gradient streams (vitlens_hip/f32.py says what runs). */
int vl_gemm_f32(const float* A, const float* W, const float* bias, float* out, const float* res, int M, int N, int K,
                int lda, int ldw, int ldo, float alpha, int act, hipStream_t stream);
int vl_attn_fwd_f32(const float* q, const float* k, const float* v, const long* strides, float* out, float* lse,
                    int B, int H, int Lq, int Lk, int dh, float scale, int causal, hipStream_t stream);
int vl_im2col_f32(const float* x, float* patches, int N, int C, int H, int W, int kh, int kw, int sh, int sw,
                  int Kp, int transpose_hw, hipStream_t stream);

int vl_device_info(int device, char* arch, int arch_len, int* cus, int* clock_khz, long* hbm_bytes);

/* Fused attention forward: softmax(q k^T [+ causal mask]) v without materialising the scores.
 * q, k, v are STRIDED [B,H,L,dh] views: element (b,h,l,d) of operand i at ptr_i[b*strides[3i] + h*strides[3i+1] +
 * l*strides[3i+2] + d] (strides in elements, multiples of 8; i = 0,1,2 for q,k,v) - normally the three column blocks of
 * the packed in-projection output [tokens, 3*width], read in place.  q is multiplied by qscale (softmax_scale*log2e;
 * pass 1 for a pre-scaled q) as it is loaded; V is transposed while it is staged into LDS.
 * out [B,Lq,H*dh] bf16 (token-major, ready for the out-projection); lse [B,H,Lq] optional (natural-log LSE of the
 * scaled scores, kept for the backward pass).  dh = 32, 64, or a multiple of 8 in (64, 128] (ViT-H/14: 80, ViT-bigG/14:
 * 104): those run zero-padded to 128 inside the kernel, only the real columns are read and written.
 * Replaces F.multi_head_attention_forward (transformer.py:241-252, causal mask :870-876) and
 * the Perceiver einsum attention (perceiver.py:128-145). */
int vl_attn_fwd_bf16(const void* q, const void* k, const void* v, const long* strides, void* out, float* lse,
                     int B, int H, int Lq, int Lk, int dh, float qscale, int causal, hipStream_t stream);

/* LayerNorm folded into the GEMMs either side of it (round 4).  For a FROZEN pre-LN block on a bf16 residual stream
 * (ResidualAttentionBlock, transformer.py:254-272: x + attn(ln_1(x)), x + mlp(ln_2(x)); LayerNorm transformer.py:17-34)
 *     LN(x) W^T + b  =  rstd_m (x (W gamma)^T - mean_m c) + (b + W beta),      c_n = sum_k (W gamma)[n, k]
 * so the consuming GEMM can read the raw residual rows and apply the row statistics in its epilogue, and the GEMM that
 * produced those rows can leave their partial sums behind: the normalised activations are never written or read.
 *   vl_gemm_main_rows          rows of an [M, K] x [N, K]^T problem that vl_gemm_bf16 (cfg -1) gives to the persistent
 *                              256x256-tile kernel; the entries below take exactly such row ranges (0: none)
 *   vl_gemm_lnfold_bf16        out bf16[M,N] = act(rstd_m * (A Wg^T - mean_m * ln_c) + bias_f); A = raw rows bf16 [M,K],
 *                              Wg = bf16(W * gamma) [N,K], bias_f = b + W beta f32 [N], ln_c f32 [N] (sums of the ROUNDED
 *                              Wg rows), ln_mean / ln_rstd f32 [M]; act = VL_ACT_NONE / VL_ACT_GELU / VL_ACT_GELU_DSAVE
 *                              (out2 = gelu', as vl_gemm_bf16_ex)
 *   vl_gemm_res_rowstats_bf16  out bf16 = res + A W^T + bias (VL_EPI_RES_BF16, in place allowed) and row_part f32
 *                              [M][N/64][2] = (sum, sum of squares) of the STORED bf16 values per row and 64-column slice
 *   vl_ln_row_stats            mean / rstd [rows]: rows < m_main from row_part (P = N/64 slices, summed in slice order),
 *                              rows >= m_main from the bf16 rows themselves (the leftover rows of a row-split GEMM); a row
 *                              whose E[x^2] - mean^2 would cancel (variance below 1e-3 of E[x^2], |mean| > ~30 sigma) is
 *                              also recomputed two-pass from its stored values: x_bf16 is always required.  y_left (optional,
 *                              with ln_w / ln_b): bf16 LayerNorm output of the rows >= y_row0 (>= m_main), row r at
 *                              y_left[(r - y_row0) * y_row_stride] - the rows the consuming GEMM runs as layernorm + small
 *                              tiles get their operand from this launch instead of one more 256-row LayerNorm launch
 * Whole 256x256 tiles, K >= 512, 16-byte aligned operands; anything else is refused (the caller keeps vl_layernorm_fwd +
 * vl_gemm_bf16 for it). */
int vl_gemm_main_rows(int M, int N);
int vl_gemm_lnfold_bf16(const void* A, const void* Wg, const float* bias_f, const float* ln_c, const float* ln_mean,
                        const float* ln_rstd, void* out, void* out2, int M, int N, int K, int lda, int ldw, int ldo, int act,
                        hipStream_t stream);
int vl_gemm_res_rowstats_bf16(const void* A, const void* W, const float* bias, void* out, const void* res, float* row_part,
                              int M, int N, int K, int lda, int ldw, int ldo, hipStream_t stream);
int vl_ln_row_stats(const float* row_part, int P, const void* x_bf16, long x_row_stride, int D, int m_main, int rows, float eps,
                    float* mean, float* rstd, const float* ln_w, const float* ln_b, void* y_left, long y_row_stride, int y_row0,
                    hipStream_t stream);

/* LayerNorm over the last dim (eps inside sqrt, biased variance): y = (x-mean)*rstd*w + b.
 * Source row for output row r is  r*row_mul + row_index[r]  when row_index != NULL (EOT gather,
 * model.py:539; cls pooling uses row_index == NULL with x_row_stride = (T+1)*D), else r.
 * Replaces LayerNorm/LayerNormFp32 (transformer.py:17-34) and PreNorm (perceiver.py:67-82). */
int vl_layernorm_fwd(const void* x, int x_dtype, long x_row_stride, const int64_t* row_index, long row_mul,
                     const float* w, const float* b, void* y, int y_dtype, long y_row_stride,
                     float* mean, float* rstd, int rows, int D, float eps, hipStream_t stream);

/* [cls ; tokens] + positional_embedding (+ adapter pos on rows 1..T) -> ln_pre, one pass.
 * tokens [B,T,D]; y [B,T+1,D].  Optional (training): xpre f32 [B,T+1,D] = the un-normalised rows,
 * mean/rstd [B*(T+1)].  Replaces transformer.py:756-772 (+ :734-745 for pos2). */
int vl_assemble_ln_pre(const void* tokens, int tok_dtype, const float* cls, const float* pos, const float* pos2,
                       const float* w, const float* b, void* y, int y_dtype, float* xpre, float* mean, float* rstd,
                       int B, int T, int D, float eps, hipStream_t stream);

/* F.normalize(dim=-1, eps): y f32 and/or bf16 copy; norms[rows] optional (model.py:522-540). */
int vl_l2_normalize(const float* x, float* y, void* y_bf16, float* norms, int rows, int D, float eps,
                    hipStream_t stream);
int vl_l2_normalize_bwd(const float* f, const float* df, const float* norms, float* dx, int rows, int D, float eps,
                        hipStream_t stream);

/* Patch extraction for Conv2d(bias=False, padding=0) as a GEMM operand: x [N,C,H,W] f32 ->
 * patches bf16 [N*gh*gw, Kp], column order (c,i,j), zero padded to Kp.  transpose_hw=1 reads the
 * stored tensor as [N,C,W,H] (AST_tokenizer.py:46-47).  Token order is row-major (gh, gw)
 * exactly as conv -> reshape(N,C,-1).permute(0,2,1) (transformer.py:674-676). */
int vl_im2col_bf16(const float* x, void* patches, int N, int C, int H, int W, int kh, int kw, int sh, int sw,
                   int Kp, int transpose_hw, hipStream_t stream);

/* out[b,l,:] = token_embedding[ids[b,l]] + positional_embedding[l]   (model.py:531-533) */
int vl_text_embed(const int64_t* ids, const float* tok_emb, const float* pos, void* out, int out_dtype,
                  int B, int L, int D, int vocab, hipStream_t stream);

int vl_cast_f32_bf16(const float* x, void* y, long n, hipStream_t stream);
/* y[r,:] = x[r,:] + table[r % T,:]   (x_vada["x"] + x_vada["pos"], transformer.py:743-745) */
int vl_add_rows(const void* x, int x_dtype, const float* table, void* y, int y_dtype, long rows, int T, int D,
                hipStream_t stream);
/* out[c,r] bf16 = in[r,c]; columns r in [R, ldo) zero-filled (operand prep for gradient GEMMs) */
int vl_transpose_to_bf16(const void* in, int in_dtype, long ldi, int R, int C, void* out, long ldo, hipStream_t stream);
/* The same for the weight-gradient path (C % 64 == 0, ldo % 8 == 0, ldi % 8 == 0, 16-byte aligned pointers), optionally
 * fused with the bias gradient: colsum_out[c] += colsum_scale * sum_r in[r, c] (deterministic two-stage sum; ws =
 * ceil(ldo/256)*C floats).  Autograd counterpart of nn.Linear's dW operands and bias gradient. */
int vl_transpose_colsum_bf16(const void* in, int in_dtype, long ldi, int R, int C, void* out, long ldo,
                             float* colsum_out, float colsum_scale, float* ws, hipStream_t stream);

/* InfoNCE pieces over logits f32 [R,C] (row r's positive is column r+label_off):
 *   vl_ce_stats      row_lse[R], col_lse[C], diag[R]; col_ws = 2*ceil(R/64)*C floats of workspace
 *   vl_ce_loss_accum *loss += w_row*mean(row_lse-diag) + w_col*mean(col_lse[r+off]-diag)
 *   vl_ce_grad       G[R,ldg] / GT[C,ldgt] bf16 = dLoss/dlogits (and transpose), *dscale += <G,logits>/scale (two-stage
 *                    fixed-order sum through ws = vl_ce_grad_ws_floats(R, C, ldg, ldgt) floats: deterministic)
 * Replaces F.cross_entropy(logits_per_x)+F.cross_entropy(logits_per_y) and autograd thereof
 * (loss.py:158-163, 300-306, 377-383). */
/* f32 [rows,D] -> bf16 [rows,3D] hi/lo split (pattern 0: hi|lo|hi, 1: hi|hi|lo) for near-fp32 logits */
int vl_split_bf16x3(const float* x, void* out, long rows, int D, int pattern, hipStream_t stream);
int vl_ce_stats(const float* logits, long ld, int R, int C, int label_off, float* row_lse, float* col_lse,
                float* diag, float* col_ws, hipStream_t stream);
int vl_ce_loss_accum(const float* row_lse, const float* col_lse, const float* diag, int R, int C, int label_off,
                     float w_row, float w_col, float* loss_inout, hipStream_t stream);
int vl_ce_grad(const float* logits, long ld, int R, int C, int label_off, const float* row_lse, const float* col_lse,
               float w_row, float w_col, void* G, long ldg, void* GT, long ldgt, float logit_scale,
               float* dscale_inout, float* ws, hipStream_t stream);
long vl_ce_grad_ws_floats(int R, int C, long ldg, long ldgt);

/* ---- backward / optimizer (trainable towers; autograd of the ops above) ---------------------- */
/* dx = dLN(dy) + dres (optional); outputs f32 `dx` and/or a bf16 copy `dx_bf16` (next GEMM operand). */
int vl_layernorm_bwd(const void* dy, int dy_dtype, long dy_stride, const void* x, int x_dtype, long x_stride,
                     const float* mean, const float* rstd, const float* w, const float* dres, float* dx,
                     void* dx_bf16, long dx_stride, int rows, int D, hipStream_t stream);
/* The same with the residual-gradient stream (dres in, dx out) in dtype g_dtype (VL_F32 | VL_BF16): bf16 is what the
 * reference's amp_bf16 autocast carries for the gradients of its bf16 residual stream. */
int vl_layernorm_bwd_g(const void* dy, int dy_dtype, long dy_stride, const void* x, int x_dtype, long x_stride,
                       const float* mean, const float* rstd, const float* w, const void* dres, void* dx, int g_dtype,
                       void* dx_bf16, long dx_stride, int rows, int D, hipStream_t stream);
/* Column reductions are deterministic two-stage sums (row slabs -> fixed-order combine, no atomics) and need a caller
 * workspace of vl_colreduce_ws_floats(rows, cols, planes) floats (planes: 2 for the LayerNorm parameters, 1 for vl_colsum). */
long vl_colreduce_ws_floats(int rows, int cols, int planes);
/* dw[j] += sum_r dy*xhat ; db[j] += sum_r dy   (accumulating; zero the buffers first) */
int vl_layernorm_bwd_params(const void* dy, int dy_dtype, long dy_stride, const void* x, int x_dtype, long x_stride,
                            const float* mean, const float* rstd, float* dw, float* db, int rows, int D,
                            float* ws, hipStream_t stream);
/* out[j] += scale * sum_r a[r,j]  (bias gradients) */
int vl_colsum(const void* a, int a_dtype, long lda, float* out, int rows, int cols, float scale, float* ws, hipStream_t stream);
int vl_gelu_bf16(const void* u, void* y, long n, hipStream_t stream);
/* y[m,j] = h[m,2j] * gelu(h[m,2j+1])  (recompute of the GEGLU output from the saved pre-activation) */
int vl_geglu_bf16(const void* h, void* y, long rows, int n_out, hipStream_t stream);
/* Attention backward (see csrc/vl_attn_bwd.hip).  q, k, v, dO, o are strided [B,H,L,dh] views (strides[15]: sb, sh, sr
 * in elements for q, k, v, dO, o in that order - multiples of 8), read in place: q/k/v out of the packed in-projection
 * output, dO out of the out-projection's input gradient [tokens, width], o = the forward's token-major output.  q is
 * multiplied by qscale (= scale*log2e) as it is loaded, as in the forward.  delta [B,H,Lq] fp32 is a caller workspace:
 * rowsum(dO*o) is produced by the first kernel and consumed by the second.  dq/dk/dv are token-major bf16 destinations
 * with row strides ld_dq / ld_dkv, already offset to their column block; scale = softmax scale. */
int vl_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* dO, const void* o, const long* strides,
                     const float* lse, float* delta, void* dq, void* dk, void* dv, long ld_dq, long ld_dkv,
                     int B, int H, int Lq, int Lk, int dh, float qscale, int causal, float scale, hipStream_t stream);
/* ONE-kernel attention backward for self-attention that fits a workgroup (csrc/vl_attn_bwd_fused.hip, round 4): head dim 64,
 * Lq == Lk == L, no causal mask, L <= 256 or L = 32 m + 1 <= 257.  Same operands and destinations as vl_attn_bwd_bf16
 * (strides[15]); delta is computed while dO is staged (no workspace).  Every score tile is evaluated once: 5 matrix products
 * and 16 exponentials per lane per 32x32 tile instead of 7 and 32.  Replaces the autograd of
 * F.multi_head_attention_forward (open_clip/transformer.py:241-252) and of the Perceiver's latent self-attention
 * (open_clip/perceiver.py:128-145).  vl_attn_bwd_fused_supported returns 1 when the fused entry takes the problem. */
int vl_attn_bwd_fused_supported(int Lq, int Lk, int dh, int causal);
int vl_attn_bwd_fused_bf16(const void* q, const void* k, const void* v, const void* dO, const void* o, const long* strides,
                           const float* lse, void* dq, void* dk, void* dv, long ld_dq, long ld_dkv, int B, int H, int L, int dh,
                           float qscale, float scale, hipStream_t stream);
/* ---- audio front end (csrc/vl_audio.hip) ----
 * Kaldi-compatible log-mel filterbank = torchaudio.compliance.kaldi.fbank(htk_compat, 16 kHz, hanning window, 128 bins,
 * no dither, 25 ms / 10 ms frames, snip_edges) + zero-padding / truncation to target_len rows + Normalize(mean, std), i.e.
 * AudioASTProcessorEval.convert2fbank + transform (open_clip/modal_audio/processors/at_processor.py:839-873).
 * wave [batch, n_samples] f32 (row stride wave_stride), window [win] f32, banks [nmel, nfft/2+1] f32 (host-built, device
 * resident), out [batch, target_len, nmel] f32.  Parity unpinned: torchaudio is not available to check against. */
int vl_kaldi_fbank(const float* wave, long wave_stride, int batch, long n_samples, const float* window, const float* banks,
                   float* out, int target_len, int win, int shift, int nfft, int nmel, float preemph, float mean, float std,
                   hipStream_t stream);
/* ---- point-cloud tokenizer (PointBERT grouping) ---- */
/* farthest point sampling: xyz [B,N,3] f32, start [B] (the reference draws it with torch.randint, misc.py:60);
 * idx [B,G] int64 (bit-exact vs misc.fps), centers [B,G,3] optional. */
int vl_fps(const float* xyz, const int64_t* start, int64_t* idx, float* centers, int B, int N, int G, hipStream_t stream);
/* pc_norm after a gather (modal_3d/processors/pc_processor.py:32-38, PCProcessorEval :60-88): out [B,G,C] f32 =
 * (pts[b, idx[b,g], :] - centroid) / max distance from the centroid over the G selected points; idx NULL = all N points. */
int vl_pc_gather_normalize(const float* pts, const int64_t* idx, float* out, int B, int N, int G, int C, hipStream_t stream);
/* ---- evaluation-time image / depth preprocessing (csrc/vl_preproc.hip) ----
 * Separable bicubic resampling restricted to a crop window, tables built by the host (vitlens_hip/preproc.py):
 * bounds [n_out,2] int32 = (first tap, tap count) and coefficients [n_out,ksize] per OUTPUT index of the full resized
 * axis; xout0/nxout (yout0/nyout) select the crop.  8-bit path = Pillow's Image.resize(BICUBIC) (22-bit fixed-point
 * int32 taps, 8-bit intermediate; byte-exact), replacing torchvision Resize+CenterCrop+ToTensor+Normalize of
 * open_clip/transform.py:138-155: the vertical pass writes out [C,nyout,W] f32 = (u8/255 - mean[c]) / std[c] (mean/std:
 * HOST arrays of C floats) and/or the resized bytes out_u8 [nyout,W,C]; W = source width of the horizontal pass, and in
 * the vertical pass src is the [nrows, W, C] intermediate whose row 0 is image row row0 (nothing beyond it is read).  Float path = ATen bicubic (antialiased or the
 * 4-tap border-clamped form) with DepthNorm's clamp(lo,hi)/divide_by fused into the read, replacing
 * modal_depth/processors/vt_processor.py:292-337; taps are clamped to the image, so bounds may start at -1. */
int vl_resample_h_u8(const uint8_t* src, long row_stride, int W, int C, int row0, int nrows, const int* bounds,
                     const int* kk, int ksize, int xout0, int nxout, uint8_t* dst, hipStream_t stream);
int vl_resample_v_u8_norm(const uint8_t* src, int W, int C, int row0, int nrows, const int* bounds, const int* kk,
                          int ksize, int yout0, int nyout, const float* mean, const float* stdv, float* out,
                          uint8_t* out_u8, hipStream_t stream);
/* The 8-bit path for a LIST of images of different sizes in two launches (blockIdx.z = image).  desc: device array
 * [n][16] int64 per image = {src pointer, row stride in bytes, source width, row0, nrows, xout0, yout0, horizontal
 * bounds offset, horizontal coefficient offset, horizontal ksize, vertical bounds offset, vertical coefficient offset,
 * vertical ksize, byte offset of the image's [nrows, crop_w, C] intermediate in tmp, 0, 0}; offsets count ints into
 * `tables` (the per-axis bounds / coefficient tables of the distinct sizes, packed).  out [n, C, crop_h, crop_w] f32. */
int vl_resample_batch_u8_norm(const int64_t* desc, int n, int C, int crop_h, int crop_w, int max_nrows, const int* tables,
                              uint8_t* tmp, const float* mean, const float* stdv, float* out, hipStream_t stream);
int vl_resample_h_f32(const float* src, long row_stride, int W, int row0, int nrows, const int* bounds,
                      const float* weights, int ksize, int xout0, int nxout, int clamp_on, float clamp_lo,
                      float clamp_hi, float divide_by, float* dst, hipStream_t stream);
int vl_resample_v_f32_norm(const float* src, int W, int H, int row0, const int* bounds, const float* weights, int ksize,
                           int yout0, int nyout, float mean, float stdv, float* out, hipStream_t stream);
/* k nearest neighbours of each centre (set semantics = topk(sorted=False), dvae.py:107-118) + gather +
 * centre subtraction: nidx [B,G,k] int32 optional, patches bf16 [B*G*k, Kp] optional (xyz in cols 0..2). */
int vl_knn_group(const float* xyz, const int64_t* center_idx, int* nidx, void* patches, int B, int N, int G,
                 int k, int Kp, hipStream_t stream);
/* Ball query + grouping of the `pnsa` tokenizer (query_ball_point + sample_and_group,
 * open_clip/modal_3d/models/pointnet/pointnet_util.py:101-161): for every centre center_idx[b,s] the first `nsample`
 * point indices in ascending order with -2ab+|a|^2+|b|^2 <= radius2 (short groups repeat their first index) -> idx
 * [B,S,nsample] int32 (optional) and patches [B*S*nsample, Kp] bf16 (optional) = (xyz_j - centre, feats[b,j,0:D], 0...):
 * the rows of the first 1x1 convolution.  xyz [B,N,3] f32, feats [B,N,D] f32 or NULL with D = 0. */
int vl_ball_group(const float* xyz, const float* feats, const int64_t* center_idx, int* idx, void* patches, int B, int N,
                  int S, int D, float radius2, int nsample, int Kp, hipStream_t stream);
int vl_group_max(const void* x, long ldx, void* out, int out_dtype, long ldo, long groups, int M, int C, hipStream_t stream);
int vl_pad3_bf16(const float* c, void* out, long R, int Kp, hipStream_t stream);
/* ---- trainable point tokenizer: nn.BatchNorm1d over [R = B*G*M, C] bf16 activations (dvae.py:184-194) ----
 * ws: caller workspace of (nchunk + 1) * 2 * C floats (row-chunk partial sums; deterministic, no atomics).
 * vl_bn_stats: biased batch mean/var per column; running_mean/var (optional) updated as nn.BatchNorm1d does
 * (momentum, unbiased variance).  vl_bn_apply: y = gamma*(x-mean)/sqrt(var+eps)+beta (+ReLU); pass the batch
 * statistics (train) or the running statistics (eval).  vl_bn_bwd: dgamma/dbeta are ACCUMULATED; dx (optional)
 * is the input gradient; train=0 treats mean/var as constants (eval-mode BN). */
int vl_bn_stats(const void* x, long ldx, int R, int C, float* ws, int nchunk, float* mean, float* var,
                float* running_mean, float* running_var, float momentum, hipStream_t stream);
/* SyncBatchNorm (torch.nn.SyncBatchNorm under --use-bn-sync, training/point_cloud/pc_tri_main.py:372-373): the two
 * passes above split where the ranks exchange data.  Forward: vl_bn_stats_local -> local [2C+1] floats = per-column
 * mean, M2 = sum (x-mean)^2, and the row count as int bits; all-gather to [W, 2C+1]; vl_bn_stats_merge -> global
 * mean / biased var (+ running statistics with the global unbiased variance) and *total = global row count.
 * Backward: vl_bn_bwd_reduce accumulates dgamma/dbeta (local, as SyncBatchNorm does) and writes sums [2C] =
 * (sum dy', sum dy'*xhat); all-reduce(sum) them; vl_bn_bwd_apply -> dx with the global sums and count. */
int vl_bn_stats_local(const void* x, long ldx, int R, int C, float* ws, int nchunk, float* local, hipStream_t stream);
int vl_bn_stats_merge(const float* gathered, int W, int C, float* mean, float* var, float* running_mean,
                      float* running_var, float momentum, int* total, hipStream_t stream);
int vl_bn_bwd_reduce(const void* dy, long lddy, const void* x, long ldx, const float* mean, const float* var,
                     const float* gamma, const float* beta, float eps, int relu, float* ws, int nchunk, float* dgamma,
                     float* dbeta, float* sums, int R, int C, hipStream_t stream);
int vl_bn_bwd_apply(const void* dy, long lddy, const void* x, long ldx, const float* mean, const float* var,
                    const float* gamma, const float* beta, float eps, int relu, const float* sums, const int* total,
                    void* dx, long lddx, int R, int C, hipStream_t stream);
int vl_bn_apply(const void* x, long ldx, const float* mean, const float* var, const float* gamma, const float* beta,
                float eps, int relu, void* out, long ldo, long R, int C, hipStream_t stream);
int vl_bn_bwd(const void* dy, long lddy, const void* x, long ldx, const float* mean, const float* var,
              const float* gamma, const float* beta, float eps, int relu, int train, float* ws, int nchunk,
              float* dgamma, float* dbeta, void* dx, long lddx, int R, int C, hipStream_t stream);
/* backward of torch.max over the M rows of each group (gradient to the first arg-max row), added to `base`
 * (optional); and the sum over the M rows of each group (backward of the expand in dvae.py:207-208). bf16. */
int vl_group_max_bwd(const void* f, long ldf, const void* dg, long lddg, const void* base, long ldb, void* out,
                     long ldo, long groups, int M, int C, hipStream_t stream);
int vl_group_sum(const void* x, long ldx, void* out, long ldo, long groups, int M, int C, hipStream_t stream);
/* torch.optim.AdamW step on one tensor (grad is multiplied by grad_scale first); step counts from 1. */
int vl_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, float grad_scale, hipStream_t stream);
int vl_clamp_scalar(float* p, float lo, float hi, hipStream_t stream);
int vl_axpy_f32(float* y, const float* x, float alpha, long n, hipStream_t stream);
/* out[i] = x[i] * exp(log_scale[0]) * mul (in place allowed): `logit_scale.exp()` (open_clip/model.py:619, loss.py:125-127)
 * applied on the device, so that a training step never reads the temperature on the host (hipGraph-capturable). */
int vl_scale_exp_f32(const float* x, float* out, long n, const float* log_scale, float mul, hipStream_t stream);
/* out[t,:] += sum_b x[b*batch_stride_rows + row_offset + t, :]   (positional-embedding gradients) */
int vl_batch_rowsum(const float* x, float* out, int B, int T, int D, long batch_stride_rows, long row_offset,
                    hipStream_t stream);

/* ---- the contrastive exchange over RCCL (csrc/vl_comm.cpp) ----
 * One process per GPU, one communicator per process; RCCL (librccl.so.1) is resolved at run time.  These are the three
 * collectives of a training step of the hot path, for a host that is not Python (the Python host makes the same calls
 * through torch.distributed, or - vitlens_hip.step.AbiComm - through these entries):
 *   vl_allgather_embed     the packed unit features of the step, [b, k*E] f32 per rank -> [W*b, k*E] in rank order
 *                          (replaces gather_features' 2-4 dist.all_gather calls, open_clip/loss.py:20-78); count = floats per rank
 *   vl_reducescatter_grad  backward of that gather under --gather-with-grad: full [W*b, E] f32 -> this rank's [b, E] slice of the
 *                          sum over ranks (loss.py:55-61: torch.distributed.nn.all_gather's autograd)
 *   vl_allreduce_grad      sum of a flat fp32 gradient bucket over the ranks, in place (DistributedDataParallel's all-reduce;
 *                          the 1 / world of its mean is applied by the optimizer step, vl_adamw_step's grad_scale)
 * vl_comm_unique_id fills 128 bytes on ONE rank; the host ships them to the others (its own channel: file, socket, MPI) and
 * every rank calls vl_comm_create (collective).  All calls are asynchronous on `stream`. */
typedef struct vl_comm* vl_comm_t;
int vl_comm_unique_id(void* id128);
int vl_comm_create(vl_comm_t* comm, const void* id128, int rank, int world);
int vl_comm_destroy(vl_comm_t comm);
int vl_allgather_embed(vl_comm_t comm, const float* local, float* gathered, long count, hipStream_t stream);
int vl_reducescatter_grad(vl_comm_t comm, const float* full, float* mine, long count_per_rank, hipStream_t stream);
int vl_allreduce_grad(vl_comm_t comm, float* buf, long count, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VITLENS_HIP_H */

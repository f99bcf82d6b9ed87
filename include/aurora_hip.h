/*
 * aurora_hip.h -- C ABI of libaurora_hip.so, the MI355X (gfx950) compute library behind
 * aurora_amd.Aurora.forward / rollout.
 *
 * The reference (microsoft/aurora) has no FFI: its operator boundary is the torch.nn.Module
 * call interface.  Every entry point below therefore replaces one
 * torch-level operator group of the reference's hot path and cites it.  All pointers are raw
 * DEVICE pointers (hipMalloc'ed / torch-allocated), `stream` is a hipStream_t passed as void*,
 * sizes are element counts unless stated, strides ("ld*") are in elements.  No torch types.
 * Every function only enqueues work on `stream` and returns 0 on success or a negative
 * AURORA_E_* code; aurora_hip_last_error() describes the last failure of the calling thread.
 *
 * dtype codes: AURORA_F32 = 0 (float), AURORA_BF16 = 1 (bfloat16, raw uint16 storage).
 */
#ifndef AURORA_HIP_H
#define AURORA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AURORA_F32 0
#define AURORA_BF16 1

#define AURORA_OK 0
#define AURORA_E_ARG (-1)     /* invalid argument (shape / alignment / dtype) */
#define AURORA_E_LAUNCH (-2)  /* HIP launch error */

#define AURORA_ACT_NONE 0
#define AURORA_ACT_GELU 1 /* exact erf GELU (torch.nn.GELU default) */
#define AURORA_ACT_SILU 2 /* x * sigmoid(x) (time_mlp / AdaLN modulation, swin3d.py:805-809, film.py:28) */

const char* aurora_hip_last_error(void);
int aurora_hip_version(void);

/* ---- dense linear layers ------------------------------------------------------------------
 * C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ residual[M,N]);  W is nn.Linear's (out,in)
 * row-major weight.  A, W, C share `dtype`; accumulation is fp32 on the MFMA units
 * (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32).  bias and residual are fp32 (nullable).
 * C2 (nullable) receives a second copy of the result in the OTHER dtype (fp32 <-> bf16).
 * ldr == 0 broadcasts one residual row to every output row.
 * Constraints: K * sizeof(dtype) % 128 == 0 (zero-pad K otherwise); lda/ldw multiples of
 * 16 bytes; operand pointers 16-byte aligned.  Outputs whose rows are not 16-byte aligned take
 * a scalar store path.
 * Replaces F.linear call sites: swin3d.py:59-66,153,169,554,609-612, perceiver.py:79-88,
 * 141-152, encoder.py:318-363, decoder.py:214-263, film.py:48, patchembed.py:112 (as GEMM
 * over unfolded patches).
 */
int aurora_hip_linear(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                      void* C, int64_t ldc, void* C2, int64_t ldc2,
                      const float* residual, int64_t ldr,
                      int64_t M, int N, int K, int dtype, int act, void* stream);

/* aurora_hip_linear with the fp32 numerics chosen PER CALL (the library keeps no mutable state):
 * f32_gemm says how a large fp32 linear (N % 256 == 0) is multiplied; -1 takes the process default.
 *   1 (default): every fp32 operand is split exactly into three bf16 terms and six bf16 MFMAs per K-slab
 *      reproduce the fp32 product to ~1e-7 relative (fp32-grade, no range restriction, 16/6 of the fp32 MFMA rate).
 *   2: two fp16 terms (round-to-nearest: a_h + a_l = a to 2^-24 |a|), three fp16 MFMAs, the weight operand scaled
 *      by 2^6 and the result scaled back exactly.  Same accuracy, but inside fp16's range only: activations must
 *      satisfy |x| < 65504 (|x| < 0.25 carries an absolute error of up to 3e-8), weights |w| < 1000.  Meant for
 *      linears whose input is bounded by construction (a LayerNorm output, the GELU of a linear of one), whatever
 *      the model inputs are.  With `guard` != NULL the decision is taken on the device, per launch: the two-term
 *      split runs only if *guard < guard_limit, else the three-term bf16 split -- the caller leaves max |activation|
 *      (or an upper bound of it, scaled into guard_limit) there, e.g. with aurora_hip_absmax, without any host
 *      synchronisation.
 *   0: native v_mfma_f32_16x16x4_f32 FMA chains (bitwise an fp32 FMA chain; erff in the GELU epilogue).
 * The environment variable AURORA_F32_GEMM=native|bf16|f16 sets the process default, which
 * aurora_hip_default_f32_gemm() reports.  The reference's counterpart is the fp32 F.linear outside autocast
 * (encoder.py / decoder.py, aurora.py:322-349). */
int aurora_hip_linear_ex(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                         void* C, int64_t ldc, void* C2, int64_t ldc2,
                         const float* residual, int64_t ldr,
                         int64_t M, int N, int K, int dtype, int act,
                         int f32_gemm, const float* guard, float guard_limit, void* stream);
int aurora_hip_default_f32_gemm(void);
/* `batch` independent problems of the same shape in ONE launch (strided batch): problem g reads A + g * stride_a,
 * W + g * stride_w, bias + g * stride_bias and writes C + g * stride_c (strides in elements; every problem stays 16-byte
 * aligned; stride 0 shares an operand).  Same kernels, modes and constraints as aurora_hip_linear_ex, no second output /
 * residual.  Replaces the reference's per-level Python loops: LevelConditioned patch embeddings and heads
 * (levelcond.py:36-69: one weight per pressure level) and the per-level bias of the atmospheric patch embedding
 * (encoder.py:318-330). */
int aurora_hip_linear_batched(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                              int64_t M, int N, int K, int dtype, int act, int f32_gemm, const float* guard,
                              float guard_limit, int batch, int64_t stride_a, int64_t stride_w, int64_t stride_bias,
                              int64_t stride_c, void* stream);

/* aurora_hip_linear for callers that can lend scratch memory: a bf16 linear with few 256 x 256 tiles and a long K (the
 * per-rank shapes of a latitude band at the coarse backbone stages, swin3d.py:59-66,153,169 at M = 2,160: 72 tiles on
 * 256 CUs, K = 8192) is then split along K inside ONE launch -- every tile's K range is cut into `split` slices, one
 * workgroup each; slices publish their fp32 accumulators to `workspace` and take a ticket of their tile, and whoever
 * draws a tile's last ticket adds the slices up in slice order and runs the epilogue.  No workgroup waits for another
 * (nothing depends on dispatch order), the result does not depend on arrival order; it differs from the un-split
 * product by the rounding of split - 1 fp32 additions per element.
 *   aurora_hip_linear_workspace: bytes of scratch the library would like for this shape (0: it would not split).
 *   workspace, workspace_bytes : 16-byte aligned scratch, contents irrelevant; too small / NULL = no split.
 *   tickets, n_tickets         : int32 words, ZERO on entry and left zero (one per tile); fewer than tiles = no split.
 *                                NOT validated: a count left behind by a launch that never finished (a device fault, a
 *                                torn-down graph) means no workgroup of that tile ever draws its last ticket -- the tile's
 *                                output rows are never written, silently.  A device-side abort returns normally to the
 *                                host, so "after a failed launch" cannot be told: re-zero the words (hipMemsetAsync) ahead
 *                                of a batch of launches -- the model handle does so at the start of EVERY sharded step.
 *                                Two launches that may run at the same time need their own ticket words.
 *   split                      : 0 lets the library choose, 1 forbids, > 1 asks for that many slices.
 * fp32 problems and shapes that would not split run exactly as aurora_hip_linear. */
int64_t aurora_hip_linear_workspace(int64_t M, int N, int K, int dtype);
int aurora_hip_linear_ws(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                         void* C2, int64_t ldc2, const float* residual, int64_t ldr, int64_t M, int N, int K, int dtype,
                         int act, void* workspace, int64_t workspace_bytes, int32_t* tickets, int n_tickets, int split,
                         void* stream);

/* The qkv linear of a Swin block (swin3d.py:153) with its result in HEAD PLANES instead of rows: the 64-column blocks of
 * C = A W^T + bias are q, k, v of the attention heads (block sel * heads + h, swin3d.py:154-156 reshapes exactly these);
 * head h owns a plane of [M rows][q | k | v = 192 elements] bf16, `plane_stride` elements (>= 192 M, a multiple of 8)
 * behind the previous head's.  The window attention gathers q, k and v of one (token, head) at a time: in a plane that is
 * 384 contiguous bytes, and a window's runs of consecutive tokens are contiguous runs of DRAM; in rows of 3 D elements
 * they are three 128-byte pieces D * 2 bytes apart, every token 3 D * 2 bytes further (aurora_hip_window_attention_planes
 * reads this layout).  N = 64 heads x 3, or x 2 / x 1 with C moved 64 / 128 elements into the row (k | v of halo rows).
 * bf16 only; same arithmetic and bits as aurora_hip_linear. */
int aurora_hip_linear_planes(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                             int64_t plane_stride, int heads, int64_t M, int N, int K, int dtype, void* stream);

/* Mode 2 with operands that are ALREADY split (flags OR-ed into f32_gemm = 2): the split is the same arithmetic wherever
 * it happens, so results are bit-identical to plain mode 2, but a GEMM whose operands arrive split spends no VALU work
 * on them -- the two-term kernel goes from 0.29 to 0.40 PFLOP/s fp32-equivalent when both do.
 * The fp16-pair layout of a row of K fp32 values (K % 32 == 0) occupies the same K * 4 bytes: per group of 32 features
 * 128 bytes, first the 32 fp16 high halves h = fp16(x), then the 32 fp16 remainders l = fp16(x - h); strides stay in
 * 4-byte units.  aurora_hip_split_f16 produces it (weights: scale = 64, the 2^6 of mode 2), aurora_hip_layernorm_split
 * and a linear with AURORA_F32_C_SPLIT write it directly.
 *   AURORA_F32_W_SPLIT: W is in the pair layout, scaled by 64.  With a guard the launch runs iff *guard < guard_limit
 *     and has NO fallback of its own: pair it with an f32_gemm = 1 call on the fp32 weights carrying the same guard
 *     (mode 1 with a guard runs iff the guard FAILS).
 *   AURORA_F32_A_SPLIT: A is in the pair layout (unscaled); needs AURORA_F32_W_SPLIT.  With a guard it is for a buffer whose
 *     FORMAT was decided by that same guard and limit on the device: written as pairs by a guarded two-term producer
 *     (AURORA_F32_C_SPLIT) iff the guard holds, as fp32 by its mode-1 twin otherwise -- consumer and producer switch together.
 *   AURORA_F32_C_SPLIT: C is written in the pair layout (after bias / activation / residual); ldc % 32 == 0, no C2.
 * Shapes: N % 256 == 0, K % 32 == 0, K >= 96. */
#define AURORA_F32_A_SPLIT 4
#define AURORA_F32_W_SPLIT 8
#define AURORA_F32_C_SPLIT 16
int aurora_hip_split_f16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int K,
                         float scale, void* stream);

/* out[0] = max |x[i]| over n contiguous fp32 values (x 16-byte aligned); NaNs are ignored.
 * aurora_hip_absmax_fold: out[0] = max(out[0], max |x[i]|) -- no zeroing, so that several producers can share one word;
 * aurora_hip_zero_words clears n <= 64 such words with one tiny launch (a step clears all of its guard words at once:
 * a 4-byte hipMemsetAsync in front of every absmax cost a latitude band ~90 us each). */
int aurora_hip_absmax(const float* x, int64_t n, float* out, void* stream);
int aurora_hip_absmax_fold(const float* x, int64_t n, float* out, void* stream);
int aurora_hip_zero_words(float* words, int n, void* stream);

/* ---- 3D shifted-window attention core ------------------------------------------------------
 * For every window w and head h: O = softmax(Q K^T / sqrt(hd) + mask) V over the window's
 * `win_tokens` (<= 144) tokens.  qkv: [B][L][3*D] with columns q | k | v, each head-major
 * (swin3d.py:153-155).  The roll / pad / window-partition / reverse / crop / un-roll index
 * permutations of swin3d.py:471-505 are folded into `tok` (int32 [n_windows][win_tokens],
 * token index inside one batch element or -1 for a zero-padded position; padded positions
 * carry q = k = v = qkv_bias, exactly as Linear(0) does in the reference).  `grp`
 * (uint8 [n_windows][win_tokens], nullable) holds the communication-group labels of
 * compute_3d_shifted_window_mask (swin3d.py:303-360): score += -100 where labels differ.
 * Output O: [B][L][D], token order, padded positions are not written.  head_dim must be 64.
 * Replaces WindowAttention.forward's SDPA (swin3d.py:154-168) + swin3d.py:177-285,471-505.
 * Latitude-band sharding: a rank passes qkv = [owned rows | halo rows received from its
 * neighbours] (L rows) and L_out = number of owned rows; outputs are written for tokens < L_out
 * only (out has L_out rows per batch element).  Un-sharded callers pass L_out = L.
 */
int aurora_hip_window_attention(const void* qkv, const float* qkv_bias, void* out,
                                const int32_t* tok, const uint8_t* grp,
                                int B, int64_t L, int64_t L_out, int D, int heads, int n_windows,
                                int win_tokens, int dtype, void* stream);
/* The same with q | k | v in head planes (aurora_hip_linear_planes): head h owns [B * L rows][q | k | v = 192 elements]
 * bf16, planes `plane_stride` elements apart (>= 192 B L); row b * L + t is token t of batch element b.
 * plane_stride = 0: the row layout above.  Same results, bit for bit. */
int aurora_hip_window_attention_planes(const void* qkv, int64_t plane_stride, const float* qkv_bias, void* out,
                                       const int32_t* tok, const uint8_t* grp,
                                       int B, int64_t L, int64_t L_out, int D, int heads, int n_windows,
                                       int win_tokens, int dtype, void* stream);

/* ---- (adaptive) layer norm with residual --------------------------------------------------
 * out[r,:] = res[r % res_mod? ,:] + LN(y[r,:]) * gain[:] + shift[:]   (res nullable)
 * y: [M][D] of `dtype` with row stride ldy; statistics in fp32, eps as given.
 * gain/shift: fp32 [D] (AdaLN: scale_bias + scale, shift -- film.py:48-49; affine LN: weight,
 * bias).  res: fp32, row stride ldr; row index is r, or r % res_mod when res_mod > 0
 * (Perceiver latents broadcast over grid columns, perceiver.py:224-232).
 * out_f32 (nullable, stride ldo) and out_t (nullable, `dtype`, stride ldt) receive the result.
 * Replaces swin3d.py:507-508, perceiver.py:224-232, encoder.py:320, perceiver.py:144-147.
 */
int aurora_hip_layernorm(const void* y, int64_t ldy, const float* gain, const float* shift,
                         const float* res, int64_t ldr, int64_t res_mod,
                         float* out_f32, int64_t ldo, void* out_t, int64_t ldt,
                         int64_t M, int D, float eps, int dtype, void* stream);
/* bf16 linear + (adaptive) LayerNorm + residual in ONE launch, for N = D = 512 (stage 0 of the backbone):
 *     x_out = x_in + LN(A W^T + bias) * gain + shift,     x_bf16 = bf16(x_out)   (nullable)
 * i.e. `x = shortcut + norm1(proj(attn), c)` and `x = x + norm2(mlp(x), c)` of a Swin block (swin3d.py:507-508,
 * film.py:38-49) without writing the linear's result to memory.  A, W bf16 (K contiguous, K % 32 == 0, K >= 96); the
 * linear's result is rounded to bf16 before the statistics, as the reference's autocast linear is; fp32 statistics
 * (two passes), fp32 residual stream; x_out may alias x_in.  One workgroup owns 128 whole rows. */
int aurora_hip_linear_layernorm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                const float* gain, const float* shift,
                                const float* x_in, int64_t ldx, float* x_out, int64_t ldo,
                                void* x_bf16, int64_t ldb, int64_t M, int N, int K, float eps, void* stream);

/* The same for fp32 rows, with the fp16-pair layout of the two-term GEMMs (see AURORA_F32_A_SPLIT) on either side: the
 * LayerNorm in front of a Perceiver MLP hands its result to fc1 already split (out_split; out_f32 may then be NULL),
 * and the LayerNorm behind the MLP takes that same array as its residual (res_is_split = 1: the residual value is
 * high half + remainder, equal to the fp32 value to 2^-23 relative) -- the fp32 copy is never written.
 * D % 32 == 0; pair-layout strides % 32 == 0 (4-byte units).  Either output may be NULL, not both. */
int aurora_hip_layernorm_split(const float* y, int64_t ldy, const float* gain, const float* shift,
                               const void* res, int64_t ldr, int64_t res_mod, int res_is_split,
                               float* out_f32, int64_t ldo, void* out_split, int64_t ld_split,
                               int64_t M, int D, float eps, void* stream);

/* ---- patch merging: 2x2 gather + LayerNorm(4D) -------------------------------------------
 * x: fp32 residual stream [B][C][H][W][D] -> out `dtype` [B][C][H2][W2][4D], H2 = ceil(H/2),
 * features ordered (h, w, D), zero padding at the bottom/right (swin3d.py:526-553).
 * The following Linear(4D, 2D) is an aurora_hip_linear call.
 */
int aurora_hip_merge_ln(const float* x, const float* ln_w, const float* ln_b, void* out,
                        int B, int C, int H, int W, int D, float eps, int dtype, void* stream);

/* ---- patch splitting: pixel shuffle + crop + LayerNorm(D/2) ------------------------------
 * y: `dtype` [B][C][H][W][2D'] with 2D' = 4*Dq (output of lin1) -> out `dtype`
 * [B][C][2H-crop_h][2W-crop_w][Dq]  (swin3d.py:574-611).
 */
int aurora_hip_split_ln(const void* y, const float* ln_w, const float* ln_b, void* out,
                        int B, int C, int H, int W, int Dq, int crop_h, int crop_w, float eps,
                        int dtype, void* stream);

/* ---- patch embedding front end: normalise + unfold ----------------------------------------
 * Builds the GEMM operand of LevelPatchEmbed (patchembed.py:100-115):
 * out[((c*B + b)*Hp + hp)*Wp + wp][k_offset + (v*T + t)*P*P + i*P + j] =
 *     f_v( (src_v[b,t,c,hp*P+i,wp*P+j] - loc_v[c]) * inv_scale_v[c] )
 * rows are ordered (level, batch, patch) so that one level's rows are contiguous (per-level
 * weights / biases are then plain row ranges).  The Batch.normalise affine map
 * (batch.py:94-116) is fused in.  Columns [K_total, Kpad) are zero-filled by the call whose
 * variables end at K_total.  `desc` is a HOST array of n_vars (<= 32) descriptors; their
 * `src`, `loc`, `inv_scale` members are device pointers.  Strides are in elements and may be
 * 0 (static variables broadcast over batch/history/level; constant planes).
 * transform codes: 0 none, 1 clamp(min=0) (aurora.py:301-317), 2 clamp + air-pollution
 * feature combiner (aurora.py:733-742) with Linear(2,1) weights tw0, tw1 and bias tb;
 * ocean-wave channels (aurora.py:892-912, NaN = missing): 3 density = !isnan, 4 value with
 * NaN -> 0, 5 sin(deg2rad(x)) and 6 cos(deg2rad(x)), both with NaN -> 0.
 */
typedef struct aurora_patch_var {
  const float* src;       /* device: first element of the variable                           */
  int64_t stride_b;       /* element strides: batch, history step, level, latitude, longitude */
  int64_t stride_t;
  int64_t stride_c;
  int64_t stride_h;
  int64_t stride_w;
  const float* loc;       /* device [n_lvl] location                                          */
  const float* inv_scale; /* device [n_lvl] 1/scale                                           */
  int32_t transform;
  float tw0, tw1, tb;
} aurora_patch_var;

int aurora_hip_patchify(const aurora_patch_var* desc, int n_vars, void* out, int64_t Kpad,
                        int k_offset, int K_total, int B, int T, int n_lvl, int Hp, int Wp, int P,
                        int dtype, void* stream);
/* The same, and max |value written| is folded into *absmax (nullable; not zeroed here: aurora_hip_zero_words) -- the
 * guard word of the operand-split decision of the linears that read `out` (aurora_hip_linear_ex), at no extra pass. */
int aurora_hip_patchify_absmax(const aurora_patch_var* desc, int n_vars, void* out, int64_t Kpad,
                               int k_offset, int K_total, int B, int T, int n_lvl, int Hp, int Wp, int P,
                               int dtype, float* absmax, void* stream);

/* ---- small-set cross attention of the Perceiver resamplers ---------------------------------
 * Per grid column col in [0, n_cols) and head: softmax(q k^T / sqrt(hd)) v over Lk keys.
 * q: fp32/bf16 [Lq][inner] shared by all columns when q_col_stride == 0, else rows
 * (col*Lq + i).  kv: key j of column (b, l) is one row, columns k | v
 * (2*inner), row index b*kv_bstride + j*kv_lstride + l.  out rows (col*Lq + i), [inner].  Replaces perceiver.py:141-152 as used by
 * encoder.py:184-196 (Lq=3, Lk=levels) and decoder.py:156-166 (Lq=levels, Lk=3).
 */
int aurora_hip_perceiver_attention(const void* q, int64_t q_col_stride, const void* kv, void* out,
                                   int B, int64_t cols_per_b, int64_t kv_bstride, int64_t kv_lstride,
                                   int Lq, int Lk, int heads, int head_dim, int dtype, void* stream);
/* The same with a device-side choice of the OUTPUT format (fp32 only): the rows are written as fp16 pairs (see
 * AURORA_F32_A_SPLIT) iff *pair_guard < pair_limit, as fp32 otherwise -- the guard and limit of the guarded two-term
 * linear that reads them next (to_out), so that producer, consumer and its three-term twin switch together. */
int aurora_hip_perceiver_attention_ex(const void* q, int64_t q_col_stride, const void* kv, void* out,
                                      int B, int64_t cols_per_b, int64_t kv_bstride, int64_t kv_lstride,
                                      int Lq, int Lk, int heads, int head_dim, int dtype,
                                      const float* pair_guard, float pair_limit, void* stream);
/* The same as one half of a device-side choice between this kernel and the re-associated pair below: the launch retires
 * at once iff *skip_guard < skip_limit (null: never). */
int aurora_hip_perceiver_attention_unless(const void* q, int64_t q_col_stride, const void* kv, void* out,
                                          int B, int64_t cols_per_b, int64_t kv_bstride, int64_t kv_lstride,
                                          int Lq, int Lk, int heads, int head_dim, int dtype,
                                          const float* pair_guard, float pair_limit, const float* skip_guard,
                                          float skip_limit, void* stream);

/* ---- the decoder's level de-aggregation, re-associated (perceiver.py:141-152 + its `to_out`, decoder.py:156-166, 225-231)
 * With queries shared by all columns (q_col_stride == 0) and Lk = 3 keys per column,
 *     to_out(concat_h sum_j p[l,h,j] v[j,h,:]) = sum_h sum_j p[l,h,j] (W_out[:, h-th 64 columns] v[j,h,:]):
 * the three values of a column are projected per head (Lk rows per column instead of Lq) and combined with the softmax
 * weights in registers; the attention output is never written.  fp32-grade: two fp16 terms per operand (the values must be
 * inside fp16's range: the caller's guard), fp32 weights p, fp32 accumulation.
 *   _probs: P[col][head][l][2] = (p_0, p_1) of p = softmax_j(q_l . k_j / 8) for l = 0, their differences to level 0 for
 *           l > 0 (fp32, 64 floats per (col, head), zero-padded; p_2 = 1 - p_0 - p_1) and Vp[col * 3 + j][inner] = (v_0 - v_2, v_1 - v_2, v_2)[j] in the fp16-pair layout
 *           (AURORA_F32_A_SPLIT); q, kv as aurora_hip_perceiver_attention.
 *   _out:   out[col * Lq + l][0..N) = sum_h W_h (Vp_2 + p_0 Vp_0 + p_1 Vp_1)[col, h] + bias, W_h = W_pairs[:, 64 h .. 64 h + 63];
 *           W_pairs: [N][ldw] pre-split weights scaled by 2^6 (aurora_hip_split_f16 with scale 64).
 * Both retire at once unless *guard < guard_limit (null: always run).  Built for Lq in {3, 4, 13},
 * Lk = 3, head_dim 64, an even number of heads, N % 128 == 0: aurora_hip_perceiver_out_supported says whether a shape is. */
int aurora_hip_perceiver_out_supported(int Lq, int Lk, int heads, int head_dim, int N);
int aurora_hip_perceiver_probs(const float* q, const float* kv, float* P, void* Vp, int B, int64_t cols_per_b,
                               int64_t kv_bstride, int64_t kv_lstride, int Lq, int Lk, int heads, int head_dim,
                               const float* guard, float guard_limit, void* stream);
/* The same two from PRE-MULTIPLIED scores.  The queries of a first Perceiver layer are model constants, so
 * q_l . (W_k x_j) = (W_k^T q_l) . x_j: the key half of `to_kv` (perceiver.py:141-143) shrinks from heads * head_dim columns to
 * Lq * heads rows made at pack time (scaled by 1 / sqrt(head_dim)).  A context row of `vs` (ld_vs floats apart, row index as kv
 * above) holds [v (heads * head_dim) | ... | score of (query l, head h) at s_off + l * heads + h].  fp32 only.
 * _attention_scores: out / pair_guard / skip_guard as aurora_hip_perceiver_attention_unless.
 * _probs_scores: P / Vp / guard as aurora_hip_perceiver_probs. */
int aurora_hip_perceiver_attention_scores(const float* vs, int64_t ld_vs, int s_off, void* out, int B, int64_t cols_per_b,
                                          int64_t kv_bstride, int64_t kv_lstride, int Lq, int Lk, int heads, int head_dim,
                                          const float* pair_guard, float pair_limit, const float* skip_guard, float skip_limit,
                                          void* stream);
int aurora_hip_perceiver_probs_scores(const float* vs, int64_t ld_vs, int s_off, float* P, void* Vp, int B, int64_t cols_per_b,
                                      int64_t kv_bstride, int64_t kv_lstride, int Lq, int Lk, int heads, int head_dim,
                                      const float* guard, float guard_limit, void* stream);
int aurora_hip_perceiver_out(const void* Vp, const void* W_pairs, int64_t ldw, const float* P, const float* bias,
                             float* out, int64_t ldo, int64_t n_cols, int Lq, int Lk, int heads, int head_dim, int N,
                             const float* guard, float guard_limit, void* stream);

/* ---- token assembly at the encoder output -------------------------------------------------
 * x[b][c][l][:] = (c == 0 ? surf[b][l][:] : agg[(b*L + l)*(Cl-1) + c-1][:])
 *                 + pos_scale[l][:] + time_emb[b][:]
 * (encoder.py:332-363).  Writes the fp32 stream and, if out_t != NULL, a `dtype` copy.
 */
int aurora_hip_assemble_tokens(const float* surf, const float* agg, const float* pos_scale,
                               const float* time_emb, float* out_f32, void* out_t,
                               int B, int Cl, int64_t L, int D, int dtype, void* stream);

/* ---- decoder back end: unpatchify + post-decoder hooks + clamp + unnormalise ----------------
 * y: fp32 [(b*L + l)*n_lvl + c][ldy] head outputs (decoder.py:214-263, util.py:18-41); variable
 * v, level c, in-patch pixel (i, j) sits in column col0 + c*lvl_stride + i*P + j (lvl_stride != 0
 * for level-conditioned heads).  Writes dst_v[b][c][hp*P+i][wp*P+j] = g(z) * scale_v[c] + loc_v[c]
 * with Batch.unnormalise (batch.py:118-140) fused, where
 *   z = y                                         plain variables
 *   z = y + (1 + y_mod) * (prev - loc) * inv      difference prediction (aurora.py:761-779),
 *                                                 y_mod at mod_col0 (>= 0), prev the raw previous
 *                                                 state of the same variable
 *   then z = min(z, 1) on the levels in clamp_max1_levels (bit c; SO2 fix aurora.py:781-794)
 *   then g = clamp(min=0) if clamp_min0 (aurora.py:368-388);
 *   ocean-wave post hook (aurora.py:914-932): angle_col0 >= 0 makes col0 / angle_col0 the sin / cos
 *   heads of a direction, z = rad2deg(atan2(sin, cos)) mod 360; dens_col0 >= 0 keeps z only where
 *   mask[h][w] > mask_thresh (water) and the density head is >= 0 (sigmoid >= 0.5), else NaN.
 * `desc`: HOST array of n_vars (<= 32) descriptors with device pointers inside.
 */
typedef struct aurora_unpatch_var {
  float* dst;             /* [B][n_lvl][H][W] contiguous                                      */
  const float* loc;       /* device [n_lvl]                                                   */
  const float* scale;     /* device [n_lvl]                                                   */
  int32_t clamp_min0;
  int32_t col0;
  int32_t lvl_stride;
  int32_t mod_col0;       /* -1: no modulation head                                           */
  const float* prev;      /* device, raw units: prev[b*prev_sb + c*prev_sc + h*prev_sh + w]   */
  int64_t prev_sb, prev_sc, prev_sh;
  const float* inv_scale; /* device [n_lvl] 1/scale (only read when mod_col0 >= 0)            */
  uint32_t clamp_max1_levels;
  int32_t angle_col0;     /* -1: not a direction                                              */
  int32_t dens_col0;      /* -1: no density channel                                           */
  const float* mask;      /* device, raw water-body mask plane (H, W), row stride mask_sh     */
  int64_t mask_sh;
  float mask_thresh;      /* raw-units threshold equivalent to "normalised value > 0"         */
} aurora_unpatch_var;

int aurora_hip_unpatchify(const float* y, int64_t ldy, const aurora_unpatch_var* desc, int n_vars,
                          int B, int n_lvl, int Hp, int Wp, int P, void* stream);

/* ---- utility ------------------------------------------------------------------------------ */
/* dst[r, 0:cols] = src[r, 0:cols] for r < rows (row strides in elements of `dtype`). */
int aurora_hip_copy2d(const void* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows,
                      int64_t cols, int dtype, void* stream);
/* dst row r = src row idx[r] (r < n_rows), rows of row_bytes bytes (multiple of 16), pitches in bytes.
 * Packs the halo rows a latitude band sends to a neighbour (new: the reference is single-device). */
int aurora_hip_gather_rows(const void* src, int64_t src_pitch_bytes, const int32_t* idx, void* dst,
                           int64_t dst_pitch_bytes, int64_t n_rows, int64_t row_bytes, void* stream);
/* dst (other dtype) = convert(src): n elements, fp32 -> bf16 (round-nearest-even) or back. */
int aurora_hip_convert(const void* src, void* dst, int64_t n, int src_dtype, void* stream);

/* =============================================================================================
 * Model handle: ONE FORECAST STEP behind the C ABI.
 *
 * Replaces `Aurora.forward` (aurora/model/aurora.py:265-392) = Perceiver3DEncoder.forward (encoder.py:198-366) +
 * Swin3DTransformerBackbone.forward (swin3d.py:884-936) + Perceiver3DDecoder.forward (decoder.py:168-276) with the
 * normalisation of Batch.normalise / unnormalise (batch.py:94-140) fused in, for EVERY public model class: Aurora,
 * AuroraPretrained, AuroraSmallPretrained, Aurora12hPretrained, AuroraHighRes, and the variants AuroraAirPollution
 * (level-conditioned embeddings / heads levelcond.py:36-69, dynamic and static inputs encoder.py:226-303, feature combiners
 * and difference prediction aurora.py:726-796, second decoder Perceiver decoder.py:232-248) and AuroraWave (density / angle
 * channels and their inverse, aurora.py:854-932).  A forecast can run on one device or as one latitude band of a
 * sharded forecast (aurora_hip_set_band below).  Call order:
 *
 *   aurora_hip_create(&config, &model)
 *   aurora_hip_pack_weights(model, name, data, shape, ndim, AURORA_F32, on_device)   for every state_dict entry, after the
 *                                       checkpoint adapters (compat.py) -- names and shapes are the reference's schema
 *   aurora_hip_finalize(model, stream)          bf16 backbone copies, AdaLN modulation, fused heads, base weight set
 *   aurora_hip_precompute(model, &grid, stream) per grid / level set: Fourier tables, window tables, statistics
 *   per step:  aurora_hip_set_time(model, hours, B, stream);  aurora_hip_step(model, &io, stream)
 *   aurora_hip_destroy(model)
 *
 * Everything is enqueued on `stream`; a step performs no host synchronisation and may be captured into a hipGraph once
 * one step has run eagerly (the workspace grows on the first step; aurora_hip_set_time stays outside the capture; replay only
 * while aurora_hip_generation() is unchanged).
 * One in-flight step per handle (the reference's own threading contract, foundry/server/mlflow_wrapper.py:121).
 * The library keeps no state outside the handles.
 */
typedef struct aurora_hip_model aurora_hip_model;

typedef struct aurora_hip_config {   /* Aurora.__init__ keywords, aurora/model/aurora.py:55-95 */
  int32_t embed_dim, patch_size, latent_levels, num_heads;   /* num_heads: heads of the Perceivers */
  int32_t n_stages;                                          /* len(encoder_depths), <= 4 */
  int32_t encoder_depths[4], encoder_heads[4], decoder_depths[4], decoder_heads[4];
  int32_t window[3];
  int32_t enc_depth, dec_depth;
  float perceiver_ln_eps;
  int32_t max_history;
  double timestep_hours;
  int32_t stabilise_level_agg, use_lora, lora_steps, lora_mode;   /* lora_mode: 0 single, 1 from_second, 2 all */
  int32_t autocast;                                          /* 1: bf16 backbone (aurora.py:327-343) */
  int32_t n_surf, n_static, n_atmos;
  const char* const* surf_vars;                              /* fixes the order of every per-variable array below */
  const char* const* static_vars;
  const char* const* atmos_vars;
  /* ---- variant keywords (aurora.py:86-95); zero-initialised = the ERA5 model family -------------------------------
   * variant: 0 base; 1 air pollution: positive variables go through the clamp + log feature combiner
   * (`{surf,atmos}_feature_combiner.<var>` weights, aurora.py:733-742), variables in `modulation_heads` with an entry in
   * difference_history predict a difference to that history state (aurora.py:761-779), SO2 is capped at the levels
   * >= 850 hPa when LoRA is on (aurora.py:781-794); 2 ocean wave: `surf_vars` are the model's channels (`<v>_sin`,
   * `<v>_cos`, `<v>_density`, ...), `surf_inputs` the variables the caller supplies after AuroraWave's
   * batch_transform_hook; density / sin / cos channels are derived on the fly and inverted after the decoder
   * (aurora.py:892-932; the water-body mask is the static variable "wmb"). */
  int32_t variant;
  int32_t n_level_condition;                                 /* levels with their own patch embedding / heads (levelcond.py) */
  const double* level_condition;
  int32_t dynamic_vars, atmos_static_vars, clamp_at_first_step, simulate_indexing_bug;
  int32_t n_separate_perceiver;  const char* const* separate_perceiver;
  int32_t n_modulation_heads;    const char* const* modulation_heads;
  const int32_t* difference_history;                         /* [n_modulation_heads] history index, -1: no difference */
  int32_t n_positive_surf;       const char* const* positive_surf_vars;
  int32_t n_positive_atmos;      const char* const* positive_atmos_vars;
  int32_t n_surf_inputs;         const char* const* surf_inputs;              /* wave only; 0: = surf_vars */
  int32_t n_density;             const char* const* density_channel_surf_vars;
  int32_t n_angle;               const char* const* angle_surf_vars;
  /* ---- how THIS handle runs its steps (results are the same bits or within the documented bounds either way); all zero =
   * the library's defaults.  Read when the handle is created and kept in it: nothing reads the environment, and two handles
   * of one process may differ.  (The Python front end fills these from the AURORA_* environment variables of INTEGRATION.md,
   * once, when it creates a handle.)  Switches: 0 default, 1 off, 2 on. */
  struct aurora_hip_tuning {
    int32_t fuse_ln;               /* D = 512 linear + AdaLN in one launch: 0 default (= 2), 1 never, 2 when its row tiles fill the chip, 3 always */
    int32_t band_split_attention;  /* a latitude band's interior windows as a launch of their own in front of the halo wait (default off) */
    int32_t qkv_planes;            /* bf16 blocks: q | k | v in head planes (default on; off: token rows -- same bits) */
    int32_t split_k;               /* few-tile / long-K bf16 linears split along K (default on; off: a band's bf16 arithmetic is the
                                      un-sharded step's bit for bit whatever the CU count) */
    int32_t perceiver_reassoc;     /* decoder de-aggregation re-associated (aurora_hip_perceiver_out; default on) */
    int32_t score_weights;         /* first Perceiver layers: to_kv's key half replaced by the Lq * heads rows W_k^T q_l (the queries
                                      are model constants), attention from those scores (default on; off: k | v, then q . k) */
    int32_t reserved[2];
  } tuning;
} aurora_hip_config;

typedef struct aurora_hip_grid {     /* HOST pointers */
  int32_t n_lat, n_lon;              /* of the data; one surplus latitude row (n_lat % patch == 1) is dropped, batch.py:142-168 */
  const double* lat;                 /* [n_lat] degrees, decreasing */
  const double* lon;                 /* [n_lon] degrees */
  int32_t n_levels;
  const double* levels;              /* [n_levels] hPa */
  int32_t levels_float32;            /* 1: the levels pass through float32 (a tuple containing a float upstream) */
  const double* surf_loc;            /* normalisation statistics (aurora/normalisation.py): [n_surf] */
  const double* surf_scale;
  const double* static_loc;          /* [n_static] */
  const double* static_scale;
  const double* atmos_loc;           /* [n_atmos][n_levels] */
  const double* atmos_scale;
  /* Optional (both or neither): the Fourier position / scale features of the patch grid, [L][embed_dim] each, L = patches
   * per level.  The reference derives them from lat / lon in float32 torch kernels whose last-ulp behaviour the expansion
   * amplifies (wavelengths down to 1e-4 km); a caller that must reproduce a particular host's numbers bit for bit
   * supplies them, otherwise they are computed here (float32 geometry as upstream, fp64 trigonometry). */
  const float* pos_encoding;
  const float* scale_encoding;
} aurora_hip_grid;

typedef struct aurora_hip_step_io {  /* DEVICE pointers; variable order = the config's (surf: surf_inputs for the wave variant) */
  int32_t B, T;                      /* batch size, history states given (<= max_history) */
  const float* const* surf;          /* [n_surf]   each (B, T, n_lat, n_lon) with element strides surf_strides; NULL entry = absent */
  int64_t surf_strides[4];
  const float* const* stat;          /* [n_static] each (n_lat, n_lon) with element strides static_strides; NULL entry = absent */
  int64_t static_strides[2];
  const float* const* atmos;         /* [n_atmos]  each (B, T, n_levels, n_lat, n_lon) with element strides atmos_strides; NULL = absent */
  int64_t atmos_strides[5];
  float* const* out_surf;            /* [n_out_surf] each (B, H', n_lon) contiguous, H' = n_lat - n_lat % patch; n_out_surf and the
                                        order are aurora_hip_output_vars' (= the surface inputs, except for the wave variant) */
  float* const* out_atmos;           /* [n_atmos]  each (B, n_levels, H', n_lon) contiguous */
  int32_t rollout_step;              /* Metadata.rollout_step of the input: selects the LoRA weight set (lora.py:105-129) and
                                        whether positive variables are clamped (aurora.py:368-388) */
} aurora_hip_step_io;
/* A band of a sharded forecast (aurora_hip_set_band) passes and receives ITS latitude rows only: n_lat above is then the
 * band's row count, aurora_hip_band_rows tells which rows of the full grid those are. */

int aurora_hip_create(const aurora_hip_config* config, aurora_hip_model** out);
void aurora_hip_destroy(aurora_hip_model* model);
/* One state_dict entry (float32; `data` on the host, or on the device if on_device != 0).  The handle keeps its own copy. */
int aurora_hip_pack_weights(aurora_hip_model* model, const char* name, const void* data, const int64_t* shape, int ndim,
                            int dtype, int on_device);
int aurora_hip_finalize(aurora_hip_model* model, void* stream);
/* Packed weight files: a self-describing binary ("AURORAHIP1", entries of name / dtype / shape / raw data) that replaces
 * the pickled `.ckpt` + adapters (aurora.py:432-456, compat.py) for hosts without Python.  _save_packed writes what a
 * finalized handle holds -- the large backbone matrices in bf16 when the handle runs the bf16 backbone (the bits the
 * GEMMs consume: 2.6 GB instead of 5 GB for the 1.3 B model), the rest as fp32 -- and _load_packed fills a freshly created
 * handle from such a file instead of aurora_hip_pack_weights calls (then _finalize as usual).  bf16-only files serve
 * autocast handles; LoRA models keep fp32 attention projections (the merge W + BA is done in fp32). */
int aurora_hip_save_packed(aurora_hip_model* model, const char* path, void* stream);
int aurora_hip_load_packed(aurora_hip_model* model, const char* path);
int aurora_hip_precompute(aurora_hip_model* model, const aurora_hip_grid* grid, void* stream);
/* The position / scale tables aurora_hip_precompute derives from lat / lon when the grid carries no tables of its own
 * (posencoding.py:61-192; HOST pointers in and out, no device work): pos_out and scale_out are
 * [(n_lat / patch) * (n_lon / patch)][embed_dim] floats; a surplus latitude row is ignored.  Float32 geometry as upstream,
 * fp64 trigonometry -- equal to the reference's torch tables except where its float32 sin / sqrt differ in the last ulp,
 * which the shortest wavelengths amplify (tests/test_encodings.py states the bound per wavelength). */
int aurora_hip_pos_scale_encoding(const double* lat, const double* lon, int n_lat, int n_lon, int patch_size,
                                  int embed_dim, float* pos_out, float* scale_out);
/* Absolute times of the batch elements in hours since the Unix epoch (encoder.py:359-363), host pointer. */
int aurora_hip_set_time(aurora_hip_model* model, const double* time_hours, int B, void* stream);
int aurora_hip_step(aurora_hip_model* model, const aurora_hip_step_io* io, void* stream);
int64_t aurora_hip_workspace_bytes(const aurora_hip_model* model);
/* Which fp32 path did the last step take?  The large fp32 linears of encoder and decoder pick their operand split on the
 * DEVICE, from guard words the step leaves behind (aurora_hip_linear_ex): out[0] max |encoder context| (only when the encoder
 * is not part of the guarded chain, else 0), out[1] / out[2] max |normalised atmospheric / surface input| (folded in by
 * patchify), out[3] max |decoder context|.  A word below AURORA_F16_SAFE_RANGE (after the layer's own scaling, see
 * DESIGN.md 3) means two fp16 terms, above it three bf16 terms -- same accuracy, different speed.  Synchronises `stream`. */
#define AURORA_F16_SAFE_RANGE 16384.0f
int aurora_hip_guard_words(const aurora_hip_model* model, float out[4], void* stream);
/* Counts the re-allocations of device memory that enqueued work points at: the workspace (it grows with the first step of
 * a larger batch / history / grid), the time buffers (a larger batch), the grid tables (aurora_hip_precompute).  A hipGraph
 * captured from aurora_hip_step is valid for as long as this number does not change. */
int64_t aurora_hip_generation(const aurora_hip_model* model);
/* sizeof() of the structs of this header as the library was compiled, in the order aurora_hip_config, aurora_hip_grid,
 * aurora_hip_step_io, aurora_hip_band, aurora_hip_halo_msg, aurora_hip_plan_info, aurora_patch_var, aurora_unpatch_var,
 * aurora_hip_profile_entry: a binding in another language checks its own struct declarations against them.  Returns the
 * number of entries. */
int aurora_hip_abi_sizes(int32_t* out, int capacity);
/* Names of the surface variables a step predicts, in the order of aurora_hip_step_io.out_surf (ocean wave: the non-angle
 * variables, then the directions, aurora.py:914-932; otherwise the surface inputs).  Returns the count; `names` may be NULL. */
int aurora_hip_output_vars(const aurora_hip_model* model, const char** names, int capacity);
/* aurora_hip_set_time with the calendar fields the dynamic variables of the air-pollution model are made of
 * (encoder.py:226-246): calendar[b] = {hour of day, weekday (Monday = 0), day of month}; NULL derives them from the time
 * stamp as UTC. */
int aurora_hip_set_time_ex(aurora_hip_model* model, const double* time_hours, const int32_t* calendar, int B, void* stream);

/* ---- one forecast across several devices: latitude bands + halo exchange (the reference is
 * single-device) ---------------------------------------------------------------------------------------------------------
 * Every rank owns a contiguous band of latitude rows at every backbone stage (boundaries on the coarsest stage, doubled
 * per finer stage, so patch merges / splits stay local; on window rows of the finer stages when that is balanced, else the
 * most balanced split whose windows stay within two neighbouring ranks -- a rank exchanges with rank - 1 and rank + 1 only,
 * aurora_hip_precompute fails if no such split exists).  Everything except window attention is local to a token, a 2 x 2
 * block or a grid column.  A shifted-window block needs k | v of the neighbouring band's first / last rows: the handle
 * gathers those rows of the block's INPUT into `send` staging buffers (half the bytes of k | v, and available before the
 * qkv GEMM), calls `post` (start sending / receiving; asynchronous to `stream`), runs its own qkv GEMM and the windows that
 * need no halo row, calls `wait` (make `stream` wait for the messages), projects the received rows to k | v itself and
 * attends the boundary windows.  The TRANSPORT is the host's: RCCL point-to-point (torch.distributed / ncclSend /
 * ncclRecv) in production, anything else in tests -- the library does not link a communication library.
 * Message buffers are the two staging buffers the host hands over (its own allocations, so that its transport can address
 * them): one that the handle fills with what is sent, one that receives; at least aurora_hip_band_staging_bytes() each
 * (valid after aurora_hip_precompute).  A message is a byte range of one of them: what goes to / comes from the previous
 * rank first, the next rank's behind it -- so one gather launch packs both and one GEMM projects both.
 * Call order: create, pack, finalize, aurora_hip_set_band, aurora_hip_precompute with the FULL grid, aurora_hip_band_rows,
 * aurora_hip_set_band_staging, then steps on the band's rows. */
typedef struct aurora_hip_halo_msg {
  int32_t peer;       /* rank */
  int32_t reserved;
  int64_t offset;     /* byte offset into the send (for a send) or the receive (for a receive) staging buffer */
  int64_t bytes;
} aurora_hip_halo_msg;
typedef int (*aurora_hip_halo_post_fn)(void* user, const aurora_hip_halo_msg* sends, int32_t n_sends,
                                       const aurora_hip_halo_msg* recvs, int32_t n_recvs, void* stream);
typedef int (*aurora_hip_halo_wait_fn)(void* user, void* stream);
typedef struct aurora_hip_band {
  int32_t rank, world;
  aurora_hip_halo_post_fn post;
  aurora_hip_halo_wait_fn wait;
  void* user;
} aurora_hip_band;
int aurora_hip_set_band(aurora_hip_model* model, const aurora_hip_band* band);   /* world <= 1: un-sharded again */
/* Data rows [row0, row1) of the full (cropped) latitude axis that this rank owns. */
int aurora_hip_band_rows(const aurora_hip_model* model, int32_t* row0, int32_t* row1);
int64_t aurora_hip_band_staging_bytes(const aurora_hip_model* model);
int aurora_hip_set_band_staging(aurora_hip_model* model, void* send, void* recv, int64_t staging_bytes);
/* The partition and the attention plans themselves, as pure host functions (no model, no device work): what the handle
 * uses internally, exposed for hosts that place data themselves and for tests (tests/test_partition.py compares them with
 * numpy plans that are replayed against global attention).
 * aurora_hip_band_partition: owned token rows [h0, h1) of `rank` at backbone stage `stage` (0 = finest) of an
 * `n_stages`-stage U-net over a token grid res0 = (levels, latitude rows, longitude columns).
 * aurora_hip_band_plan: the plan of one block flavour at one stage: `res` the stage's token grid, rows[2 r], rows[2 r + 1]
 * the owned rows of rank r there.  Fills `info`; the arrays may be NULL (query sizes first): tok / grp
 * [n_windows][win_tokens] (index into [owned rows | halo rows], -1 = padding or a position no owned query can see; grp is
 * left untouched when has_groups == 0), send_idx_prev / _next: local indices of the owned rows sent to rank - 1 / + 1,
 * in the receiver's halo order. */
typedef struct aurora_hip_plan_info {
  int32_t n_windows, win_tokens, n_own, n_halo, n_interior;   /* the first n_interior windows touch no halo row */
  int32_t recv_offset[2], recv_count[2], send_count[2];       /* [0] previous rank, [1] next rank; offsets into the halo rows */
  int32_t has_groups;
} aurora_hip_plan_info;
int aurora_hip_band_partition(int n_stages, const int32_t res0[3], const int32_t window[3], int world, int rank, int stage,
                              int32_t* h0, int32_t* h1);
int aurora_hip_band_plan(const int32_t res[3], const int32_t window[3], int shifted, int world, int rank,
                         const int32_t* rows, aurora_hip_plan_info* info, int32_t* tok, uint8_t* grp,
                         int32_t* send_idx_prev, int32_t* send_idx_next);

/* Per-launch timing of the handle's own kernels: between _begin and _end every launch of a kernel kind whose bit is set
 * in kind_mask (bit i = entry i of the table _end returns: linear_bf16, linear_f32, window_attention_bf16, layernorm,
 * merge_ln, split_ln, patchify, perceiver_attention, assemble_tokens, unpatchify, copy2d, absmax, linear_layernorm_bf16,
 * gather_rows, perceiver_out) is bracketed by a HIP event pair on the launch stream.  _end synchronises the device and
 * fills `out` (capacity >= 15): launches, summed
 * milliseconds and summed algorithmic work (FLOPs for the linears, bytes for the window attention) per kind.  This is
 * what bench.py's `roofline` is computed from.  An event pair keeps a launch from overlapping its neighbours. */
typedef struct aurora_hip_profile_entry {
  const char* kernel;
  int64_t launches;
  double ms;
  double work;
} aurora_hip_profile_entry;
int aurora_hip_profile_begin(aurora_hip_model* model, uint32_t kind_mask);
int aurora_hip_profile_end(aurora_hip_model* model, aurora_hip_profile_entry* out, int capacity, int* n_out);
/* The same, one entry per LAUNCH in launch order (launches = 1; `work` tells the shapes of a kind apart).  With capacity
 * smaller than the number of recorded launches only *n_out is set and the recording is kept: query first, then fetch. */
int aurora_hip_profile_end_list(aurora_hip_model* model, aurora_hip_profile_entry* out, int capacity, int* n_out);

/* ---- debugging aid ------------------------------------------------------------------------------
 * The four-wave bf16 GEMM tile with the hand-scheduled main loop (csrc/gemm_a4.hip; plain bf16 linears on 256 x 256 tiles with
 * K >= AURORA_GEMM_A4_MIN_K, a read-once process default) can leave s_memtime stamps of workgroups 0 and 255 in 8 x 8 device
 * words -- per wave: kernel start, loop start, loop end, 64-wide K units, kernel end -- for tools/gemm_a4_stamps.py.  NULL switches it off. */
void aurora_hip_debug_a4_stamps(void* device_words);

#ifdef __cplusplus
}
#endif
#endif /* AURORA_HIP_H */

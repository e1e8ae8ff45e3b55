/* xvit.h — C ABI of libxvit_hip.so: the MI355X (gfx950) hot path of the cross-attention 3-D ViT.
 *
 * The reference (vsahni3/cross-attention-ViT) has no FFI; its hot path sits behind plain
 * torch.nn.Module calls.  Each entry point below stands in for the eager-PyTorch op sequence
 * of the reference lines it cites (paths relative to the reference repo).  The Python host
 * side (cross-attention-vit_amd/xvit/) binds these with ctypes and re-exposes the reference's
 * own nn.Module signatures; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions (SURVEY.md §8(b), C-ABI face)
 *   - extern "C", raw device pointers + explicit sizes/strides (in ELEMENTS) + a HIP stream
 *     handle passed as void*.  No torch types cross this boundary.
 *   - The caller allocates every output and workspace; the library never allocates, frees or
 *     keeps a pointer after the call returns.  Every launch is enqueued on `stream`; there is
 *     no internal synchronisation, so all entry points are HIP-graph capturable.
 *   - Return: 0 = OK; <0 = argument/shape/unsupported error (nothing launched; text via
 *     xvit_last_error_string()); >0 = hipError_t reported by the launch.
 *   - Activations are bf16 (fp32 accumulate); the residual stream, LayerNorm statistics,
 *     parameter gradients and the loss are fp32.
 */
#ifndef XVIT_H
#define XVIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XVIT_VERSION 303 /* 0.3.3: xvit_add_cast_f32_bf16, xvit_rows_combine; 0.3.2: probability dropout in the low-rank fusion (xvit_cls_softmax_*, xvit_head_cols bias_scale, xvit_head_bias_grad), xvit_xattn_kv_wgrad removed; 0.3.1: xvit_set_dropout_epoch; 0.3.0: workspaces in xvit_attn_fwd/bwd (CLS peel), xvit_linear_f32_batched; 0.2.0: ld_alt in xvit_layernorm_fwd/bwd, dropout in xvit_attn_*, xvit_patch_embed_*, xvit_attn_fwd_fp8, xvit_linear_f32, workspaces */

enum { XVIT_OK = 0, XVIT_ERR_ARG = -1, XVIT_ERR_UNSUPPORTED = -2 };
enum { XVIT_BF16 = 0, XVIT_F32 = 1 };

typedef void* xvit_stream_t; /* hipStream_t */

int xvit_version(void);
const char* xvit_last_error_string(void);
/* Process-wide tuning knobs (diagnostics / A-B measurements; results never depend on them):
 *   "gemm_tile"      0 = automatic tile choice, 1 = always the 128x128 kernel, 2 = the 256x256 kernel whenever M, N >= 256
 *   "gemm_group"     0 = automatic, n > 0 = column tiles per super-column of the 256x256 kernel's tile walk
 *   "gemm_epilogue"  0 = automatic, 1 = always the 8-byte-per-lane epilogue (bf16 outputs normally use 16 bytes per lane) */
int xvit_set_option(const char* name, int value);
/* Dropout under HIP-graph capture.  The reference seeds its dropout masks from the host RNG at every call (nn.Dropout,
 * model_cross.py:27,47,95,101,170; it trains with p = 0.1 .. 0.25, main_mist.py:71-77); a captured step would freeze the seeds passed
 * below and replay the same masks.  With a device counter registered here (process-wide; NULL = off, the default) every dropout-carrying
 * launch hands its address to the kernel, which uses seed + counter * odd constant, read at run time: a graph that increments the counter
 * at its head draws fresh masks at every replay, identical in its forward and backward.  Results with NULL are unchanged. */
int xvit_set_dropout_epoch(const uint64_t* device_counter);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue.  Replaces every nn.Linear on the path and its autograd backward:
 *   model_cross.py:23,26 (FeedForward), :43,46 (to_qkv / to_out), :81-85 (wq/wk/wv/proj),
 *   :168 (patch_to_embedding), :177-181 (mlp_head); model.py:110-111,133-137.
 * C[M,N] = op(A) * op(B), bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16).
 *   layout NT: A[M,K] (k contiguous), B[N,K] (k contiguous)       y = x W^T      (forward)
 *   layout NN: A[M,K] (k contiguous), B[K,N] (n contiguous)       dx = dy W      (dgrad)
 *   layout TN: A stored [K,M] (m contiguous), B[K,N] (n contig.)  dW = dy^T x    (wgrad)
 * Epilogue, in this order: +bias[n]; act; dropout; +residual; row remap; store / accumulate.
 * ---------------------------------------------------------------------------------------- */
enum { XVIT_GEMM_NT = 0, XVIT_GEMM_NN = 1, XVIT_GEMM_TN = 2 };
enum { XVIT_ACT_NONE = 0,
       XVIT_ACT_GELU = 1,  /* aux (bf16, optional) receives the pre-activation z; C = gelu(z), erf form */
       XVIT_ACT_DGELU = 2  /* aux (bf16) supplies z; C = acc * gelu'(z) */ };
enum { XVIT_ACC_STORE = 0, XVIT_ACC_ADD = 1 /* fp32 C only: C += result */ };

typedef struct xvit_gemm_args {
  int32_t layout, M, N, K, batch;
  int32_t c_dtype;    /* XVIT_BF16 | XVIT_F32 */
  int32_t act;        /* XVIT_ACT_* */
  int32_t accumulate; /* XVIT_ACC_* */
  int32_t split_k;    /* >=1.  >1: the contraction is cut into split_k ranges whose fp32 partial tiles go to the
                         workspace (xvit_gemm_workspace_bytes(args) bytes); a second kernel sums them in a fixed
                         order and applies the epilogue (deterministic, no atomics) */
  /* residual row = res_row_off + (row % res_row_mod) when res_row_mod > 0 (broadcast, e.g. pos_embedding) */
  int32_t res_row_mod, res_row_off;
  /* output row = row + (row / out_seg_rows) * out_seg_skip + out_row_off when out_seg_rows > 0
     (patch rows -> token rows that leave room for the CLS row, model_cross.py:195-196) */
  int32_t out_seg_rows, out_seg_skip, out_row_off;
  int32_t aux_mode;   /* 0: aux = the pre-activation z (ACT_GELU writes it, ACT_DGELU reads it and evaluates gelu');
                         1: aux = gelu'(z): ACT_GELU writes the derivative (one more fma next to the shared exponential) and
                            ACT_DGELU only multiplies by it — the backward epilogue loses its erf / exp evaluation */
  const void* A; const void* B; void* C;
  const float* bias;     /* [N] fp32 or NULL */
  const float* residual; /* fp32 [*, N] or NULL */
  void* aux;             /* bf16 [M, N] or NULL (see act) */
  int64_t lda, ldb, ldc, ldr, ldaux;                                  /* leading dims, elements */
  int64_t stride_a, stride_b, stride_c, stride_bias, stride_r, stride_aux; /* per-batch strides */
  void* workspace;         /* caller-owned scratch for split_k > 1 (fp32 partial tiles), else NULL */
  int64_t workspace_bytes;
  float* colsum;           /* optional fp32 [N]: colsum[n] += sum over rows of the stored C before its rounding to c_dtype
                              (bias gradient of the producing Linear; atomic adds, so the caller zeroes it first);
                              per-batch stride = stride_bias */
  /* nn.Dropout fused after the activation: element (row, col) is kept iff hash(seed, row*N + col) >= p * 2^24
     (the mask xvit_dropout applies to a contiguous [M, N] tensor with the same seed), scaled by 1/(1-p) */
  float dropout_p; int32_t reserved2;
  uint64_t dropout_seed;
} xvit_gemm_args;

int xvit_gemm(const xvit_gemm_args* args, xvit_stream_t stream);
/* scratch needed by xvit_gemm for these args (0 when none); no launch, no allocation */
int64_t xvit_gemm_workspace_bytes(const xvit_gemm_args* args);

/* fp32 Linear for the single-token CLS path (model_cross.py:91 wq, :100 proj, :112-113 the FFN on the one fused token,
 * :177-181 the heads): y[M,N] = x[M,K] W[N,K]^T (+bias) (GELU) (dropout) (+residual), ALL fp32 (W is the master weight
 * itself), on the f32-input MFMA; K-chunks are summed in a fixed order through the caller's workspace (deterministic).
 * M = batch rows; K % 16 == 0.  Optional bf16 copies for the backward chain: z_bf16 = pre-activation (act = GELU),
 * y_bf16 = the stored y.  Epilogue order as xvit_gemm.  workspace: xvit_linear_f32_workspace_bytes(M, N, K) bytes. */
int64_t xvit_linear_f32_workspace_bytes(int M, int N, int K);
int xvit_linear_f32(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* y, int64_t ldy, int M, int N, int K,
                    int act, void* z_bf16, int64_t ldz, const float* residual, int64_t ldr, void* y_bf16, int64_t ldyb, float dropout_p,
                    uint64_t dropout_seed, void* workspace, int64_t workspace_bytes, xvit_stream_t stream);

/* Tiny fp32 linear for shapes the MFMA tile cannot address (the num_classes=2 head,
 * model_cross.py:181).  y[M,N] = x[M,K] W[N,K]^T + b.  x is bf16, W/b/y fp32.  */
int xvit_small_linear_fwd(const void* x_bf16, int64_t ldx, const float* W, const float* b, float* y, int M, int N, int K, xvit_stream_t stream);
/* dx[M,K] (bf16) = (dy W) * (z ? gelu'(z) : 1);  dW[N,K] += dy^T x;  db[N] += colsum(dy).  dy fp32.
 * z (bf16 [M,K], optional) is the pre-activation of the GELU that produced x (model_cross.py:178). */
int xvit_small_linear_bwd(const float* dy, const void* x_bf16, int64_t ldx, const float* W, const void* z_bf16, int64_t ldz,
                          void* dx_bf16, int64_t lddx, float* dW, float* db, int M, int N, int K, int deterministic, xvit_stream_t stream);
/* deterministic != 0: one row chunk instead of 8-row chunks meeting in fp32 atomics (bit-reproducible, slower) */

/* ------------------------------------------------------------------------------------------
 * LayerNorm (model_cross.py:11-17 PreNorm, :174/:203 final norm; model.py:186-187,207 eps=1e-6).
 * x is the fp32 residual stream [rows, d] (row stride ldx).  If x_alt != NULL, row k * seq_len (the first row of
 * sequence k) is read from x_alt + k * ld_alt instead: the "cls of i + patches of j" concatenation of model_cross.py:140
 * without materialising it (ld_alt = seq_len * ldx when x_alt is the other modality's token tensor, = d for a packed
 * [sequences, d] copy of its CLS rows).
 * Outputs: y_bf16 (the GEMM operand) and / or y_f32 (the fp32 copy the single-token CLS path feeds to xvit_linear_f32);
 * either may be NULL, not both.
 * ---------------------------------------------------------------------------------------- */
int xvit_layernorm_fwd(const float* x, const float* x_alt, int64_t ldx, int seq_len, int64_t ld_alt, const float* gamma, const float* beta,
                       float eps, void* y_bf16, int64_t ldy, float* y_f32, int64_t ldyf, float* mean, float* rstd, int rows, int d,
                       xvit_stream_t stream);
/* dx = (dres ? dres : 0) + LN'(dy); also emits a bf16 copy of dx (the next GEMMs' operand) when
 * dx_bf16 != NULL; dgamma/dbeta are ADDED (fp32 atomics).  Optional dxsum[d] += column sums of dx and
 * dressum[d] += column sums of dres: the bias gradients of the Linears on either side of the norm
 * (dx is the dy of the Linear that produced x; dres the dy of the Linear whose output joined x), for free. */
int xvit_layernorm_bwd(const void* dy_bf16, int64_t lddy, const float* x, const float* x_alt, int64_t ldx, int seq_len, int64_t ld_alt,
                       const float* mean, const float* rstd, const float* gamma, const float* dres, int64_t lddres,
                       float* dx, int64_t lddx, void* dx_bf16, int64_t lddxb, float* dgamma, float* dbeta, float* dxsum,
                       float* dressum, int rows, int d, float* workspace, int64_t workspace_bytes, xvit_stream_t stream);
/* workspace (optional, xvit_layernorm_bwd_workspace_bytes(rows, d) bytes): the deterministic form — per-block partial sums
 * are stored there and added to dgamma / dbeta / dxsum / dressum in block order by a second small kernel instead of by fp32
 * atomics (bit-reproducible run to run; SURVEY.md 8(b) "Determinism"). */
int64_t xvit_layernorm_bwd_workspace_bytes(int rows, int d);

/* ------------------------------------------------------------------------------------------
 * Fused self-attention (model_cross.py:53-60; model.py:165-172): softmax(q k^T * scale) v
 * without materialising the [B,H,N,N] scores.  Element (b, n, h, :) of q/k/v lives at
 * ptr + b*stride_b + n*stride_n + h*dh (q, k, v normally point into one [B,N,3d] tensor).
 * dh must be 64.  lse[B,H,N] = log-sum-exp of the scaled scores (saved for backward).
 * dropout_p > 0 (model.py:169 attn_dropout on the probabilities): element (b, h, q, k) is kept iff
 * hash(seed, ((b*H + h)*N + q)*N + k) >= p * 2^24 (the mask xvit_dropout applies to a contiguous [B, H, N, N] tensor with that
 * seed), scaled by 1/(1-p), applied after the softmax normalisation; the backward regenerates it from the same (p, seed).
 * ---------------------------------------------------------------------------------------- */
int xvit_attn_fwd(const void* q, const void* k, const void* v, int64_t stride_b, int64_t stride_n, void* o, int64_t o_stride_b,
                  int64_t o_stride_n, float* lse, int B, int H, int N, int dh, float scale, float dropout_p, uint64_t dropout_seed,
                  float* workspace, int64_t workspace_bytes, xvit_stream_t stream);
/* workspace: xvit_attn_bwd_workspace_bytes(B, H, N) bytes of fp32, 16-byte aligned (rowsum(do*o), -lse*log2e, CLS-peel partials).
 * dq/dk/dv use the q/k/v strides. */
int xvit_attn_bwd(const void* q, const void* k, const void* v, int64_t stride_b, int64_t stride_n, const void* o, const void* d_o,
                  int64_t o_stride_b, int64_t o_stride_n, const float* lse, float* workspace, int64_t workspace_bytes, void* dq, void* dk,
                  void* dv, int B, int H, int N, int dh, float scale, float dropout_p, uint64_t dropout_seed, xvit_stream_t stream);
/* CLS peel.  The reference's sequences are cls + P patch tokens (model_cross.py:195-196): N = 64 m + 1 at every BASELINE config with
 * cubic power-of-two volumes (513, 1025, 4097).  On a tile grid token 0 costs one more 128-query block per (b, head) and one more
 * 64-key tile per block.  With a workspace (xvit_attn_fwd_workspace_bytes > 0 <=> the shape qualifies: N % 64 == 1, no probability
 * dropout, and under the default xvit_set_option("attn_peel", 1) a grid of >= 768 workgroups; 2 = any grid, 0 = never) the kernels tile the patch tokens only; token 0 enters as the initial
 * online-softmax state / initial gradient accumulators (as a key) and as one extra MFMA block per wave with partial results merged
 * in a fixed order (as a query).  Same outputs (o, lse, dq, dk, dv for all N tokens), bit-reproducible; workspace == NULL keeps
 * token 0 on the tile grid. */
int64_t xvit_attn_fwd_workspace_bytes(int B, int H, int N);
int64_t xvit_attn_bwd_workspace_bytes(int B, int H, int N);

/* ------------------------------------------------------------------------------------------
 * CLS-query cross-attention (model_cross.py:91-99): one query row per (b, h) against N keys.
 * q [B, d] bf16 (row stride ldq); k/v as above; o [B, d] bf16; p [B,H,N] fp32 probabilities
 * (saved for backward).  HBM-bound GEMV-style kernel.  q_f32 [B, d] (optional, row stride ldqf) replaces q as the query and
 * o_f32 (optional) receives an fp32 copy of o: the single-token CLS path keeps its operands in fp32 (xvit_linear_f32).
 * ---------------------------------------------------------------------------------------- */
int xvit_cls_xattn_fwd(const void* q, int64_t ldq, const float* q_f32, int64_t ldqf, const void* k, const void* v, int64_t stride_b,
                       int64_t stride_n, void* o, int64_t ldo, float* o_f32, int64_t ldof, float* p, int B, int H, int N, int dh, float scale,
                       float dropout_p, uint64_t dropout_seed, xvit_stream_t stream);
/* dropout_p / dropout_seed: attn_drop on the probabilities (model_cross.py:97); p[] holds the pre-dropout values */
int xvit_cls_xattn_bwd(const void* q, int64_t ldq, const void* k, const void* v, int64_t stride_b, int64_t stride_n, const float* p,
                       const void* d_o, int64_t lddo, float* dq, int64_t lddq, void* dk, void* dv, float* coef, int B, int H, int N, int dh,
                       float scale, float dropout_p, uint64_t dropout_seed, xvit_stream_t stream);
/* dk / dv (bf16, laid out like k / v; both or neither) and / or coef (fp32 [B, N, 2 H]).  dK and dV of one (b, head) are rank one:
 * dk[n] = coef[b][n][head] * q_head (the softmax scale included), dv[n] = coef[b][n][H + head] * dO_head.
 *   xvit_xattn_kv_dgrad (the low-rank fusion's token gradient): dhn[b, n, :] (bf16) = dk[b, n, :] Wk + dv[b, n, :] Wv
 *     = sum_j coef[b, n, j] R[j, b, :], with R (fp32 [2 H, B, d]): R[h, b, :] = q[b, h, :] Wk[64 h .. 64 h + 63, :],
 *     R[H + h, b, :] = dO[b, h, :] Wv[64 h .., :] (xvit_head_rows).  d = 64 H, 2 H <= 32. */
int xvit_xattn_kv_dgrad(const float* coef, const float* R, void* dhn_bf16, int64_t lddh, int B, int H, int N, int d, xvit_stream_t stream);

/* The fusion's key / value path in its low-rank form (model_cross.py:88-99): with one query row per (sample, head),
 *   scores[b, n, h] = hn[b, n, :] . U[b, h, :] (+ a constant over n),  U[b, h, :] = q[b, h, :] Wk[64 h .. 64 h + 63, :]          (xvit_head_rows)
 *   out[b, 64 h + e] = rz[b, h] sum_c Wv[64 h + e, c] S[b, h, c] + bv[64 h + e],  S[b, h, :] = sum_n e[b, n, h] hn[b, n, :]       (xvit_head_cols)
 * so wk and wv are never applied to the N tokens: the two passes over hn are batched xvit_gemm calls ([N, d] x [d, 16] and
 * [16, N] x [N, d] per sample) and the softmax over n sits between them (xvit_cls_softmax_fwd: bf16 weights e = exp(scale (s - max)),
 * rz = 1 / sum e).  Backward: dp = hn . Y (Y = dO_h Wv_h, xvit_head_rows), xvit_cls_softmax_bwd -> coef (the input of
 * xvit_xattn_kv_dgrad) and bf16 ds, T = sum_n ds hn (xvit_gemm), dq = Wk_h T (xvit_head_cols), dWk_h = q_h^T T, dWv_h = dO_h^T (rz S)
 * (xvit_head_wgrad).  All per-head products are fp32 on the f32-input MFMA against the fp32 master weights; d = 64 H, H <= 16.
 *   xvit_head_rows : out[b, h, c] (fp32, element strides out_sb / out_sh; optional bf16 copy with its own strides) = sum_e x[b, 64 h + e] W[64 h + e, c]
 *   xvit_head_cols : out[b, 64 h + e] (fp32, row stride ldo; optional bf16 copy) = row_scale[b, h] * sum_c t[b, h, c] W[64 h + e, c]
 *                    + bias_scale[b, h] * bias[64 h + e]   (row_scale, bias, bias_scale optional: 1, 0, 1)
 *   xvit_head_wgrad: dW[64 h + e, c] = sum_b x[b, 64 h + e] row_scale[b, h] t[b, h, c]
 *   xvit_head_bias_grad: out[j] = sum_b x[b, j] w[b, j / 64] (the gradient of bv under dropout; samples summed in order)
 * Dropout on the probabilities (attn_drop, model_cross.py:97; the reference trains with 0.1 .. 0.25): p'[n] = m[n] p[n] / (1 - rate) with
 * the mask xvit_cls_xattn_fwd uses (xvit_dropout on a contiguous [B, H, N] tensor).  xvit_cls_softmax_fwd then also writes the KEPT weights
 * e_masked (the operand of the row-sum GEMM: S = sum_n e_masked hn) and rz becomes a [3][B][H] block: rz, rz / (1 - rate) (the row scale
 * of xvit_head_cols and of dWv) and rz / (1 - rate) * sum_n e_masked (bias_scale: the weight in front of bv, which no longer is one);
 * xvit_cls_softmax_bwd regenerates the mask: dp~ = m dp / (1 - rate) replaces dp and the second half of coef is p'.                    */
int xvit_head_rows(const float* x, int64_t ldx, const float* W, int64_t ldw, float* out, int64_t out_sb, int64_t out_sh, void* out_bf16, int64_t ob_sb,
                   int64_t ob_sh, int ob_heads, int B, int H, int d, xvit_stream_t stream);   /* head rows H .. ob_heads - 1 of the bf16 copy are zeroed */
int xvit_head_cols(const float* t, int64_t t_sb, int64_t t_sh, const float* W, int64_t ldw, const float* row_scale, int64_t rs_ld, const float* bias,
                   const float* bias_scale, int64_t bsc_ld, float* out, int64_t ldo, void* out_bf16, int64_t ldob, int B, int H, int d, xvit_stream_t stream);
int xvit_head_bias_grad(const float* x, int64_t ldx, const float* w, int64_t ldw, float* out, int B, int H, int d, xvit_stream_t stream);
int xvit_head_wgrad(const float* x, int64_t ldx, const float* t, int64_t t_sb, int64_t t_sh, const float* row_scale, int64_t rs_ld, float* dW, int64_t lddw,
                    int B, int H, int d, xvit_stream_t stream);
/* s [B, N, lds] fp32 -> e [B, N, lde] bf16 (columns >= H zeroed; lde <= 16), rz [B, H]; dropout_p > 0: also e_masked (like e) and rz is [3][B][H] */
int xvit_cls_softmax_fwd(const float* s, int64_t lds, void* e_bf16, int64_t lde, float* rz, int B, int H, int N, float scale, void* e_masked_bf16,
                         float dropout_p, uint64_t dropout_seed, xvit_stream_t stream);
/* p = e rz; ds = scale p (dp~ - sum_n p dp~) -> coef [B, N, 2 H] fp32 = (ds | p'), ds_bf16 [B, N, ldb] (columns >= H zeroed) */
int xvit_cls_softmax_bwd(const void* e_bf16, int64_t lde, const float* rz, const float* dp, int64_t ldp, float* coef, void* ds_bf16, int64_t ldb, int B, int H,
                         int N, float scale, float dropout_p, uint64_t dropout_seed, xvit_stream_t stream);

/* MX-fp8 forward attention (BASELINE.json configs[4] "fp8 MFMA QK^T/AV path"; reference ops model_cross.py:55-59):
 * same arguments and outputs as xvit_attn_fwd without dropout, but q, k, v are first quantised to OCP e4m3 with one e8m0
 * scale per 32 contraction elements (q, k along d_h; v transposed, along the keys) into the caller's workspace, and both
 * products run on v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate).  Accuracy is that of 3-bit mantissas (stated in
 * tests/test_attn_fp8_gpu.py); lse / o feed the bf16 backward unchanged.  Opt-in: nothing selects it by default. */
int64_t xvit_attn_fp8_workspace_bytes(int B, int H, int N, int dh);
int xvit_attn_fwd_fp8(const void* q, const void* k, const void* v, int64_t stride_b, int64_t stride_n, void* o, int64_t o_stride_b, int64_t o_stride_n,
                      float* lse, int B, int H, int N, int dh, float scale, void* workspace, int64_t workspace_bytes, xvit_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 3-D patchify (model_cross.py:193, modelv3.py:129): img [B, M, 1, D, H, W] (fp32 or bf16, contiguous) ->
 * rows of a bf16 patch matrix [*, pd]; token t = (h*Wn + w)*Dn + d, feature f = (p1*hp + p2)*wp + p3.
 * Patch (b, m, t) goes to row  b*stride_b + m*stride_m + t + row_off,  so the same kernel lays the rows out
 *   per modality with a CLS slot  (ModelCross: stride_b = P+1, stride_m = B*(P+1), row_off = 1) or
 *   concatenated per sample       (ModelVIT:   stride_b = M*P+1, stride_m = P,     row_off = 1).
 * zero_rows rows (0, zero_row_stride, 2*zero_row_stride, ...) are zero-filled: the CLS slots, so the patch
 * matrix lines up row-for-row with the token tensor (model_cross.py:195-196).
 * ---------------------------------------------------------------------------------------- */
int xvit_patchify(const void* img, int img_dtype, void* patches_bf16, int B, int M, int D, int H, int W, int dp, int hp, int wp,
                  int64_t stride_b, int64_t stride_m, int row_off, int zero_rows, int64_t zero_row_stride, xvit_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * Patch embedding straight from the volume (model_cross.py:193-197: rearrange -> patch_to_embedding -> + pos_embedding):
 *   x[(m*B + b)*(cls_rows + P) + cls_rows + t, :] = patch(b, m, t) W^T + bias + pos[cls_rows + t, :]
 * with the token / feature order of xvit_patchify, WITHOUT writing the [rows, dp*hp*wp] patch matrix: the GEMM's LDS-DMA
 * loaders gather the patch rows from img [B, M, 1, D, H, W] (bf16, contiguous).  CLS rows (cls_rows = 1) come out as
 * bias + pos[0] and are overwritten by xvit_cls_row_fwd, exactly as with a zero CLS row in a stored patch matrix.
 * xvit_patch_embed_wgrad: dW[d, pd] = sum over patch rows of dx[row, :]^T patch(row) (CLS rows of dx are skipped), fp32,
 * split over the contraction into the caller's workspace and summed in a fixed order (deterministic).
 * xvit_patch_embed_supported: 1 when the geometry fits the fused kernels (wp in {8,16,32,64}, hp*wp % 64 == 0, pd and d multiples of
 * 256, >= 2048 rows, volume < 2 GiB); otherwise use xvit_patchify + xvit_gemm.  ANY patch grid qualifies: where 64 consecutive tokens
 * are whole d-columns of one h-row (64 % (D/dp) == 0 and (D/dp)*(W/wp) % 64 == 0: 128^3 volumes) the weight gradient's K-step offsets
 * are wave-uniform; elsewhere (15 patches per axis at the 240^3 UCSF-PDGM shape, dataset_ucsf.py:84-88) every lane places its k-rows
 * per K-step from the running token count (three multiply-high divisions), K-steps may straddle samples.
 * ---------------------------------------------------------------------------------------- */
typedef struct xvit_patch_geom {
  int32_t B, M;          /* img [B, M, 1, D, H, W] */
  int32_t D, H, W;
  int32_t dp, hp, wp;    /* patch_size (model_cross.py:151) */
  int32_t cls_rows;      /* 1: every sample's rows start with one CLS row (ModelCross); 0: patch rows only */
} xvit_patch_geom;
int xvit_patch_embed_supported(const xvit_patch_geom* g, int d);
int xvit_patch_embed_fwd(const void* img_bf16, const xvit_patch_geom* g, const void* W_bf16, int64_t ldw, const float* bias, const float* pos, int64_t ldpos,
                         float* x, int64_t ldx, int d, xvit_stream_t stream);
int64_t xvit_patch_embed_wgrad_workspace_bytes(const xvit_patch_geom* g, int d);
int xvit_patch_embed_wgrad(const void* img_bf16, const xvit_patch_geom* g, const void* dx_bf16, int64_t lddx, float* dW, int64_t lddw, int d, void* workspace,
                           int64_t workspace_bytes, xvit_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Input stage (dataset_ucsf.py:84-88, 152-158): MONAI `ResizeWithPadOrCropd(spatial_size, constant_values=pad_value)`
 * followed by `.to(torch.float)`, on the raw int16 NIfTI voxels already on the device:
 * src int16 [nvol, Ds, Hs, Ws] -> dst bf16 [nvol, D, H, W].  Per dimension: symmetric pad with before = deficit/2
 * when the source is smaller, centre crop with start = size/2 - target/2 when it is larger (MONAI SpatialPad
 * "symmetric" + CenterSpatialCrop).  MONAI is not installed here: PARITY UNPINNED (restated from its documented rule).
 * ---------------------------------------------------------------------------------------- */
int xvit_resize_pad_crop_i16(const void* src_i16, void* dst_bf16, int nvol, int Ds, int Hs, int Ws, int D, int H, int W,
                             float pad_value, xvit_stream_t stream);

/* x[m, b, 0, :] = cls + pos[0]  (model_cross.py:195-197, the CLS row); x fp32 [M*B, N, d] */
int xvit_cls_row_fwd(const float* cls, const float* pos, float* x, int MB, int N, int d, xvit_stream_t stream);
/* dpos[n,:] += sum_{mb} dx[mb,n,:];  dcls += sum_{mb} dx[mb,0,:] */
int xvit_embed_bwd(const float* dx, float* dpos, float* dcls, int MB, int N, int d, xvit_stream_t stream);

/* ---- elementwise / reductions --------------------------------------------------------- */
int xvit_cast_f32_bf16(const float* src, void* dst_bf16, int64_t n, xvit_stream_t stream);
/* out = a + b (fp32) and its bf16 copy in one pass: the sum of a branch output's two gradients (one per reader: its own fusion and the
 * partner's, model_cross.py:140-142) handed to the block before it in both dtypes.  n % 8 == 0; out may be a or b. */
int xvit_add_cast_f32_bf16(const float* a, const float* b, float* out, void* out_bf16, int64_t n, xvit_stream_t stream);
/* dst[r, :] (and dst2[r, :], optional) = a[r, :] + b[r, :] for r < rows, d columns; a NULL operand is zero; each tensor has its own dtype
 * (XVIT_F32 / XVIT_BF16) and row stride in elements.  The CLS-row bookkeeping of the fusions — take the B CLS rows out of a [B, N, d] token
 * tensor (row stride N d), write the fused token back, add / zero / copy the same rows of the gradients (model_cross.py:140-142 and its
 * backward) — one launch per site.  dst may alias a or b. */
int xvit_rows_combine(void* dst, int dst_dtype, int64_t ld_dst, void* dst2, int dst2_dtype, int64_t ld_dst2, const void* a, int a_dtype, int64_t ld_a,
                      const void* b, int b_dtype, int64_t ld_b, int rows, int d, xvit_stream_t stream);
/* out[n] (+)= sum_r x[r, n];  x bf16 or fp32.  workspace (optional, xvit_colsum_workspace_bytes(rows, n) bytes): the row chunks'
 * partial sums are stored there and added in chunk order (bit-reproducible) instead of meeting in fp32 atomics on out. */
int xvit_colsum(const void* x, int x_dtype, int64_t ldx, float* out, int rows, int n, int accumulate, float* workspace, int64_t workspace_bytes,
                xvit_stream_t stream);
int64_t xvit_colsum_workspace_bytes(int rows, int n);
/* dropout with a counter-based mask (same (seed, element index) -> same mask in fwd and bwd):
 * y = x * keep / (1-p).  In-place allowed.  dtype XVIT_BF16 | XVIT_F32. */
int xvit_dropout(const void* x, void* y, int dtype, int64_t n, float p, uint64_t seed, xvit_stream_t stream);

/* Per-step classification statistics kept on the device (replaces log_stats -> compute_metrics, model_cross.py:243-255 /
   utils.py:18-62: six torchmetrics objects, six .item() host syncs and an AUROC per step).  logits fp32 [B, C = 2] (row stride
   ld), labels int64 [B].  pred = argmax; the step's accuracy, precision, recall, specificity, F1, NPV (0 for an empty
   denominator) and exact AUROC of softmax(logits)[:, 1] (0 if a class is absent) are folded into
   state[XVIT_METRIC_STATE] (fp64, caller-zeroed at the start of an epoch):
     [0..3] pooled tn, fp, fn, tp   [4] samples   [5] steps   [6..12] sum of batch_size * {acc, prec, rec, spec, f1, npv, auroc}
   so state[6+k] / state[4] is the batch-size-weighted epoch mean Lightning logs for on_epoch=True.  B <= 8192. */
#define XVIT_METRIC_STATE 16
int xvit_binary_metrics_step(const float* logits, int64_t ld, const int64_t* labels, int B, int C, double* state, xvit_stream_t stream);

/* Diagnostic (no reference counterpart): out[2*b] = XCC (XCD) id and out[2*b+1] = HW_ID register of the CU
   that ran workgroup b of an `nblocks`-block launch on `stream`; every block lingers `linger_us` so the
   launch spreads over all CUs the stream may use.  Maps CU-mask bits of a masked stream to XCDs. */
int xvit_cu_trace(uint32_t* out, int nblocks, int linger_us, xvit_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-tensor Adam step with torch.optim.Adam semantics (model_cross.py:276-278: L2 weight decay added to
 * the gradient, bias correction with `step`, eps outside the square root).  table_dev: device array of
 *   struct { float* p; const float* g; float* m; float* v; void* shadow_bf16 (or NULL); int64_t n; }
 * chunks_dev: device array of int32 pairs (tensor index, chunk index), one block per 16384-element chunk.
 * When shadow_bf16 is given it receives bf16(p) — the GEMM operand copy, so no cast pass follows the step.
 * grad_scale multiplies every gradient first (1 for plain training).
 * ---------------------------------------------------------------------------------------- */
#define XVIT_ADAM_CHUNK 16384
int xvit_adam_step(const void* table_dev, const void* chunks_dev, int n_chunks, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float grad_scale, xvit_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Head tail (model_cross.py:205-211): logits = mean_m logits_m;  loss = CE(logits, labels,
 * label_smoothing), mean over the batch.  Also writes dlogits_m[M,B,C] = d loss / d logits_m.
 * ---------------------------------------------------------------------------------------- */
int xvit_mean_ce(const float* logits_m, const int64_t* labels, float label_smoothing, float* logits, float* loss, float* dlogits_m,
                 int M, int B, int C, xvit_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* XVIT_H */

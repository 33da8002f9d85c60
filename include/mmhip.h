/* libmmhip -- C ABI of the MI355X (gfx950) hot path of tomoqt/multimodal.
 *
 * The reference has no FFI: its hot path is a chain of ATen calls inside nn.Modules
 * (SURVEY.md 8b).  Each entry point below replaces the ATen op sequence cited next to
 * it (paths relative to the reference repository).  Conventions:
 *   - every pointer is a DEVICE pointer unless named host_*; the caller owns all
 *     buffers, including workspaces; nothing here allocates or frees;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued there, never synced;
 *   - activations are f32 (MM_F32) or bf16 (MM_BF16) as selected by `dtype`;
 *     parameters named `const float*` are always f32 master copies;
 *   - return MM_OK or a negative code; mm_last_error() gives the message
 *     (thread-local).  Nothing throws across this boundary.
 */
#ifndef MMHIP_H
#define MMHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_OK 0
#define MM_ERR_ARG (-1)
#define MM_ERR_LAUNCH (-2)
#define MM_ERR_UNSUPPORTED (-3)

enum { MM_F32 = 0, MM_BF16 = 1 };
enum { MM_LAYOUT_BLC = 0, /* channels-last  [B][L][C] */
       MM_LAYOUT_BCL = 1  /* channel-major  [B][C][L] (nn.Conv1d's layout) */ };
enum { MM_ATTN_SOFTMAX = 0, MM_ATTN_STABLEMAX = 1 };

int mm_version(void);
const char* mm_last_error(void);

/* ---- K1: depthwise Conv1d(k=7,pad=3,groups=C) + transpose + LayerNorm(C) ------------------
 * replaces models/convnext.py:38-40 (self.dwconv; transpose(1,2); self.norm).
 * x: [B][L][C] (MM_LAYOUT_BLC) or [B][C][L] (MM_LAYOUT_BCL); y: [B][L][C].
 * w: [C][7] (= dwconv.weight (C,1,7)), b: [C], ln_w/ln_b: [C].
 * mean/rstd: [B*L] row statistics for the backward pass, may both be NULL. */
int mm_dwconv7_ln_fwd(const void* x, const float* w, const float* b, const float* ln_w, const float* ln_b,
                      void* y, float* mean, float* rstd, int64_t B, int64_t C, int64_t L, float eps,
                      int dtype, int in_layout, void* stream);

/* backward of the above (autograd of models/convnext.py:38-40), channels-last x/dy/dx.
 * dres (nullable, [B][L][C]) is added into dx (the block's residual gradient, convnext.py:47).
 * dw:[C][7] db,dln_w,dln_b:[C] are OVERWRITTEN (deterministic two-stage reduction).
 * workspace: at least mm_dwconv7_ln_bwd_workspace(B,C,L) bytes. */
size_t mm_dwconv7_ln_bwd_workspace(int64_t B, int64_t C, int64_t L);
int mm_dwconv7_ln_bwd(const void* x, const void* dy, const void* dres, const float* w, const float* b,
                      const float* ln_w, const float* mean, const float* rstd, void* dx, float* dw, float* db,
                      float* dln_w, float* dln_b, void* workspace, size_t workspace_bytes, int64_t B, int64_t C,
                      int64_t L, int dtype, void* stream);

/* ---- stem: Conv1d(1,C0,k=4,s=4) + LayerNorm(C0), channels-last output ----------------------
 * replaces models/convnext.py:67-72.  x: f32 [B][Lin]; w: [C0][4]; y: [B][Lo][C0], Lo=(Lin-4)/4+1. */
int mm_stem_ln_fwd(const float* x, const float* w, const float* b, const float* ln_w, const float* ln_b, void* y,
                   float* mean, float* rstd, int64_t B, int64_t Lin, int64_t C0, float eps, int dtype, void* stream);
size_t mm_stem_ln_bwd_workspace(int64_t B, int64_t Lin, int64_t C0);
int mm_stem_ln_bwd(const float* x, const void* dy, const float* w, const float* b, const float* ln_w,
                   const float* mean, const float* rstd, float* dx /*nullable [B][Lin]*/, float* dw, float* db,
                   float* dln_w, float* dln_b, void* workspace, size_t workspace_bytes, int64_t B, int64_t Lin,
                   int64_t C0, int dtype, void* stream);

/* ---- row LayerNorm over the last dim ------------------------------------------------------
 * replaces nn.LayerNorm at models/convnext.py:78,:131 and models/transformer_decoder.py:122,:160,:306.
 * x: [G][Lin][C]; y: [G][Lout][C] with Lout <= Lin: rows l >= Lout of each group are dropped
 * (that is Conv1d(k=2,s=2) discarding an odd tail, convnext.py:80).  Plain LN: G=1, Lin=Lout=rows. */
int mm_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                     int64_t G, int64_t Lin, int64_t Lout, int64_t C, float eps, int dtype, void* stream);
size_t mm_layernorm_bwd_workspace(int64_t rows, int64_t C);
int mm_layernorm_bwd(const void* x, const void* dy, const void* dres /*nullable, x-shaped, added into dx*/,
                     const float* w, const float* mean, const float* rstd, void* dx, float* dw, float* db,
                     void* workspace, size_t workspace_bytes, int64_t G, int64_t Lin, int64_t Lout, int64_t C,
                     int dtype, void* stream);
/* Pre-LN residual block  y = x + Linear(...LN(x)...)  (models/transformer_decoder.py:122-127,:157-163): the same backward,
 * dx = dres + LN'(dy), which also returns dres_sum[c] = sum_rows dres and dx_sum[c] = sum_rows dx -- autograd's bias
 * gradients of the Linear after the residual add and of the Linear that produced x -- without a pass of their own.
 * Wide rows only: C > 512 and C % 4 == 0; workspace as mm_layernorm_bwd_workspace(rows, C). */
int mm_layernorm_bwd_sums(const void* x, const void* dy, const void* dres, const float* w, const float* mean,
                          const float* rstd, void* dx, float* dw, float* db, float* dres_sum, float* dx_sum,
                          void* workspace, size_t workspace_bytes, int64_t rows, int64_t C, int dtype, void* stream);

/* ---- GEMM family (MFMA; bf16 or exact-f32 matrix cores) -----------------------------------
 * out[M][N] = epilogue( A[M][K] . B^T ), B given as [N][K] (nn.Linear weight layout).
 * Replaces F.linear / nn.Linear at models/convnext.py:41,:43,:80,:134 and the matmuls autograd
 * derives from them.  Operand layouts: *_kmajor = 0 means the reduction index is contiguous
 * ([rows][K], leading dimension ld*), 1 means it is the slow index ([K][rows]).
 * Epilogues (applied in f32 on the accumulator, in this order):
 *   acc += bias[n]                       if bias
 *   if pre:  pre[m][n] = acc             (saved pre-activation, training)
 *   act: 0 none | 1 GELU(erf) (convnext.py:42) | 2 relu(x)^2 (transformer_decoder.py:161)
 *   if mul:  acc *= dact(mul[m][n])      (backward: mul holds the saved pre-activation; act selects
 *                                         the DERIVATIVE: 1 GELU', 2 2*relu)  [act then means dact]
 *   if gamma: acc *= gamma[n]            (LayerScale, convnext.py:45)
 *   if rowscale: acc *= rowscale[m / rows_per_scale]   (DropPath factor, convnext.py:17)
 *   if res:  acc += res[m][n]            (residual, convnext.py:47)
 * split_k > 1 needs workspace of split_k*M*N floats and a plain epilogue (bias only).
 * split_k <= 1 with a workspace given (any size; 8*M*N floats covers every case): the library may split K on its own for
 * skinny problems with a deep K (about 1000 rows: encoder stage 3 at the config batch, decode steps), keeping the full
 * epilogue (it is applied by the reduction pass); without a workspace it never splits. */
typedef struct {
    const void* A; const void* B; void* out;
    int64_t M, N, K;
    int64_t lda, ldb, ldo;
    int a_kmajor, b_kmajor;
    int dtype;            /* dtype of A, B (MM_F32 | MM_BF16) */
    int out_dtype;        /* dtype of out, pre, res, mul */
    const float* bias;    /* [N] or NULL */
    void* pre;            /* [M][N] (ldo) or NULL */
    const void* mul;      /* [M][N] (ldo) or NULL */
    int act;
    const float* gamma;   /* [N] or NULL */
    const float* rowscale; int64_t rows_per_scale;
    const void* res;      /* [M][N] (ldo) or NULL */
    int split_k; void* workspace; size_t workspace_bytes;
} mm_gemm_args;
int mm_gemm(const mm_gemm_args* args, void* stream);

/* column sums: out[n] = sum_m x[m][n] (bias gradients); optional gamma/z terms for LayerScale:
 * replaces the bias-gradient reductions autograd derives from nn.Linear. workspace: see _workspace. */
size_t mm_colsum_workspace(int64_t M, int64_t N);
int mm_colsum(const void* x, float* out, void* workspace, size_t workspace_bytes, int64_t M, int64_t N,
              int64_t ld, int dtype, void* stream);

/* mm_colsum without a workspace argument and without any allocation.  No counterpart in the reference: it exists for
 * callers that cannot allocate or carry a workspace -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  The [chunks][N] partial sums live in a 5 MiB device array built into the library (one copy per GPU,
 * held until the process ends), so N <= 8192 (larger N: MM_ERR_ARG, use mm_colsum).  Results are bit-identical to mm_colsum.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (different streams) race. */
int mm_colsum_noalloc(const void* x, float* out, int64_t M, int64_t N, int64_t ld, int dtype, void* stream);

/* ---- K3: prefix-LM attention with partial RoPE, softmax or stablemax ----------------------
 * replaces models/transformer_decoder.py:129-153 (+ the mask of :295-298, built in-kernel).
 * q,k,v: [B][S][H][d] with row stride ld_qkv elements between consecutive s (so a fused [B][S][3E]
 * projection works); o: [B][S][H][d] contiguous.  Query i may see key j iff j <= i or
 * (i < prompt_len and j < prompt_len).  RoPE (NeoX half rotation) is applied to the first rot dims
 * of q and k at absolute position s, from cos/sin tables [S][rot/2] (f32).
 * lse: [B][H][S] f32 row normaliser saved for backward (softmax: m + log l; stablemax: sum of s),
 * may be NULL. */
int mm_prefix_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* cos_t,
                       const float* sin_t, int64_t B, int64_t H, int64_t S, int64_t d, int64_t rot,
                       int64_t prompt_len, int64_t ld_qkv, int mode, int dtype, void* stream);
/* backward of the above (autograd of transformer_decoder.py:129-153): dq,dk,dv are gradients w.r.t. the
 * UN-rotated q,k,v, laid out [B][S][H][d] with row stride ld_dqkv; o,d_o are [B][S][H][d] contiguous.
 * workspace: mm_prefix_attn_bwd_workspace(B,H,S) bytes (row sums of dO*O).  Deterministic, no atomics. */
size_t mm_prefix_attn_bwd_workspace(int64_t B, int64_t H, int64_t S);
int mm_prefix_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                       void* dq, void* dk, void* dv, const float* cos_t, const float* sin_t, void* workspace,
                       size_t workspace_bytes, int64_t B, int64_t H, int64_t S, int64_t d, int64_t rot,
                       int64_t prompt_len, int64_t ld_qkv, int64_t ld_dqkv, int mode, int dtype, void* stream);

/* RoPE pre-pass (models/transformer_decoder.py:129-135 hoisted out of the attention kernel): rotates in place the
 * first `rot` dims of every head of x [B][S][H][d] (row stride ld) at absolute position s; inverse=1 applies the
 * transpose (gradients).  With q,k pre-rotated, call the attention entry points with rot = 0. */
int mm_rope_inplace(void* x, const float* cos_t, const float* sin_t, int64_t B, int64_t S, int64_t H, int64_t d,
                    int64_t rot, int64_t ld, int inverse, int dtype, void* stream);

/* ---- K4: one decode step of attention against a KV cache ----------------------------------
 * no counterpart in the reference (it recomputes the full decoder every step,
 * train_autoregressive.py:97-101); equals row `pos` of K3 by the argument of SURVEY.md A9.
 * q,k_new,v_new: [B][H][d] for the token at absolute position pos (ld_qkv = elements between b rows).
 * RoPE is applied to q and k_new; k_new (rotated) and v_new are appended at slot pos of
 * kcache/vcache [B][H][Smax][d]; o: [B][H][d] = attention of q over slots 0..pos.
 * d <= 512, d % 4 == 0, Smax <= 1024.  Slots at or beyond pos are never read (the caches may hold anything there).
 * bf16 runs in one pass (K and V rows of a position read together, non-temporally; the softmax weights are exp2 of
 * base-2 scores against a running maximum per 16-lane group, merged at the end); f32 keeps two passes and the op order
 * of the prefill kernel. */
int mm_decode_attn(const void* q, const void* k_new, const void* v_new, void* kcache, void* vcache, void* o,
                   const float* cos_t, const float* sin_t, int64_t B, int64_t H, int64_t d, int64_t rot,
                   int64_t pos, int64_t Smax, int64_t ld_qkv, int mode, int dtype, void* stream);
/* The same step with the position read from device memory (*pos_dev, clamped to [0, Smax-1]), for a decode step captured
 * once in a HIP graph and replayed per token (multimodal_amd/generate.py). */
int mm_decode_attn_dev(const void* q, const void* k_new, const void* v_new, void* kcache, void* vcache, void* o,
                       const float* cos_t, const float* sin_t, int64_t B, int64_t H, int64_t d, int64_t rot,
                       const int32_t* pos_dev, int64_t Smax, int64_t ld_qkv, int mode, int dtype, void* stream);
/* prefill: rotate K of the prompt rows and copy K,V into the cache (same layout as above). */
int mm_kv_cache_fill(const void* k, const void* v, void* kcache, void* vcache, const float* cos_t, const float* sin_t,
                     int64_t B, int64_t H, int64_t S, int64_t d, int64_t rot, int64_t Smax, int64_t ld_qkv,
                     int dtype, void* stream);

/* ---- small fused pieces --------------------------------------------------------------------
 * LayerScale + DropPath backward (autograd of models/convnext.py:45-47):
 *   dz = dout * gamma[c] * rowscale[m/rows_per_scale];  dgamma[c] = sum_m dout*rowscale*z;  dbias[c] = sum_m dz.
 * dout,z,dz: [M][C]; deterministic two-stage reduction through `workspace`. */
size_t mm_layerscale_bwd_workspace(int64_t M, int64_t C);
int mm_layerscale_bwd(const void* dout, const void* z, const float* gamma, const float* rowscale, int64_t rows_per_scale,
                      void* dz, float* dgamma, float* dbias, void* workspace, size_t workspace_bytes, int64_t M,
                      int64_t C, int dtype, void* stream);
/* argmax over the last dim of logits [rows][V] (row stride ld) -> int64 ids (train_autoregressive.py:102);
 * the first maximal index wins, as torch.argmax does. */
int mm_argmax_rows(const void* logits, int64_t* out, int64_t rows, int64_t V, int64_t ld, int dtype, void* stream);
/* Decoder MLP activation relu(h)^2 (models/transformer_decoder.py:161) over n elements, and its backward
 * dh = da * 2*relu(h); dh may alias da. */
/* DropPath per-sample factor (models/convnext.py:13-17): out[i] = floor(keep + u[i]) / keep, u = the torch.rand((B,1,1)) draw. */
int mm_droppath_scale(const float* u, float* out, int64_t n, float keep, void* stream);
int mm_relu2_fwd(const void* h, void* a, int64_t n, int dtype, void* stream);
int mm_relu2_bwd(const void* h, const void* da, void* dh, int64_t n, int dtype, void* stream);
/* relu^2 backward over [M][N] plus colsum[n] = sum_m dh[m][n], autograd's gradient of the bias of the Linear that produced h
 * (models/transformer_decoder.py:161), in the same pass; workspace as mm_colsum_workspace(M, N); rows 16-byte aligned. */
int mm_relu2_bwd_colsum(const void* h, const void* da, void* dh, float* colsum, void* workspace, size_t workspace_bytes,
                        int64_t M, int64_t N, int dtype, void* stream);
/* out[c][r] = (out_dtype) in[r][c]: transposed compute-dtype copy of an f32 [R][C] weight (row stride ldo >= R), the
 * operand that makes dX = dY . W (autograd of nn.Linear, models/transformer_decoder.py:159-163) K-contiguous. */
int mm_cast_transpose(const float* in, void* out, int64_t R, int64_t C, int64_t ldo, int out_dtype, void* stream);
/* The same for many weights in ONE launch (after an optimizer step: every transposed bf16 operand copy of the decoder).
 * descs / tile_desc live in device memory: descs[t] describes weight t (src = f32 master or its bf16 copy, [R][C]; dst = bf16
 * [C][ldo]; tile0 = index of its first 64 x 64 tile), tile_desc[b] = t for launch block b, n_tiles = sum of ceil(R/64) *
 * ceil(C/64).  R, C, ldo multiples of 8, bases 16-byte aligned.  Same reference lines as mm_cast_transpose. */
typedef struct mm_transpose_desc {
    const void* src; void* dst;
    int64_t R, C, ldo, tile0;
    int64_t src_dtype;                 /* MM_F32 or MM_BF16 */
} mm_transpose_desc;
int mm_cast_transpose_multi(const mm_transpose_desc* descs, const int32_t* tile_desc, int64_t n_tiles, void* stream);

/* ---- optimizer: OrthoGrad projection + AdamW, multi-tensor (next-row N3) ---------------------
 * replaces ortho_grad.py:60-99 + torch.optim.AdamW.step.  All tables are DEVICE arrays the caller builds once:
 * descs[n_tensors]; the parameters are cut into chunks of chunk_elems elements: chunk_tensor[c], chunk_off[c];
 * tensor_chunk_start[n_tensors+1] (chunks of a tensor are consecutive).  g is overwritten with the projected
 * gradient (as the reference does), p/m/v are updated in place.  step counts from 1. */
typedef struct {
    float* p; float* g; float* m; float* v; int64_t n;
    void* w16;      /* optional bf16 copy of p (n elements) refreshed by the update, or NULL: the GEMM operand of the next step */
} mm_param_desc;
size_t mm_orthograd_adamw_workspace(int64_t n_tensors, int64_t n_chunks);
int mm_orthograd_adamw(const mm_param_desc* descs, const int32_t* chunk_tensor, const int64_t* chunk_off,
                       const int32_t* tensor_chunk_start, int64_t n_tensors, int64_t n_chunks, int64_t chunk_elems,
                       float lr, float beta1, float beta2, float adam_eps, float weight_decay, int64_t step,
                       float ortho_eps, int rescale, int do_ortho, void* workspace, size_t workspace_bytes, void* stream);
/* The same update with the step-dependent scalars in DEVICE memory: hyper = {lr, 1 - beta1^step, sqrt(1 - beta2^step)}.
 * For callers that replay a captured HIP graph of the train step: the host rewrites the three floats before each replay
 * (multimodal_amd.optim.OrthoGradAdamW(capturable=True), multimodal_amd.graph.CapturedTrainStep). */
int mm_orthograd_adamw_dev(const mm_param_desc* descs, const int32_t* chunk_tensor, const int64_t* chunk_off,
                           const int32_t* tensor_chunk_start, int64_t n_tensors, int64_t n_chunks, int64_t chunk_elems,
                           const float* hyper, float beta1, float beta2, float adam_eps, float weight_decay,
                           float ortho_eps, int rescale, int do_ortho, void* workspace, size_t workspace_bytes,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif

/*
 * dkd.h -- C ABI of libdkd.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the
 * DeiT teacher->student distillation training step.
 *
 * The reference (serizard/DeltaKD) has no native code and no FFI: every op below replaces the
 * ATen/cuBLAS/cuDNN kernel that the cited reference line causes through timm==0.9.12 / torch.
 * (file:line are relative to /root/reference; [3P] = inside timm, reached from that call site.)
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller (torch tensors on the
 *     Python side); the library never allocates, frees or retains caller memory after return.
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - return 0 on success, <0 on error; dkd_last_error() returns a thread-local message.
 *   - bf16 tensors are raw uint16_t payloads; "f32" = float.
 *   - global state is limited to write-once caches of device properties (CU count, a kernel attribute set on first use) and the
 *     launch probe of bench.py (mutex-protected, off by default); nothing else is shared between calls, so the entry points are
 *     safe to call from torch's main and autograd threads concurrently.  One process drives one GPU.
 *   - scratch memory is always passed in by the caller; the dkd_*_workspace_bytes() queries say how much.
 */
#ifndef DKD_H
#define DKD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DKD_OK 0
#define DKD_ERR_ARG (-1)
#define DKD_ERR_HIP (-2)
#define DKD_ERR_UNSUPPORTED (-3)

int dkd_version(void);
const char* dkd_last_error(void);
/* number of compute units / name of device 0 as seen by HIP (diagnostics for bench.py) */
int dkd_device_info(int device, int* cu_count, char* name, int name_len);

/* Row map: physical_row(m) = rpg > 0 ? (m / rpg) * gstride + (m % rpg) + off : m.
 * Expresses "all patch tokens of every sample, prefix tokens stripped" (feat[:, npre:], model/loss.py:88-98)
 * and "broadcast over the batch" (gstride = 0: pos_embed) without materialising a copy. */
typedef struct { int32_t rpg, gstride, off; } DkdRowMap;

/* ---------------------------------------------------------------- GEMM (bf16 MFMA, fp32 accumulate) */
enum {
  DKD_EPI_BIAS        = 1 << 0,  /* v += bias[n]                                             */
  DKD_EPI_GELU        = 1 << 1,  /* v = gelu_erf(v); if preact != NULL the pre-activation is stored (bf16) */
  DKD_EPI_DGELU       = 1 << 2,  /* v *= gelu'(preact[m,n])   (backward of fc1's GELU)        */
  DKD_EPI_RESID       = 1 << 3,  /* v = resid[rmap(m), n] + rowscale[m / rows_per_sample] * v */
  DKD_EPI_OUT_F32     = 1 << 4,  /* C is float (default: bf16)                               */
  DKD_EPI_TAP_F32     = 1 << 5,  /* tap (value before RESID) stored as float (default bf16)   */
  DKD_EPI_RELU        = 1 << 6,  /* v = max(v, 0)                                            */
  DKD_EPI_ACCUM       = 1 << 7,  /* C += v (f32 output only)                                  */
  DKD_EPI_RELU_GATE   = 1 << 8   /* v = preact[m,n] > 0 ? v : 0  (backward of a ReLU whose OUTPUT is given as `preact`) */
};

typedef struct {
  const void* A;      /* bf16 [M, K], row stride lda (elements), rows through amap            */
  const void* B;      /* bf16 [N, K], row stride ldb : C = A * B^T  (torch Linear weight layout) */
  void* C;            /* bf16 or f32 [M, N], row stride ldc, rows through cmap                */
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  DkdRowMap amap, cmap;
  uint32_t epi;       /* DKD_EPI_* flags                                                      */
  const float* bias;  /* f32 [N]                                                              */
  const float* resid; /* f32, row stride ldr, rows through rmap                               */
  int32_t ldr;
  DkdRowMap rmap;
  const float* rowscale; /* f32 [M / rows_per_sample] (DropPath keep/keep_prob) or NULL (=1)    */
  int32_t rows_per_sample;
  void* preact;       /* bf16 [M, N] row stride ldp: written by GELU, read by DGELU           */
  int32_t ldp;
  void* tap;          /* optional copy of (acc + bias) before RESID (the feature tap of
                         model/models.py:189-191), row stride ldt, rows = m                   */
  int32_t ldt;
  int32_t conv_hw;    /* > 0: implicit GEMM of a 3 x 3 / pad 1 convolution on the hw x hw token grid (MGD generation block,
                         model/models.py:148-151, model/loss.py:443-446): A is the activation x bf16 [B * hw * hw, Cin] (lda >= Cin),
                         K = 9 * Cin, B the weight as [N, (ky, kx, cin)]; the 3 x 3 neighbourhood is gathered by the kernel's
                         source addressing (zero padding included), no [M, 9 Cin] matrix exists.  The input gradient is the same
                         call on dY with the weight flipped and transposed ([Cin, (2 - ky, 2 - kx, cout)]).  Cin % 64 == 0. */
  /* LayerNorm folded into the GEMMs around it ([3P] timm Block: x -> norm -> Linear).  For a frozen (inference) model the separate
   * LayerNorm pass -- a pure streaming kernel at the HBM rate -- disappears:
   *   producer (the residual GEMM that writes x: proj / fc2, epilogue BIAS | RESID | OUT_F32 with identity row maps): also stores
   *     xb = bf16(x) (row stride ldxb) and adds every row's  sum x, sum x^2  over this GEMM's N columns into rowstats f32 [M][2]
   *     (atomically: the caller zeroes it; a row's N columns may come from several tiles / launches);
   *   consumer (the Linear behind the norm, epilogue BIAS or BIAS | GELU, bf16 output): A = xb, B = bf16(gamma * W), bias = W beta + b,
   *     ln_c f32 [N] = row sums of that B, ln_stats = the producer's rowstats (over exactly K columns):
   *         C[m, n] = rstd[m] (acc[m, n] - mean[m] ln_c[n]) + bias[n],   mean = s1 / K, rstd = rsqrt(s2 / K - mean^2 + ln_eps)
   * which equals Linear(LayerNorm(x)) with x rounded to bf16 BEFORE it is centred instead of after (same error against fp32:
   * tools_dev/ln_fold_probe.py, tests/test_fullsize_gpu.py).  Taken by the kernels that serve the wide teacher GEMMs; others refuse. */
  void* xb;
  int32_t ldxb;
  float* rowstats;
  const float* ln_stats;
  const float* ln_c;
  float ln_eps;
} DkdGemm;

/* C[M,N] = epilogue(A[M,K] * B[N,K]^T).  Replaces nn.Linear / Conv2d-as-GEMM forward and the dgrad GEMMs
 * ([3P] timm Attention/Mlp/PatchEmbed reached from model/models.py:195; align layers model/loss.py:89-91,426).
 * Requires K % 64 == 0 and 16-byte aligned rows. */
int dkd_gemm_nt(const DkdGemm* g, void* stream);

/* C[N1,N2] += sum_m A[m,N1] * B[m,N2]   (weight gradients; f32 atomics into C, row stride ldc).
 * A, B bf16 with row strides lda/ldb and row maps; M is the reduction length.
 * a_colsum (optional, f32 [N1]) += sum_m A[m, :]: the bias gradient, fused (A = dY is already streaming through LDS). */
int dkd_gemm_tn(const void* A, const void* B, float* C, int32_t M, int32_t N1, int32_t N2, int32_t lda, int32_t ldb,
                int32_t ldc, DkdRowMap amap, DkdRowMap bmap, float* a_colsum, void* stream);

/* Weight (and bias) gradient of that convolution without an im2col matrix: dW f32 [Cout, 3, 3, Cin] (i.e. [Cout, (ky, kx, cin)], the
 * layout the forward consumes) += sum_m dY[m, o] * x[m + (ky-1) hw + (kx-1), c] over the pixels whose neighbour lies inside the image;
 * dbias f32 [Cout] += column sums of dY (may be NULL).  dY bf16 [B*hw*hw, Cout], x bf16 [B*hw*hw, Cin], both contiguous.
 * One launch of the split-M TN kernel, the nine taps ([Cout, Cin] blocks) side by side in grid.z. */
int dkd_conv3x3_wgrad(const void* dY, const void* x, float* dW, float* dbias, int32_t B, int32_t hw, int32_t Cin, int32_t Cout,
                      void* stream);

/* Upper triangle (128 x 128 tile granularity) of the Gram matrix C[N,N] += A[M,N]^T A[M,N]: the tiles below the diagonal are not
 * touched -- the caller mirrors them (LRKD: G = T^T T of the teacher feature matrix, model/loss.py:318-321; 21 of 36 tiles for
 * N = 768). */
int dkd_gram(const void* A, float* C, int32_t M, int32_t N, int32_t lda, int32_t ldc, DkdRowMap amap, void* stream);
/* The same for L matrices of one shape at constant strides (elements) in ONE launch: A_l = A + l stride_a, C_l = C + l stride_c
 * (C_l zero on entry).  The LRKD targets' three teacher taps (model/loss.py:318-324, blocks 0 / 1 / 11). */
int dkd_gram_batched(const void* A, int64_t stride_a, float* C, int64_t stride_c, int32_t L, int32_t M, int32_t N, int32_t lda, int32_t ldc,
                     DkdRowMap amap, void* stream);

/* Up to 24 independent weight gradients in one launch (the two Linear layers of an MLP, proj + qkv, all four of a transformer block,
 * or those of several blocks): same arithmetic as dkd_gemm_tn per problem, but their blocks share the GPU -- every 128-column tile of
 * every problem gets the same number of M splits, chosen so that the launch is one round of workgroups that end together; the more
 * tiles, the fewer atomically added partial tiles per gradient.  Problems the grouped kernel does not take (see gemm.hip) are
 * launched on their own. */
typedef struct DkdTnProblem {
  const void* A;
  const void* B;
  float* C;
  float* a_colsum;
  int32_t M, N1, N2, lda, ldb, ldc;
  DkdRowMap amap, bmap;
} DkdTnProblem;
int dkd_gemm_tn_group(const DkdTnProblem* problems, int32_t n, void* stream);
/* The same launch for the weight gradients a caller deferred from dkd_block_bwd (DkdBlockGrads.defer_wgrad) -- typically those of
 * SEVERAL consecutive blocks at once (the student walks its blocks backward and flushes every six: 24 problems, 114 tiles, 4 M splits
 * instead of 26, one ring fill / atomic tail instead of six).  Autograd's accumulation of the weight gradients of nn.Linear
 * (tools/engine.py:61-62 -> torch.autograd) is what it replaces; the time is booked under the block-backward probe scope. */
int dkd_block_wgrad_group(const DkdTnProblem* problems, int32_t n, void* stream);

/* ---------------------------------------------------------------- attention ([3P] F.scaled_dot_product_attention) */
/* qkv bf16 [B, N, 3, H, 64] (the fused qkv Linear output, no head-split copy); out bf16 [B, N, H*64];
 * lse f32 [B, H, N] (natural-log sum-exp of the scaled scores, saved for backward; may be NULL). N <= 256. */
int dkd_attn_fwd(const void* qkv, void* out, float* lse, int32_t B, int32_t N, int32_t H, void* stream);
/* dqkv bf16 [B, N, 3, H, 64] from dout bf16 [B, N, H*64]. */
int dkd_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int32_t B, int32_t N,
                 int32_t H, void* stream);

/* ---------------------------------------------------------------- LayerNorm ([3P] nn.LayerNorm eps 1e-6) */
/* x f32 [M, D] (rows through xmap) -> y bf16 [M, D] contiguous rows; mean/rstd f32 [M] saved when non-NULL. */
int dkd_layernorm_fwd(const float* x, int32_t ldx, DkdRowMap xmap, const float* gamma, const float* beta, void* y,
                      float* mean, float* rstd, int32_t M, int32_t D, float eps, int32_t y_is_f32, void* stream);
/* dx f32 [M, D] = (accumulate ? dx : 0) + LN'(dy); dgamma/dbeta f32 [D] +=.  dy bf16 or f32 [M, D].
 * ws: NULL (every block adds its partial sums atomically) or 2 * D * ceil(M / 64) floats of scratch: per-block partial rows summed by
 * a second small launch (no same-address atomics: about a third faster at M = 50k). */
int dkd_layernorm_bwd(const void* dy, int32_t dy_is_f32, const float* x, int32_t ldx, DkdRowMap xmap, const float* gamma,
                      const float* mean, const float* rstd, float* dx, int32_t lddx, DkdRowMap dxmap, int32_t accumulate,
                      float* dgamma, float* dbeta, int32_t M, int32_t D, float* ws, void* stream);

/* Fused dgrad GEMM + LayerNorm backward for LayerNorm width D = 192 (the DeiT-tiny student; [3P] timm Block: the dgrad of fc1 /
 * qkv feeds the backward of norm2 / norm1): dT = A[M,K] W[192,K]^T stays on chip (f32) and
 *     dx[m,:] += LN'(dT[m,:]) ; dgamma += sum_m dT xhat ; dbeta += sum_m dT        (same arithmetic as dkd_layernorm_bwd, accumulate = 1)
 * cast_out (optional, bf16 [M,192]) = rowscale[m / rows_per_sample] * (updated dx): the scale-cast that opens the next branch.
 * ws: dkd_layernorm_bwd_workspace_bytes(M, 192) bytes.  K % 64 == 0. */
int dkd_gemm_nt_lnbwd(const void* A, const void* W, int32_t M, int32_t K, int32_t lda, int32_t ldb, const float* x, int32_t ldx,
                      const float* gamma, const float* mean, const float* rstd, float* dx, int32_t lddx, float* dgamma, float* dbeta,
                      float* ws, void* cast_out, const float* rowscale, int32_t rows_per_sample, void* stream);

/* The whole MLP branch of a D = 192 block (the DeiT-tiny student; [3P] timm Block: x = x + drop_path(mlp(norm2(x))), whose fc2 output
 * is the feature model/models.py:185-193 taps) as ONE kernel per direction (csrc/mlp192.hip): the activation between the two GEMMs
 * stays in registers.
 *   forward : y2 = LN(x1); pre = y2 W1^T + b1; h = gelu(pre); f = h W2^T + b2; tap = bf16(f); x2 = x1 + rowscale[m / rows_per_sample] f
 *   backward: dF = bf16(s2 g + gtap); dH = (dF W2) * gelu'(pre); dT = dH W1; g += LN'(dT); d_ln_w / d_ln_b +=; cast_out = bf16(s1 g)
 * fc1_w bf16 [hidden, 192] (nn.Linear layout), fc2_wt bf16 [hidden, 192] = fc2.weight^T.  hidden % 64 == 0.
 * Saved activations (forward: all five or none -- none = inference; x2 may then alias x1): y2 bf16 [Mp, 192], h bf16 [Mp, hidden] row-major
 * with Mp = M rounded up to 16 (the kernels store whole 16-row groups without predicates), mean / rstd f32 [M], and `pre`:
 * Mp * hidden bf16 in a FRAGMENT-NATIVE order (per 16-row group and 32-unit step, one uint4 per lane) that only dkd_mlp192_bwd reads.
 * backward: dF bf16 [Mp, 192] and dH bf16 [Mp, hidden] are outputs for the weight-gradient launch (dW2 = dF^T h, dW1 = dH^T y2);
 * ws: dkd_layernorm_bwd_workspace_bytes(M, 192) bytes.  gtap, s1, s2, cast_out, tap may be NULL.
 * next_*: optional (all five or none) -- the NEXT block's norm1 applied to the finished rows of x2 in the same epilogue:
 * next_y bf16 [M, 192] = LayerNorm(x2; next_ln_w, next_ln_b, eps), next_mean / next_rstd f32 [M]; that block's LayerNorm launch
 * disappears (DkdBlock.ln1_ready). */
int dkd_mlp192_fwd(const float* x1, const float* ln_w, const float* ln_b, float eps, const void* fc1_w, const float* fc1_b,
                   const void* fc2_wt, const float* fc2_b, const float* rowscale, int32_t rows_per_sample, float* x2, void* tap, void* y2,
                   void* pre, void* h, float* mean, float* rstd, const float* next_ln_w, const float* next_ln_b, void* next_y,
                   float* next_mean, float* next_rstd, int32_t M, int32_t hidden, void* stream);
/* The qkv projection and the attention of a D = 192, 3-head block in one launch ([3P] timm Attention.forward up to proj; one workgroup per
 * sample, N <= 208): qkv bf16 [B*N, 576] = y1 Wqkv^T + b (written once, for the backward), o bf16 [B*N, 192] = softmax(q k^T / 8) v per
 * head, lse f32 [B, 3, N] (may be NULL).  y1 bf16 [B*N, 192] (norm1 output), wqkv bf16 [576, 192], bqkv f32 [576].  Same results as
 * dkd_gemm_nt(BIAS) + dkd_attn_fwd up to the summation order (DkdBlock.fuse_attn selects it). */
int dkd_attn192_fwd(const void* y1, const void* wqkv, const float* bqkv, void* qkv, void* o, float* lse, int32_t B, int32_t N, void* stream);
/* The same launch carried through the rest of the branch ([3P] timm Block: x = x + drop_path(attn(norm1(x))), round 4):
 *     x1[m, :] = x[m, :] + rowscale[m / N] * (o[m, :] proj_w^T + proj_b)
 * proj_w bf16 [192, 192] (nn.Linear layout), proj_b f32 [192], x / x1 f32 [B*N, 192] (x1 may alias x), rowscale f32 [B] or NULL (= 1).
 * qkv, o and lse are written as by dkd_attn192_fwd (the backward reads them).  Same results as dkd_attn192_fwd + dkd_gemm_nt(BIAS | RESID |
 * OUT_F32, rowscale) up to the summation order. */
int dkd_attn192_fwd_proj(const void* y1, const void* wqkv, const float* bqkv, void* qkv, void* o, float* lse, const void* proj_w,
                         const float* proj_b, const float* x, const float* rowscale, float* x1, int32_t B, int32_t N, void* stream);
/* Backward of the same branch, one launch ([3P] autograd of timm Attention.forward and of the LayerNorm in front of it, reached from
 * the reference's loss_scaler call at tools/engine.py:61-62): dO = dy proj.weight (never written: each head's slice is computed into the
 * LDS image the attention backward reads), then dq, dk, dv per head into dqkv bf16 [B*N, 576] (the qkv weight gradient reads it).
 * dy bf16 [B*N, 192] (gradient w.r.t. proj's output, DropPath scale applied), proj_wt bf16 [192, 192] = proj.weight^T, qkv / o / lse as
 * dkd_attn192_fwd wrote them.  8 <= N <= 208.
 * With qkv_wt (bf16 [192, 576] = qkv.weight^T; NULL: stop at dqkv) the kernel goes on to the branch's input: dT = dqkv qkv.weight stays
 * on chip (f32) and  g[m,:] += LN'(dT[m,:]),  d_ln_w += sum_m dT xhat,  d_ln_b += sum_m dT  -- the arithmetic of dkd_gemm_nt_lnbwd -- with
 * x f32 [B*N, 192] the block's input, ln_w = norm1.weight, mean / rstd f32 [B*N] norm1's saved statistics, g f32 [B*N, 192] the gradient
 * stream, ws dkd_layernorm_bwd_workspace_bytes(B*N, 192) bytes (one partial row per workgroup, summed by the reduction launch, which
 * dkd_block_bwd may defer: DkdBlockGrads.ln_defer).
 * Same results as dkd_gemm_nt(dy, proj_wt) + dkd_attn_bwd (+ dkd_gemm_nt_lnbwd) up to one bf16 rounding of dO less and the summation
 * order (DkdBlock.fuse_attn selects it in dkd_block_bwd). */
int dkd_attn192_bwd(const void* dy, const void* proj_wt, const void* qkv, const void* o, const float* lse, void* dqkv, const void* qkv_wt,
                    const float* x, const float* ln_w, const float* mean, const float* rstd, float* g, float* d_ln_w, float* d_ln_b,
                    float* ws, int32_t B, int32_t N, void* stream);
int dkd_mlp192_bwd(float* g, const void* gtap, const float* s2, const float* s1, int32_t rows_per_sample, const void* pre,
                   const void* fc2_wt, const void* fc1_w, const float* x1, const float* ln_w, const float* mean, const float* rstd, void* dF,
                   void* dH, void* cast_out, float* d_ln_w, float* d_ln_b, float* ws, int32_t M, int32_t hidden, void* stream);

/* ---------------------------------------------------------------- data movement / elementwise */
/* img f32 [B, C, H, W] -> patches bf16 [B*(H/p)*(W/p), C*p*p] in Conv2d weight order (c, i, j). ([3P] PatchEmbed) */
int dkd_im2col_patches(const float* img, void* patches, int32_t B, int32_t C, int32_t H, int32_t W, int32_t p, void* stream);
/* x[b, t, :] = tok[t, :] + pos[t, :] for the npre prefix tokens (cls[, dist]); x f32 [B, N, D]. */
int dkd_prefix_tokens_fwd(float* x, const float* tok, const float* pos, int32_t B, int32_t N, int32_t D, int32_t npre, void* stream);
/* dtok[t,:] += sum_b dx[b,t,:] (t < npre);  dpos[t,:] += sum_b dx[b,t,:] (all t). */
int dkd_embed_bwd(const float* dx, float* dtok, float* dpos, int32_t B, int32_t N, int32_t D, int32_t npre, void* stream);
/* y bf16 [M, D] = (rowscale ? rowscale[m / rows_per_sample] : 1) * x[xmap(m)] (+ add[m] if add) ; x f32, add f32|bf16. */
int dkd_scale_cast_bf16(const float* x, int32_t ldx, DkdRowMap xmap, const float* rowscale, int32_t rows_per_sample,
                        const void* add, int32_t add_is_f32, int32_t ldadd, void* y, int32_t ldy, int32_t M, int32_t D, void* stream);
/* f32 -> bf16 flat cast; optionally also the transpose of a [rows, cols] matrix (w_t may be NULL). */
int dkd_cast_weight(const float* w, void* w_bf16, void* w_t_bf16, int32_t rows, int32_t cols, void* stream);
/* The transposed casts of a whole table of matrices in ONE launch.  `items` lives in DEVICE memory (the caller builds it once: the
 * pointers of parameters and shadows are stable); first_tile = number of 32 x 32 tiles of all earlier items; total_tiles = the sum. */
typedef struct DkdCastItem {
  const float* w;      /* f32 [rows, cols] */
  void* wt;            /* bf16 [cols, rows] */
  int32_t rows, cols, first_tile, pad_;
} DkdCastItem;
int dkd_cast_weight_group(const DkdCastItem* items, int32_t n, int32_t total_tiles, void* stream);
/* out[n] += sum_m x[amap(m), n]; x bf16 (or f32 if x_is_f32) [M, N] : bias gradients. */
int dkd_colsum(const void* x, int32_t x_is_f32, int32_t ldx, DkdRowMap xmap, float* out, int32_t M, int32_t N, void* stream);
/* y f32 [M, D] (+)= x (bf16 or f32) scattered through ymap rows (gradient of a token-strip view). */
int dkd_add_rows(const void* x, int32_t x_is_f32, int32_t ldx, float* y, int32_t ldy, DkdRowMap ymap, int32_t M, int32_t D,
                 int32_t accumulate, void* stream);

/* Mixup / CutMix on a device-resident batch (timm Mixup mode='batch' [3P], tools/engine.py:16-18):
 * in place, sample b with sample B-1-b.  cutmix = 0: x_b <- lam x_b + (1-lam) x_{B-1-b}; cutmix = 1: box [yl,yh) x [xl,xh) swapped in. */
int dkd_mixup(float* x, int32_t B, int32_t C, int32_t H, int32_t W, float lam, int32_t cutmix, int32_t yl, int32_t yh, int32_t xl,
              int32_t xh, void* stream);
/* The same mix written to dst, src left as it is (src == dst: in place; any other overlap is refused).  Same bytes moved as in place; a
 * batch that stays resident in HBM across steps can be mixed again without a copy per step. */
int dkd_mixup_to(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W, float lam, int32_t cutmix, int32_t yl, int32_t yh,
                 int32_t xl, int32_t xh, void* stream);
/* out f32 [B, C] = lam * smooth_onehot(labels[b]) + (1-lam) * smooth_onehot(labels[B-1-b])   (timm mixup_target). */
/* dkd_mixup_to that ALSO writes the mix as the bf16 patch matrix [B * (H/p) * (W/p), C * p * p] of dkd_im2col_patches (k = c p p + i p + j):
 * the patch-embedding GEMMs of student and teacher (timm PatchEmbed: Conv2d(3, D, p, p), model/models.py:195) read it, so the mixed batch
 * is read once and no gather pass runs over it.  p % 4 == 0, p | H, p | W. */
int dkd_mixup_to_patches(const float* src, float* dst, void* patches, int32_t p, int32_t B, int32_t C, int32_t H, int32_t W, float lam,
                         int32_t cutmix, int32_t yl, int32_t yh, int32_t xl, int32_t xh, void* stream);
int dkd_mixup_targets(const int64_t* labels, float* out, int32_t B, int32_t C, float lam, float smoothing, void* stream);
/* ema <- decay * ema + (1 - decay) * p over a flat parameter buffer (timm ModelEma [3P], tools/engine.py:68-69). */
int dkd_ema_update(float* ema, const float* p, int64_t n, float decay, void* stream);

/* top-k accuracy in percent (timm.utils.accuracy [3P], called at tools/engine.py:54-56): out f32 [nk] += 100 / B for every row of
 * z f32 [B, C] whose label's logit is beaten by fewer than ks[i] others (ties go to the lower index).  nk <= 4; caller zeroes out. */
int dkd_topk_correct(const float* z, const int64_t* labels, int32_t B, int32_t C, const int32_t* ks, int32_t nk, float* out, void* stream);

/* ---------------------------------------------------------------- losses (fused value + gradient) */
/* Base criterion + optional logit distillation in one pass (model/loss.py:35,57-67,241; timm SoftTargetCrossEntropy /
 * LabelSmoothingCrossEntropy [3P]).  z f32 [B, C] student logits; exactly one of soft_target f32 [B, C] / labels i64 [B].
 * kd_mode 0 none, 1 soft (KL, tau, /(B*C)), 2 hard (CE vs argmax teacher); z_kd/z_t f32 [B, C].
 * Outputs: losses f32 [5]: [0] = base, [1] = distill, [2] = w_base * base + w_kd * distill (what model/loss.py:241 returns),
 * [3] = w_base * base, [4] = w_kd * distill (sums accumulated atomically: caller zeroes all five first),
 * dz = w_base * dbase/dz, dz_kd = w_kd * ddistill/dz_kd   (f32 [B, C]). */
int dkd_logit_loss(const float* z, const float* soft_target, const int64_t* labels, float smoothing, int32_t kd_mode,
                   const float* z_kd, const float* z_t, float tau, float w_base, float w_kd, float* losses, float* dz,
                   float* dz_kd, int32_t B, int32_t C, void* stream);
/* loss[0] += w * sum(mask * (a - t)^2) / denom ; da = 2 w mask (a - t) / denom.  a: bf16|f32 [M, D] (student side, gets
 * the gradient), t: bf16|f32 [M, D] rows through tmap; mask f32 [M] or NULL.  (model/loss.py:326,449-451) */
int dkd_mse_loss(const void* a, int32_t a_is_f32, int32_t lda, const void* t, int32_t t_is_f32, int32_t ldt, DkdRowMap tmap,
                 const float* mask, float w_over_denom, float* loss, void* da, int32_t da_is_f32, int32_t ldda, int32_t M,
                 int32_t D, void* stream);
/* x~ = mask ? mask_token : x   (model/loss.py:433-440 collapsed); x, out bf16 [M, D], mask f32 [M]. */
int dkd_mask_select(const void* x, const float* mask_token, const float* mask, void* out, int32_t M, int32_t D, void* stream);
/* backward of mask_select: dx = (1-mask) * dout (bf16), dmask_token[d] += sum_{masked m} dout[m, d]. */
int dkd_mask_select_bwd(const void* dout, const float* mask, void* dx, float* dmask_token, int32_t M, int32_t D, void* stream);
/* WassKD-L1 (model/loss.py:187-199): per (b, d) sort over the P tokens of s (bf16|f32 [B, P, D]) and t ([B, P, D] rows
 * through tmap); loss[0] += w * sum|sort(s) - sort(t)| ; ds[b, pi(j), d] = w * sign(delta_j).  P <= 256. */
int dkd_sort_l1_loss(const void* s, int32_t s_is_f32, const void* t, int32_t t_is_f32, int32_t ldt, DkdRowMap tmap, float w,
                     float* loss, void* ds, int32_t ds_is_f32, int32_t B, int32_t P, int32_t D, void* stream);

/* MGD generation block (model/models.py:148-151): Conv3x3(pad 1) on the hw x hw token grid as gather + dkd_gemm_nt.
 * x bf16 [B, hw*hw, C] -> cols bf16 [B*hw*hw, 9*C] with k = (ky*3+kx)*C + c (conv weight permuted to [out, ky, kx, c]). */
int dkd_im2col3x3(const void* x, void* cols, int32_t B, int32_t hw, int32_t C, void* stream);
/* dx bf16 [B, hw*hw, C] = scatter-free transpose of im2col3x3 applied to dcols; relu_gate (bf16, optional): dx *= gate > 0. */
int dkd_col2im3x3(const void* dcols, const void* relu_gate, void* dx, int32_t B, int32_t hw, int32_t C, void* stream);
/* DiffKD (model/loss.py:138-149): loss += w_scalar[0] * w_over_denom * sum (s/|s| - t_hat)^2, ds (bf16) = gradient w.r.t. s.
 * s f32 [M, D], t_hat bf16 [M, D] (already normalised), w_scalar device f32 (mean of the noise-aware weights) or NULL. */
int dkd_normalize_mse(const float* s, const void* t_hat, const float* w_scalar, float w_over_denom, float* loss, void* ds,
                      int32_t ldds, int32_t M, int32_t D, void* stream);
/* t_hat = t/|t| (bf16), nz = noise * sigma[b] (f32), x_in = t_hat + nz + t_emb[b] (bf16); t bf16 rows through tmap. */
int dkd_diffkd_prepare(const void* t, int32_t ldt, DkdRowMap tmap, const float* noise, const float* sigma, const float* temb,
                       int32_t rows_per_sample, void* t_hat, float* nz, void* x_in, int32_t M, int32_t D, void* stream);
/* loss += w * sum (a * keep * keep_scale - t)^2, da (bf16) its gradient: Dropout(0.1) folded into the noise-prediction MSE. */
int dkd_dropout_mse(const float* a, const float* t, const float* keep, float keep_scale, float w_over_denom, float* loss, void* da,
                    int64_t n, void* stream);

/* Token scores of the saliency_mgd branch (model/misc.py:38-165; scorers model/models.py:14-56): head-averaged softmax attention
 * weights of an auxiliary attention, fp32 throughout (they are only ranked).  q, k f32 row-major projections (row strides ldq / ldk
 * floats, heads side by side: head h = columns [h * head_dim, (h + 1) * head_dim)); sample b's rows start at b * rows_per_sample, its
 * first query / key row is q_first / k_first.  scores f32 [B, L].
 *   diagonal = 1 (method 1): L queries x L keys,  scores[b, i] = mean_h softmax_j(q_i . k_j / sqrt(head_dim))[j = i]
 *   diagonal = 0 (methods 2, 3): ONE query (row q_first) x L keys (+ the key at row extra_key_row of the sample when >= 0, which takes part
 *                in the softmax but gets no score: the CLS token of method 2),  scores[b, j] = mean_h softmax(...)[j]
 * L <= 255, head_dim in {16, 32, 48, 64, 96, 128}. */
int dkd_saliency_scores(const float* q, const float* k, float* scores, int32_t B, int32_t L, int32_t H, int32_t head_dim, int32_t ldq,
                        int32_t ldk, int64_t q_rows_per_sample, int64_t k_rows_per_sample, int32_t q_first, int32_t k_first,
                        int32_t diagonal, int32_t extra_key_row, void* stream);

/* Launch probe for bench.py: between begin and end every dkd_gemm_nt launch is bracketed by HIP events recorded on ITS stream.
 * end() synchronises them and returns per kernel symbol (0 = gemm_nt_kernel<128>, 1 = gemm_nt_kernel<64>, 2 = gemm_nt256_kernel)
 * the algorithmic FLOPs (2 M N K), the summed durations (ms) and the launch counts (arrays of 3).  Off by default. */
int dkd_probe_begin(void);
int dkd_probe_end(double* flops, double* ms, int32_t* launches);
/* The same with more families (arrays of n <= 6; bytes may be NULL): 3 = one student block's backward (every launch of one
 * dkd_block_bwd call under one event pair; FLOPs = 2x the block's forward, bytes = each tensor the pass must touch once),
 * 4 = the fused loss kernels (bytes: read a, read t, write da), 5 = one student block's training forward.  Families 3 and 5 CONTAIN
 * the NT-GEMM launches that families 0-2 time individually. */
int dkd_probe_end_ex(int32_t n, double* flops, double* bytes, double* ms, int32_t* launches);

/* ---------------------------------------------------------------- transformer block drivers (host-side launch sequences) */
/* One pre-LN ViT block ([3P] timm Block: x = x + dp(attn(ln1 x)); x = x + dp(mlp(ln2 x))) as ONE call: the library issues the
 * 7 (forward; 5 with fuse_mlp) / 8-15 (backward) kernel launches itself, so the Python host pays one FFI crossing per block instead of one per
 * kernel (the step was host-bound at ~570 launches).  All buffers are caller-owned; M = B*N rows.
 * Inference: pass mean/rstd/lse/pre = NULL and x1 = x2 = x (in-place residual stream). */
typedef struct {
  int32_t B, N, D, H, hidden;
  float eps;
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *qkv_b, *proj_b, *fc1_b, *fc2_b;   /* f32 parameters                      */
  const void *qkv_w, *proj_w, *fc1_w, *fc2_w;                                     /* bf16 [out, in] shadows              */
  const void *qkv_wt, *proj_wt, *fc1_wt, *fc2_wt;                                 /* bf16 [in, out] shadows (backward)   */
  const float *s1, *s2;              /* DropPath keep/keep_prob per sample (f32 [B]) or NULL                             */
  float *x, *x1, *x2;                /* f32 [M, D]: block input, after the attention branch, block output               */
  void *y1, *qkv, *o, *y2, *pre, *h; /* bf16: LN1 out [M,D], qkv [M,3D], attention out [M,D], LN2 out, fc1 pre-act, GELU out [M,hidden] */
  void* tap;                         /* bf16 [M, D] feature tap (fc2 output before DropPath/residual) or NULL            */
  float *mean1, *rstd1, *mean2, *rstd2, *lse;   /* saved statistics (f32 [M] x4, [B,H,N]) or NULL                        */
  int32_t fuse_mlp;                  /* 1: the MLP branch runs on dkd_mlp192_fwd / dkd_mlp192_bwd (D = 192, hidden % 64 == 0, fc2_wt set also in
                                        the forward; y2 / pre / h and the backward's dF / dH sized for M rounded up to 16 rows -- the workspace
                                        queries below already are; `pre` is then in that kernel pair's private order).  The caller sets it once:
                                        the same descriptor goes to the forward and to the backward.                       */
  int32_t ln_fold;                   /* inference only (pre == NULL); bits: 1 = norm1 is folded into the qkv GEMM (qkv_w = bf16(ln1_w * W), qkv_b =
                                        W ln1_b + b, qkv_c = row sums of that qkv_w; the block input's bf16 copy is in `xb`, its row sums in stats1
                                        -- left there by the previous block); 2 = norm2 likewise (fc1_w, fc1_b, fc1_c; proj writes xb / stats2);
                                        4 = fc2 writes xb and stats_next for the next block's bit 1.  See DkdGemm.xb.  The shapes must take the
                                        wide-kernel path (dkd_gemm_nt refuses otherwise); stats buffers f32 [M][2], zeroed by the caller.   */
  const float *qkv_c, *fc1_c;
  float *stats1, *stats2, *stats_next;
  void* xb;                          /* bf16 [M, D]                                                                        */
  int32_t ln1_ready;                 /* 1: y1 / mean1 / rstd1 already hold norm1(x) -- written by the previous block's fused MLP kernel (its
                                        next_* outputs): the forward skips the LayerNorm launch                              */
  const float *next_ln1_w, *next_ln1_b;   /* with fuse_mlp, optional: the next block's norm1 parameters and buffers (see dkd_mlp192_fwd) */
  void* next_y1;
  float *next_mean1, *next_rstd1;
  int32_t fuse_attn;                 /* 1: qkv projection + attention run on dkd_attn192_fwd (D = 192, H = 3, N <= 208; not with ln_fold) */
} DkdBlock;

/* A deferred LayerNorm parameter-gradient reduction: dgamma[c] += sum_b part[b][c], dbeta[c] += sum_b part[b][D + c] over nblk
 * partial rows of 2 * D floats (what the LayerNorm-backward kernels leave in their scratch). */
typedef struct DkdLnReduce {
  const float* part;
  int32_t nblk, D;
  float* dgamma;
  float* dbeta;
} DkdLnReduce;
/* Up to 12 of them in one launch. */
int dkd_ln_bwd_reduce_group(const DkdLnReduce* items, int32_t n, void* stream);

typedef struct {
  float* g;                          /* in: d loss / d x2 (f32 [M, D]); out: d loss / d x (in place)                     */
  const void* gtap;                  /* bf16 [M, D] gradient of the tap or NULL                                          */
  float *d_ln1_w, *d_ln1_b, *d_ln2_w, *d_ln2_b, *d_qkv_w, *d_qkv_b, *d_proj_w, *d_proj_b, *d_fc1_w, *d_fc1_b, *d_fc2_w, *d_fc2_b;
  void *dF, *dH, *dqkv;              /* bf16 workspaces: [M, D] (also reused for dY2, dA, dO, dY1), [M, hidden], [M, 3D]  */
  void* dT;                          /* bf16 workspace [M, D]                                                            */
  float* ln_ws;                      /* f32 scratch, 2 * D * ceil(M / 64) floats, for the LayerNorm backward partial sums; or NULL */
  void* dF2;                         /* second bf16 [M, D] workspace or NULL: with it all four weight gradients are one launch   */
  float* ln_ws2;                     /* second LayerNorm scratch (same size as ln_ws), needed with ln_defer                       */
  DkdLnReduce* ln_defer;             /* array of 2 or NULL.  Non-NULL: the two dgamma / dbeta reductions of the block's LayerNorms are
                                        NOT launched; dkd_block_bwd describes them here ([0] norm2 over ln_ws, [1] norm1 over ln_ws2)
                                        and the caller runs dkd_ln_bwd_reduce_group over the blocks it has collected (the student
                                        flushes them with its deferred weight gradients: 2 launches per step instead of 24)       */
  int32_t defer_wgrad;               /* 1 (needs dF2): do NOT launch the four weight gradients; the caller issues them itself
                                        (dkd_block_wgrad_group over dF/h, dH/y2, dF2/o, dqkv/y1, for one block or for several
                                        whose workspaces it keeps alive until then) */
} DkdBlockGrads;

int dkd_blocks_fwd(const DkdBlock* blocks, int32_t n_blocks, void* stream);
int dkd_block_bwd(const DkdBlock* blk, const DkdBlockGrads* gr, void* stream);

/* Workspace queries (bytes; every sub-buffer starts on a 256-byte boundary).
 * dkd_layernorm_bwd_workspace_bytes: the `ws` scratch of dkd_layernorm_bwd.
 * dkd_block_fwd_workspace_bytes: the activation slabs of one block of dkd_blocks_fwd -- *bf16_bytes: y1 | qkv | o | y2 | pre | h [| tap]
 *   (training) or y | qkv | o | h (inference, shared by all blocks); *f32_bytes: x1 | x2 | mean1 | rstd1 | mean2 | rstd2 | lse (training)
 *   or 0.  Returns their sum.
 * dkd_block_bwd_workspace_bytes: the scratch of dkd_block_bwd (dF | dT | dH | dqkv | ln_ws | dF2), shared by all blocks;
 * dkd_block_bwd_workspace_carve fills those six pointers of `gr` from a buffer of that size. */
int64_t dkd_layernorm_bwd_workspace_bytes(int32_t M, int32_t D);
int64_t dkd_block_fwd_workspace_bytes(int32_t B, int32_t N, int32_t D, int32_t H, int32_t hidden, int32_t training, int32_t with_tap,
                                      int64_t* bf16_bytes, int64_t* f32_bytes);
int64_t dkd_block_bwd_workspace_bytes(int32_t B, int32_t N, int32_t D, int32_t hidden);
int dkd_block_bwd_workspace_carve(void* ws, int32_t B, int32_t N, int32_t D, int32_t hidden, DkdBlockGrads* gr);

/* ---------------------------------------------------------------- small glue of the step (one launch each instead of ATen chains) */
/* x bf16 [n] *= *scalar_dev (f32 product).  The upstream gradient of a loss term is a 0-dim DEVICE tensor in autograd's backward
 * (torch: grad_output * local gradient, model/loss.py:326-329 under loss.backward()); n % 8 == 0. */
int dkd_scale_bf16(void* x, const float* scalar_dev, int64_t n, void* stream);
/* dst bf16 [rows, Cp] = bf16(src f32 [rows, C] * (*scalar_dev, or 1 when null)), columns C .. Cp zero: the K-padded operand of the
 * classifier head's dgrad / wgrad GEMMs (timm VisionTransformer.head, nn.Linear backward). */
int dkd_cast_pad_bf16(const float* src, int32_t ld_src, const float* scalar_dev, void* dst, int32_t rows, int32_t C, int32_t Cp, void* stream);
/* DropPath keep masks of a whole step: out f32 [n, B] = (u < keep_prob[i]) / keep_prob[i], u from a counter-based generator seeded
 * by `seed` (timm.layers.DropPath: x.new_empty(shape).bernoulli_(keep_prob) / keep_prob, one draw per sample and branch). */
int dkd_droppath_scales(float* out, const float* keep_prob, int32_t n, int32_t B, uint64_t seed, void* stream);

/* ---------------------------------------------------------------- LRKD target: truncated SVD without factorising T (model/loss.py:318-324) */
/* Batched cyclic Jacobi: A f32 [batch, n, n] symmetric, n <= 128 -> evals [batch, n] (unsorted), evecs [batch, n, n]
 * (column j pairs with evals[j]).  One workgroup per matrix, LDS-resident; at most `sweeps` sweeps, stops after the first sweep that
 * rotates nothing (10 converge fp32). */
int dkd_jacobi_eigh(const float* A, float* evals, float* evecs, int32_t batch, int32_t n, int32_t sweeps, void* stream);

/* One step of block subspace iteration (block of 96 vectors) on L symmetric PSD matrices G f32 [L, Dt, Dt] of which only the
 * 128 x 128 tiles on and above the diagonal need to be valid (what dkd_gram writes), all layers in the same launches (csrc/lowrank.hip).
 * V f32 [L, Dt, 96] is updated in place:
 *   mode 0: V <- orth(G V)        a power step (cold start); orthonormalised through the eigen-decomposition of (G V)^T (G V)
 *   mode 1: the tracking step run once per batch, V orthonormal on entry (the previous batch's result):
 *           V <- orth(G V W), W = eigenvectors of V^T G V by at most `ritz_sweeps` Jacobi sweeps (0: keep the basis as it is),
 *           columns by descending Ritz value, orthonormalised in that order (Cholesky)
 *   mode 2: V <- orth(V)
 *   mode 3: V <- V W, Ritz vectors of G in span(V) run to convergence, V orthonormal on entry (ends a cold start)
 * Optional outputs of modes 1 and 3: evals f32 [L, 96] (Ritz values of V^T G V = squared singular values of T, descending), and
 * v_hi / v_lo bf16 [L, rank, Dt]: V[:, :rank]^T split as hi = bf16(v), lo = bf16(v - hi) -- the B operands of the projection GEMMs
 * U_k S_k = T V_k.   ws: dkd_lowrank_workspace_bytes(L, Dt) bytes of scratch, 256-byte aligned.  Dt % 32 == 0, Dt >= 128. */
int64_t dkd_lowrank_workspace_bytes(int32_t L, int32_t Dt);
int dkd_lowrank_step(const float* G, float* V, int32_t L, int32_t Dt, int32_t mode, int32_t ritz_sweeps, int32_t rank, void* v_hi,
                     void* v_lo, float* evals, void* ws, void* stream);

/* The per-batch solve of the timed path (round 5): `n_mult` power steps of block subspace iteration from the basis V f32 [L, Dt, 96]
 * (orthonormal on entry: the previous batch's result, or a cold start's), ending in a Rayleigh-Ritz step whose Jacobi diagonalisation
 * runs until nothing is left to rotate (at most `ritz_sweeps` sweeps) -- n_mult = 8 reproduces torch.linalg.svd's U_k S_k of
 * model/loss.py:318-326 to fp32 accuracy on changing batches (tests/test_fullsize_gpu.py), n_mult = 1 is round 4's one-step tracker.
 * A chain of short launches: 16-row panels of Y' = (G Y) C with the previous stage's orthonormalising transform applied to the finished
 * tile and the 96 x 96 Gram matrices accumulated atomically; one workgroup per layer for the Cholesky / Jacobi stages between them
 * (multiplies per stage 1, 2, 2, ..., 1).  Outputs as dkd_lowrank_step modes 1 / 3: V in place, evals f32 [L, 96] (may be null),
 * v_hi / v_lo bf16 [L, rank, Dt] (may be null).  ws: dkd_lowrank_chain_workspace_bytes(L, Dt) bytes, 256-byte aligned, whose first
 * dkd_lowrank_chain_zero_bytes(L, Dt) bytes must be ZERO when the first call is made (every call leaves them zero again).
 * Dt % 64 == 0, 128 <= Dt <= 2048 (other widths: dkd_lowrank_step).
 * Replaces: torch.linalg.svd at model/loss.py:321. */
int64_t dkd_lowrank_chain_workspace_bytes(int32_t L, int32_t Dt);
int64_t dkd_lowrank_chain_zero_bytes(int32_t L, int32_t Dt);
int dkd_lowrank_chain(const float* G, float* V, int32_t L, int32_t Dt, int32_t n_mult, int32_t ritz_sweeps, int32_t rank, void* v_hi,
                      void* v_lo, float* evals, void* ws, void* stream);

/* ---------------------------------------------------------------- optimizer ([3P] torch.optim.AdamW via timm create_optimizer) */
/* One launch over a flat parameter segment; optionally refreshes the bf16 shadow copy used by the GEMMs. */
int dkd_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int32_t step, float grad_scale, void* stream);
/* The same for a segment of parameters that may receive no gradient in a step (curkd's align stages outside their epochs,
 * model/loss.py:362-420): torch.optim.AdamW skips ``p.grad is None`` entirely.  state f32 [2] on the device: state[0] = largest |g| of
 * the unit the segment belongs to this step (0 = nothing wrote a gradient: the launch does nothing), state[1] = that unit's own step
 * count, already advanced for this step (bias corrections are taken from it). */
int dkd_adamw_step_gated(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, const float* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif

// Host-side launch sequences for one transformer block (see include/dkd.h): the library, not Python, walks the kernels.
#include <stdlib.h>
#include "common.h"

namespace {
const DkdRowMap ID = {0, 0, 0};

DkdGemm mk(const void* A, const void* B, void* C, int M, int N, int K) {
  DkdGemm g = {};
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K;
  g.lda = K; g.ldb = K; g.ldc = N;
  return g;
}
#define TRY(call)            \
  do {                       \
    int rc_ = (call);        \
    if (rc_ != DKD_OK) return rc_; \
  } while (0)

int block_fwd(const DkdBlock& b, void* st) {
  const int M = b.B * b.N, D = b.D, Hd = b.hidden;
  const int fold = b.pre ? 0 : b.ln_fold;          // LayerNorm folded into the GEMMs around it (inference; include/dkd.h, DkdGemm.xb)
  DkdGemm g;
  if (fold & 1) {
    g = mk(b.xb, b.qkv_w, b.qkv, M, 3 * D, D);
    g.ln_stats = b.stats1; g.ln_c = b.qkv_c; g.ln_eps = b.eps;
  } else {
    if (!b.ln1_ready) TRY(dkd_layernorm_fwd(b.x, D, ID, b.ln1_w, b.ln1_b, b.y1, b.mean1, b.rstd1, M, D, b.eps, 0, st));
    g = mk(b.y1, b.qkv_w, b.qkv, M, 3 * D, D);
  }
  bool proj_done = false;
  if (b.fuse_attn && !(fold & 1)) {       // qkv projection + attention in one launch: q, k, v reach the attention through registers / LDS
    // ... and, unless DKD_NO_ATTN_FWD_PROJ is set (A/B), proj + DropPath + residual behind them in the same launch (round 4)
    proj_done = !(fold & 2) && getenv("DKD_NO_ATTN_FWD_PROJ") == nullptr;
    if (proj_done) TRY(dkd_attn192_fwd_proj(b.y1, b.qkv_w, b.qkv_b, b.qkv, b.o, b.lse, b.proj_w, b.proj_b, b.x, b.s1, b.x1, b.B, b.N, st));
    else TRY(dkd_attn192_fwd(b.y1, b.qkv_w, b.qkv_b, b.qkv, b.o, b.lse, b.B, b.N, st));
  } else {
    g.epi = DKD_EPI_BIAS; g.bias = b.qkv_b;
    TRY(dkd_gemm_nt(&g, st));
    TRY(dkd_attn_fwd(b.qkv, b.o, b.lse, b.B, b.N, b.H, st));
  }
  if (!proj_done) {
    g = mk(b.o, b.proj_w, b.x1, M, D, D);
    g.epi = DKD_EPI_BIAS | DKD_EPI_RESID | DKD_EPI_OUT_F32; g.bias = b.proj_b;
    g.resid = b.x; g.ldr = D; g.rowscale = b.s1; g.rows_per_sample = b.N;
    if (fold & 2) {
      g.xb = b.xb; g.ldxb = D; g.rowstats = b.stats2;
    }
    TRY(dkd_gemm_nt(&g, st));
  }
  if (b.fuse_mlp) {                       // LN2 + fc1 + GELU + fc2 + tap + DropPath + residual: one kernel, h stays in registers
    const bool save = b.pre != nullptr;
    return dkd_mlp192_fwd(b.x1, b.ln2_w, b.ln2_b, b.eps, b.fc1_w, b.fc1_b, b.fc2_wt, b.fc2_b, b.s2, b.N, b.x2, b.tap, save ? b.y2 : nullptr,
                          b.pre, save ? b.h : nullptr, b.mean2, b.rstd2, b.next_ln1_w, b.next_ln1_b, b.next_y1, b.next_mean1, b.next_rstd1, M,
                          Hd, st);
  }
  if (fold & 2) {
    g = mk(b.xb, b.fc1_w, b.h, M, Hd, D);
    g.ln_stats = b.stats2; g.ln_c = b.fc1_c; g.ln_eps = b.eps;
  } else {
    TRY(dkd_layernorm_fwd(b.x1, D, ID, b.ln2_w, b.ln2_b, b.y2, b.mean2, b.rstd2, M, D, b.eps, 0, st));
    g = mk(b.y2, b.fc1_w, b.h, M, Hd, D);
  }
  g.epi = DKD_EPI_BIAS | DKD_EPI_GELU; g.bias = b.fc1_b; g.preact = b.pre; g.ldp = Hd;
  TRY(dkd_gemm_nt(&g, st));
  g = mk(b.h, b.fc2_w, b.x2, M, D, Hd);
  g.epi = DKD_EPI_BIAS | DKD_EPI_RESID | DKD_EPI_OUT_F32; g.bias = b.fc2_b;
  g.resid = b.x1; g.ldr = D; g.rowscale = b.s2; g.rows_per_sample = b.N;
  g.tap = b.tap; g.ldt = D;
  if (fold & 4) {
    g.xb = b.xb; g.ldxb = D; g.rowstats = b.stats_next;
  }
  TRY(dkd_gemm_nt(&g, st));
  return DKD_OK;
}
}  // namespace

extern "C" int dkd_blocks_fwd(const DkdBlock* blocks, int32_t n_blocks, void* stream) {
  DKD_CHECK_ARG(blocks && n_blocks > 0, "blocks_fwd: no blocks");
  for (int i = 0; i < n_blocks; ++i) {
    const DkdBlock& b = blocks[i];
    DKD_CHECK_ARG(b.x && b.x1 && b.x2 && b.y1 && b.qkv && b.o && b.y2 && b.h, "blocks_fwd: block %d has a null buffer", i);
    DKD_CHECK_ARG(!b.fuse_attn || (b.D == 192 && b.H == 3 && b.N <= 208), "blocks_fwd: block %d: fuse_attn needs D = 192, H = 3, N <= 208", i);
    DKD_CHECK_ARG(!b.fuse_mlp || (b.D == 192 && b.hidden % 64 == 0 && b.fc2_wt), "blocks_fwd: block %d: fuse_mlp needs D = 192, hidden %% 64 == 0 and fc2_wt", i);
    DKD_CHECK_ARG(!b.ln_fold || (!b.pre && !b.fuse_mlp && !b.s1 && !b.s2 && b.xb && (!(b.ln_fold & 1) || (b.stats1 && b.qkv_c)) &&
                                 (!(b.ln_fold & 2) || (b.stats2 && b.fc1_c)) && (!(b.ln_fold & 4) || b.stats_next)),
                  "blocks_fwd: block %d: ln_fold is for inference blocks (no saves, no DropPath) and needs xb, the statistics buffers and the row sums", i);
    if (b.pre) {        // training forward (activations saved): probed as a whole for bench.py's student roofline
      const double M = (double)b.B * b.N, D = b.D, Hd = b.hidden;
      const double flops = 2.0 * M * (4.0 * D * D + 2.0 * D * Hd) + 4.0 * (double)b.B * b.N * b.N * D;
      // x in, x1 in+out, x2 out (f32); y1, qkv, o, y2, pre, h written (+ qkv, o, y1, y2, h read back by the next kernel of the chain)
      // (with fuse_mlp y2 / pre / h are written once and not read back, x1 is read once)
      const double bytes = b.fuse_mlp ? M * (D * 4.0 * 4 + D * 2.0 * (1 + 3 + 1) * 2 + D * 2.0 + Hd * 2.0 * 2 + (b.tap ? D * 2.0 : 0.0))
                                      : M * (D * 4.0 * 4 + (D * 2.0 * (1 + 3 + 1 + 1) + Hd * 2.0 * 2) * 2 + (b.tap ? D * 2.0 : 0.0));
      DkdProbeScope probe(5, flops, bytes, as_stream(stream));
      TRY(block_fwd(b, stream));
    } else {
      TRY(block_fwd(b, stream));
    }
  }
  return DKD_OK;
}

extern "C" int dkd_block_bwd(const DkdBlock* bp, const DkdBlockGrads* gp, void* st) {
  DKD_CHECK_ARG(bp && gp, "block_bwd: null descriptor");
  const DkdBlock& b = *bp;
  const DkdBlockGrads& r = *gp;
  DKD_CHECK_ARG(r.g && r.dF && r.dH && r.dqkv && r.dT && b.pre && b.mean1 && b.lse, "block_bwd: missing buffer (was the forward run with saves?)");
  const int M = b.B * b.N, D = b.D, Hd = b.hidden;
  // algorithmic work of one block's backward (bench.py, roofline_student): dgrad + wgrad = 2x the forward GEMM FLOPs, attention
  // backward 2x its forward; bytes = every tensor the pass must touch ONCE: g in/out, x, x1 (f32), the saved bf16 activations y1, qkv, o,
  // y2, pre, h, and the tap gradient -- 30 D + 4 hidden (+ 2 D) bytes per token row
  const double Md = (double)M;
  const double bwd_flops = 2.0 * (2.0 * Md * (4.0 * D * D + 2.0 * (double)D * Hd) + 4.0 * (double)b.B * b.N * b.N * D);
  const double bwd_bytes = Md * (30.0 * D + 4.0 * Hd + (r.gtap ? 2.0 * D : 0.0));
  DkdProbeScope probe(3, bwd_flops, bwd_bytes, as_stream(st));
  // the LayerNorm parameter-gradient reductions: launched here, or described to the caller (ln_defer), who runs several blocks' together
  DKD_CHECK_ARG(!r.ln_defer || (r.ln_ws && r.ln_ws2), "block_bwd: ln_defer needs ln_ws and ln_ws2");
  struct LnCapture {
    bool on;
    explicit LnCapture(DkdLnReduce* items) : on(items != nullptr) { if (on) dkd_ln_capture_begin(items, 2); }
    ~LnCapture() { if (on) dkd_ln_capture_end(); }
  } capture(r.ln_defer);
  if (r.ln_defer) r.ln_defer[0].part = r.ln_defer[1].part = nullptr;
  float* ln_ws1 = r.ln_defer ? r.ln_ws2 : r.ln_ws;          // scratch of norm1's backward (norm2's partial rows must survive it when deferred)
  // ---- MLP branch
  const bool all4 = r.dF2 != nullptr;   // a second [M, D] buffer keeps the MLP branch's dF alive: all four weight gradients go out together
  DkdGemm g = {};
  if (b.fuse_mlp) {
    // scale-cast + dGELU dgrad + fc1 dgrad + LayerNorm backward + the scale-cast that opens the attention branch: one kernel
    DKD_CHECK_ARG(D == 192 && Hd % 64 == 0 && r.ln_ws && all4, "block_bwd: fuse_mlp needs D = 192, hidden %% 64 == 0, ln_ws and dF2");
    TRY(dkd_mlp192_bwd(r.g, r.gtap, b.s2, b.s1, b.N, b.pre, b.fc2_wt, b.fc1_w, b.x1, b.ln2_w, b.mean2, b.rstd2, r.dF, r.dH, r.dF2, r.d_ln2_w,
                       r.d_ln2_b, r.ln_ws, M, Hd, st));
  } else {
  TRY(dkd_scale_cast_bf16(r.g, D, ID, b.s2, b.N, r.gtap, 0, D, r.dF, D, M, D, st));
  g = mk(r.dF, b.fc2_wt, r.dH, M, Hd, D);
  g.epi = DKD_EPI_DGELU; g.preact = b.pre; g.ldp = Hd;
  TRY(dkd_gemm_nt(&g, st));
  }
  if (!all4) {                          // both MLP weight gradients in one launch (dF is not overwritten before the attention branch)
    const DkdTnProblem w[2] = {{r.dF, b.h, r.d_fc2_w, r.d_fc2_b, M, D, Hd, D, Hd, Hd, ID, ID},
                               {r.dH, b.y2, r.d_fc1_w, r.d_fc1_b, M, Hd, D, Hd, D, D, ID, ID}};
    TRY(dkd_gemm_tn_group(w, 2, st));
  }
  // D = 192 (the DeiT-tiny student): the dgrad GEMMs that end a branch carry the LayerNorm backward (and the scale-cast that opens
  // the next branch) as their epilogue -- dT never goes to memory
  const bool fuse_ln = D == 192 && Hd % 64 == 0 && r.ln_ws != nullptr && getenv("DKD_NO_LNBWD_FUSION") == nullptr;
  void* dFa = all4 ? r.dF2 : r.dF;      // gradient w.r.t. the attention branch's output (bf16)
  if (b.fuse_mlp) {
    // (done above)
  } else if (fuse_ln) {
    TRY(dkd_gemm_nt_lnbwd(r.dH, b.fc1_wt, M, Hd, Hd, Hd, b.x1, D, b.ln2_w, b.mean2, b.rstd2, r.g, D, r.d_ln2_w, r.d_ln2_b, r.ln_ws, dFa, b.s1,
                          b.N, st));
  } else {
    g = mk(r.dH, b.fc1_wt, r.dT, M, D, Hd);
    TRY(dkd_gemm_nt(&g, st));
    TRY(dkd_layernorm_bwd(r.dT, 0, b.x1, D, ID, b.ln2_w, b.mean2, b.rstd2, r.g, D, ID, 1, r.d_ln2_w, r.d_ln2_b, M, D, r.ln_ws, st));
    // ---- attention branch
    TRY(dkd_scale_cast_bf16(r.g, D, ID, b.s1, b.N, nullptr, 1, 0, dFa, D, M, D, st));
  }
  bool branch_done = false;             // the fused kernel has also run the qkv dgrad + norm1's backward
  if (b.fuse_attn && D == 192 && b.H == 3 && b.N >= 8 && b.N <= 208 && getenv("DKD_NO_ATTN_BWD_FUSION") == nullptr) {
    // proj dgrad + attention backward in one launch (dO is computed head by head into the LDS image the attention backward reads).  The
    // same launch can go on to the qkv dgrad + norm1's backward (DKD_ATTN_BWD_LN=1), but measured at batch 256 (profiles/r04_attn192_bwd_*)
    // that tail costs 52 us per block inside the per-sample kernel (every CU moves its 453 KB of f32 rows at its own ~25 GB/s, after
    // the GEMM, with nothing to overlap) against 36.5 us for dkd_gemm_nt_lnbwd on its own, and the step time is the same either way:
    // the default keeps the separate launch.
    branch_done = fuse_ln && getenv("DKD_ATTN_BWD_LN") != nullptr;
    if (branch_done)
      TRY(dkd_attn192_bwd(dFa, b.proj_wt, b.qkv, b.o, b.lse, r.dqkv, b.qkv_wt, b.x, b.ln1_w, b.mean1, b.rstd1, r.g, r.d_ln1_w, r.d_ln1_b,
                          ln_ws1, b.B, b.N, st));
    else
      TRY(dkd_attn192_bwd(dFa, b.proj_wt, b.qkv, b.o, b.lse, r.dqkv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                          nullptr, b.B, b.N, st));
  } else {
    g = mk(dFa, b.proj_wt, r.dT, M, D, D);
    TRY(dkd_gemm_nt(&g, st));
    TRY(dkd_attn_bwd(b.qkv, b.o, r.dT, b.lse, r.dqkv, b.B, b.N, b.H, st));
  }
  if (all4 && r.defer_wgrad) {
    // the caller launches dkd_gemm_tn_group({dF, h}, {dH, y2}, {dF2, o}, {dqkv, y1}) itself
  } else if (all4) {
    const DkdTnProblem w[4] = {{r.dF, b.h, r.d_fc2_w, r.d_fc2_b, M, D, Hd, D, Hd, Hd, ID, ID},
                               {r.dH, b.y2, r.d_fc1_w, r.d_fc1_b, M, Hd, D, Hd, D, D, ID, ID},
                               {dFa, b.o, r.d_proj_w, r.d_proj_b, M, D, D, D, D, D, ID, ID},
                               {r.dqkv, b.y1, r.d_qkv_w, r.d_qkv_b, M, 3 * D, D, 3 * D, D, D, ID, ID}};
    TRY(dkd_gemm_tn_group(w, 4, st));
  } else {                              // proj and qkv weight gradients in one launch
    const DkdTnProblem w[2] = {{dFa, b.o, r.d_proj_w, r.d_proj_b, M, D, D, D, D, D, ID, ID},
                               {r.dqkv, b.y1, r.d_qkv_w, r.d_qkv_b, M, 3 * D, D, 3 * D, D, D, ID, ID}};
    TRY(dkd_gemm_tn_group(w, 2, st));
  }
  if (branch_done) return DKD_OK;
  if (fuse_ln) {
    TRY(dkd_gemm_nt_lnbwd(r.dqkv, b.qkv_wt, M, 3 * D, 3 * D, 3 * D, b.x, D, b.ln1_w, b.mean1, b.rstd1, r.g, D, r.d_ln1_w, r.d_ln1_b, ln_ws1,
                          nullptr, nullptr, 0, st));
    return DKD_OK;
  }
  g = mk(r.dqkv, b.qkv_wt, r.dT, M, D, 3 * D);
  TRY(dkd_gemm_nt(&g, st));
  TRY(dkd_layernorm_bwd(r.dT, 0, b.x, D, ID, b.ln1_w, b.mean1, b.rstd1, r.g, D, ID, 1, r.d_ln1_w, r.d_ln1_b, M, D, ln_ws1, st));
  return DKD_OK;
}

// ---------------------------------------------------------------------------------------------- workspace queries
namespace {
inline int64_t al256(int64_t bytes) { return (bytes + 255) / 256 * 256; }
}

extern "C" int64_t dkd_layernorm_bwd_workspace_bytes(int32_t M, int32_t D) {
  return al256((int64_t)2 * D * (((M + 63) / 64) > 320 ? ((M + 63) / 64) : 320) * 4);   // (>= 320 partial rows: the per-sample kernels leave one per workgroup = per CU)
}

extern "C" int64_t dkd_block_fwd_workspace_bytes(int32_t B, int32_t N, int32_t D, int32_t H, int32_t hidden, int32_t training,
                                                 int32_t with_tap, int64_t* bf16_bytes, int64_t* f32_bytes) {
  const int64_t M = ((int64_t)B * N + 15) / 16 * 16;      // rows in whole groups of 16: what the fused MLP kernels store (DkdBlock.fuse_mlp)
  int64_t b16, f32 = 0;
  if (training) {
    b16 = al256(M * D * 2) * 3 + al256(M * 3 * D * 2) + al256(M * hidden * 2) * 2 + (with_tap ? al256(M * D * 2) : 0);
    f32 = al256(M * D * 4) * 2 + al256(M * 4) * 4 + al256((int64_t)B * H * N * 4);
  } else {
    b16 = al256(M * D * 2) * 2 + al256(M * 3 * D * 2) + al256(M * hidden * 2);
  }
  if (bf16_bytes) *bf16_bytes = b16;
  if (f32_bytes) *f32_bytes = f32;
  return b16 + f32;
}

extern "C" int64_t dkd_block_bwd_workspace_bytes(int32_t B, int32_t N, int32_t D, int32_t hidden) {
  const int64_t M = ((int64_t)B * N + 15) / 16 * 16;
  return al256(M * D * 2) * 3 + al256(M * hidden * 2) + al256(M * 3 * D * 2) + 2 * dkd_layernorm_bwd_workspace_bytes((int32_t)M, D);
}

extern "C" int dkd_block_bwd_workspace_carve(void* ws, int32_t B, int32_t N, int32_t D, int32_t hidden, DkdBlockGrads* gr) {
  DKD_CHECK_ARG(ws && gr && B > 0 && N > 0 && D > 0 && hidden > 0, "block_bwd_workspace_carve: bad arguments");
  DKD_CHECK_ARG(((uintptr_t)ws & 255) == 0, "block_bwd_workspace_carve: the workspace must be 256-byte aligned");
  const int64_t M = ((int64_t)B * N + 15) / 16 * 16;
  char* p = (char*)ws;
  gr->dF = p;      p += al256(M * D * 2);
  gr->dT = p;      p += al256(M * D * 2);
  gr->dH = p;      p += al256(M * hidden * 2);
  gr->dqkv = p;    p += al256(M * 3 * D * 2);
  gr->ln_ws = (float*)p;  p += dkd_layernorm_bwd_workspace_bytes((int32_t)M, D);
  gr->dF2 = p;     p += al256(M * D * 2);
  gr->ln_ws2 = (float*)p;
  return DKD_OK;
}

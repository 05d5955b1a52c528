// Backward of the attention branch of a D = 192, 3-head transformer block (the DeiT-tiny student) up to the qkv gradient, as ONE kernel:
// [3P] autograd of timm Attention.forward -- dO = dY Wproj;  (dq, dk, dv) = SDPA'(q, k, v, dO) per head -- reached from
// tools/engine.py:61-62 of the reference (loss_scaler -> backward) through the timm Block behind model/models.py:185-195.
// Unfused this was two launches with a round trip between them: the proj dgrad GEMM wrote dO [M, 192] to HBM (19 MB at batch 256) and the
// attention backward read it back together with q, k, v and O.  Here dO of a head is computed INTO the LDS image the attention backward
// reads, and never exists in memory.
//
// One 8-wave workgroup per SAMPLE (persistent over samples; 256 samples = 256 CUs at the headline batch), N <= 208 tokens = 13 groups of
// 16 rows, wave w owns groups w and w + 8.  Per head h:
//   P   dO_h^T [64 features, 16 rows] per group = Wproj^T[h] [64, 192] dY^T: the A operand is the head's 64 rows of proj.weight^T from an
//       LDS image (LDS-DMA, 16-B slots XOR-swizzled on the source address; rows permuted as in attn192.hip so that a lane's accumulators
//       of tiles 2 j, 2 j + 1 are 8 CONSECUTIVE features), the B operand the group's dY rows, held in registers for all three heads.
//       dO goes to its LDS image with one 16-byte write per lane; delta = rowsum(dO o O) is formed from the accumulators and the O chunk
//       of the same 8 features (the only global read of O).
//   A   dQ   (waves own query tiles)   |   the two phases of attn_bwd_head_kernel (csrc/attn.hip): S and dP recomputed per 32-key / 32-query
//   B   dK, dV (waves own key tiles)   |   step from the LDS images of q, k, v, dO; hand-laid pipeline over pinned LDS reads.
// q, k, v of a head arrive by LDS-DMA straight from the packed qkv matrix (no register staging): the images are UNPADDED 128-byte rows,
// 16-byte chunk c of row r at chunk c ^ (r & 6) -- that XOR keeps BOTH the ds_read_b128 row reads (16x16x32 A / B operands) and the
// ds_read_b64_tr_b16 transposed reads conflict-free (brute-forced against the bank rules of MI355X_MICROARCH.md, LDS) -- applied on
// the per-lane DMA source address.  Rows past N come from a page of zeros.  The transposed reads take their 4-column blocks from permuted
// features (tile t, row rho <-> feature 32 (t/2) + 8 (rho/4) + 4 (t%2) + rho%4), so dq, dk, dv leave with 16-byte stores.
// DMA of head h + 1's q, k, v starts when head h's phases are done and lands under phase P of head h + 1; every vector-memory operation
// inside the head loop is issued from asm with counted waits (a compiler-placed vmcnt would drain the DMA behind it).
#include <utility>
#include "common.h"

namespace {

constexpr int G_D = 192, G_H = 3, G_NT = 14, G_ROWS = G_NT * 16;      // 14 tiles of 16 rows (an even count: steps of 32): N <= 208 + padding
constexpr int G_IMG = G_ROWS * 128;                                   // one image: 28 672 B
constexpr int G_Q = 0, G_K = G_IMG, G_V = 2 * G_IMG, G_DO = 3 * G_IMG;
constexpr int G_W = 4 * G_IMG, G_W_BYTES = 64 * 384;                  // the head's 64 rows of proj.weight^T: 64 x 192 bf16
constexpr int G_LSE = G_W + G_W_BYTES, G_NDL = G_LSE + G_ROWS * 4;    // lse log2(e) and -rowsum(dO o O) / 8 per row
constexpr int G_HEADS_END = G_NDL + G_ROWS * 4;                       // 141 056 B: what the head loop uses
// phase C (qkv dgrad + LayerNorm backward) reuses the images' space: two buffers for a third of qkv.weight^T each ([192 rows, 192 k] bf16)
constexpr int G_CBUF = 192 * 384;                                     // 73 728 B
constexpr int G_TSTRIDE = 196;                                        // floats per row of a wave's [16][192] transposition table (phase C)
static_assert(8 * 16 * G_TSTRIDE * 4 <= 2 * G_CBUF, "the eight waves' tables live in the two weight buffers");
constexpr int G_RED = 2 * G_CBUF;                                     // f32 [2][192]: this workgroup's partial dgamma | dbeta
constexpr int G_SMEM = G_RED + 2 * G_D * 4;                           // 148 992 B
static_assert(G_RED >= G_HEADS_END, "the partial sums must survive the head loops of later samples");
constexpr float G_LOG2E = 1.4426950408889634f;

// Dev-only ablation bits (build with -DDKD_ATTN192B_ABL=n; results are then wrong, timings are the point): 1 no phase P MFMAs,
// 2 no phase A, 4 no phase B, 8 no global stores, 16 no q/k/v DMA, 32 no phase C MFMAs, 64 no LayerNorm epilogue, 128 no dgamma / dbeta sums.
#ifndef DKD_ATTN192B_ABL
#define DKD_ATTN192B_ABL 0
#endif
constexpr int GABL = DKD_ATTN192B_ABL;

struct Attn192Bwd {
  const bf16_t* dy;       // bf16 [B * N, 192]: gradient w.r.t. the branch's output (proj's output), DropPath scale applied
  const bf16_t* wpt;      // bf16 [192, 192]: proj.weight^T (row i = input feature i)
  const bf16_t* qkv;      // bf16 [B * N, 576] saved by the forward
  const bf16_t* o;        // bf16 [B * N, 192] saved by the forward
  const float* lse;       // f32 [B, 3, N]
  bf16_t* dqkv;           // bf16 [B * N, 576] out
  // phase C, optional (wqt == NULL: the kernel stops at dqkv)
  const bf16_t* wqt;      // bf16 [192, 576]: qkv.weight^T (row i = input feature i)
  const float* x;         // f32 [B * N, 192]: the block's input (what norm1 normalised)
  const float* gamma;     // f32 [192]: norm1.weight
  const float* mean;      // f32 [B * N]
  const float* rstd;      // f32 [B * N]
  float* g;               // f32 [B * N, 192]: gradient stream, += LN'(dqkv Wqkv)
  float* part;            // f32 [gridDim.x][384]: this workgroup's partial dgamma | dbeta (summed by dkd_ln_bwd_reduce)
  int B, N;
};

__device__ __attribute__((aligned(256))) uint4 g_zero_page[16];        // source of the padded rows' DMA (device globals start as zeros)

typedef uint32_t gu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t gu32x2 __attribute__((ext_vector_type(2)));
struct GTrPair {
  gu32x2 lo, hi;
  __device__ __forceinline__ bf16x8 get() const { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3)); }
};
// LDS reads whose place in the instruction stream is fixed by the source (as in attn.hip): volatile asm keeps them in program order, the
// data is handed to the compiler by an s_waitcnt lgkmcnt(N) tied ("+v") to the registers it releases.
template <int OFF>
__device__ __forceinline__ void g_issue_row(uint32_t a, gu32x4& v) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void g_issue_tr(uint32_t a, GTrPair& t) {       // rows r and r + 16 of a transposed 16-bit fragment
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.lo) : "v"(a), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.hi) : "v"(a), "n"(OFF + 2048));
}
template <int N>
__device__ __forceinline__ void g_wait(gu32x4& a) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N < 15 ? N : 15));
}
template <int N>
__device__ __forceinline__ void g_wait(gu32x4& a, gu32x4& b, gu32x4& c, gu32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N < 15 ? N : 15));
}
template <int N>
__device__ __forceinline__ void g_wait(GTrPair& a, GTrPair& b, GTrPair& c, GTrPair& d) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi) : "n"(N < 15 ? N : 15));
}
__device__ __forceinline__ bf16x8 g_bf(const gu32x4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 g_f4(const gu32x4& v) { return __builtin_bit_cast(f32x4, v); }
template <int... I, class F>
__device__ __forceinline__ void g_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void g_static_for(F&& f) {
  g_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
__device__ __forceinline__ gu32x4 g_pack8u(const f32x4& a, const f32x4& b) {
  return gu32x4{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
}
__device__ __forceinline__ bf16x8 g_pack8(const f32x4& a, const f32x4& b) { return __builtin_bit_cast(bf16x8, g_pack8u(a, b)); }
// A counted vector-memory wait whose count is chosen at RUN time (n: 12 or 13 LDS-DMA pieces left in flight; anything else waits for
// everything).  It releases NO registers.  Round 4 lesson, found by poisoning LDS / VGPRs with NaNs: this kernel first issued some global loads
// from asm ("=v" destinations) and released them with a later wait tied ("+v") to the same variables; under register pressure hipcc
// satisfied the ties with v_mov COPIES of the destinations placed in front of the wait -- copies of registers whose data had not arrived.
// Nothing the compiler can name may be in flight here: the head loop's only asynchronous traffic is LDS-DMA (no register result) and
// stores; the O rows that delta needs travel by LDS-DMA too (into the image that dO then overwrites).
__device__ __forceinline__ void g_vmwait_pieces(const int n) {
  asm volatile("s_cmp_eq_u32 %[n], 13\n\ts_cbranch_scc1 .Lg_w13_%=\n\ts_cmp_eq_u32 %[n], 12\n\ts_cbranch_scc1 .Lg_w12_%=\n\t"
               "s_waitcnt vmcnt(0)\n\ts_branch .Lg_wd_%=\n"
               ".Lg_w13_%=:\n\ts_waitcnt vmcnt(13)\n\ts_branch .Lg_wd_%=\n"
               ".Lg_w12_%=:\n\ts_waitcnt vmcnt(12)\n"
               ".Lg_wd_%=:" ::[n] "s"(n)
               : "memory", "scc");
}
__device__ __forceinline__ void g_dma16(uint32_t lds_dst, const void* src) {     // one 1-KiB piece: lane l -> LDS bytes dst + 16 l
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_dst), "v"(src) : "memory", "m0");
#pragma clang diagnostic pop
}

// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane of the row: two quad permutes, two row rotations
__device__ __forceinline__ float g_row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
  return v;
}

// Phase C of attn192_bwd_kernel as a function of its own (NOT inlined: its 96 accumulator + 48 operand registers would otherwise be
// allocated together with the head loop's and spill there; what is live across the call is a handful of scalars).
__device__ __forceinline__ const bf16_t* g_uniform_ptr(const bf16_t* q) {      // a wave-uniform pointer the compiler can keep in SGPRs
  const uint64_t v = (uint64_t)(uintptr_t)q;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const bf16_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ __noinline__ void attn192_bwd_phase_c(const Attn192Bwd p, char* smem, const size_t row0, const int lane, const int w_in) {
  const int w = __builtin_amdgcn_readfirstlane(w_in);
  const bf16_t* wqt = g_uniform_ptr(p.wqt);
  const int N = p.N, nt = (N + 15) >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  float* red = (float*)(smem + G_RED);
  const int grp[2] = {w, w + 8};
  const int ng = (grp[0] < nt ? 1 : 0) + (grp[1] < nt ? 1 : 0);
  // ================= C: dT^T [192 features, 16 rows] per group = Wqkv^T [192, 576] dqkv^T, then the LayerNorm backward on whole rows.
  // qkv.weight^T streams through two LDS buffers in three K chunks (the q, k and v thirds of the 576), rows permuted like the other
  // weight images (a lane's accumulators of tiles 2 j, 2 j + 1 = 8 consecutive features of ONE row); the B operand is the group's own
  // dqkv rows, read back from L2 (this CU has just written them: the stores were drained above, the lines were never in its L1).
  {
    // (the lane index goes through an opaque asm: everything derived from it below is computed HERE, not hoisted out of the sample loop into
    // registers that would have to live -- spilled -- across the head loop)
    int lane_c = lane;
    asm volatile("" : "+v"(lane_c));
    const int i16 = lane_c & 15, fg = lane_c >> 4, ax = (i16 >> 1) & 7;
    uint32_t csrc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int o = (9 * w + k) * 1024 + lane_c * 16;
      const int row = o / 384, cb = o % 384;
      const int ps = cb >> 4;
      const int ls = (ps & ~7) | ((ps & 7) ^ ((row >> 1) & 7));
      const int t = row >> 4, rho = row & 15;
      const int feat = 32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3);
      csrc[k] = (uint32_t)((feat * (3 * G_D) + ls * 8) * 2);         // byte offset inside qkv.weight^T for chunk 0
    }
    auto load_chunk = [&](const int ch, const int buf) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + buf * G_CBUF + (9 * w + k) * 1024);
        const uint32_t voff = csrc[k] + (uint32_t)ch * (G_D * 2);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(wqt) : "memory", "m0");
#pragma clang diagnostic pop
      }
    };
    const bf16_t* brow[2];
    bool blive[2];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
      const int r = grp[rg] * 16 + i16;
      blive[rg] = rg < ng && r < N;
      brow[rg] = p.dqkv + (row0 + (blive[rg] ? r : 0)) * (3 * G_D) + 8 * fg;
    }
    auto phase_c = [&](auto ngc) {
      constexpr int NG = decltype(ngc)::value;
      // the first two chunks do not depend on the head loop's dq / dk / dv stores: they are issued BEFORE those stores are waited for
      // (vmcnt(18) = everything older than these 18 pieces), so the stores' drain and the chunks' flight overlap; the barrier then makes
      // every wave's rows visible to the waves that read them back
      load_chunk(0, 0);
      load_chunk(1, 1);
      asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      f32x4 acc[NG][12];
#pragma unroll
      for (int rg = 0; rg < NG; ++rg)
#pragma unroll
        for (int dt = 0; dt < 12; ++dt) acc[rg][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const int buf = ch & 1;
        // the wave's B fragments of this chunk: PLAIN loads (no asm-issued register loads in this kernel: see g_vmwait_pieces); the
        // compiler's wait for them also drains the next chunk's pieces, which have had a chunk's worth of MFMAs to land.  Padded rows:
        // dT = 0 (nothing for dgamma / dbeta, no dx stored)
        bf16x8 bq[NG][6];
#pragma unroll
        for (int rg = 0; rg < NG; ++rg)
#pragma unroll
          for (int kk = 0; kk < 6; ++kk) {
            const uint4 v = *(const uint4*)(brow[rg] + ch * G_D + 32 * kk);
            bq[rg][kk] = __builtin_bit_cast(bf16x8, blive[rg] ? v : uint4{0u, 0u, 0u, 0u});
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // chunk ch has landed (this wave's pieces; the barrier: everybody's)
        __builtin_amdgcn_s_barrier();
        const uint32_t c0 = lds0 + buf * G_CBUF + i16 * 384 + 16 * (fg ^ ax), c1 = lds0 + buf * G_CBUF + i16 * 384 + 16 * ((4 | fg) ^ ax);
        const uint32_t c0h = c0 + 6 * (16 * 384), c1h = c1 + 6 * (16 * 384);
        gu32x4 fr[4];
        auto issue = [&](auto ii) {                                  // fragment ii: K step ii / 12, feature tile ii % 12
          constexpr int i = decltype(ii)::value, kk = i / 12, dt = i % 12;
          g_issue_row<(dt % 6) * (16 * 384) + (kk >> 1) * 128>(dt < 6 ? ((kk & 1) ? c1 : c0) : ((kk & 1) ? c1h : c0h), fr[i & 3]);
        };
        g_static_for<4>(issue);
        g_static_for<72>([&](auto ii) {
          constexpr int i = decltype(ii)::value, kk = i / 12, dt = i % 12;
          g_wait<(71 - i < 3 ? 71 - i : 3)>(fr[i & 3]);
#pragma unroll
          for (int rg = 0; rg < NG; ++rg)
            if (!(GABL & 32)) acc[rg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_bf(fr[i & 3]), bq[rg][kk], acc[rg][dt], 0, 0, 0);
          if constexpr (i + 4 < 72) issue(std::integral_constant<int, i + 4>{});
        });
        if (ch < 2) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();                              // everybody is done reading this buffer
          if (ch == 0) load_chunk(2, 0);
        }
      }
      // ---- LayerNorm backward on the wave's rows: dx = rstd (dT gamma - mean(dT gamma) - xhat mean(dT gamma xhat)), g += dx; a lane holds
      // features 32 j + 8 fg .. + 7 (j = 0..5) of row i16 in tiles 2 j, 2 j + 1; the row's other features sit in the lanes i16 + 16 k.
      // dgamma / dbeta (sums over ROWS, which sit on lanes): the wave transposes through a private LDS table -- it writes its 16 rows x 192
      // values (row stride 196 floats: conflict-free 16-byte writes), then every lane sums 3 columns over the 16 rows into running
      // registers; 24 writes + 96 reads per group instead of a 4-step cross-lane reduction for each of 96 values (11 us per sample).
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                  // nobody reads the weight buffers any more: they hold the tables now
      float sgam[3] = {0.f, 0.f, 0.f}, sbet[3] = {0.f, 0.f, 0.f};    // columns lane, lane + 64, lane + 128 of dgamma / dbeta
      float* tab = (float*)smem + w * (16 * G_TSTRIDE);
      auto col_sums = [&](float (&acc3)[3]) {                        // (the table was written by this wave: lgkmcnt(0), no barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {                            // 12 reads in flight at a time (registers are scarce here)
#pragma unroll
          for (int ii = i; ii < i + 4; ++ii) {
            t0 += tab[ii * G_TSTRIDE + lane_c];
            t1 += tab[ii * G_TSTRIDE + 64 + lane_c];
            t2 += tab[ii * G_TSTRIDE + 128 + lane_c];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(t2)::"memory");
        }
        acc3[0] += t0, acc3[1] += t1, acc3[2] += t2;
      };
#pragma unroll
      for (int rg = 0; rg < NG && !(GABL & 64); ++rg) {
        asm volatile("" ::: "memory");                               // (one group at a time: the loads of the next one stay below this line)
        const int r = grp[rg] * 16 + i16;
        const bool live = r < N;
        const size_t rowp = row0 + (live ? r : 0);
        const float mu = p.mean[rowp], rs = p.rstd[rowp];
        const float* xr = p.x + rowp * G_D + 8 * fg;
        float* gr = p.g + rowp * G_D + 8 * fg;
        f32x4 xh[12];                                                // x, then xhat
#pragma unroll
        for (int t = 0; t < 12; ++t) xh[t] = *(const f32x4*)(xr + 32 * (t >> 1) + 4 * (t & 1));
        float* trow = tab + i16 * G_TSTRIDE + 8 * fg;
        if (!(GABL & 128)) {                                           // dbeta: the rows' dT (padded rows hold zeros)
#pragma unroll
          for (int t = 0; t < 12; ++t) *(f32x4*)(trow + 32 * (t >> 1) + 4 * (t & 1)) = acc[rg][t];
          col_sums(sbet);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          const f32x4 gm = *(const f32x4*)(p.gamma + 32 * (t >> 1) + 8 * fg + 4 * (t & 1));
          f32x4& d = acc[rg][t];
          f32x4 dg;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xe = (xh[t][e] - mu) * rs;
            xh[t][e] = xe;
            dg[e] = d[e] * xe;
            d[e] *= gm[e];                                           // gy, kept in the accumulator's registers
            s1 += d[e];
            s2 += d[e] * xe;
          }
          if (!(GABL & 128)) *(f32x4*)(trow + 32 * (t >> 1) + 4 * (t & 1)) = dg;      // dgamma: dT xhat
        }
        f32x4 gv[12];                                                // the incoming gradient rows: in flight under the column sums
#pragma unroll
        for (int t = 0; t < 12; ++t) gv[t] = *(const f32x4*)(gr + 32 * (t >> 1) + 4 * (t & 1));
        if (!(GABL & 128)) col_sums(sgam);
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        s1 *= 1.f / G_D;
        s2 *= 1.f / G_D;
        if (live && (!(GABL & 8) || p.B < 0)) {
#pragma unroll
          for (int t = 0; t < 12; ++t) {
            const f32x4& gy = acc[rg][t];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(rs, gy[e] - s1 - xh[t][e] * s2, gv[t][e]);
            *(f32x4*)(gr + 32 * (t >> 1) + 4 * (t & 1)) = o;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {                                  // eight waves add into the workgroup's running sums
        atomicAdd(&red[64 * k + lane_c], sgam[k]);
        atomicAdd(&red[G_D + 64 * k + lane_c], sbet[k]);
      }
    };
    if (ng == 2) phase_c(std::integral_constant<int, 2>{});
    else phase_c(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// LN: the instantiation that goes on to phase C.  A template parameter, not a run-time test of p.wqt: the mere presence of the call site
// costs the head loop's register allocation 5-6 us per launch (67 against 61 us at batch 256: call ABI, SGPRs parked in VGPR lanes).
template <bool LN>
__global__ __launch_bounds__(512, 1) void attn192_bwd_kernel(const Attn192Bwd p) {
  __shared__ __attribute__((aligned(16))) char smem[G_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, fg = lane >> 4;
  const int N = p.N, nt = (N + 15) >> 4;                               // <= 13 (host)
  const int npc = (N + 7) >> 3;                                        // 8-row DMA pieces per image that hold a live row
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  float* lse_s = (float*)(smem + G_LSE);
  float* ndl_s = (float*)(smem + G_NDL);
  const float c = 0.125f * G_LOG2E;

  float* red = (float*)(smem + G_RED);
  for (int i = tid; i < 2 * G_D; i += 512) red[i] = 0.f;

  // ---- LDS-DMA pieces of the weight image: 24 pieces of 1 KiB per head, wave w issues pieces 3 w .. 3 w + 2.  (The per-lane source offsets
  // are recomputed at every use from an opaque copy of the lane index: kept in registers across the head loop they were spilled.)
  auto load_weights = [&](const int h) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int o = (3 * w + k) * 1024 + ln * 16;
      const int row = o / 384, cb = o % 384;                           // image row (0..63), byte inside the row
      const int ps = cb >> 4;
      const int ls = (ps & ~7) | ((ps & 7) ^ ((row >> 1) & 7));        // this physical 16-B slot holds logical slot ls
      const int t = row >> 4, rho = row & 15;
      const int feat = 32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3);
      const uint32_t dst = lds0 + G_W + (3 * w + k) * 1024;
      const uint32_t voff = (uint32_t)(((h * 64 + feat) * G_D + ls * 8) * 2);    // byte offset inside proj.weight^T
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(p.wpt) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };
  // ---- q, k, v and O pieces: 4 npc pieces of 8 rows, wave w issues pieces w, w + 8, ...  (lane: row 8 pc + lane / 8, physical chunk lane % 8).
  // The head's O rows go into the dO image: phase P reads a row's chunk from exactly the place it then writes dO to.
  const int npieces = 4 * npc;
  const int nq = (npieces - w + 7) >> 3;                               // pieces this wave issues per head (wave-uniform)
  auto load_qkvo = [&](const size_t row0, const int h) {
    if (GABL & 16) return;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int prow = ln >> 3;
    const int plc = (ln & 7) ^ (prow & 6);                             // the logical chunk this lane's 16 bytes hold
    for (int idx = w; idx < npieces; idx += 8) {
      const int which = idx / npc, pc = idx - which * npc;             // 0 q, 1 k, 2 v, 3 O
      const int row = pc * 8 + prow;
      const bf16_t* live = which < 3 ? p.qkv + (row0 + row) * (3 * G_D) + which * G_D + h * 64 + plc * 8 : p.o + (row0 + row) * G_D + h * 64 + plc * 8;
      g_dma16(lds0 + which * G_IMG + pc * 1024, row < N ? live : (const bf16_t*)g_zero_page);
    }
  };

  // weight-image fragments (phase P): tile ft, K step kk: row 16 ft + i16, logical slot 4 kk + fg
  const int ax = (i16 >> 1) & 7;
  const uint32_t wa0 = lds0 + G_W + i16 * 384 + 16 * (fg ^ ax), wa1 = lds0 + G_W + i16 * 384 + 16 * ((4 | fg) ^ ax);
  // image fragments: row reads (row i16 + 16 tile, chunk 4 ks + fg) and transposed reads (row 4 fg + i16 / 4, features 32 j + 8 (i16 % 4)
  // + 4 (dt % 2) .. + 3 for tile dt = 2 j + dt % 2); "Lo" bases address q (+0) and k (+G_IMG), "Hi" bases v (+0) and dO (+G_IMG)
  uint32_t rowLo[2], rowHi[2], trLo[2], trHi[2];
  {
    const int swr = i16 & 6, trow = 4 * fg + (i16 >> 2), swt = trow & 6, tp = i16 & 3;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      rowLo[k] = lds0 + i16 * 128 + 16 * ((4 * k + fg) ^ swr);
      rowHi[k] = rowLo[k] + 2 * G_IMG;
      trLo[k] = lds0 + trow * 128 + 16 * ((4 * k + tp) ^ swt);
      trHi[k] = trLo[k] + 2 * G_IMG;
    }
  }
  const uint32_t st_a = lds0 + G_LSE + 16 * fg;                        // lse_s[4 fg ..]; ndl_s is G_ROWS * 4 bytes further

  const int grp[2] = {w, w + 8};
  const int ng = (grp[0] < nt ? 1 : 0) + (grp[1] < nt ? 1 : 0);        // wave-uniform
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    const size_t row0 = (size_t)b * N;
    // ---- rows the DMA never writes (rows >= 8 npc of q, k, v and of the dO image, whose rows first hold O: phase P multiplies what it
    // finds there by the padded rows' dO = 0, and 0 x NaN is NaN) must read as zeros -- phase C of the previous sample has used the space,
    // and LDS starts with whatever the CU's previous kernel left in it
    for (int i = npc * 64 + tid; i < G_IMG / 16; i += 512) {
      *(uint4*)(smem + G_Q + i * 16) = uint4{0u, 0u, 0u, 0u};
      *(uint4*)(smem + G_K + i * 16) = uint4{0u, 0u, 0u, 0u};
      *(uint4*)(smem + G_V + i * 16) = uint4{0u, 0u, 0u, 0u};
    }
    for (int i = npc * 64 + tid; i < G_IMG / 16; i += 512) *(uint4*)(smem + G_DO + i * 16) = uint4{0u, 0u, 0u, 0u};
    // (and their -delta / 8: phase B reads the statistics of all 14 query tiles; a NaN left in LDS by an earlier kernel would reach dK
    // through 0 x NaN although those queries' q and dO rows are zero)
    if (nt * 16 + tid < G_ROWS) ndl_s[nt * 16 + tid] = 0.f;
    __builtin_amdgcn_s_waitcnt(0xC07F);                                // lgkmcnt(0)
    __syncthreads();
    load_weights(0);
    // ---- the wave's dY rows as MFMA B operands, for all three heads (lane (row i16, k group fg): features 32 kk + 8 fg .. + 7), and the rows'
    // log-sum-exps: plain loads, used (masked / scaled) right here -- the compiler's wait for them also covers head 0's weight pieces
    // and comes BEFORE any q / k / v / O piece is issued, so it drains nothing
    bf16x8 xt[2][6];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
      const int r = grp[rg] * 16 + i16;
      const bool live = rg < ng && r < N;                              // (padded rows: any valid address, zeroed in registers)
      const bf16_t* src = p.dy + (row0 + (live ? r : 0)) * G_D + 8 * fg;
#pragma unroll
      for (int kk = 0; kk < 6; ++kk) {
        const uint4 v = *(const uint4*)(src + 32 * kk);
        xt[rg][kk] = __builtin_bit_cast(bf16x8, live ? v : uint4{0u, 0u, 0u, 0u});
      }
    }
    float rl[3];                                                        // lse log2(e) of row tid, per head
#pragma unroll
    for (int h = 0; h < 3; ++h) rl[h] = tid < N ? p.lse[((size_t)b * G_H + h) * N + tid] * G_LOG2E : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (belt and braces: nothing is in flight when the pieces go out)

#pragma unroll 1
    for (int h = 0; h < G_H; ++h) {
      load_qkvo(row0, h);                                              // lands under phase P's GEMM
      // the weight image of this head -- issued before the pieces just issued (and, from the second head on, behind the previous head's
      // stores) -- has landed
      g_vmwait_pieces(nq);
      __builtin_amdgcn_s_barrier();

      // ================= P: dO of the wave's groups for this head
      auto phase_p = [&](auto ngc) {
        constexpr int NG = decltype(ngc)::value;
        f32x4 acc[NG][4];
#pragma unroll
        for (int rg = 0; rg < NG; ++rg)
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) acc[rg][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gu32x4 fr[4];
        auto issue = [&](auto ii) {                                    // fragment ii: K step ii / 4, feature tile ii % 4
          constexpr int i = decltype(ii)::value, kk = i >> 2, dt = i & 3;
          g_issue_row<dt * (16 * 384) + (kk >> 1) * 128>((kk & 1) ? wa1 : wa0, fr[i & 3]);
        };
        if (!(GABL & 1)) {
          g_static_for<4>(issue);
          g_static_for<24>([&](auto ii) {
            constexpr int i = decltype(ii)::value, kk = i >> 2, dt = i & 3;
            g_wait<(23 - i < 3 ? 23 - i : 3)>(fr[i & 3]);
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) acc[rg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_bf(fr[i & 3]), xt[rg][kk], acc[rg][dt], 0, 0, 0);
            if constexpr (i + 4 < 24) issue(std::integral_constant<int, i + 4>{});
          });
        }
        // every wave's q / k / v / O pieces have landed (the O rows of this wave's groups were fetched by other waves)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) {
          const int r = grp[rg] * 16 + i16;
          float d = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j) {                                // tiles 2 j, 2 j + 1: features 32 j + 8 fg .. + 7 of the head
            const gu32x4 pk = g_pack8u(acc[rg][2 * j], acc[rg][2 * j + 1]);
            gu32x4* slot = (gu32x4*)(smem + G_DO + r * 128 + 16 * ((4 * j + fg) ^ (r & 6)));
            const bf16x8 a = __builtin_bit_cast(bf16x8, pk), o8 = g_bf(*slot);     // the O chunk of the same 8 features, then dO in its place
            *slot = pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf((float)a[e], (float)o8[e], d);
          }
          d += __shfl_xor(d, 16, 64);
          d += __shfl_xor(d, 32, 64);
          if (fg == 0) ndl_s[r] = d * -0.125f;
        }
      };
      if (ng == 2) phase_p(std::integral_constant<int, 2>{});
      else phase_p(std::integral_constant<int, 1>{});
      if (tid < G_ROWS) lse_s[tid] = h == 0 ? rl[0] : (h == 1 ? rl[1] : rl[2]);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // q, k, v landed; dO / statistics written
      __builtin_amdgcn_s_barrier();
      if (h + 1 < G_H) load_weights(h + 1);                            // nobody reads this head's weight image any more

      // ================= A: dQ.  s[r] = S^T[key 4 fg + r][query i16].  Padded keys need no mask: their K rows are zero in LDS, so whatever
      // (finite) dS they get multiplies zeros in dQ.
      for (int qt = w; qt < nt && !(GABL & 2); qt += 8) {
        const int qrow = qt * 16 + i16;
        bf16x8 qf[2], dof[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = qrow * 128 + 16 * ((4 * ks + fg) ^ (qrow & 6));
          qf[ks] = *(const bf16x8*)(smem + G_Q + off);
          dof[ks] = *(const bf16x8*)(smem + G_DO + off);
        }
        const float nd8 = ndl_s[qrow] * 8.f;          // -delta
        const float lq3 = lse_s[qrow] + 3.f;          // p/8 = exp2(s c - lse log2e - 3): the 1/sqrt(64) of dS rides in the exponent
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing of the compiler's is in flight when the pinned reads start
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gu32x4 ka[2][2], va[2][2];                    // [16-key tile of the step][k half]
        g_issue_row<G_K>(rowLo[0], ka[0][0]), g_issue_row<G_K>(rowLo[1], ka[0][1]);
        g_issue_row<G_K + 2048>(rowLo[0], ka[1][0]), g_issue_row<G_K + 2048>(rowLo[1], ka[1][1]);
        g_issue_row<0>(rowHi[0], va[0][0]), g_issue_row<0>(rowHi[1], va[0][1]);
        g_issue_row<2048>(rowHi[0], va[1][0]), g_issue_row<2048>(rowHi[1], va[1][1]);
        g_static_for<G_NT / 2>([&](auto step) {
          constexpr int t = decltype(step)::value, O = t * 4096;
          g_wait<4>(ka[0][0], ka[0][1], ka[1][0], ka[1][1]);
          g_wait<0>(va[0][0], va[0][1], va[1][0], va[1][1]);
          GTrPair kt_[4];
          g_issue_tr<G_K + O>(trLo[0], kt_[0]), g_issue_tr<G_K + O + 8>(trLo[0], kt_[1]);
          g_issue_tr<G_K + O>(trLo[1], kt_[2]), g_issue_tr<G_K + O + 8>(trLo[1], kt_[3]);
          f32x4 s[2], dp[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            s[hf] = f32x4{0.f, 0.f, 0.f, 0.f}, dp[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              s[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_bf(ka[hf][ks]), qf[ks], s[hf], 0, 0, 0);
              dp[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_bf(va[hf][ks]), dof[ks], dp[hf], 0, 0, 0);
            }
          }
          if constexpr (t + 1 < G_NT / 2) {
            constexpr int P = O + 4096;
            g_issue_row<G_K + P>(rowLo[0], ka[0][0]), g_issue_row<G_K + P>(rowLo[1], ka[0][1]);
            g_issue_row<G_K + P + 2048>(rowLo[0], ka[1][0]), g_issue_row<G_K + P + 2048>(rowLo[1], ka[1][1]);
            g_issue_row<P>(rowHi[0], va[0][0]), g_issue_row<P>(rowHi[1], va[0][1]);
            g_issue_row<P + 2048>(rowHi[0], va[1][0]), g_issue_row<P + 2048>(rowHi[1], va[1][1]);
          }
          f32x4 ds[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[hf][r] = __builtin_amdgcn_exp2f(fmaf(s[hf][r], c, -lq3)) * (dp[hf][r] + nd8);
          const bf16x8 dsf = g_pack8(ds[0], ds[1]);
          g_wait<(t + 1 < G_NT / 2) ? 8 : 0>(kt_[0], kt_[1], kt_[2], kt_[3]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_[dt].get(), dsf, dq[dt], 0, 0, 0);
        });
        if (qrow < N && (!(GABL & 8) || p.B < 0)) {
          bf16_t* dp_ = p.dqkv + (row0 + qrow) * (3 * G_D) + h * 64 + 8 * fg;
#pragma unroll
          for (int j = 0; j < 2; ++j) *(gu32x4*)(dp_ + 32 * j) = g_pack8u(dq[2 * j], dq[2 * j + 1]);
        }
      }

      // ================= B: dK, dV.  s[r] = S[query 4 fg + r][key i16]
      for (int rnd = 0; rnd * 8 < nt && !(GABL & 4); ++rnd) {
        const int kt = rnd * 8 + ((w - 5 * rnd) & 7);  // the second round's tiles go to other waves (SIMDs) than phase A's
        if (kt >= nt) continue;
        const int krow = kt * 16 + i16;
        bf16x8 kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = krow * 128 + 16 * ((4 * ks + fg) ^ (krow & 6));
          kf[ks] = *(const bf16x8*)(smem + G_K + off);
          vf[ks] = *(const bf16x8*)(smem + G_V + off);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
          dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        gu32x4 qa[2][2], da[2][2], lq[2], nl[2];      // [16-query tile of the step][k half]; row statistics of the two tiles
        g_issue_row<0>(rowLo[0], qa[0][0]), g_issue_row<0>(rowLo[1], qa[0][1]);
        g_issue_row<2048>(rowLo[0], qa[1][0]), g_issue_row<2048>(rowLo[1], qa[1][1]);
        g_issue_row<G_IMG>(rowHi[0], da[0][0]), g_issue_row<G_IMG>(rowHi[1], da[0][1]);
        g_issue_row<G_IMG + 2048>(rowHi[0], da[1][0]), g_issue_row<G_IMG + 2048>(rowHi[1], da[1][1]);
        g_issue_row<0>(st_a, lq[0]), g_issue_row<64>(st_a, lq[1]);
        g_issue_row<G_ROWS * 4>(st_a, nl[0]), g_issue_row<G_ROWS * 4 + 64>(st_a, nl[1]);
        g_static_for<G_NT / 2>([&](auto step) {
          constexpr int t = decltype(step)::value, O = t * 4096;
          g_wait<8>(qa[0][0], qa[0][1], qa[1][0], qa[1][1]);
          g_wait<4>(da[0][0], da[0][1], da[1][0], da[1][1]);
          GTrPair td[4], tq[4];
          g_issue_tr<G_IMG + O>(trHi[0], td[0]), g_issue_tr<G_IMG + O + 8>(trHi[0], td[1]);
          g_issue_tr<G_IMG + O>(trHi[1], td[2]), g_issue_tr<G_IMG + O + 8>(trHi[1], td[3]);
          g_issue_tr<O>(trLo[0], tq[0]), g_issue_tr<O + 8>(trLo[0], tq[1]);
          g_issue_tr<O>(trLo[1], tq[2]), g_issue_tr<O + 8>(trLo[1], tq[3]);
          f32x4 s[2], dp[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            s[hf] = f32x4{0.f, 0.f, 0.f, 0.f}, dp[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              s[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_bf(qa[hf][ks]), kf[ks], s[hf], 0, 0, 0);
              dp[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_bf(da[hf][ks]), vf[ks], dp[hf], 0, 0, 0);
            }
          }
          g_wait<16>(lq[0], lq[1], nl[0], nl[1]);     // the statistics were requested before the 16 transposed reads of this step
          f32x4 pp[2], dss[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const f32x4 l = g_f4(lq[hf]), n = g_f4(nl[hf]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pr = __builtin_amdgcn_exp2f(fmaf(s[hf][r], c, -l[r]));
              pp[hf][r] = pr;
              dss[hf][r] = pr * fmaf(dp[hf][r], 0.125f, n[r]);
            }
          }
          const bf16x8 pf = g_pack8(pp[0], pp[1]);
          const bf16x8 dsf = g_pack8(dss[0], dss[1]);
          if constexpr (t + 1 < G_NT / 2) {
            constexpr int P = O + 4096, S4 = (t + 1) * 128;
            g_issue_row<P>(rowLo[0], qa[0][0]), g_issue_row<P>(rowLo[1], qa[0][1]);
            g_issue_row<P + 2048>(rowLo[0], qa[1][0]), g_issue_row<P + 2048>(rowLo[1], qa[1][1]);
            g_issue_row<G_IMG + P>(rowHi[0], da[0][0]), g_issue_row<G_IMG + P>(rowHi[1], da[0][1]);
            g_issue_row<G_IMG + P + 2048>(rowHi[0], da[1][0]), g_issue_row<G_IMG + P + 2048>(rowHi[1], da[1][1]);
            g_issue_row<S4>(st_a, lq[0]), g_issue_row<S4 + 64>(st_a, lq[1]);
            g_issue_row<G_ROWS * 4 + S4>(st_a, nl[0]), g_issue_row<G_ROWS * 4 + S4 + 64>(st_a, nl[1]);
          }
          constexpr int LATER = (t + 1 < G_NT / 2) ? 12 : 0;
          g_wait<LATER + 8>(td[0], td[1], td[2], td[3]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(td[dt].get(), pf, dv[dt], 0, 0, 0);
          g_wait<LATER>(tq[0], tq[1], tq[2], tq[3]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq[dt].get(), dsf, dk[dt], 0, 0, 0);
        });
        if (krow < N && (!(GABL & 8) || p.B < 0)) {
          bf16_t* kp_ = p.dqkv + (row0 + krow) * (3 * G_D) + G_D + h * 64 + 8 * fg;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            *(gu32x4*)(kp_ + 32 * j) = g_pack8u(dk[2 * j], dk[2 * j + 1]);
            *(gu32x4*)(kp_ + G_D + 32 * j) = g_pack8u(dv[2 * j], dv[2 * j + 1]);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                    // everybody is done with this head's images
    }
    if constexpr (LN) attn192_bwd_phase_c(p, smem, row0, lane, w);
  }
  if constexpr (LN) {                                                   // this workgroup's partial dgamma | dbeta
    __syncthreads();
    for (int i = tid; i < 2 * G_D; i += 512) p.part[(size_t)blockIdx.x * (2 * G_D) + i] = red[i];
  }
}

}  // namespace

extern "C" int dkd_attn192_bwd(const void* dy, const void* proj_wt, const void* qkv, const void* o, const float* lse, void* dqkv,
                               const void* qkv_wt, const float* x, const float* ln_w, const float* mean, const float* rstd, float* g,
                               float* d_ln_w, float* d_ln_b, float* ws, int32_t B, int32_t N, void* stream) {
  DKD_CHECK_ARG(dy && proj_wt && qkv && o && lse && dqkv, "attn192_bwd: null operand");
  DKD_CHECK_ARG(!qkv_wt || (x && ln_w && mean && rstd && g && d_ln_w && d_ln_b && ws),
                "attn192_bwd: with qkv_wt (the qkv dgrad + LayerNorm backward) x, ln_w, mean, rstd, g, d_ln_w, d_ln_b and ws are needed");
  DKD_CHECK_ARG(!qkv_wt || ((((uintptr_t)qkv_wt | (uintptr_t)x | (uintptr_t)ln_w | (uintptr_t)g) & 15) == 0), "attn192_bwd: operands must be 16-byte aligned");
  DKD_CHECK_ARG(B > 0 && N >= 8 && N <= 208, "attn192_bwd: need 8 <= N <= 208 tokens (N=%d)", N);
  DKD_CHECK_ARG((((uintptr_t)dy | (uintptr_t)proj_wt | (uintptr_t)qkv | (uintptr_t)o | (uintptr_t)dqkv) & 15) == 0, "attn192_bwd: operands must be 16-byte aligned");
  DKD_CHECK_ARG((long)B * N * 576 < (1L << 31), "attn192_bwd: qkv too large for 32-bit offsets");
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      dkd_set_error("attn192_bwd: cannot query the device");
      return DKD_ERR_HIP;
    }
    n_cu = prop.multiProcessorCount;
  }
  Attn192Bwd p;
  p.dy = (const bf16_t*)dy; p.wpt = (const bf16_t*)proj_wt; p.qkv = (const bf16_t*)qkv; p.o = (const bf16_t*)o; p.lse = lse;
  p.dqkv = (bf16_t*)dqkv; p.B = B; p.N = N;
  p.wqt = (const bf16_t*)qkv_wt; p.x = x; p.gamma = ln_w; p.mean = mean; p.rstd = rstd; p.g = g; p.part = ws;
  const int grid = B < n_cu ? B : n_cu;
  DKD_CHECK_ARG(!qkv_wt || (int64_t)grid * 2 * G_D * 4 <= dkd_layernorm_bwd_workspace_bytes(B * N, G_D),
                "attn192_bwd: ws (dkd_layernorm_bwd_workspace_bytes) too small for %d partial rows", grid);
  if (qkv_wt) hipLaunchKernelGGL(attn192_bwd_kernel<true>, dim3(grid), dim3(512), 0, as_stream(stream), p);
  else hipLaunchKernelGGL(attn192_bwd_kernel<false>, dim3(grid), dim3(512), 0, as_stream(stream), p);
  DKD_CHECK_LAUNCH("attn192_bwd");
  if (qkv_wt) return dkd_ln_bwd_reduce(ws, grid, d_ln_w, d_ln_b, G_D, stream);
  return DKD_OK;
}

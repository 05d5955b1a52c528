// qkv projection + scaled-dot-product attention of a D = 192, 3-head transformer block (the DeiT-tiny student) as ONE kernel:
// [3P] timm Attention.forward up to (not including) proj -- qkv = y1 Wqkv^T + b;  o = softmax(q k^T / 8) v per head -- reached from
// model/models.py:195 of the reference through timm's Block.  Unfused, the [M, 576] qkv matrix went to HBM from the GEMM and came back
// into the attention kernel (58 MB each way per block at batch 256) behind a launch boundary; it is still WRITTEN once here (the backward
// reads it), but q, k, v of a head reach the attention arithmetic through registers / LDS.
//
// One 8-wave workgroup per SAMPLE (persistent over samples: 256 samples = 256 CUs at the headline batch), N <= 208 tokens = 13 groups
// of 16 rows; wave w owns groups w and w + 8.  Per head h:
//   P1  q^T, k^T, v^T [64 features, 16 rows] per group = W_h [192 features, 192] y1^T: the A operand is the head's weight rows from an LDS
//       image (LDS-DMA, 16-B slots XOR-swizzled on the source address as in mlp192.hip), the B operand the group's y1 rows held in
//       registers in MFMA B layout for the whole kernel.  Computed TRANSPOSED, a lane's accumulators are 4 consecutive features of ONE row:
//       the image's rows are PERMUTED (tile t, row rho <-> feature 32 (t/2) + 8 (rho/4) + 4 (t%2) + rho%4) so that the 4 + 4 values of tiles 2j, 2j+1
//       in a lane are 8 consecutive features:
//         q  stays in registers and IS the B operand of S^T = K Q^T (natural k-slot order),
//         k  goes to the K image in LDS (plain ds_read_b128 fragment reads), v to the V image (read transposed by ds_read_b64_tr_b16 for
//            O^T = V^T P^T), both in 128-B rows XOR-swizzled as in attn_fwd_ring_kernel,
//       and all three are stored to the packed qkv matrix, 16 bytes per lane.
//   P2  the attention of csrc/attn.hip (attn_fwd_kernel) on the wave's own query groups; o and the log-sum-exp go to memory.
// The next head's weight image streams in during P2.  Two workgroup barriers per head.
#include <utility>
#include "common.h"

namespace {

constexpr int Q_D = 192, Q_H = 3, Q_NKT = 14, Q_ROWS = Q_NKT * 16;     // 14 key tiles (an even count for the paired P V steps): N <= 208 + a padding tile
constexpr int Q_W_BYTES = 192 * 384;                                   // the head's [q | k | v] weight rows: 192 features x 192 bf16
constexpr int Q_MAT = Q_ROWS * 128;                                    // K / V image: unpadded 128-B rows, XOR-swizzled (see below)
constexpr int Q_K_OFF = Q_W_BYTES, Q_V_OFF = Q_K_OFF + Q_MAT, Q_B_OFF = Q_V_OFF + Q_MAT;
constexpr int Q_SMEM = Q_B_OFF + 3 * Q_D * 4;                          // 131 072 + 2 304 B (the qkv bias: a global load in front of every MFMA chain exposed its latency)
constexpr float Q_LOG2E = 1.4426950408889634f, Q_LN2 = 0.6931471805599453f;

// Dev-only ablation bits (build with -DDKD_ATTN192_ABL=n; results are then wrong, timings are the point): 1 no P1 MFMAs / fragment reads,
// 2 no P2 (attention), 4 no weight LDS-DMA, 8 no global stores (behind a condition the compiler cannot fold: nothing is eliminated),
// 16 no barriers.
#ifndef DKD_ATTN192_ABL
#define DKD_ATTN192_ABL 0
#endif
constexpr int QABL = DKD_ATTN192_ABL;

struct Attn192 {
  const bf16_t* y1;       // bf16 [B * N, 192]: norm1 output
  const bf16_t* wqkv;     // bf16 [576, 192]
  const float* bqkv;      // f32 [576]
  bf16_t* qkv;            // bf16 [B * N, 576] out
  bf16_t* o;              // bf16 [B * N, 192] out
  float* lse;             // f32 [B, 3, N] out or NULL
  // PROJ instantiation only: the branch carried through proj and the residual, x1 = x + rowscale[b] (o Wproj^T + bproj)
  const bf16_t* wproj;    // bf16 [192, 192]: proj.weight (row f = output feature f)
  const float* bproj;     // f32 [192]
  const float* x;         // f32 [B * N, 192]: the block's input (the residual)
  const float* rowscale;  // f32 [B] DropPath scale of the branch, or NULL (= 1)
  float* x1;              // f32 [B * N, 192] out (may alias x)
  int B, N;
};

// ---- LDS reads whose position in the instruction stream is fixed by the source (as in attn.hip / mlp192.hip): volatile asm keeps them in
// program order; the data is handed to the compiler by an s_waitcnt lgkmcnt(N) tied ("+v") to the registers it releases, N = pinned reads
// issued after them (the LDS returns in order; compiler-generated LDS operations in between can only make the wait longer).
typedef uint32_t qu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t qu32x2 __attribute__((ext_vector_type(2)));
struct QTrPair {
  qu32x2 lo, hi;
  __device__ __forceinline__ bf16x8 get() const { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3)); }
};
template <int OFF>
__device__ __forceinline__ void q_issue_row(uint32_t a, qu32x4& v) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void q_issue_tr(uint32_t a, QTrPair& t) {       // rows r and r + 16 of a transposed 16-bit fragment
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.lo) : "v"(a), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.hi) : "v"(a), "n"(OFF + 2048));
}
template <int N>
__device__ __forceinline__ void q_wait(qu32x4& a) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N < 15 ? N : 15));
}
template <int N>
__device__ __forceinline__ void q_wait(qu32x4& a, qu32x4& b, qu32x4& c, qu32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N < 15 ? N : 15));
}
template <int N>
__device__ __forceinline__ void q_wait(QTrPair& a, QTrPair& b, QTrPair& c, QTrPair& d) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi) : "n"(N < 15 ? N : 15));
}
__device__ __forceinline__ bf16x8 q_bf(const qu32x4& v) { return __builtin_bit_cast(bf16x8, v); }
template <int... I, class F>
__device__ __forceinline__ void q_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void q_static_for(F&& f) {
  q_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
__device__ __forceinline__ bf16x8 q_pack8(const f32x4& a, const f32x4& b) {
  qu32x4 u = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, u);
}

// NF: number of leading key tiles known to be full (N >= 16 NF); tile NF gets its padding mask through the MFMA accumulator's initial value
// (-inf where key >= N, else 0), later tiles are all padding and skipped.  NF = -1: any N, masks applied with selects (attn.hip).
// PROJ: after the three heads the kernel goes on to x1 = x + rowscale (o Wproj^T + bproj) (round 4): proj.weight streams into the weight
// image's space during the last head's attention, the wave re-reads its own o rows (it has just stored them: L2) as the B operand of
// a transposed product whose accumulators are 8 consecutive output features of one row per tile pair, and the f32 residual epilogue
// runs straight from registers.  A template parameter, not a run-time test: see attn192_bwd.hip on what a conditional tail costs.
template <int NF, bool PROJ>
__global__ __launch_bounds__(512, 1) void attn192_fwd_kernel(const Attn192 p) {
  __shared__ __attribute__((aligned(16))) char smem[Q_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int N = p.N, nqt = (N + 15) >> 4;                              // <= 13 (host)
  char* Ks = smem + Q_K_OFF;
  char* Vs = smem + Q_V_OFF;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

  // ---- the K / V images start as zeros: the key tiles past the last group are never written and must read as finite numbers
  for (int i = tid; i < (Q_B_OFF - Q_K_OFF) / 16; i += 512) *(uint4*)(smem + Q_K_OFF + i * 16) = uint4{0u, 0u, 0u, 0u};
  const float* bl = (const float*)(smem + Q_B_OFF);
  for (int i = tid; i < 3 * Q_D; i += 512) ((float*)(smem + Q_B_OFF))[i] = p.bqkv[i];

  // ---- LDS-DMA pieces of the weight image: 72 pieces of 1 KiB per head, wave w issues pieces 9 w .. 9 w + 8.  The per-lane source offsets
  // are recomputed at every use from an opaque copy of the lane index (nine registers held across the head loop were spilled once the
  // PROJ tail raised the pressure).
  auto load_weights = [&](const int h) {
    if (QABL & 4) return;
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int o = (9 * w + c) * 1024 + ln * 16;
      const int row = o / 384, cb = o % 384;                             // image row = feature (which * 64 + d), byte inside the row
      const int ps = cb >> 4;
      const int ls = (ps & ~7) | ((ps & 7) ^ ((row >> 1) & 7));          // this physical 16-B slot holds logical slot ls
      // image row i of a part (q, k or v): tile t = i / 16, row rho in the tile  ->  feature 32 (t / 2) + 8 (rho / 4) + 4 (t % 2) + rho % 4 of the head:
      // the 4 + 4 accumulator values a lane holds for tiles 2 j and 2 j + 1 are then 8 CONSECUTIVE features (16-byte stores, natural k-slot order)
      const int i = row & 63, t = i >> 4, rho = i & 15;
      const int feat = 32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3);
      const uint32_t dst = lds0 + (9 * w + c) * 1024;
      const uint32_t voff = (uint32_t)((((row >> 6) * Q_D + feat + h * 64) * Q_D + ls * 8) * 2);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(p.wqkv) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };
  auto load_proj_weights = [&]() {          // the same 72 pieces, from proj.weight [192, 192] with all 12 row tiles permuted
    int ln = lane;
    asm volatile("" : "+v"(ln));             // (source offsets recomputed here: not worth nine registers across the head loop)
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int o = (9 * w + c) * 1024 + ln * 16;
      const int row = o / 384, cb = o % 384;
      const int ps = cb >> 4;
      const int ls = (ps & ~7) | ((ps & 7) ^ ((row >> 1) & 7));
      const int t = row >> 4, rho = row & 15;
      const int feat = 32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3);
      const uint32_t dst = lds0 + (9 * w + c) * 1024;
      const uint32_t voff = (uint32_t)((feat * Q_D + ls * 8) * 2);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(p.wproj) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };

  // weight-image fragments: tile ft, K step kk: row 16 ft + li, logical slot 4 kk + lg -> physical (slot & ~7) | ((slot & 7) ^ ((li >> 1) & 7))
  const int ax = (li >> 1) & 7;
  const uint32_t wa0 = lds0 + li * 384 + 16 * (lg ^ ax), wa1 = lds0 + li * 384 + 16 * ((4 | lg) ^ ax);
  // K image: row r, 16-B slot s (features 8 s .. + 7) at physical slot s ^ ((r >> 1) & 7); V image: row r, 32-B granule dt (features
  // 16 dt .. + 15) at granule dt ^ ((r >> 1) & 3)
  const uint32_t kb[2] = {lds0 + Q_K_OFF + li * 128 + (((0 * 4 + lg) ^ ax) * 16), lds0 + Q_K_OFF + li * 128 + (((1 * 4 + lg) ^ ax) * 16)};
  uint32_t vb[4];
  {
    const int row = 4 * lg + (li >> 2);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vb[dt] = lds0 + Q_V_OFF + row * 128 + ((dt ^ ((row >> 1) & 3)) * 32) + 8 * (li & 3);
  }
  const float c = 0.125f * Q_LOG2E;
  const bf16x8 ones = __builtin_bit_cast(bf16x8, qu32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
  f32x4 pinit = {0.f, 0.f, 0.f, 0.f};              // accumulator start of the partial key tile
  if (NF >= 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pinit[r] = (NF * 16 + 4 * lg + r >= N) ? -INFINITY : 0.f;
  }

  const int grp[2] = {w, w + 8};
  const int ng = (grp[0] < nqt ? 1 : 0) + (grp[1] < nqt ? 1 : 0);      // wave-uniform
  load_weights(0);
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    const size_t row0 = (size_t)b * N;
    // ---- the wave's y1 rows as MFMA B operands, for all three heads (lane (row li, k group lg): features 32 kk + 8 lg .. + 7); rows past
    // N repeat row N - 1 (finite numbers the padding mask keeps out of the softmax)
    bf16x8 xt[2][6];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
      const int r = grp[rg] * 16 + li;
      const bool live = rg < ng;                                       // (a group the wave does not own: any valid address, zeroed in registers --
      const int rc = !live ? 0 : (r < N ? r : N - 1);                  //  `live ? *ptr : zero` compiled to flat loads from a scratch copy of the zeros)
#pragma unroll
      for (int kk = 0; kk < 6; ++kk) {
        const uint4 v = *(const uint4*)(p.y1 + (row0 + rc) * Q_D + 32 * kk + 8 * lg);
        xt[rg][kk] = __builtin_bit_cast(bf16x8, live ? v : uint4{0u, 0u, 0u, 0u});
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                                // vmcnt(0) lgkmcnt(0): weights of head 0 landed, zero fill / bias written
    if (!(QABL & 16)) __syncthreads();

#pragma unroll 1
    for (int h = 0; h < Q_H; ++h) {
      // ================= P1: q, k, v of the wave's groups.  Per part (q, k, v): 24 weight fragments (6 K steps x 4 feature tiles), each read
      // ONCE for both groups, pinned four ahead of the MFMAs that consume them.
      bf16x8 qf[2][2];
      auto p1 = [&](auto ngc) {
        constexpr int NG = decltype(ngc)::value;
#pragma unroll
        for (int which = 0; which < 3; ++which) {                      // 0 q, 1 k, 2 v
          f32x4 acc[NG][4];
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const f32x4 bias = *(const f32x4*)(bl + which * Q_D + h * 64 + 32 * (dt >> 1) + 8 * lg + 4 * (dt & 1));     // the lane's 4 features of tile dt
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) acc[rg][dt] = bias;
          }
          __builtin_amdgcn_s_waitcnt(0xC07F);                          // (the compiler's bias reads are done before the pinned ones start)
          const uint32_t w0 = wa0 + which * (64 * 384), w1 = wa1 + which * (64 * 384);
          qu32x4 fr[4];
          auto issue = [&](auto ii) {                                  // fragment ii: K step ii / 4, feature tile ii % 4
            constexpr int i = decltype(ii)::value, kk = i >> 2, dt = i & 3;
            q_issue_row<dt * (16 * 384) + (kk >> 1) * 128>((kk & 1) ? w1 : w0, fr[i & 3]);
          };
          if (!(QABL & 1)) {
            q_static_for<4>(issue);
            q_static_for<24>([&](auto ii) {
              constexpr int i = decltype(ii)::value, kk = i >> 2, dt = i & 3;
              q_wait<(23 - i < 3 ? 23 - i : 3)>(fr[i & 3]);
#pragma unroll
              for (int rg = 0; rg < NG; ++rg) acc[rg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q_bf(fr[i & 3]), xt[rg][kk], acc[rg][dt], 0, 0, 0);
              if constexpr (i + 4 < 24) issue(std::integral_constant<int, i + 4>{});
            });
          }
#pragma unroll
          for (int rg = 0; rg < NG; ++rg) {
            const int r = grp[rg] * 16 + li;                           // row inside the sample (key / query index)
            bf16_t* qrow = p.qkv + (row0 + r) * (3 * Q_D) + which * Q_D + h * 64 + 8 * lg;
#pragma unroll
            for (int j = 0; j < 2; ++j) {                              // tiles 2 j, 2 j + 1: features 32 j + 8 lg .. + 7 of the head
              const uint4 pk = {pack2bf(acc[rg][2 * j][0], acc[rg][2 * j][1]), pack2bf(acc[rg][2 * j][2], acc[rg][2 * j][3]),
                                pack2bf(acc[rg][2 * j + 1][0], acc[rg][2 * j + 1][1]), pack2bf(acc[rg][2 * j + 1][2], acc[rg][2 * j + 1][3])};
              if (QABL & 32) {                                         // (ablation: the same bytes as perfectly coalesced 1-KiB stores, wrong places)
                const size_t off = (size_t)((((grp[rg] * 3 + h) * 3 + which) * 2 + j) * 1024 + lane * 16) % (size_t)(N * 1152 - 1024);
                *(uint4*)((char*)(p.qkv + row0 * (3 * Q_D)) + off) = pk;
              } else
              if (r < N && (!(QABL & 8) || p.B < 0)) *(uint4*)(qrow + 32 * j) = pk;
              if (which == 0) qf[rg][j] = __builtin_bit_cast(bf16x8, pk);
              if (which == 1) *(uint4*)(Ks + r * 128 + (((4 * j + lg) ^ ((r >> 1) & 7)) * 16)) = pk;
              if (which == 2) *(uint4*)(Vs + r * 128 + (((2 * j + (lg >> 1)) ^ ((r >> 1) & 3)) * 32) + (lg & 1) * 16) = pk;
            }
          }
        }
      };
      if (ng == 2) p1(std::integral_constant<int, 2>{});
      else if (ng == 1) p1(std::integral_constant<int, 1>{});
      __builtin_amdgcn_s_waitcnt(0xC07F);                              // lgkmcnt(0): K / V rows written
      if (!(QABL & 16)) __syncthreads();                               // everybody's K / V rows; nobody reads this head's weights any more
      if (PROJ && h + 1 == Q_H) load_proj_weights();                   // lands during the last head's P2
      else load_weights(h + 1 < Q_H ? h + 1 : 0);                      // lands during P2 (after the last head: head 0's again, for the next sample)

      // ================= P2: attention of the wave's query groups (the tile body of attn_fwd_ring_kernel, csrc/attn.hip)
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {
        if (rg >= ng || (QABL & 2)) break;
        const int q = grp[rg] * 16 + li;
        f32x4 s[Q_NKT];
        constexpr int NGK = Q_NKT / 2;
        qu32x4 kr[3][4];
        auto issue_k = [&](auto gi) {
          constexpr int g = decltype(gi)::value;
          q_issue_row<(2 * g) * 2048>(kb[0], kr[g % 3][0]), q_issue_row<(2 * g) * 2048>(kb[1], kr[g % 3][1]);
          q_issue_row<(2 * g + 1) * 2048>(kb[0], kr[g % 3][2]), q_issue_row<(2 * g + 1) * 2048>(kb[1], kr[g % 3][3]);
        };
        q_static_for<3>(issue_k);
        q_static_for<NGK>([&](auto gi) {
          constexpr int g = decltype(gi)::value, after = NGK - 1 - g < 2 ? NGK - 1 - g : 2;
          q_wait<4 * after>(kr[g % 3][0], kr[g % 3][1], kr[g % 3][2], kr[g % 3][3]);
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            constexpr f32x4 zero = {0.f, 0.f, 0.f, 0.f}, ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            const int t = 2 * g + hf;
            if (NF >= 0 && t > NF) {
              s[t] = ninf;
            } else {
              f32x4 a = (NF >= 0 && t == NF) ? pinit : zero;
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q_bf(kr[g % 3][2 * hf]), qf[rg][0], a, 0, 0, 0);
              s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q_bf(kr[g % 3][2 * hf + 1]), qf[rg][1], a, 0, 0, 0);
            }
          }
          if constexpr (g + 3 < NGK) issue_k(std::integral_constant<int, g + 3>{});
        });
        // the first V fragments are requested before the softmax arithmetic: they do not depend on it
        QTrPair vr[3][4];
        auto issue_v = [&](auto ki) {
          constexpr int kp = decltype(ki)::value;
          q_issue_tr<kp * 4096>(vb[0], vr[kp % 3][0]), q_issue_tr<kp * 4096>(vb[1], vr[kp % 3][1]);
          q_issue_tr<kp * 4096>(vb[2], vr[kp % 3][2]), q_issue_tr<kp * 4096>(vb[3], vr[kp % 3][3]);
        };
        q_static_for<2>(issue_v);
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < Q_NKT; ++kt) {
          if (NF < 0 && kt * 16 + 16 > N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kt * 16 + 4 * lg + r >= N) s[kt][r] = -INFINITY;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxc = mx * c;
#pragma unroll
        for (int kt = 0; kt < Q_NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c, -mxc));
        // the row sum is a fifth output tile of the P V product, against a fragment of ones (the sum of exactly the bf16 probabilities
        // that multiply V)
        f32x4 osum = {0.f, 0.f, 0.f, 0.f};
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        q_static_for<NGK>([&](auto ki) {
          constexpr int kp = decltype(ki)::value;
          q_wait<(kp + 1 < NGK) ? 8 : 0>(vr[kp % 3][0], vr[kp % 3][1], vr[kp % 3][2], vr[kp % 3][3]);
          if constexpr (kp + 2 < NGK) issue_v(std::integral_constant<int, kp + 2>{});
          const bf16x8 pf = q_pack8(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[kp % 3][dt].get(), pf, o[dt], 0, 0, 0);
          osum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, osum, 0, 0, 0);
        });
        const float sum = osum[0];        // every row of the ones tile holds the sums of its column = this lane's query
        if (q < N && (!(QABL & 8) || p.B < 0)) {
          const float inv = 1.f / sum;
          bf16_t* op = p.o + (row0 + q) * Q_D + h * 64 + 4 * lg;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const uint2 pk = {pack2bf(o[dt][0] * inv, o[dt][1] * inv), pack2bf(o[dt][2] * inv, o[dt][3] * inv)};
            if (QABL & 32) *(uint2*)((char*)(p.o + row0 * Q_D) + (size_t)(((grp[rg] * 3 + h) * 4 + dt) * 512 + lane * 8) % (size_t)(N * 384 - 512)) = pk;
            else *(uint2*)(op + dt * 16) = pk;
          }
          if (p.lse && lg == 0) p.lse[((size_t)b * Q_H + h) * N + q] = mxc * Q_LN2 + __logf(sum);
        }
      }
      // the next weight image has landed: everything up to the 9 pieces, i.e. all but the stores of o issued after them (4 per group; an lse
      // store each only makes the wait longer) -- not vmcnt(0), which would also sit out the drain of those stores
      if (QABL & 2) __builtin_amdgcn_s_waitcnt(0x0070);
      else if (ng == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
      else if (ng == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      if (!(QABL & 16)) __syncthreads();                               // everybody is done with this head's K / V
    }
    if constexpr (PROJ) {
      // ================= proj + residual.  (The weight image holds proj.weight since the last head's end-of-head wait.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's o stores are in L2: it reads its own rows back
      int ln = lane;
      asm volatile("" : "+v"(ln));                                     // (addresses of this tail are computed HERE, not hoisted above the head loop)
      const int li = ln & 15, lg = ln >> 4;
      bf16x8 ob[2][6];
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {
        const int r = grp[rg] * 16 + li;
        const bool live = rg < ng && r < N;
        const bf16_t* src = p.o + (row0 + (live ? r : 0)) * Q_D + 8 * lg;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
          const uint4 v = *(const uint4*)(src + 32 * kk);
          ob[rg][kk] = __builtin_bit_cast(bf16x8, live ? v : uint4{0u, 0u, 0u, 0u});
        }
      }
      const float sc = p.rowscale ? p.rowscale[b] : 1.f;
      auto proj = [&](auto ngc) {
        constexpr int NG = decltype(ngc)::value;
        f32x4 acc[NG][12];
#pragma unroll
        for (int dt = 0; dt < 12; ++dt) {
          const f32x4 bias = *(const f32x4*)(p.bproj + 32 * (dt >> 1) + 8 * lg + 4 * (dt & 1));       // the lane's 4 features of tile dt
#pragma unroll
          for (int rg = 0; rg < NG; ++rg) acc[rg][dt] = bias;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the compiler's loads are done before the pinned reads start)
        const uint32_t w0h = wa0 + 6 * (16 * 384), w1h = wa1 + 6 * (16 * 384);
        qu32x4 fr[4];
        auto issue = [&](auto ii) {                                    // fragment ii: K step ii / 12, feature tile ii % 12
          constexpr int i = decltype(ii)::value, kk = i / 12, dt = i % 12;
          q_issue_row<(dt % 6) * (16 * 384) + (kk >> 1) * 128>(dt < 6 ? ((kk & 1) ? wa1 : wa0) : ((kk & 1) ? w1h : w0h), fr[i & 3]);
        };
        q_static_for<4>(issue);
        q_static_for<72>([&](auto ii) {
          constexpr int i = decltype(ii)::value, kk = i / 12, dt = i % 12;
          q_wait<(71 - i < 3 ? 71 - i : 3)>(fr[i & 3]);
#pragma unroll
          for (int rg = 0; rg < NG; ++rg) acc[rg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q_bf(fr[i & 3]), ob[rg][kk], acc[rg][dt], 0, 0, 0);
          if constexpr (i + 4 < 72) issue(std::integral_constant<int, i + 4>{});
        });
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) {
          const int r = grp[rg] * 16 + li;
          if (r >= N) continue;
          const float* xr = p.x + (row0 + r) * Q_D + 8 * lg;
          float* yr = p.x1 + (row0 + r) * Q_D + 8 * lg;
          f32x4 xv[12];
#pragma unroll
          for (int dt = 0; dt < 12; ++dt) xv[dt] = *(const f32x4*)(xr + 32 * (dt >> 1) + 4 * (dt & 1));      // (all loads before the first store: x1 may alias x)
#pragma unroll
          for (int dt = 0; dt < 12; ++dt) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(sc, acc[rg][dt][e], xv[dt][e]);
            *(f32x4*)(yr + 32 * (dt >> 1) + 4 * (dt & 1)) = o;
          }
        }
      };
      if (ng == 2) proj(std::integral_constant<int, 2>{});
      else if (ng == 1) proj(std::integral_constant<int, 1>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(QABL & 16)) __syncthreads();                               // everybody is done with the proj image
      load_weights(0);                                                 // head 0's weights for the next sample (if any)
    }
  }
}

}  // namespace

namespace {
int attn192_launch(const void* y1, const void* wqkv, const float* bqkv, void* qkv, void* o, float* lse, const void* proj_w, const float* proj_b,
                   const float* x, const float* rowscale, float* x1, int32_t B, int32_t N, void* stream) {
  DKD_CHECK_ARG(y1 && wqkv && bqkv && qkv && o, "attn192_fwd: null operand");
  DKD_CHECK_ARG(B > 0 && N > 0 && N <= 208, "attn192_fwd: need 0 < N <= 208 tokens (N=%d)", N);
  DKD_CHECK_ARG((((uintptr_t)y1 | (uintptr_t)wqkv | (uintptr_t)bqkv | (uintptr_t)qkv | (uintptr_t)o) & 15) == 0, "attn192_fwd: operands must be 16-byte aligned");
  DKD_CHECK_ARG((long)B * N * 576 < (1L << 31), "attn192_fwd: qkv too large for 32-bit offsets");
  DKD_CHECK_ARG(!proj_w || (proj_b && x && x1 && (((uintptr_t)proj_w | (uintptr_t)proj_b | (uintptr_t)x | (uintptr_t)x1) & 15) == 0),
                "attn192_fwd_proj: proj_w needs proj_b, x and x1, 16-byte aligned");
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      dkd_set_error("attn192_fwd: cannot query the device");
      return DKD_ERR_HIP;
    }
    n_cu = prop.multiProcessorCount;
  }
  Attn192 p;
  p.y1 = (const bf16_t*)y1; p.wqkv = (const bf16_t*)wqkv; p.bqkv = bqkv; p.qkv = (bf16_t*)qkv; p.o = (bf16_t*)o; p.lse = lse;
  p.wproj = (const bf16_t*)proj_w; p.bproj = proj_b; p.x = x; p.rowscale = rowscale; p.x1 = x1;
  p.B = B; p.N = N;
  const int grid = B < n_cu ? B : n_cu;
  // 197 / 198 tokens: twelve full key tiles, the thirteenth partial (its padding mask rides in the MFMA accumulator), the fourteenth skipped
  const bool nf12 = N / 16 == 12;
  if (proj_w) {
    if (nf12) hipLaunchKernelGGL((attn192_fwd_kernel<12, true>), dim3(grid), dim3(512), 0, as_stream(stream), p);
    else hipLaunchKernelGGL((attn192_fwd_kernel<-1, true>), dim3(grid), dim3(512), 0, as_stream(stream), p);
  } else {
    if (nf12) hipLaunchKernelGGL((attn192_fwd_kernel<12, false>), dim3(grid), dim3(512), 0, as_stream(stream), p);
    else hipLaunchKernelGGL((attn192_fwd_kernel<-1, false>), dim3(grid), dim3(512), 0, as_stream(stream), p);
  }
  DKD_CHECK_LAUNCH("attn192_fwd");
  return DKD_OK;
}
}  // namespace

extern "C" int dkd_attn192_fwd(const void* y1, const void* wqkv, const float* bqkv, void* qkv, void* o, float* lse, int32_t B, int32_t N,
                               void* stream) {
  return attn192_launch(y1, wqkv, bqkv, qkv, o, lse, nullptr, nullptr, nullptr, nullptr, nullptr, B, N, stream);
}

extern "C" int dkd_attn192_fwd_proj(const void* y1, const void* wqkv, const float* bqkv, void* qkv, void* o, float* lse, const void* proj_w,
                                    const float* proj_b, const float* x, const float* rowscale, float* x1, int32_t B, int32_t N, void* stream) {
  DKD_CHECK_ARG(proj_w, "attn192_fwd_proj: null proj_w");
  return attn192_launch(y1, wqkv, bqkv, qkv, o, lse, proj_w, proj_b, x, rowscale, x1, B, N, stream);
}

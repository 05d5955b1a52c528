// The MLP branch of a D = 192 transformer block (the DeiT-tiny student) as ONE kernel per direction ([3P] timm Block: x = x + dp(mlp(ln2 x)),
// reached from model/models.py:195; its fc2 output is the feature the reference taps at model/models.py:185-193).
//
//   forward :  y2 = LN2(x1);  pre = y2 W1^T + b1;  h = gelu(pre);  f = h W2^T + b2 (the tap);  x2 = x1 + rowscale f
//   backward:  dF = bf16(s2 g + gtap);  dH = (dF W2) * gelu'(pre);  dT = dH W1;  g += LN2'(dT);  dgamma, dbeta;  dFa = bf16(s1 g)
//
// Unfused this was LN + two GEMM launches (forward) / scale-cast + dGELU GEMM + fused dgrad-LayerNorm GEMM (backward), with y2, h (fwd)
// and dF, dH (bwd) written and read back between them: 387 -> 252 MB and 465 -> 310 MB of HBM traffic per block at batch 256.
//
// Structure (both directions are the SAME skeleton with the two [hidden, 192] weight matrices swapped):
//  * a wave owns up to two GROUPS of 16 token rows for the whole kernel.  Its "input" operand X^T (forward: LN2 output, backward: dF)
//    is built in registers directly in the MFMA B-operand layout (lane = (row l & 15, k group l >> 4): 8 consecutive features per
//    32-wide K step, 6 K steps) -- the LayerNorm statistics are two xor-shuffles over the 4 lanes of a row.
//  * the hidden dimension is walked in K steps of 32 units.  GEMM 1 is computed TRANSPOSED,  P^T[hid, row] = Wa[hid, :] X^T  (A operand
//    = 16 rows of the "A-side" weight from LDS, B operand = the X^T registers), so a lane's accumulators are 4 consecutive hidden
//    units of ONE row for two 16-unit tiles: after the pointwise step (bias + GELU, or * gelu'(pre)) they ARE the A operand of GEMM 2
//    (row l & 15, k slots {4g+r, 16+4g+r}) -- the activation never leaves the registers.  The k-slot permutation is exactly the one
//    ds_read_b64_tr_b16 produces for the "B-side" weight [hid, 192] read transposed (as in the TN weight-gradient kernels).
//  * weights: forward A-side = fc1 W1 [hid, 192], B-side = fc2 W2^T [hid, 192]; backward A-side = W2^T, B-side = W1.  A K step's slice of
//    both (2 x 12 KiB, contiguous in memory) is streamed by LDS-DMA through a ring of 4 slots shared by the workgroup's 8 waves
//    (counted vmcnt waits, one barrier per K step); 16-B slots (A side) / 32-B granules (B side) are XOR-swizzled on the per-lane
//    SOURCE address so that the ds_read_b128 / ds_read_b64_tr_b16 fragment reads are bank-conflict free.
//  * one workgroup per CU, each with a contiguous run of <= 16 groups: the weights cross the L2 -> LDS path once per CU.
//  * `pre` is saved in a FRAGMENT-NATIVE layout (uint4 per lane per group and K step, 1 KiB coalesced per wave store): only the
//    backward of this same kernel reads it, lane for lane.  h / dH (operands of the weight-gradient kernel) go out row-major in full
//    128-byte lines through a small per-wave LDS staging buffer, two K steps at a time.  All row-major bf16 outputs are padded to a
//    multiple of 16 rows so that the stores of the counted region need no predicates.
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

constexpr int F_D = 192;
constexpr int F_SLICE = 32 * 384;            // one weight's 32 hidden rows x 192 bf16
constexpr int F_SLOT = 2 * F_SLICE;          // A-side slice, then B-side slice
constexpr int F_RING = 4;
constexpr int F_STG_ROW = 144;               // staging row: 128 B of data (2 K steps x 32 units) + 16 B pad
constexpr int F_STG_GRP = 16 * F_STG_ROW;
constexpr int F_STG_WAVE = 2 * F_STG_GRP;
constexpr int F_WAVES = 8;
constexpr int F_CS = 196;                    // f32 row stride of the epilogue staging (16 rows x 192 per wave)
constexpr int F_MAX_HIDDEN = 2048;           // the fc1 bias lives in LDS (a global load inside the loop would drain the DMA ring: see kstep)
constexpr int F_BIAS_OFF = F_RING * F_SLOT + F_WAVES * F_STG_WAVE;
constexpr int F_SMEM = F_BIAS_OFF + F_MAX_HIDDEN * 4;             // 96 KiB + 36 KiB + 8 KiB
static_assert(F_SMEM >= F_WAVES * 16 * F_CS * 4, "epilogue staging must fit");
static_assert(F_SMEM >= 2 * F_WAVES * F_D * 4, "dgamma / dbeta reduction must fit");

// Dev-only ablation bits (build with -DDKD_MLP_ABL=n; results are then wrong, timings are the point): 1 no pre stores / loads, 2 no row-major
// h / dH stores (nor their staging), 4 no pointwise arithmetic (GELU / gelu'), 8 no MFMAs, 16 no LDS fragment reads, 32 no LDS-DMA in
// the loop, 64 no barriers, 128 no prologue / epilogue global stores.
#ifndef DKD_MLP_ABL
#define DKD_MLP_ABL 0
#endif
constexpr int ABL = DKD_MLP_ABL;
#if DKD_MLP_ABL & 256
// bit 256: in-kernel stamps (s_memtime = shader cycles, s_memrealtime = 100 MHz) per wave into a buffer of their own, read back by the
// dev-only export at the end of this file (tools_dev/mlp192_stamps.py).  Slots: 0 start, 1 after the prologue, 2.. one per K step (taken
// after the step's barrier), 60 loop end, 61 kernel end, 62 / 63 realtime at start / end.
__device__ unsigned long long dkd_mlp_stamps[1024 * 8 * 64];
#define STAMP(slot_)                                                                                          \
  do {                                                                                                        \
    if (lane == 0) dkd_mlp_stamps[((size_t)blockIdx.x * 8 + w) * 64 + (slot_)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define STAMP_RT(slot_)                                                                                           \
  do {                                                                                                            \
    if (lane == 0) dkd_mlp_stamps[((size_t)blockIdx.x * 8 + w) * 64 + (slot_)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define STAMP(slot_) do { } while (0)
#define STAMP_RT(slot_) do { } while (0)
#endif
constexpr uint32_t vm_imm(int n) { return (uint32_t)((n & 15) | ((n >> 4) << 14) | 0x0F70); }   // s_waitcnt vmcnt(n) only

struct Mlp192 {
  int M, hidden, rows_per_sample, n_groups;
  const bf16_t* wa;        // A-side weight bf16 [hidden, 192]: forward fc1.weight, backward fc2.weight^T
  const bf16_t* wb;        // B-side weight bf16 [hidden, 192]: forward fc2.weight^T, backward fc1.weight
  uint4* pre;              // fragment-native pre-activation [n_groups][hidden / 32][64 lanes] (forward: out, backward: in)
  bf16_t* hid_out;         // row-major [16 n_groups, hidden]: forward h = gelu(pre), backward dH
  bf16_t* in16_out;        // row-major [16 n_groups, 192]: forward y2 (LN output), backward dF
  const float* rowscale;   // forward: DropPath scale of the residual; backward: the same scale applied to g (dF = s2 g + gtap); NULL = 1
  const float* x1;         // f32 [M, 192]: the LayerNorm input (forward: also the residual)
  const float* ln_w;
  const float* ln_b;       // forward only
  const float* b1;         // forward only
  const float* b2;         // forward only
  float eps;
  float* mean;             // forward: out (may be NULL); backward: in
  float* rstd;
  float* x2;               // forward: f32 [M, 192] out (may alias x1)
  bf16_t* tap;             // forward: optional bf16 [M, 192]
  float* g;                // backward: f32 [M, 192] in/out
  const bf16_t* gtap;      // backward: optional bf16 [M, 192]
  float* part;             // backward: per-workgroup partial sums [gridDim.x][2 * 192] (dgamma | dbeta)
  bf16_t* cast_out;        // backward: optional bf16 [M, 192] = rowscale_out[sample] * (updated g)
  const float* rowscale_out;
  // forward, optional: the NEXT block's norm1 applied to the rows of x2 while they are complete in this kernel's epilogue
  const float* nln_w;
  const float* nln_b;
  bf16_t* ny;              // bf16 [M, 192] = LN(x2)
  float* nmean;
  float* nrstd;
};

__device__ __forceinline__ bf16x8 as_bf16x8(const uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ void nt_store16_(bf16_t* q, const uint4 v) {      // streaming store: consumed by a much later kernel
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, (u32x4*)q);
}

// Fragment reads are PINNED (volatile asm, program order) and run DEPTH fragments ahead of the MFMAs that consume them: left to itself
// the compiler (at the register limit here) issues each ds_read right before its first use, and a wave then sits out one LDS round trip
// per two MFMAs -- the ablation builds showed the MFMA + LDS core alone at 43 of the forward's 77 us.  A read is handed to the compiler
// by an s_waitcnt lgkmcnt(N) tied ("+v") to its registers; N counts only the pinned reads issued after it (the LDS returns in order:
// compiler-generated LDS operations in between can only make the wait longer).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void lds_issue_b128(const uint32_t a, u32x4_t& v) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_issue_tr(const uint32_t a, u32x2_t& lo, u32x2_t& hi) {       // rows r and r + 16
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 16 * 384));
}
template <int N>
__device__ __forceinline__ void lds_release(u32x4_t& v) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N < 15 ? N : 15));
}
template <int N>
__device__ __forceinline__ void lds_release(u32x2_t& lo, u32x2_t& hi) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(N < 15 ? N : 15));
}
template <int... I, class Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, Fn&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
constexpr int F_DEPTH = 4;               // fragments in flight per GEMM

// MODE 0 forward, 1 backward.  SAVE (forward): write y2 / pre / h / mean / rstd (training); false = inference, nothing saved.
template <int MODE, bool SAVE>
__global__ __launch_bounds__(512, 2) void mlp192_kernel(const Mlp192 p) {
  __shared__ __attribute__((aligned(16))) char smem[F_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int KSn = p.hidden >> 5;

  // ---- this workgroup's groups, this wave's (up to two) groups
  const int g_begin = (int)((long)blockIdx.x * p.n_groups / gridDim.x), g_end = (int)((long)(blockIdx.x + 1) * p.n_groups / gridDim.x);
  const int cnt = g_end - g_begin;                                     // <= 16 (host)
  const int ng = __builtin_amdgcn_readfirstlane((w < cnt ? 1 : 0) + (w + F_WAVES < cnt ? 1 : 0));
  const int grp[2] = {g_begin + w, g_begin + w + F_WAVES};

  // ---- LDS-DMA pieces of this wave: pieces 3w .. 3w+2 of the 24 per slot (waves 0-3: A-side slice, waves 4-7: B-side slice)
  const bf16_t* wbase = w < 4 ? p.wa : p.wb;
  uint32_t srcoff[3], dstoff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int pidx = (w & 3) * 3 + c;                                  // piece inside its slice
    const int o = pidx * 1024 + lane * 16;
    const int row = o / 384, cb = o % 384;
    int col;
    if (w < 4) {                                                       // 16-B slots, XOR (row >> 1) & 7 inside aligned groups of 8
      const int ps = cb >> 4;
      const int ls = (ps & ~7) | ((ps & 7) ^ ((row >> 1) & 7));
      col = ls * 8;
    } else {                                                           // 32-B granules, XOR (row >> 1) & 3 inside groups of 4
      const int G = cb >> 5;
      const int gl = (G & ~3) | ((G & 3) ^ ((row >> 1) & 3));
      col = gl * 16 + ((cb >> 4) & 1) * 8;
    }
    srcoff[c] = (uint32_t)(row * F_D + col);
    dstoff[c] = (uint32_t)((w < 4 ? 0 : F_SLICE) + pidx * 1024);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  auto piece = [&](const int c, const int slice, const int slot) {
    const uint32_t dst = lds0 + slot * F_SLOT + dstoff[c];
    const uint32_t voff = (srcoff[c] + (uint32_t)slice * (32 * F_D)) * 2;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(wbase) : "memory", "m0");
#pragma clang diagnostic pop
  };

  STAMP(0);
  STAMP_RT(62);
  // forward: the fc1 bias into LDS.  (Read from global memory inside the loop, the compiler's wait for it -- it knows nothing of the
  // LDS-DMA pieces issued through asm -- was vmcnt(0) right behind the pieces of the step: the whole ring drained once per K step.)
  if (MODE == 0) {
    float* bl = (float*)(smem + F_BIAS_OFF);
    for (int i = tid; i < p.hidden; i += 512) bl[i] = p.b1[i];
    __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0): written before this wave meets the first barrier
  }

  // ---- prologue: the input operand X^T of each group, in registers (MFMA B layout: lane (row li, k group lg) holds features
  // 32 kk + 8 lg .. + 7 of K step kk)
  bf16x8 xt[2][6];
#pragma unroll
  for (int rg = 0; rg < 2; ++rg) {
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) xt[rg][kk] = as_bf16x8(uint4{0u, 0u, 0u, 0u});
    if (rg < ng) {
      const int row = grp[rg] * 16 + li;
      const int rowc = row < p.M ? row : p.M - 1;
      f32x4 v[12];
      if (MODE == 0) {
        const float* xr = p.x1 + (size_t)rowc * F_D + 8 * lg;
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
          v[2 * kk] = *(const f32x4*)(xr + 32 * kk);
          v[2 * kk + 1] = *(const f32x4*)(xr + 32 * kk + 4);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mu = s / F_D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mu;
            q += d * d;
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rs = rsqrtf(q / F_D + p.eps);
        if (SAVE && lg == 0 && row < p.M) {
          p.mean[row] = mu;
          p.rstd[row] = rs;
        }
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
          const float* gp = p.ln_w + 32 * kk + 8 * lg;
          const float* bp = p.ln_b + 32 * kk + 8 * lg;
          const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4), b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
          f32x4 y0, y1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y0[e] = (v[2 * kk][e] - mu) * rs * g0[e] + b0[e];
            y1[e] = (v[2 * kk + 1][e] - mu) * rs * g1[e] + b1[e];
          }
          const uint4 pk = {pack2bf(y0[0], y0[1]), pack2bf(y0[2], y0[3]), pack2bf(y1[0], y1[1]), pack2bf(y1[2], y1[3])};
          xt[rg][kk] = as_bf16x8(pk);
          if (SAVE && !(ABL & 128)) *(uint4*)(p.in16_out + (size_t)row * F_D + 32 * kk + 8 * lg) = pk;      // padded rows: no predicate
        }
      } else {
        const float sc = p.rowscale ? p.rowscale[rowc / p.rows_per_sample] : 1.f;
        const float* gr = p.g + (size_t)rowc * F_D + 8 * lg;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
          v[2 * kk] = *(const f32x4*)(gr + 32 * kk);
          v[2 * kk + 1] = *(const f32x4*)(gr + 32 * kk + 4);
        }
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
          f32x4 y0 = sc * v[2 * kk], y1 = sc * v[2 * kk + 1];
          if (p.gtap) {
            const uint4 t = *(const uint4*)(p.gtap + (size_t)rowc * F_D + 32 * kk + 8 * lg);
            y0 += f32x4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)};
            y1 += f32x4{__uint_as_float(t.z << 16), __uint_as_float(t.z & 0xffff0000u), __uint_as_float(t.w << 16), __uint_as_float(t.w & 0xffff0000u)};
          }
          const uint4 pk = {pack2bf(y0[0], y0[1]), pack2bf(y0[2], y0[3]), pack2bf(y1[0], y1[1]), pack2bf(y1[2], y1[3])};
          xt[rg][kk] = as_bf16x8(pk);
          if (!(ABL & 128)) *(uint4*)(p.in16_out + (size_t)row * F_D + 32 * kk + 8 * lg) = pk;
        }
      }
    }
  }

  STAMP(1);
  // ---- fragment read addresses (bytes inside a slot)
  // A side: tile t, K step kk: row 16 t + li, logical slot 4 kk + lg -> physical (slot & ~7) | ((slot & 7) ^ ((row >> 1) & 7)); (row >> 1) & 7
  // is the same for both tiles
  const int ax = (li >> 1) & 7;
  const int a_base = li * 384, a_o0 = 16 * (lg ^ ax), a_o1 = 16 * ((4 | lg) ^ ax);
  // B side (transposed reads): rows 4 lg + (li >> 2) (+16), 8-B piece li & 3 of the 32-B granule of column tile j
  const int frow = 4 * lg + (li >> 2), fcol = 8 * (li & 3), fsw = (frow >> 1) & 3;
  const int b_base = F_SLICE + frow * 384 + fcol;
  int b_o4[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) b_o4[m] = (m ^ fsw) * 32;

  f32x4 acc[2][12];
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[rg][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  char* stg = smem + F_RING * F_SLOT + w * F_STG_WAVE;

  // backward: the pre-activation fragments run two K steps ahead of their use
  uint4 preq[2][2];                       // [K step parity][group]
  auto load_pre = [&](const int s, const int par) {
    const int sc = s < KSn ? s : KSn - 1;
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
      preq[par][rg] = uint4{0u, 0u, 0u, 0u};
      if (rg < ng) {                      // (wave-uniform)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = __builtin_nontemporal_load((const u32x4*)(p.pre + ((size_t)grp[rg] * KSn + sc) * 64 + lane));
        preq[par][rg] = uint4{t.x, t.y, t.z, t.w};
      }
    }
  };

  // ---- ring prologue: slices 0, 1, 2 (slice s lives in slot s & 3)
#pragma unroll
  for (int u = 0; u < F_RING - 1; ++u) {
    const int sl = u < KSn ? u : KSn - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) piece(c, sl, u);
  }
  if (MODE == 1 && !(ABL & 1)) {
    load_pre(0, 0);
    load_pre(1, 1);
  } else if (MODE == 1) {
    preq[0][0] = preq[0][1] = preq[1][0] = preq[1][1] = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  }

  // One K step (32 hidden units).  NG: the wave's group count (compile time: the counted waits depend on it).
  // Vector-memory operations a wave issues per K step s, in order: 3 DMA pieces (slice s + 3), then NG "early" operations (forward:
  // the pre stores; backward: the pre loads of step s + 2), then -- odd s only -- 2 NG row-major stores of h / dH.  The wait that opens
  // step s needs slice s, issued in step s - 3: younger than it are the pieces of steps s - 2, s - 1 (6) and the other operations of
  // steps s - 3 .. s - 1 = 3 NG + 2 NG x (odd steps among them: 2 if s is even, 1 if odd).  Steps 0 .. 2 wait with the count of the ring
  // prologue (6), which can only over-wait.  Inference (no saves): the pieces are the only operations.
  auto kstep = [&](auto NGc, auto PARc, const int s) {
    constexpr int NG = decltype(NGc)::value, par = decltype(PARc)::value;
    constexpr bool saves = MODE == 1 || SAVE;
    constexpr int E = (saves && !(ABL & 1)) ? NG : 0;            // "early" operations per step
    constexpr int H = (saves && !(ABL & 2)) ? 2 * NG : 0;        // row-major stores per odd step
    constexpr int PC = (ABL & 32) ? 0 : 3;                       // pieces per step
    if (s < 3) __builtin_amdgcn_s_waitcnt(vm_imm(2 * PC));
    else if (par == 0) __builtin_amdgcn_s_waitcnt(vm_imm(2 * PC + 3 * E + 2 * H));
    else __builtin_amdgcn_s_waitcnt(vm_imm(2 * PC + 3 * E + H));
    if (!(ABL & 64)) __builtin_amdgcn_s_barrier();   // everybody's pieces of slice s have landed; everybody has left slot (s - 1) & 3
    __builtin_amdgcn_sched_barrier(0);
    if (s < 56) STAMP(2 + s);
    if (!(ABL & 32)) {
      const int nxt = s + 3 < KSn ? s + 3 : KSn - 1;      // past the end: re-issued into a slot nobody reads (keeps the counts uniform)
#pragma unroll
      for (int c = 0; c < 3; ++c) piece(c, nxt, (s + 3) & 3);
    }
    // GEMM 1, transposed: P^T[tile t][hid 4 lg + r][row li].  Fragment f = 6 t + kk; even / odd kk use the two swizzled slot offsets.
    f32x4 P[2][2];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) P[rg][0] = P[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t slot_a = lds0 + (s & 3) * F_SLOT;
    const uint32_t aA0 = slot_a + a_base + a_o0, aA1 = slot_a + a_base + a_o1;
    uint32_t aB[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) aB[m] = slot_a + b_base + b_o4[m];
    u32x4_t af[F_DEPTH];
    u32x2_t blo[F_DEPTH], bhi[F_DEPTH];
    u32x4_t bbq[2];                        // forward: the bias of this step's 2 x 4 hidden units of the lane (pinned LDS reads, oldest)
    if (MODE == 0) {
      const uint32_t ab = lds0 + F_BIAS_OFF + (32 * s + 4 * lg) * 4;
      lds_issue_b128<0>(ab, bbq[0]);
      lds_issue_b128<64>(ab, bbq[1]);
    }
    auto issue_a = [&](auto Fc) {
      constexpr int f = decltype(Fc)::value, t = f / 6, kk = f % 6;
      lds_issue_b128<t * (16 * 384) + (kk >> 1) * 128>((kk & 1) ? aA1 : aA0, af[f % F_DEPTH]);
    };
    auto issue_b = [&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      lds_issue_tr<(j >> 2) * 128>(aB[j & 3], blo[j % F_DEPTH], bhi[j % F_DEPTH]);
    };
    if (!(ABL & 16)) static_for<F_DEPTH>([&](auto f) { issue_a(f); });
    static_for<12>([&](auto Fc) {
      constexpr int f = decltype(Fc)::value, t = f / 6, kk = f % 6;
      constexpr int after = (11 - f) < (F_DEPTH - 1) ? (11 - f) : (F_DEPTH - 1);
      bf16x8 a;
      if (ABL & 16) a = xt[0][kk];
      else {
        lds_release<after>(af[f % F_DEPTH]);
        a = __builtin_bit_cast(bf16x8, af[f % F_DEPTH]);
      }
#pragma unroll
      for (int rg = 0; rg < NG; ++rg) {
        if (ABL & 8) P[rg][t][0] += __builtin_bit_cast(f32x4, a)[kk & 3];
        else P[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xt[rg][kk], P[rg][t], 0, 0, 0);
      }
      if (!(ABL & 16)) {
        if constexpr (f + F_DEPTH < 12) issue_a(std::integral_constant<int, f + F_DEPTH>{});
      }
    });
    // the first B-side fragments of GEMM 2 are requested now: they arrive under the pointwise arithmetic
    if (!(ABL & 16)) static_for<F_DEPTH>([&](auto j) { issue_b(j); });

    // pointwise step -> the A operand of GEMM 2 (k slots: tile 0 units 4 lg + r, then tile 1)
    bf16x8 hA[2];
    f32x4 bb[2];
    if (MODE == 0) {                       // (issued before every fragment read of this step: long since returned)
      lds_release<(ABL & 16) ? 1 : 2 * F_DEPTH + 1>(bbq[0]);
      lds_release<(ABL & 16) ? 0 : 2 * F_DEPTH>(bbq[1]);
      bb[0] = __builtin_bit_cast(f32x4, bbq[0]);
      bb[1] = __builtin_bit_cast(f32x4, bbq[1]);
    }
#pragma unroll
    for (int rg = 0; rg < NG; ++rg) {
      f32x4 v0, v1;
      if (MODE == 0) {
        v0 = P[rg][0] + bb[0];
        v1 = P[rg][1] + bb[1];
        if (SAVE && !(ABL & 1)) {
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 pk = {pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]), pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3])};
          __builtin_nontemporal_store(pk, (u32x4*)(p.pre + ((size_t)grp[rg] * KSn + s) * 64 + lane));
        }
#pragma unroll
        for (int e = 0; e < ((ABL & 4) ? 0 : 4); e += 2) {
          const dkd_f32x2 y0 = gelu_erf_fast2(dkd_f32x2{v0[e], v0[e + 1]}), y1 = gelu_erf_fast2(dkd_f32x2{v1[e], v1[e + 1]});
          v0[e] = y0[0], v0[e + 1] = y0[1], v1[e] = y1[0], v1[e + 1] = y1[1];
        }
      } else {
        const uint4 pr = preq[par][rg];
        const f32x4 q0 = {__uint_as_float(pr.x << 16), __uint_as_float(pr.x & 0xffff0000u), __uint_as_float(pr.y << 16), __uint_as_float(pr.y & 0xffff0000u)};
        const f32x4 q1 = {__uint_as_float(pr.z << 16), __uint_as_float(pr.z & 0xffff0000u), __uint_as_float(pr.w << 16), __uint_as_float(pr.w & 0xffff0000u)};
        v0 = P[rg][0];
        v1 = P[rg][1];
#pragma unroll
        for (int e = 0; e < ((ABL & 4) ? 0 : 4); e += 2) {
          const dkd_f32x2 d0 = dgelu_erf_fast2(dkd_f32x2{q0[e], q0[e + 1]}), d1 = dgelu_erf_fast2(dkd_f32x2{q1[e], q1[e + 1]});
          v0[e] *= d0[0], v0[e + 1] *= d0[1], v1[e] *= d1[0], v1[e + 1] *= d1[1];
        }
      }
      const uint4 hk = {pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]), pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3])};
      hA[rg] = as_bf16x8(hk);
      if ((MODE == 1 || SAVE) && !(ABL & 2)) {   // row-major copy for the weight-gradient kernel: staged, two K steps make a 128-B line
        char* q = stg + rg * F_STG_GRP + li * F_STG_ROW + par * 64 + 8 * lg;
        *(uint2*)q = uint2{hk.x, hk.y};
        *(uint2*)(q + 32) = uint2{hk.z, hk.w};
      }
    }
    if (MODE == 1 && !(ABL & 1)) load_pre(s + 2, par);   // (after the last use of preq[par] above)

    // GEMM 2: acc[row 4 lg + r][col 16 j + li] += h[row li][k slots] x Wb[k slots][col]
    static_for<12>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      constexpr int after = (11 - j) < (F_DEPTH - 1) ? (11 - j) : (F_DEPTH - 1);
      bf16x8 b;
      if (ABL & 16) b = xt[1][j >> 1];
      else {
        lds_release<2 * after>(blo[j % F_DEPTH], bhi[j % F_DEPTH]);
        b = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo[j % F_DEPTH], bhi[j % F_DEPTH], 0, 1, 2, 3));
      }
#pragma unroll
      for (int rg = 0; rg < NG; ++rg) {
        if (ABL & 8) acc[rg][j][0] += __builtin_bit_cast(f32x4, b)[j & 3] + __builtin_bit_cast(f32x4, hA[rg])[j & 3];
        else acc[rg][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hA[rg], b, acc[rg][j], 0, 0, 0);
      }
      if (!(ABL & 16)) {
        if constexpr (j + F_DEPTH < 12) issue_b(std::integral_constant<int, j + F_DEPTH>{});
      }
    });

    if (par == 1 && (MODE == 1 || SAVE) && !(ABL & 2)) { // full lines of h / dH: 8 lanes x 16 B per row, 8 rows per store instruction
#pragma unroll
      for (int rg = 0; rg < NG; ++rg)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int rl = (lane >> 3) + 8 * qq, c = lane & 7;
          const uint4 v = *(const uint4*)(stg + rg * F_STG_GRP + rl * F_STG_ROW + c * 16);
          bf16_t* dst = p.hid_out + (size_t)(grp[rg] * 16 + rl) * p.hidden + 32 * (s - 1) + 8 * c;
          if (MODE == 0) nt_store16_(dst, v);
          else *(uint4*)dst = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0): this wave is done reading the slot before it meets the next barrier
  };

  auto run = [&](auto NGc) {
    for (int s = 0; s < KSn; s += 2) {     // hidden % 64 == 0 (host)
      kstep(NGc, std::integral_constant<int, 0>{}, s);
      kstep(NGc, std::integral_constant<int, 1>{}, s + 1);
    }
  };
  if (ng == 2) run(std::integral_constant<int, 2>{});
  else if (ng == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});

  STAMP(60);
  // ---- epilogue: whole rows of the GEMM 2 result, one group at a time through this wave's f32 [16][196] staging
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the padding pieces have landed too: the ring becomes staging
  __syncthreads();
  float* cs = (float*)smem + w * (16 * F_CS);
  const int sl = li, gq = lg;                // row pass: 16 lanes per row (float4 columns sl, sl + 16, sl + 32), 4 rows per pass
  f32x4 gam[3], ag[3], ab[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gam[i] = MODE == 1 ? *(const f32x4*)(p.ln_w + 4 * (sl + 16 * i)) : *(const f32x4*)(p.b2 + 4 * (sl + 16 * i));   // forward: the fc2 bias
    ag[i] = ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int rg = 0; rg < 2; ++rg) {
    if (rg >= ng) break;                     // (wave-uniform; rg itself stays a compile-time register index)
#pragma unroll
    for (int j = 0; j < 12; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[(4 * lg + r) * F_CS + 16 * j + li] = acc[rg][j][r];
    // (wave-private staging: the LDS executes a wave's operations in order, no barrier needed)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rl = 4 * q + gq;
      const int row = grp[rg] * 16 + rl;
      const bool live = row < p.M;
      const int rowc = live ? row : p.M - 1;
      if (MODE == 0) {
        const float sc = p.rowscale ? p.rowscale[rowc / p.rows_per_sample] : 1.f;
        const float* xr = p.x1 + (size_t)rowc * F_D;
        f32x4 xv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) xv[i] = *(const f32x4*)(xr + 4 * (sl + 16 * i));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int c4 = 4 * (sl + 16 * i);
          const f32x4 f = *(const f32x4*)&cs[rl * F_CS + c4] + gam[i];
          xv[i] += sc * f;                   // the new residual row
          if (live && (!(ABL & 128) || f[0] == 1234.5f)) {
            if (p.tap) *(uint2*)(p.tap + (size_t)row * F_D + c4) = uint2{pack2bf(f[0], f[1]), pack2bf(f[2], f[3])};
            *(f32x4*)(p.x2 + (size_t)row * F_D + c4) = xv[i];
          }
        }
        if (p.ny) {                          // (uniform) LayerNorm of the finished row for the next block: its norm1 launch disappears
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i) s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
          const float mu = s / F_D;
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = xv[i][e] - mu;
              q += d * d;
            }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
          const float rs = rsqrtf(q / F_D + p.eps);
          if (live) {
            if (sl == 0) {
              p.nmean[row] = mu;
              p.nrstd[row] = rs;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const int c4 = 4 * (sl + 16 * i);
              const f32x4 gw = *(const f32x4*)(p.nln_w + c4), gb = *(const f32x4*)(p.nln_b + c4);
              f32x4 y;
#pragma unroll
              for (int e = 0; e < 4; ++e) y[e] = (xv[i][e] - mu) * rs * gw[e] + gb[e];
              *(uint2*)(p.ny + (size_t)row * F_D + c4) = uint2{pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
            }
          }
        }
      } else {
        const float mu = p.mean[rowc], rs = p.rstd[rowc];
        const float* xr = p.x1 + (size_t)rowc * F_D;
        float* dr = p.g + (size_t)rowc * F_D;
        f32x4 xv[3], dv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          xv[i] = *(const f32x4*)(xr + 4 * (sl + 16 * i));
          dv[i] = *(const f32x4*)(dr + 4 * (sl + 16 * i));
        }
        f32x4 xh[3], gy[3];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const f32x4 d = *(const f32x4*)&cs[rl * F_CS + 4 * (sl + 16 * i)];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[i][e] = (xv[i][e] - mu) * rs;
            gy[i][e] = d[e] * gam[i][e];
            if (live) {
              ab[i][e] += d[e];
              ag[i][e] += d[e] * xh[i][e];
            }
            s1 += gy[i][e];
            s2 += gy[i][e] * xh[i][e];
          }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        s1 *= 1.f / F_D;
        s2 *= 1.f / F_D;
        if (live && (!(ABL & 128) || s1 == 1234.5f)) {
          const float sc = p.cast_out ? (p.rowscale_out ? p.rowscale_out[row / p.rows_per_sample] : 1.f) : 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int c4 = 4 * (sl + 16 * i);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (gy[i][e] - s1 - xh[i][e] * s2) + dv[i][e];
            *(f32x4*)(dr + c4) = o;
            if (p.cast_out) *(uint2*)(p.cast_out + (size_t)row * F_D + c4) = uint2{pack2bf(sc * o[0], sc * o[1]), pack2bf(sc * o[2], sc * o[3])};
          }
        }
      }
    }
  }
  STAMP(61);
  STAMP_RT(63);
  if (MODE == 1) {
    // per-column partial sums of dgamma / dbeta: the 4 row lanes of a wave, then the 8 waves through LDS
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          ag[i][e] += __shfl_xor(ag[i][e], o, 64);
          ab[i][e] += __shfl_xor(ab[i][e], o, 64);
        }
    __syncthreads();
    float* red = (float*)smem;               // [2][8][192]
    if (gq == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        *(f32x4*)&red[(0 * F_WAVES + w) * F_D + 4 * (sl + 16 * i)] = ag[i];
        *(f32x4*)&red[(1 * F_WAVES + w) * F_D + 4 * (sl + 16 * i)] = ab[i];
      }
    }
    __syncthreads();
    if (tid < 2 * F_D) {
      const int which = tid / F_D, c = tid % F_D;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < F_WAVES; ++k) t += red[(which * F_WAVES + k) * F_D + c];
      p.part[(size_t)blockIdx.x * 2 * F_D + tid] = t;
    }
  }
}

int mlp192_grid(int M, int* n_groups) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n_cu = prop.multiProcessorCount;
  }
  const int G = cdiv(M, 16);
  *n_groups = G;
  const int lo = cdiv(G, 16);                        // at most 16 groups per workgroup
  int grid = cdiv(G, 4) < n_cu ? cdiv(G, 4) : n_cu;  // one workgroup per CU, at least 4 groups each (grid <= ceil(M / 64): the LN workspace)
  if (grid < lo) grid = lo;
  return grid;
}

}  // namespace

extern "C" int dkd_mlp192_fwd(const float* x1, const float* ln_w, const float* ln_b, float eps, const void* fc1_w, const float* fc1_b,
                              const void* fc2_wt, const float* fc2_b, const float* rowscale, int32_t rows_per_sample, float* x2, void* tap,
                              void* y2, void* pre, void* h, float* mean, float* rstd, const float* next_ln_w, const float* next_ln_b,
                              void* next_y, float* next_mean, float* next_rstd, int32_t M, int32_t hidden, void* stream) {
  DKD_CHECK_ARG(x1 && ln_w && ln_b && fc1_w && fc1_b && fc2_wt && fc2_b && x2, "mlp192_fwd: null operand");
  DKD_CHECK_ARG(M > 0 && hidden > 0 && hidden % 64 == 0 && hidden <= F_MAX_HIDDEN, "mlp192_fwd: hidden=%d must be a multiple of 64, <= %d", hidden,
                F_MAX_HIDDEN);
  DKD_CHECK_ARG(!rowscale || rows_per_sample > 0, "mlp192_fwd: rowscale needs rows_per_sample");
  const bool save = y2 || pre || h || mean || rstd;
  DKD_CHECK_ARG(!save || (y2 && pre && h && mean && rstd), "mlp192_fwd: the saved activations (y2, pre, h, mean, rstd) come all or none");
  DKD_CHECK_ARG(!next_y || (next_ln_w && next_ln_b && next_mean && next_rstd && ((uintptr_t)next_y & 7) == 0 && ((uintptr_t)next_ln_w & 15) == 0 &&
                            ((uintptr_t)next_ln_b & 15) == 0),
                "mlp192_fwd: the next block's LayerNorm needs its weight, bias, y, mean and rstd");
  DKD_CHECK_ARG((((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)ln_w | (uintptr_t)ln_b | (uintptr_t)fc1_w | (uintptr_t)fc2_wt | (uintptr_t)fc1_b |
                  (uintptr_t)fc2_b | (uintptr_t)tap | (uintptr_t)y2 | (uintptr_t)pre | (uintptr_t)h) & 15) == 0,
                "mlp192_fwd: operands must be 16-byte aligned");
  DKD_CHECK_ARG((long)hidden * F_D * 2 < (1L << 31), "mlp192_fwd: weight too large for 32-bit offsets");
  Mlp192 p = {};
  p.M = M; p.hidden = hidden; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
  const int grid = mlp192_grid(M, &p.n_groups);
  if (grid < 0) {
    dkd_set_error("mlp192_fwd: cannot query the device");
    return DKD_ERR_HIP;
  }
  p.wa = (const bf16_t*)fc1_w; p.wb = (const bf16_t*)fc2_wt;
  p.pre = (uint4*)pre; p.hid_out = (bf16_t*)h; p.in16_out = (bf16_t*)y2; p.rowscale = rowscale;
  p.x1 = x1; p.ln_w = ln_w; p.ln_b = ln_b; p.b1 = fc1_b; p.b2 = fc2_b; p.eps = eps; p.mean = mean; p.rstd = rstd; p.x2 = x2; p.tap = (bf16_t*)tap;
  p.nln_w = next_ln_w; p.nln_b = next_ln_b; p.ny = (bf16_t*)next_y; p.nmean = next_mean; p.nrstd = next_rstd;
  if (save) hipLaunchKernelGGL((mlp192_kernel<0, true>), dim3(grid), dim3(512), 0, as_stream(stream), p);
  else hipLaunchKernelGGL((mlp192_kernel<0, false>), dim3(grid), dim3(512), 0, as_stream(stream), p);
  DKD_CHECK_LAUNCH("mlp192_fwd");
  return DKD_OK;
}

extern "C" int dkd_mlp192_bwd(float* g, const void* gtap, const float* s2, const float* s1, int32_t rows_per_sample, const void* pre,
                              const void* fc2_wt, const void* fc1_w, const float* x1, const float* ln_w, const float* mean, const float* rstd,
                              void* dF, void* dH, void* cast_out, float* d_ln_w, float* d_ln_b, float* ws, int32_t M, int32_t hidden,
                              void* stream) {
  DKD_CHECK_ARG(g && pre && fc2_wt && fc1_w && x1 && ln_w && mean && rstd && dF && dH && d_ln_w && d_ln_b && ws, "mlp192_bwd: null operand");
  DKD_CHECK_ARG(M > 0 && hidden > 0 && hidden % 64 == 0, "mlp192_bwd: hidden=%d must be a multiple of 64", hidden);
  DKD_CHECK_ARG((!s1 && !s2) || rows_per_sample > 0, "mlp192_bwd: row scales need rows_per_sample");
  DKD_CHECK_ARG((((uintptr_t)g | (uintptr_t)gtap | (uintptr_t)pre | (uintptr_t)fc2_wt | (uintptr_t)fc1_w | (uintptr_t)x1 | (uintptr_t)ln_w |
                  (uintptr_t)dF | (uintptr_t)dH | (uintptr_t)cast_out) & 15) == 0,
                "mlp192_bwd: operands must be 16-byte aligned");
  DKD_CHECK_ARG((long)hidden * F_D * 2 < (1L << 31), "mlp192_bwd: weight too large for 32-bit offsets");
  Mlp192 p = {};
  p.M = M; p.hidden = hidden; p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
  const int grid = mlp192_grid(M, &p.n_groups);
  if (grid < 0) {
    dkd_set_error("mlp192_bwd: cannot query the device");
    return DKD_ERR_HIP;
  }
  p.wa = (const bf16_t*)fc2_wt; p.wb = (const bf16_t*)fc1_w;
  p.pre = (uint4*)pre; p.hid_out = (bf16_t*)dH; p.in16_out = (bf16_t*)dF; p.rowscale = s2;
  p.x1 = x1; p.ln_w = ln_w; p.mean = (float*)mean; p.rstd = (float*)rstd;
  p.g = g; p.gtap = (const bf16_t*)gtap; p.part = ws; p.cast_out = (bf16_t*)cast_out; p.rowscale_out = s1;
  hipLaunchKernelGGL((mlp192_kernel<1, true>), dim3(grid), dim3(512), 0, as_stream(stream), p);
  DKD_CHECK_LAUNCH("mlp192_bwd");
  return dkd_ln_bwd_reduce(ws, grid, d_ln_w, d_ln_b, F_D, stream);
}

#if DKD_MLP_ABL & 256
extern "C" int dkd_mlp192_read_stamps(void* host, int64_t bytes) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dkd_mlp_stamps), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

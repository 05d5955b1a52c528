// Scaled-dot-product attention forward / backward for ViT-sized sequences (N <= 256, head_dim 64) on gfx950.
//
// Replaces F.scaled_dot_product_attention inside timm's Attention ([3P], reached from model/models.py:195 of the
// reference).  A whole head's K and V (<= 256 x 64 bf16) fit in LDS, so there is no online-softmax rescale: one
// workgroup per (batch, head); each wave owns 16-query tiles.
//
// MFMA orientation (cdna_hip_programming section 3, "accumulator tile as the next MFMA's operand"): the score tile is
// computed TRANSPOSED, S^T = K Q^T, so a lane owns one query (column = lane&15) and 4 keys per 16-key tile.  Row max / sum
// are then 2 xor-shuffles (lanes +16, +32), and the probabilities feed the next product O^T = V^T P^T straight from the
// accumulator registers as its B operand (k-slot permutation key(g,j) = 4g+j | 16+4g+(j-4), matched on the V side by two
// ds_read_b64_tr_b16 hardware-transposed reads).  Nothing but K/V ever touches LDS.
// The qkv operand is the packed output of the qkv Linear ([B, N, 3, H, 64]) read in place: no head-split copy.
#include <utility>

#include "common.h"

namespace {

constexpr int KV_LD = 160;  // bytes per LDS row: 128 B data + 32 B pad (b128 row reads and tr_b16 reads both conflict-free)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ bf16x8 tr_pair(const char* p) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p + 16 * KV_LD));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 u = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ void store4bf(bf16_t* p, const f32x4& v, float s) {
  uint2 pk = {pack2bf(v[0] * s, v[1] * s), pack2bf(v[2] * s, v[3] * s)};
  *(uint2*)p = pk;
}

// stage `rows_valid` rows of 64 bf16 (row stride ld elements) into an LDS image of `rows_total` rows, zero padded
template <int NTHREADS = 256>
__device__ __forceinline__ void stage_rows(char* dst, const bf16_t* src, int ld, int rows_valid, int rows_total, int tid) {
  for (int idx = tid; idx < rows_total * 8; idx += NTHREADS) {
    const int row = idx >> 3, ch = idx & 7;
    s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < rows_valid) v = *(const s16x8*)(src + (size_t)row * ld + ch * 8);
    *(s16x8*)(dst + row * KV_LD + ch * 16) = v;
  }
}

// 4 waves per (batch, head), two workgroups per CU (LDS: 72 KB each).  Measured alternatives on the teacher shape (3072 heads,
// N = 198): 8 waves/1 WG per CU 166 us, 8 waves/2 WGs (<= 128 VGPRs: spills) 330 us, this 149 us.  The kernel is latency-bound
// (Q load -> QK^T -> softmax -> PV -> store per tile), not MFMA-bound.
constexpr int FWD_WAVES = 4;
template <int NKT>
__global__ __launch_bounds__(64 * FWD_WAVES, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i16 = lane & 15, fg = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int D = H * 64, ld = 3 * D;
  const bf16_t* base = qkv + (size_t)b * N * ld + h * 64;
  char* Ks = smem;
  char* Vs = smem + NKT * 16 * KV_LD;
  // Q fragments of all of this wave's query tiles are requested BEFORE K/V are staged: loaded inside the tile loop, each tile began
  // with a strided global load whose full latency was exposed (two waves per SIMD do not hide ~2 us), 3-4 times per wave.
  constexpr int MAXQ = (NKT + FWD_WAVES - 1) / FWD_WAVES;
  const int nqt = (N + 15) >> 4;
  bf16x8 qall[MAXQ][2];
#pragma unroll
  for (int it = 0; it < MAXQ; ++it) {
    const int q = (w + it * FWD_WAVES) * 16 + i16;
    const int qc = q < N ? q : N - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qall[it][ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 32 + 8 * fg);
  }
  stage_rows<64 * FWD_WAVES>(Ks, base + D, ld, N, NKT * 16, tid);
  stage_rows<64 * FWD_WAVES>(Vs, base + 2 * D, ld, N, NKT * 16, tid);
  __syncthreads();

  const float c = 0.125f * LOG2E;
#pragma unroll
  for (int it = 0; it < MAXQ; ++it) {
    const int qt = w + it * FWD_WAVES;
    if (qt >= nqt) break;
    const int q = qt * 16 + i16;
    const bf16x8 (&qf)[2] = qall[it];
    // Softmax VALU budget (the kernel is VALU-bound, not MFMA-bound): per score one max, one fma (scale folded into the exp2
    // argument), one exp2, one add; masking touches only the key tiles that straddle or exceed N.
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {                     // all QK^T MFMAs back to back (no VALU consumer in between)
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 16 + i16) * KV_LD + (ks * 32 + 8 * fg) * 2);
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt * 16 + 16 > N) {                              // wave-uniform: only the ragged / padded key tiles are masked
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 16 + 4 * fg + r >= N) s[kt][r] = -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxc = mx * c;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c, -mxc));
        s[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < NKT / 2; ++kp) {
      const bf16x8 pf = pack8(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 vf = tr_pair(Vs + (32 * kp + 4 * fg + (i16 >> 2)) * KV_LD + (dt * 16 + 4 * (i16 & 3)) * 2);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
      }
    }
    if (q < N) {
      const float inv = 1.f / sum;
      bf16_t* op = out + ((size_t)b * N + q) * D + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4bf(op + dt * 16, o[dt], inv);
      if (lse && fg == 0) lse[((size_t)b * H + h) * N + q] = mxc * LN2 + __logf(sum);
    }
  }
}

// ---- LDS reads whose position in the instruction stream is fixed by the source --------------------------------------------------------
// The compiler issues a ds_read just before its first use (it is at the register limit here), so a wave with one partner on its SIMD
// sits in s_waitcnt for most of every step.  These wrappers are volatile asm: they stay in program order, and the data is only handed
// to the compiler by lds_wait<N>(...), an s_waitcnt lgkmcnt(N) that is tied ("+v") to the registers it releases.  Rules for the caller:
// every read issued must be released by a later lds_wait before its register dies, N counts the reads issued after the ones being
// released (LDS returns in order), and no compiler-generated LDS access may sit between an issue and its release.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct TrPair {
  u32x2 lo, hi;
  __device__ __forceinline__ bf16x8 get() const { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3)); }
};
template <int OFF>
__device__ __forceinline__ void lds_issue_row(uint32_t a, u32x4& v) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
}
template <int OFF, int PAIR = 16 * KV_LD>
__device__ __forceinline__ void lds_issue_tr(uint32_t a, TrPair& t) {       // rows r and r + 16 of a transposed 16-bit fragment
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.lo) : "v"(a), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.hi) : "v"(a), "n"(OFF + PAIR));
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N < 15 ? N : 15));   // the counter has 4 bits
}
template <int N>
__device__ __forceinline__ void lds_wait(TrPair& a, TrPair& b, TrPair& c, TrPair& d) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi) : "n"(N < 15 ? N : 15));
}
__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 as_f4(const u32x4& v) { return __builtin_bit_cast(f32x4, v); }

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// ---- forward, persistent: one 8-wave workgroup per CU walks heads; K/V of head k+1 stream into the second LDS buffer by LDS-DMA
// while head k is computed.  The kernel above spends 33 us staging (HBM-bound: 155 MB of K/V) and 64 us computing per teacher
// layer and the two do not overlap -- both workgroups of a CU stage at the same time, then both compute (127 us; the layer's HBM
// floor is 311 MB / 5 TB/s = 62 us).
//  * LDS rows are unpadded (a DMA writes 1 KiB linearly): K 16-B slots are XOR-swizzled by (row>>1)&7 (ds_read_b128 row reads),
//    V 32-B granules by (row>>1)&3 (ds_read_b64_tr_b16 reads), both on the per-lane DMA source address and the fragment reads.
//    Rows past N are sourced from row N-1 (finite; their scores are masked / their probabilities are 0).
//  * ordering: the Q fragments of head k+1 are loaded (ordinary global loads) right AFTER the DMA of head k+1 is issued; the
//    vector-memory counter is in order, so the wait the compiler places before their first use -- forced to sit before the
//    barrier that opens head k+1 -- also covers the DMA, without draining the output stores issued after them.
// NF: number of leading key tiles known to be full (N >= 16 NF); tile NF gets its padding mask through the MFMA accumulator's initial
// value (-inf where key >= N, else 0), later tiles are all padding and skipped.  NF = -1: any N, masks applied with selects -- which
// the compiler spreads over all NKT tiles (2 v_cndmask per score plus spilled condition masks: as much VALU work again as the softmax).
template <int NKT, int NF>
__global__ __launch_bounds__(512, 1) void attn_fwd_ring_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                float* __restrict__ lse, int N, int H, int n_heads) {
  constexpr int NW = 8, ROWS = NKT * 16, MAT = ROWS * 128, BUF = 2 * MAT, NPIECE = 2 * (ROWS / 8);
  constexpr int MAXQ = (NKT + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][K | V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, fg = lane >> 4;
  const int D = H * 64, ld = 3 * D;
  const int nqt = (N + 15) >> 4;
  const float c = 0.125f * LOG2E;
  const int my_heads = ((int)blockIdx.x < n_heads) ? (n_heads - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  if (my_heads == 0) return;

  auto head_base = [&](int k) {
    const int hd = blockIdx.x + k * gridDim.x;
    return qkv + (size_t)(hd / H) * N * ld + (hd % H) * 64;
  };
  auto issue_dma = [&](int k) {
    const bf16_t* base = head_base(k);
    const uint32_t dst0 = (uint32_t)(uintptr_t)LDS_PTR(smem) + (k & 1) * BUF;
    for (int p = w; p < NPIECE; p += NW) {
      const bool isv = p >= ROWS / 8;
      const int row = (isv ? p - ROWS / 8 : p) * 8 + (lane >> 3);
      const int ph = lane & 7;
      const int col = isv ? (((ph >> 1) ^ ((row >> 1) & 3)) * 2 + (ph & 1)) * 8 : (ph ^ ((row >> 1) & 7)) * 8;
      const bf16_t* src = base + (isv ? 2 * D : D) + (size_t)(row < N ? row : N - 1) * ld + col;
      const uint32_t dst = dst0 + p * 1024;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };
  auto load_q = [&](bf16x8 (&qf)[MAXQ][2], int k) {
    const bf16_t* base = head_base(k);
#pragma unroll
    for (int it = 0; it < MAXQ; ++it) {
      const int q = (w + it * NW) * 16 + i16;
      const int qc = q < N ? q : N - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[it][ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 32 + 8 * fg);
    }
  };

  // fragment addresses inside a buffer: the swizzle keys depend on the lane only (row = 16 kt + i16 resp. 32 kp + 4 fg + i16/4),
  // so every read is one of these bases plus an immediate
  int k_off[2], v_off[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) k_off[ks] = i16 * 128 + (((ks * 4 + fg) ^ ((i16 >> 1) & 7)) * 16);
  {
    const int row = 4 * fg + (i16 >> 2);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) v_off[dt] = row * 128 + ((dt ^ ((row >> 1) & 3)) * 32) + 8 * (i16 & 3);
  }
  const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
  f32x4 pinit = {0.f, 0.f, 0.f, 0.f};              // accumulator start of the partial key tile
  if (NF >= 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pinit[r] = (NF * 16 + 4 * fg + r >= N) ? -INFINITY : 0.f;
  }
  bf16x8 qcur[MAXQ][2], qnext[MAXQ][2];
  issue_dma(0);
  load_q(qcur, 0);
  for (int k = 0; k < my_heads; ++k) {
    // the Q fragments of this head are in registers => (in-order counter) its K/V pieces issued before them have landed
#pragma unroll
    for (int it = 0; it < MAXQ; ++it)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) asm volatile("" ::"v"(qcur[it][ks]));
    __syncthreads();                     // everybody's pieces; and everybody is done with the other buffer (head k-1)
    if (k + 1 < my_heads) {
      issue_dma(k + 1);
      load_q(qnext, k + 1);
    }
    const uint32_t buf0 = (uint32_t)(uintptr_t)LDS_PTR(smem) + (k & 1) * BUF;
    const uint32_t kb[2] = {buf0 + k_off[0], buf0 + k_off[1]};
    const uint32_t vb[4] = {buf0 + MAT + v_off[0], buf0 + MAT + v_off[1], buf0 + MAT + v_off[2], buf0 + MAT + v_off[3]};
    const int hd = blockIdx.x + k * gridDim.x;
    const int b = hd / H, h = hd % H;
#pragma unroll
    for (int it = 0; it < MAXQ; ++it) {
      const int qt = w + it * NW;
      if (qt < nqt) {
        const int q = qt * 16 + i16;
        // S^T = K q: the K row fragments come through a ring of three register sets, two 16-key tiles per set, requested three sets
        // ahead of the MFMAs that consume them (pinned reads: see lds_issue_row)
        f32x4 s[NKT];
        constexpr int NG = NKT / 2;
        u32x4 kr[3][4];
        auto issue_k = [&](auto gi) {
          constexpr int g = decltype(gi)::value;
          lds_issue_row<(2 * g) * 2048>(kb[0], kr[g % 3][0]), lds_issue_row<(2 * g) * 2048>(kb[1], kr[g % 3][1]);
          lds_issue_row<(2 * g + 1) * 2048>(kb[0], kr[g % 3][2]), lds_issue_row<(2 * g + 1) * 2048>(kb[1], kr[g % 3][3]);
        };
        static_for<(NG < 3 ? NG : 3)>(issue_k);
        static_for<NG>([&](auto gi) {
          constexpr int g = decltype(gi)::value, after = NG - 1 - g < 2 ? NG - 1 - g : 2;
          lds_wait<4 * after>(kr[g % 3][0], kr[g % 3][1], kr[g % 3][2], kr[g % 3][3]);
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            constexpr f32x4 zero = {0.f, 0.f, 0.f, 0.f}, ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            const int t = 2 * g + hf;
            if (NF >= 0 && t > NF) {
              s[t] = ninf;
            } else {
              f32x4 a = (NF >= 0 && t == NF) ? pinit : zero;
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(kr[g % 3][2 * hf]), qcur[it][0], a, 0, 0, 0);
              s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(kr[g % 3][2 * hf + 1]), qcur[it][1], a, 0, 0, 0);
            }
          }
          if constexpr (g + 3 < NG) issue_k(std::integral_constant<int, g + 3>{});
        });
        // the first V fragments are requested before the softmax arithmetic: they do not depend on it
        TrPair vr[3][4];
        auto issue_v = [&](auto ki) {
          constexpr int kp = decltype(ki)::value;
          lds_issue_tr<kp * 4096, 2048>(vb[0], vr[kp % 3][0]), lds_issue_tr<kp * 4096, 2048>(vb[1], vr[kp % 3][1]);
          lds_issue_tr<kp * 4096, 2048>(vb[2], vr[kp % 3][2]), lds_issue_tr<kp * 4096, 2048>(vb[3], vr[kp % 3][3]);
        };
        static_for<(NG < 2 ? NG : 2)>(issue_v);
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          if (NF < 0 && kt * 16 + 16 > N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kt * 16 + 4 * fg + r >= N) s[kt][r] = -INFINITY;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxc = mx * c;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c, -mxc));
        // the row sum is a fifth output tile of the P V product, against a fragment of ones: the loop is VALU-bound (exp, max, the
        // bf16 packing), the matrix pipe is not, and the sum then is the sum of exactly the bf16 probabilities that multiply V
        f32x4 osum = {0.f, 0.f, 0.f, 0.f};
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        static_for<NG>([&](auto ki) {
          constexpr int kp = decltype(ki)::value;
          lds_wait<(kp + 1 < NG) ? 8 : 0>(vr[kp % 3][0], vr[kp % 3][1], vr[kp % 3][2], vr[kp % 3][3]);
          if constexpr (kp + 2 < NG) issue_v(std::integral_constant<int, kp + 2>{});
          const bf16x8 pf = pack8(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[kp % 3][dt].get(), pf, o[dt], 0, 0, 0);
          osum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, osum, 0, 0, 0);
        });
        const float sum = osum[0];        // every row of the ones tile holds the sums of its column = this lane's query
        if (q < N) {
          const float inv = 1.f / sum;
          bf16_t* op = out + ((size_t)b * N + q) * D + h * 64 + 4 * fg;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) store4bf(op + dt * 16, o[dt], inv);
          if (lse && fg == 0) lse[((size_t)b * H + h) * N + q] = mxc * LN2 + __logf(sum);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < MAXQ; ++it)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qcur[it][ks] = qnext[it][ks];
  }
}

// dQ: waves own query tiles; K (row + transposed reads) and V (row reads) in LDS.
template <int NKT>
__device__ __forceinline__ void attn_bwd_dq_body(const int bh, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                 const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                 bf16_t* __restrict__ dqkv, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i16 = lane & 15, fg = lane >> 4;
  const int b = bh / H, h = bh % H;
  const int D = H * 64, ld = 3 * D;
  const bf16_t* base = qkv + (size_t)b * N * ld + h * 64;
  char* Ks = smem;
  char* Vs = smem + NKT * 16 * KV_LD;
  stage_rows(Ks, base + D, ld, N, NKT * 16, tid);
  stage_rows(Vs, base + 2 * D, ld, N, NKT * 16, tid);
  __syncthreads();

  const float c = 0.125f * LOG2E;
  const int nqt = (N + 15) >> 4;
  for (int qt = w; qt < nqt; qt += 4) {
    const int q = qt * 16 + i16;
    const int qc = q < N ? q : N - 1;
    const size_t orow = ((size_t)b * N + qc) * D + h * 64;
    bf16x8 qf[2], dof[2];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 32 + 8 * fg);
      dof[ks] = *(const bf16x8*)(dout + orow + ks * 32 + 8 * fg);
      const bf16x8 of = *(const bf16x8*)(out + orow + ks * 32 + 8 * fg);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += (float)dof[ks][e] * (float)of[e];
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const float lq = lse[((size_t)b * H + h) * N + qc] * LOG2E;

    f32x4 ds[NKT];
    const float lq3 = lq + 3.f;          // p/8 = exp2(s c - lse log2e - 3): the 1/sqrt(64) of dS rides in the exponent
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      ds[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt * 16 < N) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = (kt * 16 + i16) * KV_LD + (ks * 32 + 8 * fg) * 2;
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Ks + off), qf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Vs + off), dof[ks], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[r], c, -lq3)) * (dp[r] - delta);
        if (kt * 16 + 16 > N) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kt * 16 + 4 * fg + r >= N) ds[kt][r] = 0.f;
        }
      }
    }
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < NKT / 2; ++kp) {
      const bf16x8 dsf = pack8(ds[2 * kp], ds[2 * kp + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 ktf = tr_pair(Ks + (32 * kp + 4 * fg + (i16 >> 2)) * KV_LD + (dt * 16 + 4 * (i16 & 3)) * 2);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dq[dt], 0, 0, 0);
      }
    }
    if (q < N) {
      bf16_t* dp_ = dqkv + ((size_t)b * N + q) * ld + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4bf(dp_ + dt * 16, dq[dt], 1.f);
    }
  }
}

// dK, dV: waves own key tiles; Q and dO (row + transposed reads), lse and delta in LDS.
template <int NQT>
__device__ __forceinline__ void attn_bwd_dkv_body(const int bh, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                  const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                  bf16_t* __restrict__ dqkv, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i16 = lane & 15, fg = lane >> 4;
  const int b = bh / H, h = bh % H;
  const int D = H * 64, ld = 3 * D;
  const bf16_t* base = qkv + (size_t)b * N * ld + h * 64;
  const bf16_t* obase = out + (size_t)b * N * D + h * 64;
  const bf16_t* dobase = dout + (size_t)b * N * D + h * 64;
  char* Qs = smem;
  char* dOs = smem + NQT * 16 * KV_LD;
  float* lse_s = (float*)(smem + 2 * NQT * 16 * KV_LD);
  float* dl_s = lse_s + NQT * 16;
  stage_rows(Qs, base, ld, N, NQT * 16, tid);
  stage_rows(dOs, dobase, D, N, NQT * 16, tid);
  __syncthreads();
  if (tid < NQT * 16) {
    float d = 0.f, l = 0.f;
    if (tid < N) {
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        const bf16x8 ov = *(const bf16x8*)(obase + (size_t)tid * D + ch * 8);
        const bf16x8 dv = *(const bf16x8*)(dOs + tid * KV_LD + ch * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)ov[e] * (float)dv[e];
      }
      l = lse[((size_t)b * H + h) * N + tid] * LOG2E;
    }
    dl_s[tid] = d * 0.125f;
    lse_s[tid] = l;
  }
  __syncthreads();

  const float c = 0.125f * LOG2E;
  const int nkt = (N + 15) >> 4;
  for (int kt = w; kt < nkt; kt += 4) {
    const int key = kt * 16 + i16;
    const int kc = key < N ? key : N - 1;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks] = *(const bf16x8*)(base + (size_t)kc * ld + D + ks * 32 + 8 * fg);
      vf[ks] = *(const bf16x8*)(base + (size_t)kc * ld + 2 * D + ks * 32 + 8 * fg);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int qp = 0; qp < NQT / 2; ++qp) {
      f32x4 pp[2], dss[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int qt = 2 * qp + hf;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = (qt * 16 + i16) * KV_LD + (ks * 32 + 8 * fg) * 2;
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Qs + off), kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(dOs + off), vf[ks], dp, 0, 0, 0);
        }
        const f32x4 lq = *(const f32x4*)&lse_s[qt * 16 + 4 * fg];
        const f32x4 dl = *(const f32x4*)&dl_s[qt * 16 + 4 * fg];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c, -lq[r]));
          pp[hf][r] = p;
          dss[hf][r] = p * fmaf(dp[r], 0.125f, -dl[r]);
        }
      }
      const bf16x8 pf = pack8(pp[0], pp[1]);
      const bf16x8 dsf = pack8(dss[0], dss[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int off = (32 * qp + 4 * fg + (i16 >> 2)) * KV_LD + (dt * 16 + 4 * (i16 & 3)) * 2;
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(dOs + off), pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(Qs + off), dsf, dk[dt], 0, 0, 0);
      }
    }
    if (key < N) {
      bf16_t* kp_ = dqkv + ((size_t)b * N + key) * ld + D + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store4bf(kp_ + dt * 16, dk[dt], 1.f);
        store4bf(kp_ + D + dt * 16, dv[dt], 1.f);
      }
    }
  }
}

// Both halves of the backward in ONE launch: 2 B H workgroups, two per CU.  Launched one after the other each half is 768 workgroups
// on 512 slots = 1.5 rounds, i.e. two rounds of which the second is half empty; together they are exactly 3 rounds at DeiT-tiny's
// 768 heads.  Blocks b and b + 8 (the same XCD under round-robin dispatch, so the same L2) are the two halves of one head: its
// q, k, v, dO, O are fetched from HBM once.
template <int NT>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                       const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                       bf16_t* __restrict__ dqkv, int N, int H, int n_heads) {
  const int bid = blockIdx.x;
  const int role = (bid >> 3) & 1;
  const int bh = (bid >> 4) * 8 + (bid & 7);
  if (bh >= n_heads) return;
  if (role == 0) attn_bwd_dq_body<NT>(bh, qkv, out, dout, lse, dqkv, N, H);
  else attn_bwd_dkv_body<NT>(bh, qkv, out, dout, lse, dqkv, N, H);
}

// The whole backward of one head in one workgroup of 8 waves, persistent over heads (one workgroup per CU).  The two-kernel form above
// spends two thirds of its wave cycles waiting: every workgroup stages its operands, then every wave fetches its own q / dO / O (or
// k / v) fragments from global memory tile by tile, and nothing covers those round trips.  Here q, k, v, dO of the head are ALL in LDS
// (4 x NT x 16 rows x 160 B = 140 KB at NT = 14), every MFMA operand is an LDS read, delta = rowsum(dO o O) is formed while the rows
// pass through registers, and the NEXT head's rows are fetched into registers while this head's dK / dV phase runs, so the only global
// latency a CU ever waits for is its first head's.  Phase A (dQ: waves own query tiles) and phase B (dK, dV: waves own key tiles) only
// read LDS, so there is no barrier between them; the second-round tiles of the two phases go to different SIMDs (13 tiles over 8
// waves leaves 5 second-round tiles per phase).
//
// Both phases walk NT / 2 steps of 32 rows with a hand-laid software pipeline over the pinned LDS reads above:
//     release rows(t) -> request transposed(t) -> S, dP MFMAs -> request rows(t + 1) -> exp / dS on the VALU
//                     -> release transposed(t) -> dQ | dK, dV MFMAs
// so each LDS round trip has the MFMA + VALU work of half a step to hide behind.
template <int NT>
__global__ __launch_bounds__(512, 1) void attn_bwd_head_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                            const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16_t* __restrict__ dqkv, int N, int H, int n_heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = NT * 16, REG = ROWS * KV_LD, STEP = 32 * KV_LD;
  static_assert(NT / 2 * STEP + 16 * KV_LD + 128 < 65536 && NT % 2 == 0, "LDS immediates are 16 bit");
  char* Qs = smem;
  char* Ks = Qs + REG;
  char* Vs = Ks + REG;
  char* dOs = Vs + REG;
  float* lse_s = (float*)(dOs + REG);
  float* ndl_s = lse_s + ROWS;                 // -rowsum(dO o O) / 8
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i16 = lane & 15, fg = lane >> 4;
  const int D = H * 64, ld = 3 * D;
  const float c = 0.125f * LOG2E;
  const int nt = (N + 15) >> 4;

  // Register image of the next head.  FI full passes of 512 row chunks (16 B) hold q, k, v, dO, O of the same chunk per thread; the last
  // 32 rows (ROWS * 8 is an odd multiple of 256) are 256 chunks x 5 tensors packed into 3 registers: threads 0-255 hold q and v,
  // threads 256-511 hold k, dO and O of chunk tid - 256, so dO and O of one row chunk always meet in one thread for delta.
  constexpr int FI = ROWS * 8 / 512;
  constexpr bool TAIL = (ROWS * 8) % 512 != 0;
  constexpr int NR = FI * 5 + (TAIL ? 3 : 0);
  bf16x8 reg[NR];
  float rl = 0.f;
  uint32_t oq[FI + 1], oo[FI + 1];               // element offsets of this thread's chunks inside a head's q rows / O rows
  bool ok[FI + 1];
#pragma unroll
  for (int it = 0; it <= FI; ++it) {
    const int idx = it < FI ? tid + 512 * it : 512 * FI + (tid & 255), row = idx >> 3, ch = idx & 7;
    ok[it] = row < N;
    oq[it] = (uint32_t)(row * ld + ch * 8);
    oo[it] = (uint32_t)(row * D + ch * 8);
  }
  auto fetch = [&](int bh) {
    const int b = bh / H, h = bh % H;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * 64;
    const bf16_t* obase = out + (size_t)b * N * D + h * 64;
    const bf16_t* dobase = dout + (size_t)b * N * D + h * 64;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < NR; ++r) reg[r] = z;
#pragma unroll
    for (int it = 0; it < FI; ++it)
      if (ok[it]) {
        reg[5 * it + 0] = *(const bf16x8*)(base + oq[it]);
        reg[5 * it + 1] = *(const bf16x8*)(base + D + oq[it]);
        reg[5 * it + 2] = *(const bf16x8*)(base + 2 * D + oq[it]);
        reg[5 * it + 3] = *(const bf16x8*)(dobase + oo[it]);
        reg[5 * it + 4] = *(const bf16x8*)(obase + oo[it]);
      }
    if constexpr (TAIL) {
      if (ok[FI]) {
        const bool up = tid >= 256;
        reg[5 * FI + 0] = *(const bf16x8*)(base + (up ? D : 0) + oq[FI]);
        reg[5 * FI + 1] = up ? *(const bf16x8*)(dobase + oo[FI]) : *(const bf16x8*)(base + 2 * D + oq[FI]);
        if (up) reg[5 * FI + 2] = *(const bf16x8*)(obase + oo[FI]);
      }
    }
    rl = tid < N ? lse[((size_t)b * H + h) * N + tid] * LOG2E : 0.f;
  };
  auto rowdot = [&](const bf16x8& a, const bf16x8& b_) {      // partial rowsum(dO o O) of one chunk, summed over the row's 8 chunk lanes
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += (float)a[e] * (float)b_[e];
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    return d * -0.125f;
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < FI; ++it) {
      const int idx = tid + 512 * it, row = idx >> 3, ch = idx & 7, off = row * KV_LD + ch * 16;
      const float d = rowdot(reg[5 * it + 3], reg[5 * it + 4]);
      *(bf16x8*)(Qs + off) = reg[5 * it + 0];
      *(bf16x8*)(Ks + off) = reg[5 * it + 1];
      *(bf16x8*)(Vs + off) = reg[5 * it + 2];
      *(bf16x8*)(dOs + off) = reg[5 * it + 3];
      if (ch == 0) ndl_s[row] = d;
    }
    if constexpr (TAIL) {
      const int idx = 512 * FI + (tid & 255), row = idx >> 3, ch = idx & 7, off = row * KV_LD + ch * 16;
      const bool up = tid >= 256;
      const float d = rowdot(reg[5 * FI + 1], reg[5 * FI + 2]);
      *(bf16x8*)((up ? Ks : Qs) + off) = reg[5 * FI + 0];
      *(bf16x8*)((up ? dOs : Vs) + off) = reg[5 * FI + 1];
      if (up && ch == 0) ndl_s[row] = d;
    }
    if (tid < ROWS) lse_s[tid] = rl;
  };

  // LDS byte addresses of this lane's fragments: one VGPR per region, everything else is an instruction immediate
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const uint32_t rowoff = lds0 + i16 * KV_LD + fg * 16;                              // + 16 tile KV_LD + 64 ks
  const uint32_t troff = lds0 + (4 * fg + (i16 >> 2)) * KV_LD + (i16 & 3) * 8;       // + 32 step KV_LD + 32 dt (+ 16 rows in the pair)
  const uint32_t q_row = rowoff, k_row = rowoff + REG, v_row = rowoff + 2 * REG, do_row = rowoff + 3 * REG;
  const uint32_t q_tr = troff, k_tr = troff + REG, do_tr = troff + 3 * REG;
  const uint32_t st_a = lds0 + 4 * REG + 16 * fg;                                    // lse_s[4 fg ..]; ndl_s is ROWS * 4 bytes further

  int bh = blockIdx.x;
  if (bh >= n_heads) return;
  fetch(bh);
  for (; bh < n_heads; bh += gridDim.x) {
    const int b = bh / H, h = bh % H;
    stage();
    __syncthreads();

    // ---- phase A: dQ.  s[r] = S^T[key 4 fg + r][query i16].  Padded keys need no mask: their K rows are zero in LDS, so whatever
    // (finite) dS they get multiplies zeros in dQ.
    for (int qt = w; qt < nt; qt += 8) {
      const int qrow = qt * 16 + i16;
      bf16x8 qf[2], dof[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = *(const bf16x8*)(Qs + qrow * KV_LD + fg * 16 + ks * 64);
        dof[ks] = *(const bf16x8*)(dOs + qrow * KV_LD + fg * 16 + ks * 64);
      }
      const float nd8 = ndl_s[qrow] * 8.f;          // -delta
      const float lq3 = lse_s[qrow] + 3.f;          // p/8 = exp2(s c - lse log2e - 3): the 1/sqrt(64) of dS rides in the exponent
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing of the compiler's is in flight when the pinned reads start
      f32x4 dq[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x4 ka[2][2], va[2][2];                      // [16-key tile of the step][k half]
      lds_issue_row<0>(k_row, ka[0][0]), lds_issue_row<64>(k_row, ka[0][1]);
      lds_issue_row<16 * KV_LD>(k_row, ka[1][0]), lds_issue_row<16 * KV_LD + 64>(k_row, ka[1][1]);
      lds_issue_row<0>(v_row, va[0][0]), lds_issue_row<64>(v_row, va[0][1]);
      lds_issue_row<16 * KV_LD>(v_row, va[1][0]), lds_issue_row<16 * KV_LD + 64>(v_row, va[1][1]);
      static_for<NT / 2>([&](auto step) {
        constexpr int t = decltype(step)::value, O = t * STEP;
        lds_wait<4>(ka[0][0], ka[0][1], ka[1][0], ka[1][1]);
        lds_wait<0>(va[0][0], va[0][1], va[1][0], va[1][1]);
        TrPair kt_[4];
        lds_issue_tr<O>(k_tr, kt_[0]), lds_issue_tr<O + 32>(k_tr, kt_[1]), lds_issue_tr<O + 64>(k_tr, kt_[2]), lds_issue_tr<O + 96>(k_tr, kt_[3]);
        f32x4 s[2], dp[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          s[hf] = f32x4{0.f, 0.f, 0.f, 0.f}, dp[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            s[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(ka[hf][ks]), qf[ks], s[hf], 0, 0, 0);
            dp[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(va[hf][ks]), dof[ks], dp[hf], 0, 0, 0);
          }
        }
        if constexpr (t + 1 < NT / 2) {
          constexpr int P = O + STEP;
          lds_issue_row<P>(k_row, ka[0][0]), lds_issue_row<P + 64>(k_row, ka[0][1]);
          lds_issue_row<P + 16 * KV_LD>(k_row, ka[1][0]), lds_issue_row<P + 16 * KV_LD + 64>(k_row, ka[1][1]);
          lds_issue_row<P>(v_row, va[0][0]), lds_issue_row<P + 64>(v_row, va[0][1]);
          lds_issue_row<P + 16 * KV_LD>(v_row, va[1][0]), lds_issue_row<P + 16 * KV_LD + 64>(v_row, va[1][1]);
        }
        f32x4 ds[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[hf][r] = __builtin_amdgcn_exp2f(fmaf(s[hf][r], c, -lq3)) * (dp[hf][r] + nd8);
        const bf16x8 dsf = pack8(ds[0], ds[1]);
        lds_wait<(t + 1 < NT / 2) ? 8 : 0>(kt_[0], kt_[1], kt_[2], kt_[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_[dt].get(), dsf, dq[dt], 0, 0, 0);
      });
      if (qrow < N) {
        bf16_t* dp_ = dqkv + ((size_t)b * N + qrow) * ld + h * 64 + 4 * fg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4bf(dp_ + dt * 16, dq[dt], 1.f);
      }
    }

    const int nxt = bh + gridDim.x;
    if (nxt < n_heads) fetch(nxt);                  // lands during phase B

    // ---- phase B: dK, dV.  s[r] = S[query 4 fg + r][key i16]
    for (int rnd = 0; rnd * 8 < nt; ++rnd) {
      const int kt = rnd * 8 + ((w - 5 * rnd) & 7);
      if (kt >= nt) continue;
      const int krow = kt * 16 + i16;
      bf16x8 kf[2], vf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = *(const bf16x8*)(Ks + krow * KV_LD + fg * 16 + ks * 64);
        vf[ks] = *(const bf16x8*)(Vs + krow * KV_LD + fg * 16 + ks * 64);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f32x4 dk[4], dv[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      u32x4 qa[2][2], da[2][2], lq[2], nl[2];        // [16-query tile of the step][k half]; row statistics of the two tiles
      lds_issue_row<0>(q_row, qa[0][0]), lds_issue_row<64>(q_row, qa[0][1]);
      lds_issue_row<16 * KV_LD>(q_row, qa[1][0]), lds_issue_row<16 * KV_LD + 64>(q_row, qa[1][1]);
      lds_issue_row<0>(do_row, da[0][0]), lds_issue_row<64>(do_row, da[0][1]);
      lds_issue_row<16 * KV_LD>(do_row, da[1][0]), lds_issue_row<16 * KV_LD + 64>(do_row, da[1][1]);
      lds_issue_row<0>(st_a, lq[0]), lds_issue_row<64>(st_a, lq[1]);
      lds_issue_row<ROWS * 4>(st_a, nl[0]), lds_issue_row<ROWS * 4 + 64>(st_a, nl[1]);
      static_for<NT / 2>([&](auto step) {
        constexpr int t = decltype(step)::value, O = t * STEP;
        lds_wait<8>(qa[0][0], qa[0][1], qa[1][0], qa[1][1]);
        lds_wait<4>(da[0][0], da[0][1], da[1][0], da[1][1]);
        TrPair td[4], tq[4];
        lds_issue_tr<O>(do_tr, td[0]), lds_issue_tr<O + 32>(do_tr, td[1]), lds_issue_tr<O + 64>(do_tr, td[2]), lds_issue_tr<O + 96>(do_tr, td[3]);
        lds_issue_tr<O>(q_tr, tq[0]), lds_issue_tr<O + 32>(q_tr, tq[1]), lds_issue_tr<O + 64>(q_tr, tq[2]), lds_issue_tr<O + 96>(q_tr, tq[3]);
        f32x4 s[2], dp[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          s[hf] = f32x4{0.f, 0.f, 0.f, 0.f}, dp[hf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            s[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(qa[hf][ks]), kf[ks], s[hf], 0, 0, 0);
            dp[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(da[hf][ks]), vf[ks], dp[hf], 0, 0, 0);
          }
        }
        lds_wait<16>(lq[0], lq[1], nl[0], nl[1]);    // the statistics were requested before the 16 transposed reads of this step
        f32x4 pp[2], dss[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x4 l = as_f4(lq[hf]), n = as_f4(nl[hf]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[hf][r], c, -l[r]));
            pp[hf][r] = p;
            dss[hf][r] = p * fmaf(dp[hf][r], 0.125f, n[r]);
          }
        }
        const bf16x8 pf = pack8(pp[0], pp[1]);
        const bf16x8 dsf = pack8(dss[0], dss[1]);
        if constexpr (t + 1 < NT / 2) {
          constexpr int P = O + STEP, S4 = (t + 1) * 128;
          lds_issue_row<P>(q_row, qa[0][0]), lds_issue_row<P + 64>(q_row, qa[0][1]);
          lds_issue_row<P + 16 * KV_LD>(q_row, qa[1][0]), lds_issue_row<P + 16 * KV_LD + 64>(q_row, qa[1][1]);
          lds_issue_row<P>(do_row, da[0][0]), lds_issue_row<P + 64>(do_row, da[0][1]);
          lds_issue_row<P + 16 * KV_LD>(do_row, da[1][0]), lds_issue_row<P + 16 * KV_LD + 64>(do_row, da[1][1]);
          lds_issue_row<S4>(st_a, lq[0]), lds_issue_row<S4 + 64>(st_a, lq[1]);
          lds_issue_row<ROWS * 4 + S4>(st_a, nl[0]), lds_issue_row<ROWS * 4 + S4 + 64>(st_a, nl[1]);
        }
        constexpr int LATER = (t + 1 < NT / 2) ? 12 : 0;
        lds_wait<LATER + 8>(td[0], td[1], td[2], td[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(td[dt].get(), pf, dv[dt], 0, 0, 0);
        lds_wait<LATER>(tq[0], tq[1], tq[2], tq[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq[dt].get(), dsf, dk[dt], 0, 0, 0);
      });
      if (krow < N) {
        bf16_t* kp_ = dqkv + ((size_t)b * N + krow) * ld + D + h * 64 + 4 * fg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          store4bf(kp_ + dt * 16, dk[dt], 1.f);
          store4bf(kp_ + D + dt * 16, dv[dt], 1.f);
        }
      }
    }
    __syncthreads();
  }
}

template <typename K>
int set_smem(K kernel, int bytes) {
  if (bytes <= 65536) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    dkd_set_error("attention: cannot raise dynamic LDS to %d bytes: %s", bytes, hipGetErrorString(e));
    return DKD_ERR_HIP;
  }
  return 0;
}

#define DISPATCH_NT(NT, CALL)                         \
  switch (NT) {                                       \
    case 2: { constexpr int T = 2; CALL; } break;     \
    case 4: { constexpr int T = 4; CALL; } break;     \
    case 8: { constexpr int T = 8; CALL; } break;     \
    case 14: { constexpr int T = 14; CALL; } break;   \
    default: { constexpr int T = 16; CALL; } break;   \
  }

int pick_tiles(int N) {
  const int need = (N + 15) / 16;
  const int opts[5] = {2, 4, 8, 14, 16};
  for (int i = 0; i < 5; ++i)
    if (opts[i] >= need) return opts[i];
  return -1;
}

}  // namespace

extern "C" int dkd_attn_fwd(const void* qkv, void* out, float* lse, int32_t B, int32_t N, int32_t H, void* stream) {
  DKD_CHECK_ARG(qkv && out, "attn_fwd: null operand");
  DKD_CHECK_ARG(B > 0 && H > 0 && N > 0 && N <= 256, "attn_fwd: need 0 < N <= 256 (got N=%d B=%d H=%d)", N, B, H);
  const int nt = pick_tiles(N);
  const int smem = 2 * nt * 16 * KV_LD;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      dkd_set_error("attn_fwd: cannot query the device");
      return DKD_ERR_HIP;
    }
    n_cu = prop.multiProcessorCount;
    if (const char* lim = getenv("DKD_CU_LIMIT")) {      // (dev: see gemm.hip)
      const int v = atoi(lim);
      if (v >= 8 && v < n_cu) n_cu = v;
    }
  }
  if (nt >= 8 && B * H >= 2 * n_cu) {    // enough heads per CU for the double-buffered persistent kernel to pay
    const int smem_ring = 2 * 2 * nt * 16 * 128;
    // the padding mask rides in the MFMA accumulator when N sits in the last two key tiles of the instantiation (197 / 198 in 14)
    const int nf = N / 16 >= nt ? nt - 1 : N / 16;
#define RING_LAUNCH(NF_)                                                                                                             \
  {                                                                                                                                  \
    if (int rc = set_smem(attn_fwd_ring_kernel<T, NF_>, smem_ring)) return rc;                                                       \
    hipLaunchKernelGGL((attn_fwd_ring_kernel<T, NF_>), dim3(n_cu), dim3(512), smem_ring, as_stream(stream), (const bf16_t*)qkv,      \
                       (bf16_t*)out, lse, N, H, B * H);                                                                              \
  }
    DISPATCH_NT(nt, {
      if (nf == T - 1) RING_LAUNCH(T - 1)
      else if (nf == T - 2) RING_LAUNCH(T - 2)
      else RING_LAUNCH(-1)
    });
#undef RING_LAUNCH
    DKD_CHECK_LAUNCH("attn_fwd_ring");
    return DKD_OK;
  }
  DISPATCH_NT(nt, {
    if (int rc = set_smem(attn_fwd_kernel<T>, smem)) return rc;
    hipLaunchKernelGGL(attn_fwd_kernel<T>, dim3(B * H), dim3(64 * FWD_WAVES), smem, as_stream(stream), (const bf16_t*)qkv, (bf16_t*)out, lse, N, H);
  });
  DKD_CHECK_LAUNCH("attn_fwd");
  return DKD_OK;
}

extern "C" int dkd_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int32_t B, int32_t N,
                            int32_t H, void* stream) {
  DKD_CHECK_ARG(qkv && out && dout && lse && dqkv, "attn_bwd: null operand");
  DKD_CHECK_ARG(B > 0 && H > 0 && N > 0 && N <= 256, "attn_bwd: need 0 < N <= 256 (got N=%d)", N);
  const int nt = pick_tiles(N);
  const int smem1 = 2 * nt * 16 * KV_LD;
  const int smem2 = smem1 + 2 * nt * 16 * 4;
  const int nbh = B * H;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      dkd_set_error("attn_bwd: cannot query the device");
      return DKD_ERR_HIP;
    }
    n_cu = prop.multiProcessorCount;
    if (const char* lim = getenv("DKD_CU_LIMIT")) {      // (dev: see gemm.hip)
      const int v = atoi(lim);
      if (v >= 8 && v < n_cu) n_cu = v;
    }
  }
  static const bool no_head = getenv("DKD_ATTN_BWD_SPLIT") != nullptr;
  if ((nt == 8 || nt == 14) && nbh >= n_cu && !no_head) {     // one persistent workgroup per CU, the whole head in LDS
    const int smem = 4 * nt * 16 * KV_LD + 2 * nt * 16 * 4;
    DISPATCH_NT(nt, {
      if constexpr (T == 8 || T == 14) {
        if (int rc = set_smem(attn_bwd_head_kernel<T>, smem)) return rc;
        hipLaunchKernelGGL(attn_bwd_head_kernel<T>, dim3(n_cu), dim3(512), smem, as_stream(stream), (const bf16_t*)qkv, (const bf16_t*)out,
                           (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, nbh);
      }
    });
    DKD_CHECK_LAUNCH("attn_bwd_head");
    return DKD_OK;
  }
  const int grid = ((nbh + 7) / 8) * 16;           // groups of 16 blocks: 8 heads x {dQ half, dK/dV half}
  DISPATCH_NT(nt, {
    if (int rc = set_smem(attn_bwd_kernel<T>, smem2)) return rc;
    hipLaunchKernelGGL(attn_bwd_kernel<T>, dim3(grid), dim3(256), smem2, as_stream(stream), (const bf16_t*)qkv, (const bf16_t*)out,
                       (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, nbh);
  });
  DKD_CHECK_LAUNCH("attn_bwd");
  return DKD_OK;
}

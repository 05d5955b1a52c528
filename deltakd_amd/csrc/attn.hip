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
template <int NKT>
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
    const char* Ks = smem + (k & 1) * BUF;
    const char* Vs = Ks + MAT;
    const int hd = blockIdx.x + k * gridDim.x;
    const int b = hd / H, h = hd % H;
#pragma unroll
    for (int it = 0; it < MAXQ; ++it) {
      const int qt = w + it * NW;
      if (qt < nqt) {
        const int q = qt * 16 + i16;
        f32x4 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(Ks + k_off[ks] + kt * 2048);
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qcur[it][ks], s[kt], 0, 0, 0);
          }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          if (kt * 16 + 16 > N) {
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
            const float pr = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c, -mxc));
            s[kt][r] = pr;
            sum += pr;
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
            const char* pv = Vs + v_off[dt] + kp * 4096;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(pv));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(pv + 16 * 128));
            const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
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
  }
  if (nt >= 8 && B * H >= 2 * n_cu) {    // enough heads per CU for the double-buffered persistent kernel to pay
    const int smem_ring = 2 * 2 * nt * 16 * 128;
    DISPATCH_NT(nt, {
      if (int rc = set_smem(attn_fwd_ring_kernel<T>, smem_ring)) return rc;
      hipLaunchKernelGGL(attn_fwd_ring_kernel<T>, dim3(n_cu), dim3(512), smem_ring, as_stream(stream), (const bf16_t*)qkv, (bf16_t*)out, lse, N,
                         H, B * H);
    });
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
  const int grid = ((nbh + 7) / 8) * 16;           // groups of 16 blocks: 8 heads x {dQ half, dK/dV half}
  DISPATCH_NT(nt, {
    if (int rc = set_smem(attn_bwd_kernel<T>, smem2)) return rc;
    hipLaunchKernelGGL(attn_bwd_kernel<T>, dim3(grid), dim3(256), smem2, as_stream(stream), (const bf16_t*)qkv, (const bf16_t*)out,
                       (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, nbh);
  });
  DKD_CHECK_LAUNCH("attn_bwd");
  return DKD_OK;
}

// bf16 MFMA GEMMs for the ViT forward/backward (gfx950).
//
//  gemm_nt : C[M,N] = epi(A[M,K] * B[N,K]^T)   -- every nn.Linear forward and every dgrad (with W^T shadow copies)
//  gemm_tn : C[N1,N2] += A[M,N1]^T * B[M,N2]   -- every weight gradient (split over M, f32 atomics)
//
// Design (MI355X_MICROARCH / cdna_hip_programming sections 3, 5):
//  * 128 x BN x 64 tiles, 4 waves (2x2) -- and a 256 x 256 x 64 / 8-wave tile for the wide teacher GEMMs --
//    v_mfma_f32_16x16x32_bf16, fp32 accumulators in registers.
//  * NT operands are both K-contiguous: tiles go HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR
//    round trip); the LDS image is lane-linear so the bank-conflict swizzle (16-B slot ^= (row>>1)&7) is applied to the
//    per-lane SOURCE address and again on the ds_read_b128 fragment reads.
//  * TN operands are both K(=m)-strided: tiles are register-staged into 288-B-stride rows and the MFMA fragments are
//    read with ds_read_b64_tr_b16 (hardware transpose), so no transposed activation copy ever exists in HBM.
//  * epilogue goes through LDS so that bias / GELU / residual / DropPath-scale / feature-tap traffic is 16-B coalesced.
//  * blockIdx -> tile map is XCD-aware (each XCD's L2 sees a contiguous run of tiles sharing A panels).
#include <mutex>
#include <vector>
#include "common.h"

namespace {

constexpr int BM = 128, BK = 64;
constexpr int CS_LD = 132;  // f32 staging row stride (floats) of the wgrad kernel's atomic epilogue

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective "each XCD gets a contiguous chunk" remap (guide T1); bid % 8 labels the XCD group.
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ float epi_scalar(const DkdGemm& g, float v, int m, int n) {
  if (g.epi & DKD_EPI_BIAS) v += g.bias[n];
  if (g.epi & DKD_EPI_GELU) {
    if (g.preact) ((bf16_t*)g.preact)[(size_t)m * g.ldp + n] = f2bf(v);
    v = gelu_erf(v);
  }
  if (g.epi & DKD_EPI_DGELU) v *= dgelu_erf(bf2f(((const bf16_t*)g.preact)[(size_t)m * g.ldp + n]));
  if (g.epi & DKD_EPI_RELU) v = fmaxf(v, 0.f);
  if (g.tap) {
    if (g.epi & DKD_EPI_TAP_F32) ((float*)g.tap)[(size_t)m * g.ldt + n] = v;
    else ((bf16_t*)g.tap)[(size_t)m * g.ldt + n] = f2bf(v);
  }
  if (g.epi & DKD_EPI_RESID) {
    const float sc = g.rowscale ? g.rowscale[m / g.rows_per_sample] : 1.f;
    v = g.resid[(size_t)map_row(g.rmap, m) * g.ldr + n] + sc * v;
  }
  return v;
}

// Epilogue operands that come from memory (residual row, GELU pre-activation) are fetched for a whole batch of output
// vectors BEFORE any of them is stored: C may alias resid (in-place residual), so the compiler cannot hoist a load above an
// earlier store and a load->store->load chain would expose one full memory latency per vector.
// Each lane owns 8 consecutive columns of one row: every bf16 access is a full 16-B dwordx4 (store-issue bound otherwise).
typedef __attribute__((ext_vector_type(8))) float f32x8;
struct EpiIn {
  f32x4 r0, r1;
  uint4 pre;
};
__device__ __forceinline__ f32x8 unpack8(const uint4 p) {
  return f32x8{__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u), __uint_as_float(p.y << 16), __uint_as_float(p.y & 0xffff0000u),
               __uint_as_float(p.z << 16), __uint_as_float(p.z & 0xffff0000u), __uint_as_float(p.w << 16), __uint_as_float(p.w & 0xffff0000u)};
}
__device__ __forceinline__ uint4 pack8(const f32x8& v) {
  return uint4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
}
__device__ __forceinline__ EpiIn epi_prefetch(const DkdGemm& g, const int vec_ok, const int m, const int n) {
  EpiIn in;
  in.r0 = in.r1 = f32x4{0.f, 0.f, 0.f, 0.f};
  in.pre = uint4{0u, 0u, 0u, 0u};
  if (vec_ok) {
    if (g.epi & DKD_EPI_RESID) {
      const float* rp = &g.resid[(size_t)map_row(g.rmap, m) * g.ldr + n];
      in.r0 = *(const f32x4*)rp;
      in.r1 = *(const f32x4*)(rp + 4);
    }
    if (g.epi & DKD_EPI_DGELU) in.pre = *(const uint4*)&((const bf16_t*)g.preact)[(size_t)m * g.ldp + n];
  }
  return in;
}
__device__ __forceinline__ void epi_finish(const DkdGemm& g, const int vec_ok, f32x8 v, const EpiIn& in, const int m, const int n) {
  const bool out_f32 = g.epi & DKD_EPI_OUT_F32;
  const size_t crow = (size_t)map_row(g.cmap, m) * g.ldc;
  if (vec_ok) {
    if (g.epi & DKD_EPI_BIAS) {
      const f32x4 b0 = *(const f32x4*)&g.bias[n], b1 = *(const f32x4*)&g.bias[n + 4];
      v += f32x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }
    if (g.epi & DKD_EPI_GELU) {
      if (g.preact) *(uint4*)&((bf16_t*)g.preact)[(size_t)m * g.ldp + n] = pack8(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
    }
    if (g.epi & DKD_EPI_DGELU) {
      const f32x8 p = unpack8(in.pre);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= dgelu_erf(p[e]);
    }
    if (g.epi & DKD_EPI_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (g.tap) {
      if (g.epi & DKD_EPI_TAP_F32) {
        float* tp = &((float*)g.tap)[(size_t)m * g.ldt + n];
        *(f32x4*)tp = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(tp + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        *(uint4*)&((bf16_t*)g.tap)[(size_t)m * g.ldt + n] = pack8(v);
      }
    }
    if (g.epi & DKD_EPI_RESID) {
      const float sc = g.rowscale ? g.rowscale[m / g.rows_per_sample] : 1.f;
      v = f32x8{in.r0[0], in.r0[1], in.r0[2], in.r0[3], in.r1[0], in.r1[1], in.r1[2], in.r1[3]} + sc * v;
    }
    if (out_f32) {
      float* cp = (float*)g.C + crow + n;
      f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
      if (g.epi & DKD_EPI_ACCUM) {
        lo += *(const f32x4*)cp;
        hi += *(const f32x4*)(cp + 4);
      }
      *(f32x4*)cp = lo;
      *(f32x4*)(cp + 4) = hi;
    } else {
      *(uint4*)((bf16_t*)g.C + crow + n) = pack8(v);
    }
  } else {
    for (int e = 0; e < 8 && n + e < g.N; ++e) {
      float x = epi_scalar(g, v[e], m, n + e);
      if (out_f32) {
        float* cp = (float*)g.C + crow + n + e;
        if (g.epi & DKD_EPI_ACCUM) x += *cp;
        *cp = x;
      } else {
        ((bf16_t*)g.C)[crow + n + e] = f2bf(x);
      }
    }
  }
}

// ---- epilogue shared by the 128-row NT kernels: ONE pass through LDS (f32 [128][BN], unpadded: the accumulator-layout
// ds_write_b32 is only 2-way per 32-lane group = free, the row reads are contiguous), then 16-B coalesced fused stores.
template <int BN, int NJ>
__device__ __forceinline__ void nt_epilogue(const DkdGemm& g, const int vec_ok, char* smem, f32x4 (&acc)[4][NJ], const int m0,
                                            const int n0, const int tid, const int wr, const int wc) {
  const int lane = tid & 63, frow = lane & 15, fg = lane >> 4;
  float* cs = (float*)smem;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[(wr * 64 + i * 16 + fg * 4 + r) * BN + wc * (BN / 2) + j * 16 + frow] = acc[i][j][r];
  __syncthreads();
  constexpr int TPR = BN / 8;            // threads per output row (8 columns each)
  constexpr int RPP = 256 / TPR;         // rows per sweep of the 256 threads
  const int col8 = (tid % TPR) * 8, r0 = tid / TPR;
  const int n = n0 + col8;
  if (n >= g.N) return;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    constexpr int SW = 64 / RPP;         // sweeps per 64-row half
    EpiIn in[SW];
#pragma unroll
    for (int s = 0; s < SW; ++s) {
      const int m = m0 + half * 64 + r0 + RPP * s;
      if (m < g.M) in[s] = epi_prefetch(g, vec_ok, m, n);
    }
#pragma unroll
    for (int s = 0; s < SW; ++s) {
      const int rl = half * 64 + r0 + RPP * s;
      const int m = m0 + rl;
      if (m < g.M) {
        const f32x4 lo = *(const f32x4*)&cs[rl * BN + col8], hi = *(const f32x4*)&cs[rl * BN + col8 + 4];
        epi_finish(g, vec_ok, f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, in[s], m, n);
      }
    }
  }
}

template <int BN>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const DkdGemm g, const int vec_ok) {
  constexpr int NJ = BN / 32;             // 16-col MFMA tiles per wave along N
  constexpr int A_BYTES = BM * 128;       // 16 KiB
  constexpr int B_BYTES = BN * 128;
  constexpr int BUF = A_BYTES + B_BYTES;
  constexpr int SMEM = (2 * BUF) > (128 * BN * 4) ? (2 * BUF) : (128 * BN * 4);
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (L / tiles_n) * BM, n0 = (L % tiles_n) * BN;
  const int KT = g.K / BK;

  // per-lane source rows for the LDS-DMA staging: 1 KiB chunk = 8 rows x 128 B; lane -> (row = lane>>3, slot = lane&7)
  const bf16_t* arow[4];
  int aslot[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = w * 32 + c * 8 + (lane >> 3);
    int m = m0 + r;
    m = m < g.M ? m : g.M - 1;
    arow[c] = (const bf16_t*)g.A + (size_t)map_row(g.amap, m) * g.lda;
    aslot[c] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }
  constexpr int BCH = BN / 32;  // B chunks per wave (BN rows / 8 rows per chunk / 4 waves)
  const bf16_t* brow[BCH];
  int bslot[BCH];
#pragma unroll
  for (int c = 0; c < BCH; ++c) {
    const int r = w * (BN / 4) + c * 8 + (lane >> 3);
    int n = n0 + r;
    n = n < g.N ? n : g.N - 1;
    brow[c] = (const bf16_t*)g.B + (size_t)n * g.ldb;
    bslot[c] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }

  auto stage = [&](int kt, int buf) {
    char* abase = smem + buf * BUF;
    char* bbase = abase + A_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      __builtin_amdgcn_global_load_lds(GLB_PTR(arow[c] + k0 + aslot[c]), LDS_PTR(abase + (w * 32 + c * 8) * 128), 16, 0, 0);
#pragma unroll
    for (int c = 0; c < BCH; ++c)
      __builtin_amdgcn_global_load_lds(GLB_PTR(brow[c] + k0 + bslot[c]), LDS_PTR(bbase + (w * (BN / 4) + c * 8) * 128), 16, 0, 0);
  };

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fg = lane >> 4, fswz = (frow >> 1) & 7;

  stage(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) stage(kt + 1, (kt + 1) & 1);
    const char* abase = smem + (kt & 1) * BUF;
    const char* bbase = abase + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ps = ((kk * 4 + fg) ^ fswz) * 16;
      bf16x8 a[4], b[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(abase + (wr * 64 + i * 16 + frow) * 128 + ps);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = *(const bf16x8*)(bbase + (wc * (BN / 2) + j * 16 + frow) * 128 + ps);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  nt_epilogue<BN, NJ>(g, vec_ok, smem, acc, m0, n0, tid, wr, wc);
}

// ---- 256 x 256 x 64 tile, 8 waves (2 x 4, 128 x 64 per wave), one workgroup per CU: for the wide teacher GEMMs.
// The 128^2 kernel moves 1 B of operand from L2 into LDS per 64 FLOP and its K loop is bound by that path (main loop alone:
// ~950 TF/s, no gain from deeper pipelines: profiles/r01_*); this tile halves the L2->LDS bytes and the LDS-DMA / ds_read
// instructions per MFMA.
template <int H>
__device__ __forceinline__ void epi_pass256(const DkdGemm& g, const int vec_ok, float* cs, f32x4 (&acc)[8][4], const int m0, const int n0,
                                            const int tid, const int wr, const int wc) {
  const int lane = tid & 63, frow = lane & 15, fg = lane >> 4;
  __syncthreads();
  if (wr == (H >> 1)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[(i * 16 + fg * 4 + r) * 256 + wc * 64 + j * 16 + frow] = acc[(H & 1) * 4 + i][j][r];
  }
  __syncthreads();
  const int col8 = (tid & 31) * 8, r0 = tid >> 5;      // 32 threads per 256-column row, 16 rows per sweep
  const int n = n0 + col8;
  if (n >= g.N) return;
  EpiIn in[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int m = m0 + H * 64 + r0 + 16 * s;
    if (m < g.M) in[s] = epi_prefetch(g, vec_ok, m, n);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int rl = r0 + 16 * s;
    const int m = m0 + H * 64 + rl;
    if (m < g.M) {
      const f32x4 lo = *(const f32x4*)&cs[rl * 256 + col8], hi = *(const f32x4*)&cs[rl * 256 + col8 + 4];
      epi_finish(g, vec_ok, f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, in[s], m, n);
    }
  }
}

__global__ __launch_bounds__(512, 1) void gemm_nt256_kernel(const DkdGemm g, const int vec_ok) {
  constexpr int TILE = 256 * 128;      // 32 KiB per operand tile (256 rows x 64 bf16)
  constexpr int BUF = 2 * TILE;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];   // 128 KiB; the epilogue reuses 64 KiB of it
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int tiles_n = (g.N + 255) / 256;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (L / tiles_n) * 256, n0 = (L % tiles_n) * 256;
  const int KT = g.K / 64;

  const bf16_t* arow[4];
  const bf16_t* brow[4];
  int slot[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = w * 32 + c * 8 + (lane >> 3);
    int m = m0 + r, n = n0 + r;
    m = m < g.M ? m : g.M - 1;
    n = n < g.N ? n : g.N - 1;
    arow[c] = (const bf16_t*)g.A + (size_t)map_row(g.amap, m) * g.lda;
    brow[c] = (const bf16_t*)g.B + (size_t)n * g.ldb;
    slot[c] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }
  auto stage = [&](int kt) {
    char* abase = smem + (kt & 1) * BUF;
    char* bbase = abase + TILE;
    const int k0 = kt * 64;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      __builtin_amdgcn_global_load_lds(GLB_PTR(arow[c] + k0 + slot[c]), LDS_PTR(abase + (w * 32 + c * 8) * 128), 16, 0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      __builtin_amdgcn_global_load_lds(GLB_PTR(brow[c] + k0 + slot[c]), LDS_PTR(bbase + (w * 32 + c * 8) * 128), 16, 0, 0);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fg = lane >> 4, fswz = (frow >> 1) & 7;

  stage(0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < KT) stage(kt + 1);
    const char* abase = smem + (kt & 1) * BUF;
    const char* bbase = abase + TILE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ps = ((kk * 4 + fg) ^ fswz) * 16;
      bf16x8 a[8], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(bbase + (wc * 64 + j * 16 + frow) * 128 + ps);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(abase + (wr * 128 + i * 16 + frow) * 128 + ps);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }
  float* cs = (float*)smem;
  epi_pass256<0>(g, vec_ok, cs, acc, m0, n0, tid, wr, wc);
  epi_pass256<1>(g, vec_ok, cs, acc, m0, n0, tid, wr, wc);
  epi_pass256<2>(g, vec_ok, cs, acc, m0, n0, tid, wr, wc);
  epi_pass256<3>(g, vec_ok, cs, acc, m0, n0, tid, wr, wc);
}

// ------------------------------------------------------------------------------------------------ TN (wgrad)
constexpr int TN_LD = 288;                 // bytes per LDS row (256 B of data + 32 B pad: tr reads conflict-free)
constexpr int TN_TILE = 64 * TN_LD;        // 18 KiB per operand tile
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* C,
                                                         int M, int N1, int N2, int lda, int ldb, int ldc, DkdRowMap amap,
                                                         DkdRowMap bmap, int kt_per_split, float* __restrict__ a_colsum) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TN_TILE];  // [buf][A|B] ; reused by the epilogue (33 KiB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int tiles2 = (N2 + 127) / 128;
  const int t1 = blockIdx.x / tiles2, t2 = blockIdx.x % tiles2;
  const int n1_0 = t1 * 128, n2_0 = t2 * 128;
  const int KT_all = (M + 63) / 64;
  const int kt_begin = blockIdx.y * kt_per_split;
  const int kt_end = min(KT_all, kt_begin + kt_per_split);
  if (kt_begin >= kt_end) return;

  const int lrow = tid >> 4, lcol = (tid & 15) * 8;  // staging: thread -> (row lrow + 16 c, 8 elements at lcol)
  s16x8 ra[4], rb[4];
  // fused bias gradient: column sums of the A operand (dY) ride along in the blocks of the first N2 tile
  const bool do_colsum = a_colsum != nullptr && t2 == 0;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int kt) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int m = kt * 64 + lrow + 16 * c;
      s16x8 va = {0, 0, 0, 0, 0, 0, 0, 0}, vb = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < M) {
        const bf16_t* pa = A + (size_t)map_row(amap, m) * lda + n1_0 + lcol;
        const bf16_t* pb = B + (size_t)map_row(bmap, m) * ldb + n2_0 + lcol;
        if (n1_0 + lcol + 8 <= N1) va = *(const s16x8*)pa;
        else
          for (int e = 0; e < 8; ++e)
            if (n1_0 + lcol + e < N1) va[e] = (short)pa[e];
        if (n2_0 + lcol + 8 <= N2) vb = *(const s16x8*)pb;
        else
          for (int e = 0; e < 8; ++e)
            if (n2_0 + lcol + e < N2) vb[e] = (short)pb[e];
      }
      ra[c] = va;
      rb[c] = vb;
      if (do_colsum) {
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += __uint_as_float(((uint32_t)(uint16_t)va[e]) << 16);
      }
    }
  };
  auto lstore = [&](int buf) {
    char* abase = smem + buf * 2 * TN_TILE;
    char* bbase = abase + TN_TILE;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *(s16x8*)(abase + (lrow + 16 * c) * TN_LD + lcol * 2) = ra[c];
      *(s16x8*)(bbase + (lrow + 16 * c) * TN_LD + lcol * 2) = rb[c];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int i16 = lane & 15, fg = lane >> 4;
  // k-slot (g, j) of a 32-deep MFMA step holds m = 4g + j (j<4) | 16 + 4g + (j-4): the same permutation for both operands
  const int tr_off = (4 * fg + (i16 >> 2)) * TN_LD + 8 * (i16 & 3);

  gload(kt_begin);
  lstore(0);
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    if (kt + 1 < kt_end) gload(kt + 1);
    const char* abase = smem + cur * 2 * TN_TILE;
    const char* bbase = abase + TN_TILE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* p = abase + ks * 32 * TN_LD + tr_off + (wr * 64 + i * 16) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p + 16 * TN_LD));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        a[i] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const char* p = bbase + ks * 32 * TN_LD + tr_off + (wc * 64 + j * 16) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p + 16 * TN_LD));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        b[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < kt_end) lstore(cur ^ 1);
    __syncthreads();
  }

  float* cs = (float*)smem;
  if (do_colsum) {   // 16 row-lanes (tid >> 4) hold partial sums of the same 8 columns: combine through LDS
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[lrow * 128 + lcol + e] = csum[e];
    __syncthreads();
    if (tid < 128 && n1_0 + tid < N1) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += cs[r * 128 + tid];
      atomicAdd(&a_colsum[n1_0 + tid], t);
    }
  }
  // epilogue: stage through LDS, then 256-B contiguous f32 atomics per wave-instruction
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
    if (wr == h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[(i * 16 + fg * 4 + r) * CS_LD + wc * 64 + j * 16 + i16] = acc[i][j][r];
    }
    __syncthreads();
    for (int s = 0; s < 32; ++s) {
      const int rl = (tid >> 7) + 2 * s, cl = tid & 127;
      const int n1 = n1_0 + h * 64 + rl, n2 = n2_0 + cl;
      if (n1 < N1 && n2 < N2) atomicAdd(&C[(size_t)n1 * ldc + n2], cs[rl * CS_LD + cl]);
    }
  }
}

// ---- wgrad tile 128 x 192 for the narrow student (D = 192): the 192-wide operand is covered by ONE tile, so the wide operand
// (dY or the saved activation, 4x larger) streams through exactly once instead of once per 128-column tile (the 128^2 kernel
// runs at ~3.6 TB/s of re-read traffic: fabric-bound, not MFMA-bound).  SWAP: the caller's A is the 192-wide one; the kernel
// then computes (B^T A) and writes it transposed, and the fused bias column sums come from the kernel's B operand.
constexpr int T192_LDB = 416;              // 384 B of data + 32 B pad (tr reads conflict-free: 104 dwords = 40 mod 64)
constexpr int T192_CS = 197;               // odd f32 staging stride: row- and column-order reads both conflict-free
template <bool SWAP>
__global__ __launch_bounds__(256, 2) void gemm_tn192_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* C, int M,
                                                            int N1, int N2, int lda, int ldb, int ldc, DkdRowMap amap, DkdRowMap bmap,
                                                            int kt_per_split, float* __restrict__ colsum) {
  constexpr int A_BYTES = 64 * TN_LD, B_BYTES = 64 * T192_LDB;
  constexpr int SMEM = (A_BYTES + B_BYTES) > (64 * T192_CS * 4) ? (A_BYTES + B_BYTES) : (64 * T192_CS * 4);
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int n1_0 = blockIdx.x * 128;
  const int KT_all = (M + 63) / 64;
  const int kt_begin = blockIdx.y * kt_per_split;
  const int kt_end = min(KT_all, kt_begin + kt_per_split);
  if (kt_begin >= kt_end) return;

  const int arow = tid >> 4, acol = (tid & 15) * 8;          // A staging: rows arow + 16 c (c < 4)
  const int brow = tid / 24, bcol = (tid % 24) * 8;          // B staging: threads 0..239, rows brow + 10 c (c < 7)
  const bool bld = tid < 240;
  s16x8 ra[4], rb[7];
  const bool sum_a = colsum != nullptr && !SWAP, sum_b = colsum != nullptr && SWAP && blockIdx.x == 0;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int kt) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int m = kt * 64 + arow + 16 * c;
      s16x8 va = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < M) {
        const bf16_t* pa = A + (size_t)map_row(amap, m) * lda + n1_0 + acol;
        if (n1_0 + acol + 8 <= N1) va = *(const s16x8*)pa;
        else
          for (int e = 0; e < 8; ++e)
            if (n1_0 + acol + e < N1) va[e] = (short)pa[e];
      }
      ra[c] = va;
      if (sum_a) {
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += __uint_as_float(((uint32_t)(uint16_t)va[e]) << 16);
      }
    }
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      const int r = brow + 10 * c;
      const int m = kt * 64 + r;
      s16x8 vb = {0, 0, 0, 0, 0, 0, 0, 0};
      if (bld && r < 64 && m < M) {
        const bf16_t* pb = B + (size_t)map_row(bmap, m) * ldb + bcol;
        if (bcol + 8 <= N2) vb = *(const s16x8*)pb;
        else
          for (int e = 0; e < 8; ++e)
            if (bcol + e < N2) vb[e] = (short)pb[e];
      }
      rb[c] = vb;
      if (sum_b) {
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += __uint_as_float(((uint32_t)(uint16_t)vb[e]) << 16);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) *(s16x8*)(smem + (arow + 16 * c) * TN_LD + acol * 2) = ra[c];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      const int r = brow + 10 * c;
      if (bld && r < 64) *(s16x8*)(smem + A_BYTES + r * T192_LDB + bcol * 2) = rb[c];
    }
  };

  f32x4 acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int i16 = lane & 15, fg = lane >> 4;
  const int tra = (4 * fg + (i16 >> 2)) * TN_LD + 8 * (i16 & 3);
  const int trb = (4 * fg + (i16 >> 2)) * T192_LDB + 8 * (i16 & 3);

  gload(kt_begin);
  lstore();
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    if (kt + 1 < kt_end) gload(kt + 1);                      // global loads fly under the MFMAs of this tile
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[6];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* p = smem + ks * 32 * TN_LD + tra + (wr * 64 + i * 16) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p + 16 * TN_LD));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        a[i] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const char* p = smem + A_BYTES + ks * 32 * T192_LDB + trb + (wc * 96 + j * 16) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p + 16 * T192_LDB));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        b[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                         // everyone is done reading the single LDS buffer
    if (kt + 1 < kt_end) lstore();
    __syncthreads();
  }

  float* cs = (float*)smem;
  if (sum_a) {          // 16 row-lanes hold partial sums of the same 8 columns of this block's A slice
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[arow * 128 + acol + e] = csum[e];
    __syncthreads();
    if (tid < 128 && n1_0 + tid < N1) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += cs[r * 128 + tid];
      atomicAdd(&colsum[n1_0 + tid], t);
    }
    __syncthreads();
  }
  if (sum_b) {          // 10 row-lanes hold partial sums of the same 8 columns of B (the caller's A): first tile row only
    if (bld) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[brow * 192 + bcol + e] = csum[e];
    }
    __syncthreads();
    if (tid < 192 && tid < N2) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 10; ++r) t += cs[r * 192 + tid];
      atomicAdd(&colsum[tid], t);
    }
    __syncthreads();
  }
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
    if (wr == h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[(i * 16 + fg * 4 + r) * T192_CS + wc * 96 + j * 16 + i16] = acc[i][j][r];
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * 192; idx += 256) {
      int rl, cl;
      if (SWAP) { cl = idx >> 6; rl = idx & 63; }            // lanes run along n1 (contiguous in the caller's transposed C)
      else { rl = idx / 192; cl = idx % 192; }
      const int n1 = n1_0 + h * 64 + rl;
      if (n1 < N1 && cl < N2) {
        float* dst = SWAP ? &C[(size_t)cl * ldc + n1] : &C[(size_t)n1 * ldc + cl];
        atomicAdd(dst, cs[rl * T192_CS + cl]);
      }
    }
  }
}

}  // namespace

// ---- optional launch probe (bench.py): HIP events around every NT-GEMM launch, on the stream it is launched on.
namespace {
struct ProbeRec {
  int sym;
  double flops;
  hipEvent_t e0, e1;
};
std::mutex g_probe_mu;
bool g_probe_on = false;
std::vector<ProbeRec> g_probe;
struct ProbeScope {
  bool on;
  ProbeRec rec;
  hipStream_t st;
  ProbeScope(int sym, double flops, hipStream_t s) : on(false), st(s) {
    std::lock_guard<std::mutex> lk(g_probe_mu);
    if (!g_probe_on) return;
    on = true;
    rec.sym = sym;
    rec.flops = flops;
    (void)hipEventCreate(&rec.e0);
    (void)hipEventCreate(&rec.e1);
    (void)hipEventRecord(rec.e0, st);
  }
  ~ProbeScope() {
    if (!on) return;
    (void)hipEventRecord(rec.e1, st);
    std::lock_guard<std::mutex> lk(g_probe_mu);
    g_probe.push_back(rec);
  }
};
}  // namespace

extern "C" int dkd_probe_begin(void) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_probe.clear();
  g_probe_on = true;
  return DKD_OK;
}

// sym 0 = gemm_nt_kernel<128>, 1 = gemm_nt_kernel<64>, 2 = gemm_nt256_kernel.  Arrays of 3.
extern "C" int dkd_probe_end(double* flops, double* ms, int32_t* launches) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_probe_on = false;
  for (int i = 0; i < 3; ++i) {
    flops[i] = 0.0;
    ms[i] = 0.0;
    launches[i] = 0;
  }
  for (auto& r : g_probe) {
    (void)hipEventSynchronize(r.e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.e0, r.e1);
    flops[r.sym] += r.flops;
    ms[r.sym] += t;
    launches[r.sym] += 1;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_probe.clear();
  return DKD_OK;
}

extern "C" int dkd_gemm_nt(const DkdGemm* gp, void* stream) {
  DKD_CHECK_ARG(gp && gp->A && gp->B && gp->C, "gemm_nt: null operand");
  const DkdGemm& g = *gp;
  DKD_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
  DKD_CHECK_ARG(g.K % BK == 0, "gemm_nt: K=%d must be a multiple of %d", g.K, BK);
  DKD_CHECK_ARG(g.lda % 8 == 0 && g.ldb % 8 == 0, "gemm_nt: lda=%d / ldb=%d must be multiples of 8 (16-B rows)", g.lda, g.ldb);
  DKD_CHECK_ARG(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0, "gemm_nt: A/B must be 16-byte aligned");
  DKD_CHECK_ARG(!(g.epi & DKD_EPI_BIAS) || g.bias, "gemm_nt: BIAS without bias pointer");
  DKD_CHECK_ARG(!(g.epi & DKD_EPI_RESID) || g.resid, "gemm_nt: RESID without resid pointer");
  DKD_CHECK_ARG(!(g.epi & DKD_EPI_DGELU) || g.preact, "gemm_nt: DGELU without preact pointer");
  DKD_CHECK_ARG(!(g.epi & DKD_EPI_ACCUM) || (g.epi & DKD_EPI_OUT_F32), "gemm_nt: ACCUM needs f32 output");
  DKD_CHECK_ARG(!g.rowscale || g.rows_per_sample > 0, "gemm_nt: rowscale needs rows_per_sample");
  int vec_ok = (g.N % 8 == 0) && (g.ldc % 8 == 0) && (((uintptr_t)g.C & 15) == 0);
  if (g.epi & DKD_EPI_BIAS) vec_ok = vec_ok && (((uintptr_t)g.bias & 15) == 0);
  if (g.epi & DKD_EPI_RESID) vec_ok = vec_ok && (g.ldr % 8 == 0) && (((uintptr_t)g.resid & 15) == 0);
  if (g.preact) vec_ok = vec_ok && (g.ldp % 8 == 0) && (((uintptr_t)g.preact & 15) == 0);
  if (g.tap) vec_ok = vec_ok && (g.ldt % 8 == 0) && (((uintptr_t)g.tap & 15) == 0);
  const bool narrow = (g.N % 128 != 0) && (g.N % 128 <= 64);
  const int tiles_m = cdiv(g.M, BM);
  // wide GEMMs with enough 256^2 tiles to keep 256 CUs balanced (>= 4 rounds): qkv / fc1 of the teacher
  const bool wide = g.N % 256 == 0 && (long)cdiv(g.M, 256) * (g.N / 256) >= 1024;
  ProbeScope probe(wide ? 2 : (narrow ? 1 : 0), 2.0 * g.M * g.N * g.K, as_stream(stream));
  if (wide) {
    hipLaunchKernelGGL(gemm_nt256_kernel, dim3(cdiv(g.M, 256) * (g.N / 256)), dim3(512), 0, as_stream(stream), g, vec_ok);
    DKD_CHECK_LAUNCH("gemm_nt256");
    return DKD_OK;
  }
  if (narrow) {
    hipLaunchKernelGGL(gemm_nt_kernel<64>, dim3(tiles_m * cdiv(g.N, 64)), dim3(256), 0, as_stream(stream), g, vec_ok);
  } else {
    hipLaunchKernelGGL(gemm_nt_kernel<128>, dim3(tiles_m * cdiv(g.N, 128)), dim3(256), 0, as_stream(stream), g, vec_ok);
  }
  DKD_CHECK_LAUNCH("gemm_nt");
  return DKD_OK;
}

extern "C" int dkd_gemm_tn(const void* A, const void* B, float* C, int32_t M, int32_t N1, int32_t N2, int32_t lda, int32_t ldb,
                           int32_t ldc, DkdRowMap amap, DkdRowMap bmap, float* a_colsum, void* stream) {
  DKD_CHECK_ARG(A && B && C, "gemm_tn: null operand");
  DKD_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0, "gemm_tn: empty problem");
  DKD_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0, "gemm_tn: lda=%d / ldb=%d must be multiples of 8", lda, ldb);
  DKD_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_tn: A/B must be 16-byte aligned");
  const int KT = cdiv(M, 64);
  const bool wide_b = N2 > 128 && N2 <= 192;                       // B is the 192-wide operand
  const bool wide_a = !wide_b && N1 > 128 && N1 <= 192 && N2 > 192; // A is: swap roles, write transposed
  if (wide_b || wide_a) {
    const int t1 = cdiv(wide_b ? N1 : N2, 128);
    int sp = cdiv(512, t1);
    if (sp > cdiv(KT, 4)) sp = cdiv(KT, 4);
    if (sp < 1) sp = 1;
    const int per1 = cdiv(KT, sp);
    sp = cdiv(KT, per1);
    if (wide_b)
      hipLaunchKernelGGL(gemm_tn192_kernel<false>, dim3(t1, sp), dim3(256), 0, as_stream(stream), (const bf16_t*)A, (const bf16_t*)B, C, M, N1,
                         N2, lda, ldb, ldc, amap, bmap, per1, a_colsum);
    else
      hipLaunchKernelGGL(gemm_tn192_kernel<true>, dim3(t1, sp), dim3(256), 0, as_stream(stream), (const bf16_t*)B, (const bf16_t*)A, C, M, N2,
                         N1, ldb, lda, ldc, bmap, amap, per1, a_colsum);
    DKD_CHECK_LAUNCH("gemm_tn192");
    return DKD_OK;
  }
  const int tiles = cdiv(N1, 128) * cdiv(N2, 128);
  // enough M-splits to fill 256 CUs x 2 blocks, but at least 4 k-tiles per block
  int splits = cdiv(512, tiles);
  if (splits > cdiv(KT, 4)) splits = cdiv(KT, 4);
  if (splits < 1) splits = 1;
  const int per = cdiv(KT, splits);
  splits = cdiv(KT, per);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), 0, as_stream(stream), (const bf16_t*)A, (const bf16_t*)B, C, M,
                     N1, N2, lda, ldb, ldc, amap, bmap, per, a_colsum);
  DKD_CHECK_LAUNCH("gemm_tn");
  return DKD_OK;
}

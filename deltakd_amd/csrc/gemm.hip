// bf16 MFMA GEMMs for the ViT forward/backward (gfx950).
//
//  gemm_nt : C[M,N] = epi(A[M,K] * B[N,K]^T)   -- every nn.Linear forward and every dgrad (with W^T shadow copies)
//  gemm_tn : C[N1,N2] += A[M,N1]^T * B[M,N2]   -- every weight gradient (split over M, f32 atomics)
//
// Design (MI355X_MICROARCH / cdna_hip_programming sections 3, 5):
//  * 128 x BN x 64 tiles, 4 waves (2x2), v_mfma_f32_16x16x32_bf16, fp32 accumulators in registers.
//  * NT operands are both K-contiguous: tiles go HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR
//    round trip); the LDS image is lane-linear so the bank-conflict swizzle (16-B slot ^= (row>>1)&7) is applied to the
//    per-lane SOURCE address and again on the ds_read_b128 fragment reads.
//  * TN operands are both K(=m)-strided: tiles are register-staged into 288-B-stride rows and the MFMA fragments are
//    read with ds_read_b64_tr_b16 (hardware transpose), so no transposed activation copy ever exists in HBM.
//  * epilogue goes through LDS so that bias / GELU / residual / DropPath-scale / feature-tap traffic is 16-B coalesced.
//  * blockIdx -> tile map is XCD-aware (each XCD's L2 sees a contiguous run of tiles sharing A panels).
#include "common.h"

namespace {

constexpr int BM = 128, BK = 64;
constexpr int CS_LD = 132;  // f32 epilogue staging row stride (floats)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective "each XCD gets a contiguous chunk" remap (guide T1); bid % 8 labels the XCD group.
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

struct EpiCtx {
  const DkdGemm* g;
};

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

template <int BN>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const DkdGemm g, const int vec_ok) {
  constexpr int NJ = BN / 32;             // 16-col MFMA tiles per wave along N
  constexpr int A_BYTES = BM * 128;       // 16 KiB
  constexpr int B_BYTES = BN * 128;
  constexpr int BUF = A_BYTES + B_BYTES;
  constexpr int SMEM = (2 * BUF) > (64 * CS_LD * 4) ? (2 * BUF) : (64 * CS_LD * 4);
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

  // ---- epilogue: 2 passes of 64 rows through LDS (f32), then 16-B coalesced fused stores
  float* cs = (float*)smem;
  const int col4 = (tid & 31) * 4;
  const bool out_f32 = g.epi & DKD_EPI_OUT_F32;
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
    if (wr == h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[(i * 16 + fg * 4 + r) * CS_LD + wc * (BN / 2) + j * 16 + frow] = acc[i][j][r];
    }
    __syncthreads();
    if (col4 < BN) {
      for (int s = 0; s < 8; ++s) {
        const int rl = (tid >> 5) + 8 * s;
        const int m = m0 + h * 64 + rl, n = n0 + col4;
        if (m >= g.M || n >= g.N) continue;
        const size_t crow = (size_t)map_row(g.cmap, m) * g.ldc;
        f32x4 v = *(const f32x4*)&cs[rl * CS_LD + col4];
        if (vec_ok) {
          if (g.epi & DKD_EPI_BIAS) v += *(const f32x4*)&g.bias[n];
          if (g.epi & DKD_EPI_GELU) {
            if (g.preact) {
              uint2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
              *(uint2*)&((bf16_t*)g.preact)[(size_t)m * g.ldp + n] = pk;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          }
          if (g.epi & DKD_EPI_DGELU) {
            const uint2 pk = *(const uint2*)&((const bf16_t*)g.preact)[(size_t)m * g.ldp + n];
            v[0] *= dgelu_erf(__uint_as_float(pk.x << 16));
            v[1] *= dgelu_erf(__uint_as_float(pk.x & 0xffff0000u));
            v[2] *= dgelu_erf(__uint_as_float(pk.y << 16));
            v[3] *= dgelu_erf(__uint_as_float(pk.y & 0xffff0000u));
          }
          if (g.epi & DKD_EPI_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (g.tap) {
            if (g.epi & DKD_EPI_TAP_F32) *(f32x4*)&((float*)g.tap)[(size_t)m * g.ldt + n] = v;
            else {
              uint2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
              *(uint2*)&((bf16_t*)g.tap)[(size_t)m * g.ldt + n] = pk;
            }
          }
          if (g.epi & DKD_EPI_RESID) {
            const float sc = g.rowscale ? g.rowscale[m / g.rows_per_sample] : 1.f;
            const f32x4 rv = *(const f32x4*)&g.resid[(size_t)map_row(g.rmap, m) * g.ldr + n];
            v = rv + sc * v;
          }
          if (out_f32) {
            float* cp = (float*)g.C + crow + n;
            if (g.epi & DKD_EPI_ACCUM) v += *(const f32x4*)cp;
            *(f32x4*)cp = v;
          } else {
            uint2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *(uint2*)((bf16_t*)g.C + crow + n) = pk;
          }
        } else {
          for (int e = 0; e < 4 && n + e < g.N; ++e) {
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
    }
  }
}

// ------------------------------------------------------------------------------------------------ TN (wgrad)
constexpr int TN_LD = 288;                 // bytes per LDS row (256 B of data + 32 B pad: tr reads conflict-free)
constexpr int TN_TILE = 64 * TN_LD;        // 18 KiB per operand tile
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* C,
                                                         int M, int N1, int N2, int lda, int ldb, int ldc, DkdRowMap amap,
                                                         DkdRowMap bmap, int kt_per_split) {
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

  // epilogue: stage through LDS, then 256-B contiguous f32 atomics per wave-instruction
  float* cs = (float*)smem;
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

}  // namespace

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
  int vec_ok = (g.N % 4 == 0) && (g.ldc % 4 == 0) && (((uintptr_t)g.C & 15) == 0);
  if (g.epi & DKD_EPI_BIAS) vec_ok = vec_ok && (((uintptr_t)g.bias & 15) == 0);
  if (g.epi & DKD_EPI_RESID) vec_ok = vec_ok && (g.ldr % 4 == 0) && (((uintptr_t)g.resid & 15) == 0);
  if (g.preact) vec_ok = vec_ok && (g.ldp % 4 == 0) && (((uintptr_t)g.preact & 7) == 0);
  if (g.tap) vec_ok = vec_ok && (g.ldt % 4 == 0) && (((uintptr_t)g.tap & 15) == 0);
  const bool narrow = (g.N % 128 != 0) && (g.N % 128 <= 64);
  const int tiles_m = cdiv(g.M, BM);
  if (narrow) {
    hipLaunchKernelGGL(gemm_nt_kernel<64>, dim3(tiles_m * cdiv(g.N, 64)), dim3(256), 0, as_stream(stream), g, vec_ok);
  } else {
    hipLaunchKernelGGL(gemm_nt_kernel<128>, dim3(tiles_m * cdiv(g.N, 128)), dim3(256), 0, as_stream(stream), g, vec_ok);
  }
  DKD_CHECK_LAUNCH("gemm_nt");
  return DKD_OK;
}

extern "C" int dkd_gemm_tn(const void* A, const void* B, float* C, int32_t M, int32_t N1, int32_t N2, int32_t lda, int32_t ldb,
                           int32_t ldc, DkdRowMap amap, DkdRowMap bmap, void* stream) {
  DKD_CHECK_ARG(A && B && C, "gemm_tn: null operand");
  DKD_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0, "gemm_tn: empty problem");
  DKD_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0, "gemm_tn: lda=%d / ldb=%d must be multiples of 8", lda, ldb);
  DKD_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_tn: A/B must be 16-byte aligned");
  const int tiles = cdiv(N1, 128) * cdiv(N2, 128);
  const int KT = cdiv(M, 64);
  // enough M-splits to fill 256 CUs x 2 blocks, but at least 4 k-tiles per block
  int splits = cdiv(512, tiles);
  if (splits > cdiv(KT, 4)) splits = cdiv(KT, 4);
  if (splits < 1) splits = 1;
  const int per = cdiv(KT, splits);
  splits = cdiv(KT, per);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), 0, as_stream(stream), (const bf16_t*)A, (const bf16_t*)B, C, M,
                     N1, N2, lda, ldb, ldc, amap, bmap, per);
  DKD_CHECK_LAUNCH("gemm_tn");
  return DKD_OK;
}

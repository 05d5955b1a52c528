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
#include "common.h"

namespace {

constexpr int BM = 128, BK = 64;
__device__ uint4 dkd_zero16 = {0u, 0u, 0u, 0u};      // source of LDS-DMA granules that must read as zeros (edges, conv padding)
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
  if (g.epi & DKD_EPI_RELU_GATE) v = bf2f(((const bf16_t*)g.preact)[(size_t)m * g.ldp + n]) > 0.f ? v : 0.f;
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
    if (g.epi & (DKD_EPI_DGELU | DKD_EPI_RELU_GATE)) in.pre = *(const uint4*)&((const bf16_t*)g.preact)[(size_t)m * g.ldp + n];
  }
  return in;
}
__device__ __forceinline__ void epi_finish(const DkdGemm& g, const int vec_ok, f32x8 v, const EpiIn& in, const int m, const int n,
                                           const bool bias_done = false) {
  const bool out_f32 = g.epi & DKD_EPI_OUT_F32;
  const size_t crow = (size_t)map_row(g.cmap, m) * g.ldc;
  if (vec_ok) {
    if ((g.epi & DKD_EPI_BIAS) && !bias_done) {
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
    if (g.epi & DKD_EPI_RELU_GATE) {
      const f32x8 p = unpack8(in.pre);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = p[e] > 0.f ? v[e] : 0.f;
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

#ifndef DKD_NT_C_A
#define DKD_NT_C_A 1
#define DKD_NT_C_B 4
#endif
// streaming (nontemporal) 16-byte store: the bf16 activations these epilogues write (qkv, fc1 pre-activation and GELU output, dgrad
// results) are consumed by a LATER kernel, 16 lanes of a wave cover whole 128-byte lines of a row
__device__ __forceinline__ void nt_store16(bf16_t* p, const uint4 v) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, (u32x4*)p);
}

// ---- epilogue shared by the 128-row NT kernels: ONE pass through LDS (f32 [128][BN], unpadded: the accumulator-layout
// ds_write_b32 is only 2-way per 32-lane group = free, the row reads are contiguous), then 16-B coalesced fused stores.
// FAST 3: C(f32) = resid + acc + bias with identity row maps, no row scale (the teacher's proj / fc2; bf16 tap optional) -- compiled
// without the generic path's per-vector flag tests and row-map arithmetic.  7: C(f32) = resid + rowscale[sample] * (acc + bias), the
// student's proj / fc2 under DropPath.  0: generic.
// residual rows of one thread of the FAST 3 / 7 epilogue (2 halves x SW sweeps x 8 columns), requested before the K loop
template <int BN>
struct ResidRegs {
  static constexpr int SW = 64 / (256 / (BN / 8));
  f32x4 lo[2][SW], hi[2][SW];
  uint4 pre[2][SW];                      // FAST 5: the bf16 pre-activation rows of the dGELU epilogue
};
template <int BN>
__device__ __forceinline__ void preact_prefetch(const DkdGemm& g, ResidRegs<BN>& r, const int m0, const int n0, const int tid) {
  constexpr int TPR = BN / 8, RPP = 256 / TPR, SW = 64 / RPP;
  const int n = n0 + (tid % TPR) * 8, r0 = tid / TPR;
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int s = 0; s < SW; ++s) {
      const int m = m0 + half * 64 + r0 + RPP * s;
      r.pre[half][s] = uint4{0u, 0u, 0u, 0u};
      if (m < g.M && n < g.N) {          // read once, never again: do not let it displace dH / W in the L2
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = __builtin_nontemporal_load((const u32x4*)&((const bf16_t*)g.preact)[(size_t)m * g.ldp + n]);
        r.pre[half][s] = uint4{t.x, t.y, t.z, t.w};
      }
    }
}
template <int BN>
__device__ __forceinline__ void resid_prefetch(const DkdGemm& g, ResidRegs<BN>& r, const int m0, const int n0, const int tid) {
  constexpr int TPR = BN / 8, RPP = 256 / TPR, SW = 64 / RPP;
  const int n = n0 + (tid % TPR) * 8, r0 = tid / TPR;
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int s = 0; s < SW; ++s) {
      const int m = m0 + half * 64 + r0 + RPP * s;
      r.lo[half][s] = f32x4{0.f, 0.f, 0.f, 0.f}, r.hi[half][s] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m < g.M && n < g.N) {
        const float* rp = &g.resid[(size_t)m * g.ldr + n];
        r.lo[half][s] = *(const f32x4*)rp;
        r.hi[half][s] = *(const f32x4*)(rp + 4);
      }
    }
}

template <int BN, int NJ, int FAST>
__device__ __forceinline__ void nt_epilogue(const DkdGemm& g, const int vec_ok, char* smem, f32x4 (&acc)[4][NJ], const int m0,
                                            const int n0, const int tid, const int wr, const int wc, const ResidRegs<BN>* prf = nullptr) {
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
  if (FAST == 1 || FAST == 4 || FAST == 5 || FAST == 6) {
    // bf16 outputs with identity row maps: 1: C = acc + bias;  4: preact = acc + bias, C = gelu(preact);  5: C = acc * gelu'(preact);
    // 6: C = acc
    constexpr int SW = 64 / RPP;
    f32x8 b8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (FAST != 5 && FAST != 6) {
      const f32x4 b0 = *(const f32x4*)&g.bias[n], b1 = *(const f32x4*)&g.bias[n + 4];
      b8 = f32x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint4 pre[SW];
      if (FAST == 5) {                   // requested before the K loop (preact_prefetch)
#pragma unroll
        for (int s = 0; s < SW; ++s) pre[s] = prf->pre[half][s];
      }
#pragma unroll
      for (int s = 0; s < SW; ++s) {
        const int rl = half * 64 + r0 + RPP * s;
        const int m = m0 + rl;
        if (m < g.M) {
          const f32x4 lo = *(const f32x4*)&cs[rl * BN + col8], hi = *(const f32x4*)&cs[rl * BN + col8 + 4];
          f32x8 v = f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]} + b8;
          if (FAST == 4) {
            nt_store16((bf16_t*)g.preact + (size_t)m * g.ldp + n, pack8(v));
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const dkd_f32x2 y = gelu_erf_fast2(dkd_f32x2{v[e], v[e + 1]});
              v[e] = y[0], v[e + 1] = y[1];
            }
          }
          if (FAST == 5) {
            const f32x8 p = unpack8(pre[s]);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const dkd_f32x2 d = dgelu_erf_fast2(dkd_f32x2{p[e], p[e + 1]});
              v[e] *= d[0], v[e + 1] *= d[1];
            }
          }
          // forward outputs (qkv, GELU output) stream; a backward result is read again by the very next kernels (dgrad GEMM + wgrad)
          if (FAST == DKD_NT_C_A || FAST == DKD_NT_C_B) nt_store16((bf16_t*)g.C + (size_t)m * g.ldc + n, pack8(v));
          else *(uint4*)((bf16_t*)g.C + (size_t)m * g.ldc + n) = pack8(v);
        }
      }
    }
    return;
  }
  if (FAST == 3 || FAST == 7) {           // 7: the same with the DropPath row scale (the student's proj / fc2)
    constexpr int SW = 64 / RPP;
    const f32x4 b0 = *(const f32x4*)&g.bias[n], b1 = *(const f32x4*)&g.bias[n + 4];
    const bool tap = g.tap != nullptr;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 r0v[SW], r1v[SW];
#pragma unroll
      for (int s = 0; s < SW; ++s) {
        r0v[s] = prf->lo[half][s];
        r1v[s] = prf->hi[half][s];
      }
#pragma unroll
      for (int s = 0; s < SW; ++s) {
        const int rl = half * 64 + r0 + RPP * s;
        const int m = m0 + rl;
        if (m < g.M) {
          const f32x4 lo = *(const f32x4*)&cs[rl * BN + col8] + b0, hi = *(const f32x4*)&cs[rl * BN + col8 + 4] + b1;
          if (tap)
            *(uint4*)&((bf16_t*)g.tap)[(size_t)m * g.ldt + n] = uint4{pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(hi[0], hi[1]),
                                                                       pack2bf(hi[2], hi[3])};
          float* cp = (float*)g.C + (size_t)m * g.ldc + n;
          if (FAST == 7) {
            const float sc = g.rowscale[m / g.rows_per_sample];
            *(f32x4*)cp = r0v[s] + sc * lo;
            *(f32x4*)(cp + 4) = r1v[s] + sc * hi;
          } else {
            const f32x4 o0 = r0v[s] + lo, o1 = r1v[s] + hi;
            *(f32x4*)cp = o0;
            *(f32x4*)(cp + 4) = o1;
            if (g.xb) {                  // LayerNorm folded into the next GEMM (DkdGemm.xb): bf16 copy of the new x + this tile's row sums
              *(uint4*)&((bf16_t*)g.xb)[(size_t)m * g.ldxb + n] = uint4{pack2bf(o0[0], o0[1]), pack2bf(o0[2], o0[3]), pack2bf(o1[0], o1[1]),
                                                                         pack2bf(o1[2], o1[3])};
              float s1 = (o0[0] + o0[1]) + (o0[2] + o0[3]) + (o1[0] + o1[1]) + (o1[2] + o1[3]);
              float s2 = (o0[0] * o0[0] + o0[1] * o0[1]) + (o0[2] * o0[2] + o0[3] * o0[3]) + (o1[0] * o1[0] + o1[1] * o1[1]) +
                         (o1[2] * o1[2] + o1[3] * o1[3]);
#pragma unroll
              for (int o = 1; o < TPR; o <<= 1) {       // the TPR lanes of this row (TPR = 16 or 8: inside one wave)
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
              }
              // parked in LDS behind the tile staging: one lane per row adding its two sums (4 active lanes per wave instruction,
              // 128 instructions per tile) cost proj +25 us; the tile's 256 sums go out below as 4 full-width atomic instructions
              if (tid % TPR == 0) *(float2*)&cs[128 * BN + 2 * rl] = float2{s1, s2};
            }
          }
        }
      }
    }
    if (FAST == 3 && g.xb) {              // (uniform) rowstats[2 (m0 + r) + {0, 1}] += the row's sums over this tile's columns
      __syncthreads();
      const int m = m0 + (tid >> 1);
      if (m < g.M) atomicAdd(&g.rowstats[2 * (size_t)m0 + tid], cs[128 * BN + tid]);
    }
    return;
  }
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

// CONV: implicit GEMM of a 3 x 3 / pad 1 convolution on the hw x hw token grid (MGD generation block, model/models.py:148-151):
// A is the activation x [B * hw * hw, Cin] itself, K = 9 * Cin with k = tap * Cin + c (tap = ky * 3 + kx); the A row a lane sources for
// output row m and K step kt is row m + (ky - 1) * hw + (kx - 1) when that pixel lies inside the image, else a page of zeros -- the
// gather lives in the LDS-DMA source addressing, the [M, 9 Cin] im2col matrix is never written.  Cin % 64 == 0 (a 64-wide K step
// stays inside one tap).
template <int BN, int FAST = 0, bool CONV = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const DkdGemm g, const int vec_ok) {
  constexpr int NJ = BN / 32;             // 16-col MFMA tiles per wave along N
  constexpr int A_BYTES = BM * 128;       // 16 KiB
  constexpr int B_BYTES = BN * 128;
  constexpr int BUF = A_BYTES + B_BYTES;
  constexpr int SMEM = ((2 * BUF) > (128 * BN * 4) ? (2 * BUF) : (128 * BN * 4)) + (FAST == 3 ? 1024 : 0);   // + the tile's row sums (LN fold)
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
  int apy[4], apx[4];                     // CONV: pixel of the lane's row
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = w * 32 + c * 8 + (lane >> 3);
    int m = m0 + r;
    m = m < g.M ? m : g.M - 1;
    arow[c] = (const bf16_t*)g.A + (size_t)map_row(g.amap, m) * g.lda;
    aslot[c] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
    if (CONV) {
      const int p = m % (g.conv_hw * g.conv_hw);
      apy[c] = p / g.conv_hw;
      apx[c] = p % g.conv_hw;
    }
  }
  const int conv_cin = CONV ? g.K / 9 : 0;
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
    if (CONV) {
      const int tap = k0 / conv_cin, c0 = k0 - tap * conv_cin;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = (unsigned)(apy[c] + dy) < (unsigned)g.conv_hw && (unsigned)(apx[c] + dx) < (unsigned)g.conv_hw;
        const bf16_t* src = ok ? arow[c] + (long)(dy * g.conv_hw + dx) * g.lda + c0 + aslot[c] : (const bf16_t*)&dkd_zero16;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(abase + (w * 32 + c * 8) * 128), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        __builtin_amdgcn_global_load_lds(GLB_PTR(arow[c] + k0 + aslot[c]), LDS_PTR(abase + (w * 32 + c * 8) * 128), 16, 0, 0);
    }
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

  // f32 residual epilogues: the rows this thread will add to are requested NOW and arrive under the K loop; read in the epilogue
  // (after the tile has been staged) their round trip was exposed once per tile -- K = 768 is only 12 steps
  ResidRegs<BN> rres;
  if (FAST == 3 || FAST == 7) resid_prefetch<BN>(g, rres, m0, n0, tid);
  if (FAST == 5) preact_prefetch<BN>(g, rres, m0, n0, tid);

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

  nt_epilogue<BN, NJ, FAST>(g, vec_ok, smem, acc, m0, n0, tid, wr, wc, &rres);
}

// ---- 256 x 256 tile, 8 waves (2 x 4, 128 x 64 per wave), PERSISTENT, one workgroup per CU: for the wide teacher GEMMs.
// The 128^2 kernel moves 1 B of operand from L2 into LDS per 64 FLOP and its K loop is bound by that path; this tile halves
// the L2->LDS bytes and the LDS-DMA / ds_read instructions per MFMA.  What the counters said about the first version of it
// (double-buffered 64-wide K steps, profiles/r01_d_*): MFMA busy 55 %, waves 40 % of their time in s_waitcnt, LDS array 21 %
// busy with no bank conflicts -- the loop waits for the LDS-DMA stage loads.  With 2 stage buffers at most 64 KiB per CU is
// in flight and only right after issue; Little's law at ~2k cycles of loaded latency needs that much in flight ALL the time.
//  * K is walked in 32-wide "units" (A half 16 KiB + W half 16 KiB, 64-B LDS rows); a ring of 5 units uses all 160 KiB of
//    LDS: one being multiplied, four in flight (128 KiB).  Unit q+5 is issued during unit q, right after the barrier that
//    frees q's slot, one 1-KiB piece per 8 MFMAs (an LDS-DMA instruction holds its wave's issue for 60-180 cycles; the SIMD's
//    other wave issues MFMAs underneath); the wait is the counted vmcnt(12) = "all but the three newest units".
//  * the ring runs across tiles: the loads of the next tile's first units are in flight during a tile's last phases and its
//    epilogue, so there is no prologue bubble per tile and the epilogue overlaps the next tile's loads.
//  * the fragments of the NEXT unit are read into a second register set while the MFMAs of the current one issue; every wave
//    leaves the barrier with 32 MFMAs' operands already in registers.
//  * the MFMA is issued with the operands swapped (acc = W-fragment x A-fragment): a lane's 4 accumulator registers are 4
//    consecutive output COLUMNS of one row, and the W rows a fragment reads are permuted so that fragments j, j+1 hold
//    adjacent 4-column groups -- the epilogue stores 8 consecutive columns (16 B bf16 / 2 x 16 B f32) per lane straight from
//    registers: no LDS staging (the ring owns the LDS), no barriers, no cross-wave coupling.
//  * 64-B LDS rows: 16-B slot s of row r sits at physical slot s ^ g[key(r)], g = {0,2,3,1}, key = (r>>2)&3 for A rows and
//    (r>>3)&3 for W rows (= (lane>>2)&3 of the reading lane for both); every ds_read_b128 lane group then covers all 64 banks.
// EOPS: vector-memory instructions one wave issues in an interior tile's epilogue when that is known at compile time (16: bf16 C
// and nothing else), 0 otherwise.  Known, the waits that follow an epilogue count its stores as outstanding instead of draining
// them: they retire under the next tile's first phases.
// ABL: dev-only ablation bits (build with -DDKD_NT256_ABL=n; results are then wrong, timings are the point): 1 no epilogue,
// 2 two units per tile, 4 no LDS-DMA in the loop, 8 no barriers, 16 no fragment reads, 32 no MFMAs, 64 LDS-DMA pieces of full 128-byte lines,
// 128 every second workgroup of an XCD skips its bf16 stores, 256 ordinary instead of streaming stores, 512 the W pieces as plain
// register loads instead of LDS-DMA.
// WN: waves along N.  4: the 256 x 256 tile, 8 waves, ring of 5 units, one workgroup per CU (qkv / fc1 of the teacher).
//     2: a 256 x 128 tile, 4 waves, ring of 3 units (72 KiB), two workgroups per CU -- for N = 768 (proj / fc2), where 256-wide
//        tiles leave 256 CUs with 2.3 rounds of work; it moves 25 % less operand data through the LDS-DMA path than the
//        128 x 128 kernel, and the second workgroup computes while the first one's f32 residual epilogue drains.
constexpr uint32_t vmcnt_imm(int n) { return (uint32_t)((n & 15) | ((n >> 4) << 14) | 0x0F70); }   // s_waitcnt vmcnt(n) only
// FAST: 0 = the generic fused epilogue (runtime flags, row maps); 1 = C(bf16) = acc + bias; 2 = C(bf16) = gelu(acc + bias);
// 3 = C(f32) = resid + acc + bias (+ bf16 tap); all with identity row maps -- the two epilogues the teacher's qkv / fc1 use, compiled without the per-vector flag tests, row-map
// divisions and spilled-SGPR reads of the generic path (they, not the GELU arithmetic, were most of the epilogue's VALU time).
// FOLD (FAST 1 / 2 only): the LayerNorm in front of this Linear is folded into it (DkdGemm.ln_stats / ln_c): the epilogue is
// C = rstd[m] (acc - mean[m] c[n]) + bias[n] with the row statistics from the producer GEMM's row sums.
template <int ABL, int EOPS, int WN, int FAST, bool FOLD = false>
__global__ __launch_bounds__(128 * WN, WN == 4 ? 1 : 2) void gemm_nt256_kernel(const DkdGemm g, const int n_tiles_cg) {
  constexpr int BN = 64 * WN, NW = 2 * WN;          // tile columns, waves
  constexpr int WHALF = 16384, UNIT = WHALF + BN * 64, RING = WN == 4 ? 5 : 3;
  constexpr int AP = 8 / WN, PIECES = AP + 2;       // 1-KiB LDS-DMA pieces per wave per unit: A rows, then 2 of W rows
  static_assert(EOPS == 0 || ((EOPS == 16 || (FOLD && EOPS == 24)) && WN == 4), "the counted waits below encode (RING-2 | RING-1) * PIECES + EOPS < 64");
  __shared__ __attribute__((aligned(16))) char smem[RING * UNIT];   // WN = 4: all 160 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w / WN, wc = w % WN;
  const int tiles_n = g.N / BN;
  const int P = g.K / 32;               // units per tile (even: K % 64 == 0)

  // this block's tiles: XCD x = blockIdx % 8 owns a contiguous chunk of the tile list (its L2 sees neighbouring A panels and all
  // of W); the chunk is dealt round-robin to the XCD's resident blocks
  const int x = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  // n_tiles_cg = n_tiles | col_groups << 24.  col_groups = 1: the chunk is a range of the row-major tile list (a band of row panels,
  // ALL column tiles: the XCD's L2 has to hold all of W).  col_groups = 2 / 4: the XCDs form a (8 / col_groups) x col_groups grid and
  // an XCD owns a band of row panels x 1/col_groups of the column tiles -- for a W that does not fit the 4 MiB L2 beside the A panels
  // (teacher fc1: 4.7 MB; it was re-fetched from the fabric once per round, 5.8x the algorithmic read traffic).
  const int n_tiles = n_tiles_cg & 0xFFFFFF, CG = n_tiles_cg >> 24;
  int chunk_begin, chunk_cnt, sub_cols = 0, sub_row0 = 0, sub_col0 = 0;
  if (CG <= 1) {
    const int tq = n_tiles >> 3, tr = n_tiles & 7;
    chunk_begin = x < tr ? x * (tq + 1) : tr * (tq + 1) + (x - tr) * tq;
    chunk_cnt = tq + (x < tr ? 1 : 0);
  } else {
    const int RG = 8 / CG, rg = x / CG, cg = x % CG, tiles_m = n_tiles / tiles_n;
    sub_cols = tiles_n / CG;
    sub_col0 = cg * sub_cols;
    sub_row0 = (int)((long)rg * tiles_m / RG);
    chunk_cnt = ((int)((long)(rg + 1) * tiles_m / RG) - sub_row0) * sub_cols;
    chunk_begin = 0;
  }
  const int my_tiles = slot_in_xcd < chunk_cnt ? (chunk_cnt - slot_in_xcd + nslots - 1) / nslots : 0;
  if (my_tiles == 0) return;
  const int total_units = my_tiles * P;
  auto tile_of = [&](int k) {
    const int j = slot_in_xcd + k * nslots;
    return CG <= 1 ? chunk_begin + j : (sub_row0 + j / sub_cols) * tiles_n + sub_col0 + j % sub_cols;
  };

  // ---- load cursor (runs RING units ahead of the multiply cursor)
  const bf16_t* Ab = (const bf16_t*)g.A;
  const bf16_t* Bb = (const bf16_t*)g.B;
  uint32_t aoff[AP], boff[2];           // element offsets of this lane's 16-B source granules (< 2^31: host-checked)
  int ld_tile = 0, ld_p = 0, ld_unit = 0;
  auto set_load_tile = [&](int k) {
    const int L = tile_of(k);
    const int lm0 = (L / tiles_n) * 256, ln0 = (L % tiles_n) * BN;
#pragma unroll
    for (int c = 0; c < AP; ++c) {
      const int r = w * (256 / NW) + c * 16 + (lane >> 2);
      int m = lm0 + r;
      m = m < g.M ? m : g.M - 1;
      const int ga = (0x78 >> (2 * ((r >> 2) & 3))) & 3;
      aoff[c] = (uint32_t)map_row(g.amap, m) * (uint32_t)g.lda + (((lane & 3) ^ ga) * 8);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int r = w * 32 + c * 16 + (lane >> 2);
      const int gw = (0x78 >> (2 * ((r >> 3) & 3))) & 3;
      boff[c] = (uint32_t)(ln0 + r) * (uint32_t)g.ldb + (((lane & 3) ^ gw) * 8);
    }
  };
  set_load_tile(0);
  // piece c of the load cursor's unit -> ring slot `slot`.  c < AP: A rows w*(256/NW) + c*16 ..+15; then two pieces of W rows.
  // Issued through inline asm: the compiler orders every LDS read behind a pending LDS-DMA it knows of with vmcnt(0), which would
  // drain the ring at each fragment read; ordering is this kernel's job (counted vmcnt + barrier), and the compiler's own
  // vmcnt bookkeeping for the epilogue's loads can only over-wait because no DMA is issued between such a load and its use.
  f32x4 abl_junk = {0.f, 0.f, 0.f, 0.f};      // (ablation bit 512 only)
  auto piece = [&](int c, int slot) {
    const bool is_a = c < AP;
    const int row0 = is_a ? w * (256 / NW) + c * 16 : w * 32 + (c - AP) * 16;
    const uint32_t dst = (uint32_t)(uintptr_t)LDS_PTR(smem) + slot * UNIT + (is_a ? 0 : WHALF) + row0 * 64;
    // SGPR base + 32-bit byte offset: half the address data of the 64-bit-per-lane form.  M0 (the LDS destination) is set inside
    // the statement and named as a clobber; nothing else in this kernel makes the compiler use M0.
    uint32_t voff = ((is_a ? aoff[is_a ? c : 0] : boff[is_a ? 0 : c - AP]) + ld_p * 32) * 2;
    if (ABL & 64) {
      // (ablation: the same bytes per unit as FULL 128-byte line segments -- 8 rows x 128 B per piece, rows 0-7 of the piece's 16 on even
      // units and rows 8-15 on odd ones, columns of both units -- to see what 64-byte segments cost the texture path; LDS contents are garbage)
      const uint32_t base = is_a ? aoff[is_a ? c : 0] : boff[is_a ? 0 : c - AP];
      const uint32_t ld = is_a ? (uint32_t)g.lda : (uint32_t)g.ldb;
      const uint32_t rowbase = base - (((lane >> 2) * ld) + ((base % ld) % 32));           // row 0 of the piece, column 0 of the tile's K range
      voff = (rowbase + ((lane >> 3) + 8 * (ld_p & 1)) * ld + (lane & 7) * 8 + (ld_p >> 1) * 64) * 2;
    }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    if (is_a) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(Ab) : "memory", "m0");
    else if (ABL & 512) {
      // (ablation bit 512: the W pieces as ordinary 16-byte-per-lane loads into a junk register quad -- same bytes, same slots in the
      // in-order vmcnt queue, no LDS write: is the ~23 B/clk of the LDS-DMA stream the LDS side's or the memory pipeline's?  timings only)
      // The destination stays LIVE for the whole kernel ("+v" here, consumed behind a vmcnt(0) at the kernel's end): a dead asm-load
      // destination is handed to another value while the load is in flight (attn192_bwd defect 1 -- and the first build of
      // this very ablation, which faulted).
      asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(abl_junk) : "v"(voff), "s"(Bb) : "memory");
    } else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(Bb) : "memory", "m0");
#pragma clang diagnostic pop
  };
  // past the end of the block's work the cursor stays on the last unit: the re-issued pieces land in a slot nobody reads and keep
  // the vmcnt arithmetic uniform
  auto advance_load = [&]() {
    if (ld_unit + 1 < total_units) {
      ++ld_unit;
      if (++ld_p == P) {
        ld_p = 0;
        set_load_tile(++ld_tile);
      }
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fg = lane >> 4;
  const int phys = fg ^ ((0x78 >> (2 * ((frow >> 2) & 3))) & 3);
  const int a_lds = (wr * 128 + frow) * 64 + phys * 16;
  const int w_lds = WHALF + (wc * 64 + 8 * (frow >> 2) + (frow & 3)) * 64 + phys * 16;   // wr in {0,1}, wc < WN

  // A fragments are refreshed IN PLACE: a[i] is dead once its 4 MFMAs have issued, so the next unit's fragment i is read into
  // it right behind them (96 -> 64 fragment registers; a second full set spilled).  W fragments (used by every i) have two sets.
  bf16x8 a[8], bF[4], bG[4];
  auto lda_frag = [&](const int i, const int slot) { a[i] = *(const bf16x8*)(smem + slot * UNIT + a_lds + i * 1024); };
  auto ldb_frag = [&](bf16x8 (&b)[4], const int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(smem + slot * UNIT + w_lds + (j >> 1) * 2048 + (j & 1) * 256);
  };

  int slot = 0;                         // ring slot of the unit being multiplied
  int pend = 0;                         // phases left in which the previous epilogue's stores may be outstanding
  bool exact = false;                   // the previous epilogue issued exactly EOPS instructions in every wave
  // one unit: a[] and X hold its fragments (already waited for); a[] and Y receive the next unit's (not across a tile boundary:
  // the epilogue needs the registers)
  auto phase = [&](const bf16x8 (&X)[4], bf16x8 (&Y)[4], const bool last_of_tile) {
    // my pieces of the next unit have landed: all but the RING-2 newest units -- plus, in the first RING-2 phases after an
    // epilogue with a known instruction count, its stores (they sit between those units in the in-order counter)
    if (EOPS > 0 && pend > 0) {
      __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 2) * PIECES + EOPS));
      --pend;
    } else {
      __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 2) * PIECES));
    }
    if (!(ABL & 8)) __builtin_amdgcn_s_barrier();         // ... and everybody's; and everybody is done reading this unit's slot
    __builtin_amdgcn_sched_barrier(0);
    const int nslot = slot == RING - 1 ? 0 : slot + 1;
    if (!last_of_tile && !(ABL & 16)) ldb_frag(Y, nslot);
    // One LDS-DMA piece per 8 (WN = 4) or ~5 (WN = 2) MFMAs, every wave at the same points.  (Tried: each wave at its own MFMA slot so that at most one
    // wave of the CU is in a DMA issue -- 25 % slower; all four pieces in a burst behind the barrier -- the same.  In shader
    // cycles this loop already runs at the rate of its LDS-DMA stream alone: DESIGN.md, 'wide NT kernel'.)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(ABL & 32)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X[j], a[i], acc[i][j], 0, 0, 0);
      if (!last_of_tile && !(ABL & 16)) lda_frag(i, nslot);
      // PIECES = 4: after row groups 1,3,5,7;  6: after 0,1,3,4,5,7
      const bool issue_here = PIECES == 4 ? (i & 1) : (i % 4 != 2);      // (constants once the loop is unrolled)
      const int which = PIECES == 4 ? (i >> 1) : (i - (i > 2) - (i > 6));
      if (issue_here && !last_of_tile && !(ABL & 4)) piece(which, slot);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!last_of_tile) advance_load();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the next unit's fragments arrived under these MFMAs
  };

  // FOLD: the row sums of a tile's 8 rows per lane (DkdGemm.ln_stats) are requested one tile AHEAD, right behind the previous tile's
  // epilogue stores, through asm the compiler's vmcnt bookkeeping does not see: read in the epilogue itself, its wait for them was
  // vmcnt(0) -- the LDS-DMA ring drained once per tile (+16 us per launch).  They count as 8 more known operations of that
  // epilogue (EOPS = 24) and have long arrived when the next epilogue reads them (every phase waits for all but the newest units).
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 stq[8];
  auto request_stats = [&](const int kt) {
    const int L = tile_of(kt < my_tiles ? kt : my_tiles - 1);
    const int rm0 = (L / tiles_n) * 256;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int m = rm0 + wr * 128 + i * 16 + frow;
      m = m < g.M ? m : g.M - 1;
      const float* sp = g.ln_stats + 2 * (size_t)m;
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(stq[i]) : "v"(sp) : "memory");
    }
  };
  if (FOLD) request_stats(0);

  for (int u = 0; u < RING; ++u) {
#pragma unroll
    for (int c = 0; c < PIECES; ++c) piece(c, u);
    advance_load();
  }

  for (int k = 0; k < my_tiles; ++k) {
    // ---- tile prologue: the tile's first unit has landed.  Outstanding in the in-order counter at this point: RING-1 units, the
    // previous tile's epilogue stores, and the unit issued after them -> "all but the newest unit" (k > 0), which over-waits by
    // the stores' acknowledge unless their number is known; RING-1 units ahead on the very first tile.
    if (k == 0) __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 1) * PIECES));
    else if (EOPS > 0 && exact) __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 1) * PIECES + EOPS));   // RING-2 units + the stores + the newest unit
    else __builtin_amdgcn_s_waitcnt(vmcnt_imm(PIECES));
    pend = (EOPS > 0 && exact && k > 0) ? RING - 2 : 0;
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ldb_frag(bF, slot);
#pragma unroll
    for (int i = 0; i < 8; ++i) lda_frag(i, slot);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int p = 0; p < ((ABL & 2) ? 2 : P); p += 2) {
      phase(bF, bG, false);
      slot = slot == RING - 1 ? 0 : slot + 1;
      phase(bG, bF, p + 2 >= P);
      // (the slot of a tile's last unit is refilled after the epilogue, below)
      if (p + 2 < P) slot = slot == RING - 1 ? 0 : slot + 1;
    }
    // ---- epilogue, straight from registers: lane holds C[m0 + wr*128 + i*16 + frow][n0 + wc*64 + 32*jp + 8*fg .. +7]
    const int L = tile_of(k);
    const int m0 = (L / tiles_n) * 256, n0 = (L % tiles_n) * BN;
    exact = m0 + 256 <= g.M && !((ABL & 128) && (slot_in_xcd & 1));      // no row of the tile is masked off: every wave issues all its stores
    if (ABL & 1) {
      float t = 0.f;
      for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (t == 1234.5f) ((float*)g.C)[0] = t;
    } else {
      const int nb = n0 + wc * 64 + fg * 8;
      f32x8 bias8[2];                    // in registers: C may alias anything, so the compiler would reload it per vector
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        bias8[jp] = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (g.epi & DKD_EPI_BIAS) {
          const f32x4 b0 = *(const f32x4*)&g.bias[nb + jp * 32], b1 = *(const f32x4*)&g.bias[nb + jp * 32 + 4];
          bias8[jp] = f32x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        }
      }
      f32x8 c8[2];                       // FOLD: row sums of the gamma-scaled weight for this lane's columns; mean / rstd of its 8 rows
      float mu8[8], rs8[8];
      if (FOLD) {
        const float invK = 1.f / (float)g.K;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const f32x4 c0 = *(const f32x4*)&g.ln_c[nb + jp * 32], c1 = *(const f32x4*)&g.ln_c[nb + jp * 32 + 4];
          c8[jp] = f32x8{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("" : "+v"(stq[i]));          // (requested a tile ago: see request_stats)
          mu8[i] = __uint_as_float(stq[i].x) * invK;
          rs8[i] = rsqrtf(fmaxf(__uint_as_float(stq[i].y) * invK - mu8[i] * mu8[i], 0.f) + g.ln_eps);   // (one-pass variance: clamp the rounding)
        }
      }
      constexpr bool FAST3 = FAST == 3;
      if (FAST3) {
        // C(f32) = resid + acc + bias, identity row maps, optional bf16 tap: 32 contiguous bytes per lane, 128 per row and block.
        // C may alias resid (it does: the residual stream is updated in place), so the compiler keeps every load behind the stores
        // before it -- row group by row group that was 8 exposed memory round trips per tile, all eight waves idle in each.  The
        // loads are issued two row groups at a time, ahead of their stores (more spills: the accumulators hold 128 registers): 4 round trips.
        constexpr int GB = 2;
#pragma unroll
        for (int i0 = 0; i0 < 8; i0 += GB) {
          f32x4 rr[GB][4];
          float st1[GB] = {0.f, 0.f}, st2[GB] = {0.f, 0.f};
#pragma unroll
          for (int gi = 0; gi < GB; ++gi) {
            const int m = m0 + wr * 128 + (i0 + gi) * 16 + frow;
            if (m < g.M) {
              const float* rp = &g.resid[(size_t)m * g.ldr + nb];
              rr[gi][0] = *(const f32x4*)rp, rr[gi][1] = *(const f32x4*)(rp + 4), rr[gi][2] = *(const f32x4*)(rp + 32), rr[gi][3] = *(const f32x4*)(rp + 36);
            }
          }
#pragma unroll
          for (int gi = 0; gi < GB; ++gi) {
            const int i = i0 + gi, m = m0 + wr * 128 + i * 16 + frow;
            if (m < g.M) {
              const f32x8 v0 = f32x8{acc[i][0][0], acc[i][0][1], acc[i][0][2], acc[i][0][3], acc[i][1][0], acc[i][1][1], acc[i][1][2],
                                     acc[i][1][3]} + bias8[0];
              const f32x8 v1 = f32x8{acc[i][2][0], acc[i][2][1], acc[i][2][2], acc[i][2][3], acc[i][3][0], acc[i][3][1], acc[i][3][2],
                                     acc[i][3][3]} + bias8[1];
              if (g.tap) {
                bf16_t* tp = &((bf16_t*)g.tap)[(size_t)m * g.ldt + nb];
                *(uint4*)tp = pack8(v0);
                *(uint4*)(tp + 32) = pack8(v1);
              }
              float* cp = (float*)g.C + (size_t)m * g.ldc + nb;
              const f32x4 o0 = rr[gi][0] + f32x4{v0[0], v0[1], v0[2], v0[3]}, o1 = rr[gi][1] + f32x4{v0[4], v0[5], v0[6], v0[7]};
              const f32x4 o2 = rr[gi][2] + f32x4{v1[0], v1[1], v1[2], v1[3]}, o3 = rr[gi][3] + f32x4{v1[4], v1[5], v1[6], v1[7]};
              *(f32x4*)cp = o0;
              *(f32x4*)(cp + 4) = o1;
              *(f32x4*)(cp + 32) = o2;
              *(f32x4*)(cp + 36) = o3;
              if (g.xb) {                // LayerNorm folded into the next GEMM: bf16 copy of the new x + row sums of these 16 columns
                bf16_t* xp = &((bf16_t*)g.xb)[(size_t)m * g.ldxb + nb];
                *(uint4*)xp = uint4{pack2bf(o0[0], o0[1]), pack2bf(o0[2], o0[3]), pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3])};
                *(uint4*)(xp + 32) = uint4{pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3]), pack2bf(o3[0], o3[1]), pack2bf(o3[2], o3[3])};
                const f32x4 sq = o0 * o0 + o1 * o1 + o2 * o2 + o3 * o3, sm = o0 + o1 + o2 + o3;
                st1[gi] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
                st2[gi] = (sq[0] + sq[1]) + (sq[2] + sq[3]);
              }
            }
          }
          if (g.xb) {                    // (wave-uniform) the four lanes fg of a row hold its 64 columns of this wave: combine, one adds
#pragma unroll
            for (int gi = 0; gi < GB; ++gi) {
              float a1 = st1[gi], a2 = st2[gi];
              a1 += __shfl_xor(a1, 16, 64);
              a2 += __shfl_xor(a2, 16, 64);
              a1 += __shfl_xor(a1, 32, 64);
              a2 += __shfl_xor(a2, 32, 64);
              const int m = m0 + wr * 128 + (i0 + gi) * 16 + frow;
              if (fg == 0 && m < g.M) {
                atomicAdd(&g.rowstats[2 * (size_t)m], a1);
                atomicAdd(&g.rowstats[2 * (size_t)m + 1], a2);
              }
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < (FAST3 ? 0 : 8); ++i) {
        const int m = m0 + wr * 128 + i * 16 + frow;
        if (FAST == 1 || FAST == 2) {
          // bf16 output, streaming stores of FULL 128-byte lines: a row's 64 columns of this wave sit in 4 lanes x 2 column
          // blocks; lanes of an even/odd row pair swap one block (DPP), so that one store instruction writes both halves of the
          // even row's line and the next one the odd row's.  (Half-line streaming stores wrote 30 % more bytes to memory.)
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          uint4 pk[2];
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const f32x4 lo = acc[i][2 * jp], hi = acc[i][2 * jp + 1];
            f32x8 v = f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (FOLD) v = rs8[i] * (v - mu8[i] * c8[jp]) + bias8[jp];
            else v += bias8[jp];
            if (FAST == 2) {
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                const dkd_f32x2 y = gelu_erf_fast2(dkd_f32x2{v[e], v[e + 1]});
                v[e] = y[0], v[e + 1] = y[1];
              }
            }
            pk[jp] = pack8(v);
          }
          const bool odd = frow & 1;
          const uint4 send = odd ? pk[0] : pk[1];
          const uint4 recv = {(uint32_t)__shfl_xor((int)send.x, 1, 64), (uint32_t)__shfl_xor((int)send.y, 1, 64),
                              (uint32_t)__shfl_xor((int)send.z, 1, 64), (uint32_t)__shfl_xor((int)send.w, 1, 64)};
          const uint4 d0 = odd ? recv : pk[0], d1 = odd ? pk[1] : recv;      // line of the even row, line of the odd row
          const int me = m - (odd ? 1 : 0);
          bf16_t* cp = (bf16_t*)g.C + (size_t)me * g.ldc + nb + (odd ? 32 : 0);
          // (ablation bit 128: every second workgroup of an XCD keeps its results -- are the epilogue's stores bound per CU or by the
          // chip's HBM write rate?  timings only)
          const bool keep_out = !((ABL & 128) && (slot_in_xcd & 1));
          if (ABL & 256) {                 // (ablation bit 256: ordinary stores -- does the consumer find the tensor in the Infinity Cache?)
            if (me < g.M) *(u32x4*)cp = u32x4{d0.x, d0.y, d0.z, d0.w};
            if (me + 1 < g.M) *(u32x4*)(cp + g.ldc) = u32x4{d1.x, d1.y, d1.z, d1.w};
          } else {
            if (me < g.M && keep_out) __builtin_nontemporal_store(u32x4{d0.x, d0.y, d0.z, d0.w}, (u32x4*)cp);
            if (me + 1 < g.M && keep_out) __builtin_nontemporal_store(u32x4{d1.x, d1.y, d1.z, d1.w}, (u32x4*)(cp + g.ldc));
          }
        } else if (m < g.M) {
          EpiIn in[2];
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) in[jp] = epi_prefetch(g, 1, m, nb + jp * 32);
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const f32x4 lo = acc[i][2 * jp], hi = acc[i][2 * jp + 1];
            epi_finish(g, 1, f32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]} + bias8[jp], in[jp], m, nb + jp * 32, true);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (FOLD) request_stats(k + 1);       // (behind this tile's stores, ahead of the refill pieces: part of the counted EOPS)
    // refill the slot the tile's last unit occupied (every wave passed that unit's barrier long ago... but not its READS: the
    // last phase's fragment reads were of the NEXT slot; this slot's reads completed before the last phase's barrier)
#pragma unroll
    for (int c = 0; c < PIECES; ++c) piece(c, slot);
    advance_load();
    slot = slot == RING - 1 ? 0 : slot + 1;
  }
  if (ABL & 512) {
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    asm volatile("" ::"v"(abl_junk));
  }
}

// ------------------------------------------------------------------------------------------------ dgrad GEMM + LayerNorm backward
// The narrow student (D = 192): the dgrad GEMMs that end a branch -- dT = dH W1 (K = hidden) and dT = dqkv Wqkv (K = 3 D) -- have
// N = D, so a 128 x 192 tile holds whole rows and the LayerNorm backward that consumes dT can BE the epilogue:
//     dx = rstd (dT gamma - mean(dT gamma) - xhat mean(dT gamma xhat)),   g += dx,   dgamma += dT xhat,   dbeta += dT
// dT never goes to memory (round trip of 2 x 19 MB and a launch per LayerNorm at bs 256), it stays f32 instead of being rounded to
// bf16, and the scale-cast that opens the next branch (dF = bf16(rowscale g)) rides along as a second output.
// Mainloop = gemm_nt_kernel's (LDS-DMA double buffering) on a 128 x 192 tile; LDS 2 x 40 KiB: two workgroups per CU.
// Epilogue: the tile through LDS (f32 [64][196]), then the row pass of ln_bwd_kernel (16 lanes per row, 12 columns per lane) with the
// per-column partial sums of dgamma / dbeta written to the workspace rows of ln_bwd_reduce_kernel.
struct LnBwdEpi {
  const float* x;          // f32 [M, ldx] input of the LayerNorm (forward)
  const float* gamma;
  const float* mean;
  const float* rstd;
  float* dx;               // f32 [M, lddx]: += LN'(dT)
  float* part;             // f32 [gridDim.x][2 * 192] per-block partial sums (dgamma | dbeta)
  bf16_t* cast_out;        // optional bf16 [M, 192]: rowscale[row / rows_per_sample] * (updated dx)
  const float* rowscale;   // optional (NULL = 1)
  int ldx, lddx, rows_per_sample;
};
constexpr int LNB_D = 192, LNB_CS = 196, LNB_BM = 128;
__global__ __launch_bounds__(256, 2) void gemm_nt_lnbwd_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const int M,
                                                            const int K, const int lda, const int ldb, const LnBwdEpi ln) {
  // 128 x 192 tile, the four waves SIDE BY SIDE along N (wave tile 128 x 48: 8 x 3 MFMA tiles, 96 accumulator registers): every wave
  // owns rows of every 32-row quarter of the epilogue, so the accumulators drain quarter by quarter and the global loads of the 8
  // rows a wave normalises per quarter (x and the gradient rows: 48 registers) are in flight before the quarter is staged.  Measured on the
  // way here: 2 x 2 waves on 128 rows -- no registers for that, the row pass was a chain of exposed round trips, 62 us = no faster
  // than the three kernels it replaces; 64-row tiles -- W is streamed through the LDS-DMA path (~25 B/clk/CU) once per 64 rows, 54 us.
  constexpr int BN = LNB_D, NJ = 3;
  constexpr int A_BYTES = LNB_BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;      // 16 + 24 KiB per stage
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];      // 80 KiB; the epilogue uses 64 * 196 * 4 = 49 KiB of it
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = xcd_remap(blockIdx.x, gridDim.x) * LNB_BM;
  const int KT = K / BK;

  const bf16_t* arow[4];
  int aslot[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = w * 32 + c * 8 + (lane >> 3);
    int m = m0 + r;
    m = m < M ? m : M - 1;
    arow[c] = A + (size_t)m * lda;
    aslot[c] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }
  constexpr int BCH = BN / 32;                 // 6 chunks of 8 W rows per wave
  const bf16_t* brow[BCH];
  int bslot[BCH];
#pragma unroll
  for (int c = 0; c < BCH; ++c) {
    const int r = w * (BN / 4) + c * 8 + (lane >> 3);
    brow[c] = W + (size_t)r * ldb;
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

  f32x4 acc[8][NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
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
      bf16x8 b[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = *(const bf16x8*)(bbase + (w * (BN / 4) + j * 16 + frow) * 128 + ps);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bf16x8 a = *(const bf16x8*)(abase + (i * 16 + frow) * 128 + ps);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: LayerNorm backward on whole rows, 32 rows at a time (f32 [32][196] through LDS).  16 lanes per row (lane owns
  // float4 columns sl, sl + 16, sl + 32).
  const int sl = lane & 15, gq = lane >> 4;
  float* cs = (float*)smem;
  f32x4 gam[3], ag[3], ab[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gam[i] = *(const f32x4*)(ln.gamma + 4 * (sl + 16 * i));
    ag[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {                // 32 rows at a time: a wave owns 8 of them = 2 passes of 4 rows
    // x, the gradient rows and the statistics of the wave's 8 rows are requested before the quarter is staged
    f32x4 xv[2][3], dv[2][3];
    float mus[2], rss[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = m0 + q * 32 + w * 8 + it * 4 + gq;
      const bool live = row < M;
      const float* xr = ln.x + (size_t)(live ? row : 0) * ln.ldx;
      const float* dr = ln.dx + (size_t)(live ? row : 0) * ln.lddx;
      mus[it] = live ? ln.mean[row] : 0.f;
      rss[it] = live ? ln.rstd[row] : 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        xv[it][i] = *(const f32x4*)(xr + 4 * (sl + 16 * i));
        dv[it][i] = *(const f32x4*)(dr + 4 * (sl + 16 * i));
      }
    }
    __syncthreads();                           // the K loop's last fragment reads / the previous quarter's row pass are done
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[(i * 16 + fg * 4 + r) * LNB_CS + w * (BN / 4) + j * 16 + frow] = acc[q * 2 + i][j][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int rl = w * 8 + it * 4 + gq;
      const int row = m0 + q * 32 + rl;
      const bool live = row < M;
      const float mu = mus[it], rs = rss[it];
      f32x4 xh[3], gy[3];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int c4 = 4 * (sl + 16 * i);
        xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        gy[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live) {
          const f32x4 d = *(const f32x4*)&cs[rl * LNB_CS + c4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[i][e] = (xv[it][i][e] - mu) * rs;
            ab[i][e] += d[e];
            ag[i][e] += d[e] * xh[i][e];
            gy[i][e] = d[e] * gam[i][e];
            s1 += gy[i][e];
            s2 += gy[i][e] * xh[i][e];
          }
        }
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
      }
      s1 *= 1.f / LNB_D;
      s2 *= 1.f / LNB_D;
      if (live) {
        float* dr = ln.dx + (size_t)row * ln.lddx;
        const float sc = ln.cast_out ? (ln.rowscale ? ln.rowscale[row / ln.rows_per_sample] : 1.f) : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int c4 = 4 * (sl + 16 * i);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs * (gy[i][e] - s1 - xh[i][e] * s2) + dv[it][i][e];
          *(f32x4*)(dr + c4) = o;
          if (ln.cast_out) {
            const uint2 pk = {pack2bf(sc * o[0], sc * o[1]), pack2bf(sc * o[2], sc * o[3])};
            *(uint2*)(ln.cast_out + (size_t)row * LNB_D + c4) = pk;
          }
        }
      }
    }
  }
  // per-column partial sums of dgamma / dbeta: combine the 4 row groups of a wave, then the 4 waves through LDS
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
  float* red = (float*)smem;                   // [2][4][192]
  if (gq == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      *(f32x4*)&red[(0 * 4 + w) * LNB_D + 4 * (sl + 16 * i)] = ag[i];
      *(f32x4*)&red[(1 * 4 + w) * LNB_D + 4 * (sl + 16 * i)] = ab[i];
    }
  }
  __syncthreads();
  if (tid < LNB_D) {
    const float pg = red[0 * LNB_D + tid] + red[1 * LNB_D + tid] + red[2 * LNB_D + tid] + red[3 * LNB_D + tid];
    const float pb = red[4 * LNB_D + tid] + red[5 * LNB_D + tid] + red[6 * LNB_D + tid] + red[7 * LNB_D + tid];
    ln.part[(size_t)blockIdx.x * 2 * LNB_D + tid] = pg;
    ln.part[(size_t)blockIdx.x * 2 * LNB_D + LNB_D + tid] = pb;
  }
}

// ------------------------------------------------------------------------------------------------ TN (wgrad)
constexpr int TN_LD = 288;                 // bytes per LDS row (256 B of data + 32 B pad: tr reads conflict-free)
constexpr int TN_TILE = 64 * TN_LD;        // 18 KiB per operand tile
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* C,
                                                         int M, int N1, int N2, int lda, int ldb, int ldc, DkdRowMap amap,
                                                         DkdRowMap bmap, int kt_per_split, float* __restrict__ a_colsum,
                                                         int upper_only, int conv_hw = 0, int conv_dy = 0, int conv_dx = 0,
                                                         long batch_stride_ab = 0, long batch_stride_c = 0) {
  // conv_hw > 0: weight gradient of a 3 x 3 / pad 1 convolution, tap = blockIdx.z -- the B row paired with reduction index m is the
  // input pixel m + dy * hw + dx when it lies inside the image, zeros otherwise (no im2col matrix); the tap's [N1, N2] block of
  // C = dW[N1][9][N2] starts at column tap * N2
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (conv_hw == 0 && bz > 0) {           // batched problems of one shape (dkd_gram_batched): operands / results at constant strides
    A += (size_t)bz * batch_stride_ab;
    B += (size_t)bz * batch_stride_ab;
    C += (size_t)bz * batch_stride_c;
  }
  if (conv_hw > 0) {
    conv_dy = bz / 3 - 1;
    conv_dx = bz % 3 - 1;
    C += bz * N2;
    if (bz != 4) a_colsum = nullptr;
  }
  __shared__ __attribute__((aligned(16))) char smem[4 * TN_TILE];  // [buf][A|B] ; reused by the epilogue (33 KiB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int tiles2 = (N2 + 127) / 128;
  int t1 = bx / tiles2, t2 = bx % tiles2;
  if (upper_only) {                      // Gram matrix: blockIdx.x enumerates the tile pairs (t1 <= t2) row by row
    int p = bx, row = 0, len = tiles2;
    while (p >= len) {
      p -= len;
      ++row;
      --len;
    }
    t1 = row;
    t2 = row + p;
  }
  const int n1_0 = t1 * 128, n2_0 = t2 * 128;
  const int KT_all = (M + 63) / 64;
  const int kt_begin = by * kt_per_split;
  const int kt_end = min(KT_all, kt_begin + kt_per_split);
  if (kt_begin >= kt_end) return;

  const int lrow = tid >> 4, lcol = (tid & 15) * 8;  // staging: thread -> (row lrow + 16 c, 8 elements at lcol)
  // Operand rows travel global -> registers -> LDS.  A 64-row step is ~0.25 us of MFMA work and a loaded HBM round trip 1-2 us, so the
  // loads run TN_D steps ahead of the multiply in a ring of register sets (one step ahead -- the first version -- left the kernel
  // waiting for memory most of the time: 450 TFLOP/s on the conv wgrads, 95 us for the LRKD Gram matrix).
  constexpr int TN_D = 4;
  s16x8 ra[TN_D][4], rb[TN_D][4];
  // fused bias gradient: column sums of the A operand (dY) ride along in the blocks of the first N2 tile
  const bool do_colsum = a_colsum != nullptr && t2 == 0;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int kt, const int set) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int m = kt * 64 + lrow + 16 * c;
      s16x8 va = {0, 0, 0, 0, 0, 0, 0, 0}, vb = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < M) {
        const bf16_t* pa = A + (size_t)map_row(amap, m) * lda + n1_0 + lcol;
        long brow = map_row(bmap, m);
        bool b_ok = true;
        if (conv_hw > 0) {
          const int p = m % (conv_hw * conv_hw), y = p / conv_hw + conv_dy, x = p % conv_hw + conv_dx;
          b_ok = (unsigned)y < (unsigned)conv_hw && (unsigned)x < (unsigned)conv_hw;
          brow = m + conv_dy * conv_hw + conv_dx;
        }
        const bf16_t* pb = B + (size_t)brow * ldb + n2_0 + lcol;
        if (n1_0 + lcol + 8 <= N1) va = *(const s16x8*)pa;
        else
          for (int e = 0; e < 8; ++e)
            if (n1_0 + lcol + e < N1) va[e] = (short)pa[e];
        if (b_ok) {
          if (n2_0 + lcol + 8 <= N2) vb = *(const s16x8*)pb;
          else
            for (int e = 0; e < 8; ++e)
              if (n2_0 + lcol + e < N2) vb[e] = (short)pb[e];
        }
      }
      ra[set][c] = va;
      rb[set][c] = vb;
      if (do_colsum) {
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += __uint_as_float(((uint32_t)(uint16_t)va[e]) << 16);
      }
    }
  };
  auto lstore = [&](int buf, const int set) {
    char* abase = smem + buf * 2 * TN_TILE;
    char* bbase = abase + TN_TILE;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *(s16x8*)(abase + (lrow + 16 * c) * TN_LD + lcol * 2) = ra[set][c];
      *(s16x8*)(bbase + (lrow + 16 * c) * TN_LD + lcol * 2) = rb[set][c];
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

#pragma unroll
  for (int d = 0; d < TN_D; ++d)
    if (kt_begin + d < kt_end) gload(kt_begin + d, d);
  lstore(0, 0);
  __syncthreads();
  for (int kb = kt_begin; kb < kt_end; kb += TN_D) {
#pragma unroll
    for (int d = 0; d < TN_D; ++d) {
      const int kt = kb + d;
      if (kt < kt_end) {                 // (uniform)
        const int cur = (kt - kt_begin) & 1;
        // set d held step kt, which is in LDS: it takes step kt + TN_D
        if (kt + TN_D < kt_end) gload(kt + TN_D, d);
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
        if (kt + 1 < kt_end) lstore(cur ^ 1, (d + 1) % TN_D);
        __syncthreads();
      }
    }
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

// ---- 128 x 192 wgrad tile fed by an LDS-DMA ring.
// The register-staged kernel above runs one 64-row k step per ~4 us: its loads are issued one step ahead (the registers hold no
// more) and HBM latency under load is several times a step's 0.4 us of MFMA work, so two co-resident workgroups leave the CU idle
// most of the time (main loop alone: 2 TB/s of operand reads).  Here the operands go HBM -> LDS by global_load_lds in 32-row
// units through a ring of four (80 KiB per workgroup, two workgroups per CU); three units are in flight behind the one being
// multiplied, and the waits are counted vmcnt as in the wide NT kernel.
//  * LDS-DMA writes 1 KiB linearly, so rows cannot be padded; 32-B granules are XOR-swizzled instead: A rows (256 B, 8 granules):
//    granule ^ (row & 7); B rows (384 B, 12 granules): within each group of four, (granule & 3) ^ ((row >> 1) & 3).  Both make the
//    8 rows a ds_read_b64_tr_b16 lane group touches fall on 8 distinct 8-bank sets.  The swizzle is applied to the per-lane SOURCE
//    address of the DMA and to the fragment read address.
//  * rows past M and columns past N1 / N2 are sourced from a 16-byte page of zeros.
//  * the fused bias gradient (column sums of dY) is one more MFMA per row tile against a fragment of ones.

// WR wave rows of 64 tile rows each: WR = 2 is the 128 x 192 tile (4 waves, 80 KiB ring, two workgroups per CU), WR = 4 the 256 x 192
// tile (8 waves, 112 KiB, one per CU): per 32-row unit it moves 28 KiB for twice the FLOPs of the small tile's 20 KiB -- the kernel
// runs at the rate of its LDS-DMA stream, and in a grouped launch the tiles of several gradients supply the parallelism.
template <int WR> struct TndCfg {
  static constexpr int AW = 64 * WR;                 // A columns (tile rows of C) per workgroup
  static constexpr int A_ROWB = AW * 2;              // bytes per A row in LDS
  static constexpr int A_ST = 32 * A_ROWB;           // one unit of A
  static constexpr int UNIT = A_ST + 32 * 384;
  static constexpr int RING = WR == 2 ? 4 : 5;       // units resident (one multiplied, the rest in flight): 80 KiB / 140 KiB
  static constexpr int SMEM = RING * UNIT;
  static constexpr int WAVES = 2 * WR;
};
static_assert(TndCfg<2>::SMEM >= 64 * T192_CS * 4, "epilogue staging must fit");

// One problem of the kernel, in kernel-operand order (the host has already swapped the operands when the wide one is A).
struct TnProb {
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  float* colsum;
  int M, N1, N2, lda, ldb, ldc;
  DkdRowMap amap, bmap;
  int units_per_split, tiles1, swap, n_blocks;     // n_blocks: tiles1 * splits rounded up to a multiple of 8
};

// `bid` / `nblk`: this block's index inside its problem and the problem's block count (both in launch order, so bid % 8 is the XCD).
// PB: B pieces this wave issues per unit (3 with 4 waves; with 8 waves the 12 pieces are 2 each for waves 0-3, 1 each for waves 4-7 --
// the counted waits need the per-wave piece count at compile time, so the two halves run two instantiations).
template <int WR, int PB>
__device__ __forceinline__ void tn192d_body(char* smem, const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* C, int M,
                                            int N1, int N2, int lda, int ldb, int ldc, const DkdRowMap& amap, const DkdRowMap& bmap,
                                            int units_per_split, float* __restrict__ colsum, int tiles1, const bool SWAP, int bid,
                                            int nblk, int rot = 0, int tile_in = -1, int split_in = 0) {
  using Cfg = TndCfg<WR>;
  constexpr int A_ST = Cfg::A_ST, UNIT = Cfg::UNIT, RING = Cfg::RING, PIECES = 2 + PB, A_ROWB = Cfg::A_ROWB, AW = Cfg::AW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  // 1-D grid, XCD-aware: each XCD gets a contiguous run of (split, tile) ids with the tiles of a split adjacent, so the blocks
  // that stream the same rows of the 192-wide operand run on the same XCD at the same time and share them through its L2
  // (dealt round-robin, every XCD fetched those rows again: twice the HBM/fabric traffic for a 768 x 192 gradient)
  // (rot: a group launch pads every problem's block range to a multiple of 8 and the padding sits at the END of the remapped list, i.e.
  // on the last XCDs; rotating the XCD labels by the problem's index spreads the idle slots over all eight -- with one 8-wave workgroup
  // per CU and 12 real blocks of 16, two XCDs had no work at all and the others ran two rounds)
  // (tile_in >= 0: the caller placed this block -- the group launch's one-XCD-per-split table, gemm_tn192g_kernel)
  const int L = tile_in >= 0 ? 0 : xcd_remap((bid & ~7) | ((bid + rot) & 7), nblk);
  const int tile = tile_in >= 0 ? tile_in : L % tiles1, split = tile_in >= 0 ? split_in : L / tiles1;
  const int n1_0 = tile * AW;
  const int U_all = (M + 31) / 32;
  const int u_begin = split * units_per_split;
  const int u_end = min(U_all, u_begin + units_per_split);
  if (u_begin >= u_end) return;
  const int n_units = u_end - u_begin;

  // ---- per-lane DMA sources.  piece c < 2: A piece w*2 + c (1 KiB of A_ROWB-byte rows); else one of the 12 B pieces (1 KiB of 384-B rows)
  const int b_first = WR == 2 ? w * 3 : (w < 4 ? w * 2 : 8 + (w - 4));
  int prow[PIECES], pcol[PIECES];        // row inside the unit, source column (elements); pcol < 0: zero page
#pragma unroll
  for (int c = 0; c < PIECES; ++c) {
    if (c < 2) {
      const int off = (w * 2 + c) * 1024 + lane * 16;
      const int row = off / A_ROWB, p = (off % A_ROWB) >> 4;       // 16-byte chunk inside the row
      const int G = p >> 1;                                        // 32-byte granule
      const int g = (G & ~7) | ((G & 7) ^ (row & 7));
      const int col = n1_0 + g * 16 + (p & 1) * 8;
      prow[c] = row;
      pcol[c] = col < N1 ? col : -1;
    } else {
      const int off = (b_first + (c - 2)) * 1024 + lane * 16;
      const int row = off / 384, cb = off % 384;
      const int G = cb >> 5;
      const int g = (G & ~3) | ((G & 3) ^ ((row >> 1) & 3));
      const int col = g * 16 + ((cb >> 4) & 1) * 8;
      prow[c] = row;
      pcol[c] = col < N2 ? col : -1;
    }
  }
  // mapped source row of each piece for the load cursor's unit, advanced incrementally (no division per unit)
  int mrow[PIECES], mgrp[PIECES];
  auto seek = [&](int c, int m) {        // m -> (group, row in group) of the operand's row map
    const DkdRowMap& mp = c < 2 ? amap : bmap;
    if (mp.rpg > 0) {
      mgrp[c] = m / mp.rpg;
      mrow[c] = m % mp.rpg;
    } else {
      mgrp[c] = 0;
      mrow[c] = m;
    }
  };
#pragma unroll
  for (int c = 0; c < PIECES; ++c) seek(c, u_begin * 32 + prow[c]);
  int ld_u = 0;                          // load cursor (unit index inside this block's range)
  const bf16_t* zero = (const bf16_t*)&dkd_zero16;
  auto piece = [&](int c, int slot) {
    const DkdRowMap& mp = c < 2 ? amap : bmap;
    const int m = (u_begin + ld_u) * 32 + prow[c];
    const long srow = mp.rpg > 0 ? (long)mgrp[c] * mp.gstride + mrow[c] + mp.off : (long)mrow[c];
    const bf16_t* base = c < 2 ? A : B;
    const int ld = c < 2 ? lda : ldb;
    const bf16_t* src = (pcol[c] >= 0 && m < M) ? base + srow * ld + pcol[c] : zero;
    const uint32_t dst = (uint32_t)(uintptr_t)LDS_PTR(smem) + slot * UNIT + (c < 2 ? (w * 2 + c) * 1024 : A_ST + (b_first + (c - 2)) * 1024);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory", "m0");
#pragma clang diagnostic pop
  };
  auto advance_load = [&]() {            // past the end the cursor stays: re-issued pieces land in a free slot nobody reads
    if (ld_u + 1 < n_units) {
      ++ld_u;
#pragma unroll
      for (int c = 0; c < PIECES; ++c) {
        const DkdRowMap& mp = c < 2 ? amap : bmap;
        mrow[c] += 32;
        if (mp.rpg > 0)
          while (mrow[c] >= mp.rpg) {
            mrow[c] -= mp.rpg;
            ++mgrp[c];
          }
      }
    }
  };

  f32x4 acc[4][6], acc1[6];              // acc1: the ones-row products (bias gradient); [0..3] for sum_a, [0..5] for sum_b
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 6; ++j) acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool sum_a = colsum != nullptr && !SWAP && wc == 0, sum_b = colsum != nullptr && SWAP && tile == 0 && wr == 0;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  const int i16 = lane & 15, fg = lane >> 4;
  const int frow = 4 * fg + (i16 >> 2);                              // fragment row (lo); hi = +16: same row & 7, same (row>>1)&3
  const int fcol = 8 * (i16 & 3);
  int a_off[4], b_off[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int G = wr * 4 + i;
    a_off[i] = frow * A_ROWB + (((G & ~7) | ((G & 7) ^ (frow & 7))) * 32) + fcol;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int g = wc * 6 + j;
    b_off[j] = A_ST + frow * 384 + (((g & ~3) | ((g & 3) ^ ((frow >> 1) & 3))) * 32) + fcol;
  }

  for (int u = 0; u < RING - 1; ++u) {
#pragma unroll
    for (int c = 0; c < PIECES; ++c) piece(c, u);
    advance_load();
  }
  int slot = 0;
  for (int u = 0; u < n_units; ++u) {
    __builtin_amdgcn_s_waitcnt(vmcnt_imm((RING - 2) * PIECES));     // this unit has landed: all but the two newest
    __builtin_amdgcn_s_barrier();                                    // ... everybody's pieces; and everybody left unit u-1's slot
    __builtin_amdgcn_sched_barrier(0);
    {
      const int free_slot = slot == 0 ? RING - 1 : slot - 1;
#pragma unroll
      for (int c = 0; c < PIECES; ++c) piece(c, free_slot);
      advance_load();
    }
    const char* base = smem + slot * UNIT;
    bf16x8 a[4], b[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(base + a_off[i]));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(base + a_off[i] + 16 * A_ROWB));
      a[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(base + b_off[j]));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(base + b_off[j] + 16 * 384));
      b[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if (sum_a) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, acc1[i], 0, 0, 0);
    }
    if (sum_b) {
#pragma unroll
      for (int j = 0; j < 6; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b[j], acc1[j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                              // lgkmcnt(0): fragment reads done before the next barrier
    slot = slot == RING - 1 ? 0 : slot + 1;
  }
  __builtin_amdgcn_s_waitcnt(0x0070);                                // the padding pieces too: the ring becomes epilogue staging
  __syncthreads();

  // bias gradient: column 0 of a ones-product tile holds the sums of the 16 rows (sum_a); row 0 the sums of the 16 columns (sum_b)
  if (sum_a && i16 == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n1 = n1_0 + wr * 64 + i * 16 + fg * 4 + r;
        if (n1 < N1) atomicAdd(&colsum[n1], acc1[i][r]);
      }
  }
  if (sum_b && fg == 0) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int n2 = wc * 96 + j * 16 + i16;
      if (n2 < N2) atomicAdd(&colsum[n2], acc1[j][0]);
    }
  }
  float* cs = (float*)smem;
  for (int h = 0; h < WR; ++h) {
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
    for (int idx = tid; idx < 64 * 192; idx += 64 * Cfg::WAVES) {
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

template <bool SWAP>
__global__ __launch_bounds__(256, 2) void gemm_tn192d_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* C, int M,
                                                             int N1, int N2, int lda, int ldb, int ldc, DkdRowMap amap, DkdRowMap bmap,
                                                             int units_per_split, float* __restrict__ colsum, int tiles1) {
  __shared__ __attribute__((aligned(16))) char smem[TndCfg<2>::SMEM];
  tn192d_body<2, 3>(smem, A, B, C, M, N1, N2, lda, ldb, ldc, amap, bmap, units_per_split, colsum, tiles1, SWAP, blockIdx.x, gridDim.x);
}

// Up to 24 independent weight gradients in ONE launch (the two of an MLP, the four of a transformer block, those of several blocks):
// each alone is ~1.5 blocks per CU that all start and end together, so its ring fill and its atomic epilogue overlap nothing; side by
// side the blocks of one problem fill in while another's drain.  Block ranges are padded to multiples of 8 so blockIdx % 8 stays the XCD.
constexpr int TN_GROUP_MAX = 24;        // 24 x 96 B of kernel arguments (the limit is 4 KiB)
struct TnGroup {
  TnProb p[TN_GROUP_MAX];
  int n;
  // xsplits > 0: XCD x = blockIdx % 8 works on M split x % xsplits of the problems gstart[x / xsplits] .. gstart[x / xsplits + 1], one tile
  // per block in slot order: ALL tiles of a (problem, split) run on one XCD, so the rows of its 192-wide operand come from memory once.
  // (Round 3's per-problem remap spread a split's 6 tiles over two XCDs, 5 over two or three: the PMC counters of round 4 showed
  // 2.60 GB per six-block launch against 1.86 GB algorithmic -- those operands fetched about twice.)
  int xsplits, gstart[9];
};
template <int WR>
__global__ __launch_bounds__(128 * WR, WR == 2 ? 2 : 1) void gemm_tn192g_kernel(const TnGroup grp) {
  __shared__ __attribute__((aligned(16))) char smem[TndCfg<WR>::SMEM];
  if constexpr (WR == 2) {
    if (grp.xsplits > 0) {
      const int x = blockIdx.x & 7, gi = x / grp.xsplits;
      int k = grp.gstart[gi], rem = blockIdx.x >> 3;
      const int kend = grp.gstart[gi + 1];
      while (k < kend && rem >= grp.p[k].tiles1) {
        rem -= grp.p[k].tiles1;
        ++k;
      }
      if (k >= kend) return;
      const TnProb& q = grp.p[k];
      tn192d_body<2, 3>(smem, q.A, q.B, q.C, q.M, q.N1, q.N2, q.lda, q.ldb, q.ldc, q.amap, q.bmap, q.units_per_split, q.colsum, q.tiles1,
                        q.swap != 0, 0, 0, 0, rem, x % grp.xsplits);
      return;
    }
  }
  int bid = blockIdx.x, k = 0;
  while (k + 1 < grp.n && bid >= grp.p[k].n_blocks) {
    bid -= grp.p[k].n_blocks;
    ++k;
  }
  const TnProb& q = grp.p[k];
  if (bid >= q.n_blocks) return;
  if constexpr (WR == 2) {
    tn192d_body<2, 3>(smem, q.A, q.B, q.C, q.M, q.N1, q.N2, q.lda, q.ldb, q.ldc, q.amap, q.bmap, q.units_per_split, q.colsum, q.tiles1,
                      q.swap != 0, bid, q.n_blocks, k);
  } else {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4)
      tn192d_body<4, 2>(smem, q.A, q.B, q.C, q.M, q.N1, q.N2, q.lda, q.ldb, q.ldc, q.amap, q.bmap, q.units_per_split, q.colsum, q.tiles1,
                        q.swap != 0, bid, q.n_blocks, k);
    else
      tn192d_body<4, 1>(smem, q.A, q.B, q.C, q.M, q.N1, q.N2, q.lda, q.ldb, q.ldc, q.amap, q.bmap, q.units_per_split, q.colsum, q.tiles1,
                        q.swap != 0, bid, q.n_blocks, k);
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
  DKD_CHECK_ARG(!(g.epi & DKD_EPI_RELU_GATE) || g.preact, "gemm_nt: RELU_GATE without the gate (preact) pointer");
  DKD_CHECK_ARG((g.xb != nullptr) == (g.rowstats != nullptr), "gemm_nt: xb and rowstats come together (LayerNorm fold, producer side)");
  DKD_CHECK_ARG((g.ln_stats != nullptr) == (g.ln_c != nullptr), "gemm_nt: ln_stats and ln_c come together (LayerNorm fold, consumer side)");
  DKD_CHECK_ARG(!g.xb || (g.ldxb % 8 == 0 && ((uintptr_t)g.xb & 15) == 0 && ((uintptr_t)g.rowstats & 7) == 0), "gemm_nt: xb rows must be 16-byte aligned");
  DKD_CHECK_ARG(!(g.conv_hw > 0 && (g.xb || g.ln_stats)), "gemm_nt: no LayerNorm fold on the convolution path");
  if (g.conv_hw > 0) {
    DKD_CHECK_ARG(g.K % 9 == 0 && (g.K / 9) % 64 == 0 && g.lda >= g.K / 9 && g.amap.rpg == 0 && g.M % (g.conv_hw * g.conv_hw) == 0,
                  "gemm_nt(conv3x3): need K = 9 * Cin, Cin %% 64 == 0, identity A row map, M a multiple of hw^2 (K=%d M=%d hw=%d)", g.K, g.M,
                  g.conv_hw);
    int vec_ok_c = (g.N % 8 == 0) && (g.ldc % 8 == 0) && (((uintptr_t)g.C & 15) == 0);
    if (g.epi & DKD_EPI_BIAS) vec_ok_c = vec_ok_c && (((uintptr_t)g.bias & 15) == 0);
    if (g.preact) vec_ok_c = vec_ok_c && (g.ldp % 8 == 0) && (((uintptr_t)g.preact & 15) == 0);
    DkdProbeScope probe(0, 2.0 * g.M * g.N * g.K, 0.0, as_stream(stream));
    hipLaunchKernelGGL((gemm_nt_kernel<128, 0, true>), dim3(cdiv(g.M, BM) * cdiv(g.N, 128)), dim3(256), 0, as_stream(stream), g, vec_ok_c);
    DKD_CHECK_LAUNCH("gemm_nt(conv3x3)");
    return DKD_OK;
  }
  int vec_ok = (g.N % 8 == 0) && (g.ldc % 8 == 0) && (((uintptr_t)g.C & 15) == 0);
  if (g.epi & DKD_EPI_BIAS) vec_ok = vec_ok && (((uintptr_t)g.bias & 15) == 0);
  if (g.epi & DKD_EPI_RESID) vec_ok = vec_ok && (g.ldr % 8 == 0) && (((uintptr_t)g.resid & 15) == 0);
  if (g.preact) vec_ok = vec_ok && (g.ldp % 8 == 0) && (((uintptr_t)g.preact & 15) == 0);
  if (g.tap) vec_ok = vec_ok && (g.ldt % 8 == 0) && (((uintptr_t)g.tap & 15) == 0);
  // 128 x 64 tiles (48 KiB of LDS: three workgroups per CU) also for the two-output GELU epilogue of a short-K GEMM (the student's fc1:
  // K = 192, 154 MB of stores): measured 46 -> 43 us; the same tiles lose on the dGELU dgrad (56 -> 59) and on K = 768 (128 -> 138)
  const bool narrow = ((g.N % 128 != 0) && (g.N % 128 <= 64)) ||
                      (g.N % 64 == 0 && g.K <= 256 && g.epi == (DKD_EPI_BIAS | DKD_EPI_GELU) && g.preact != nullptr);
  const int tiles_m = cdiv(g.M, BM);
  // wide GEMMs with enough 256 x 256 tiles for >= 4 rounds on 256 CUs: qkv / fc1 of the teacher.  (The WN = 2 instance of the
  // kernel -- 256 x 128 tiles, two workgroups per CU -- was measured for the teacher's N = 768 GEMMs: proj 148 us, fc2 353 us
  // against 132 / 297 us for the 128 x 128 kernel below, so they stay there.)
  const long a_last = g.amap.rpg > 0 ? (long)((g.M - 1) / g.amap.rpg) * g.amap.gstride + (g.M - 1) % g.amap.rpg + g.amap.off : g.M - 1;
  const bool fits32 = (a_last + 1) * g.lda < (1L << 31) && (long)g.N * g.ldb < (1L << 31);   // its 32-bit source offsets
  const bool ring_ok = vec_ok && fits32 && g.K >= 512;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      dkd_set_error("gemm_nt: cannot query the device");
      return DKD_ERR_HIP;
    }
    n_cu = prop.multiProcessorCount & ~7;
    // (dev, DKD_CU_LIMIT=n: persistent kernels on n workgroups -- two half-batch teacher pipelines side by side on disjoint halves of
    // the chip, tools_dev/teacher_halves_probe.py)
    if (const char* lim = getenv("DKD_CU_LIMIT")) {
      const int v = atoi(lim) & ~7;
      if (v >= 8 && v < n_cu) n_cu = v;
    }
  }
  const long t256 = g.N % 256 == 0 ? (long)cdiv(g.M, 256) * (g.N / 256) : 0;
  bool wide = ring_ok && t256 >= 4 * n_cu;
  // N = 768 (teacher proj / fc2): 594 tiles of 256 x 256 are 2.3 rounds on 256 CUs.  The whole rounds go to the persistent kernel,
  // the remaining rows to the 128 x 128 kernel (a second call on the row range behind them).
  const bool fast3w = vec_ok && g.epi == (DKD_EPI_BIAS | DKD_EPI_RESID | DKD_EPI_OUT_F32) && !g.rowscale && g.amap.rpg == 0 &&
                      g.cmap.rpg == 0 && g.rmap.rpg == 0 && !g.preact && !(g.tap && (g.epi & DKD_EPI_TAP_F32));
  static const int split_k_min = getenv("DKD_SPLIT_KMIN") ? atoi(getenv("DKD_SPLIT_KMIN")) : 1536;     // (dev: A/B of proj on this path)
  if (!wide && ring_ok && fast3w && t256 >= 2 * n_cu && t256 < 4 * n_cu && g.K >= split_k_min) {   // fc2: 330 -> 305 us; proj (K = 768): no gain
    const int tn = g.N / 256;
    const int panels1 = (int)((t256 / n_cu) * n_cu / tn);           // whole rounds' worth of 256-row panels
    const int M1 = panels1 * 256;
    if (M1 > 0 && M1 < g.M) {
      DkdGemm g2 = g;
      g2.M = g.M - M1;
      g2.A = (const char*)g.A + (size_t)M1 * g.lda * 2;
      g2.C = (char*)g.C + (size_t)M1 * g.ldc * 4;
      g2.resid = g.resid + (size_t)M1 * g.ldr;
      if (g.tap) g2.tap = (char*)g.tap + (size_t)M1 * g.ldt * 2;
      if (g.xb) {
        g2.xb = (char*)g.xb + (size_t)M1 * g.ldxb * 2;
        g2.rowstats = g.rowstats + 2 * (size_t)M1;
      }
      DkdGemm g1 = g;
      g1.M = M1;
      {
        DkdProbeScope probe1(2, 2.0 * g1.M * g1.N * g1.K, 0.0, as_stream(stream));
        hipLaunchKernelGGL((gemm_nt256_kernel<0, 0, 4, 3>), dim3(n_cu), dim3(512), 0, as_stream(stream), g1, panels1 * tn);
        DKD_CHECK_LAUNCH("gemm_nt256");
      }
      return dkd_gemm_nt(&g2, stream);
    }
  }
  // LayerNorm fold: only the kernels that serve the wide teacher GEMMs implement it (the caller checks with the same shape rules:
  // deltakd_amd.vit.ln_fold_supported)
  if (g.ln_stats && !(wide && !(g.epi & (DKD_EPI_RESID | DKD_EPI_DGELU | DKD_EPI_OUT_F32)) && !g.tap && !g.preact && g.cmap.rpg == 0 && g.amap.rpg == 0 &&
                      !g.rowscale && (g.epi == DKD_EPI_BIAS || g.epi == (DKD_EPI_BIAS | DKD_EPI_GELU)))) {
    dkd_set_error("gemm_nt: LayerNorm fold (consumer) needs the wide-kernel path: N %% 256 == 0, >= 1024 tiles, K >= 512, bias (+ GELU) epilogue, bf16 output");
    return DKD_ERR_UNSUPPORTED;
  }
  if (g.xb && !((wide && fast3w) || (!wide && !narrow && fast3w && g.N % 128 == 0))) {
    dkd_set_error("gemm_nt: LayerNorm fold (producer) needs the f32 residual epilogue with identity row maps and N %% 128 == 0");
    return DKD_ERR_UNSUPPORTED;
  }
  DkdProbeScope probe(wide ? 2 : (narrow ? 1 : 0), 2.0 * g.M * g.N * g.K, 0.0, as_stream(stream));
  if (wide) {
#ifndef DKD_NT256_ABL
#define DKD_NT256_ABL 0
#endif
    // W beyond what an XCD's L2 keeps beside the A panels: split the column tiles over two XCD groups (see the kernel)
    static const int cg_env = getenv("DKD_NT256_CG") ? atoi(getenv("DKD_NT256_CG")) : 0;
    const int tn256 = g.N / 256;
    int cgr = (long)g.N * g.K * 2 > (3L << 20) && tn256 % 2 == 0 ? 2 : 1;
    if (cg_env > 0 && tn256 % cg_env == 0 && 8 % cg_env == 0) cgr = cg_env;
    const int n_tiles = (cdiv(g.M, 256) * tn256) | (cgr << 24);
    const dim3 grid(n_cu);               // persistent: one workgroup per CU, tiles dealt per XCD inside the kernel
    // bf16 C and nothing else written or read by the epilogue: 16 stores per wave per tile, counted exactly by the waits
    const bool plain16 = !(g.epi & (DKD_EPI_RESID | DKD_EPI_DGELU | DKD_EPI_OUT_F32)) && !g.tap && !g.preact;
    const bool ident = g.cmap.rpg == 0 && !g.rowscale && (g.epi & DKD_EPI_BIAS);
    const int fast = !(plain16 && ident) ? 0 : (g.epi == DKD_EPI_BIAS ? 1 : (g.epi == (DKD_EPI_BIAS | DKD_EPI_GELU) ? 2 : 0));
    // dev A/B (DKD_NT256_WN2=1): the bf16-epilogue GEMMs on 256 x 128 tiles, two 4-wave workgroups per CU (ring of 3 units each) -- one
    // workgroup's epilogue can run under the other's K loop, at 85 instead of 128 FLOP per byte through the LDS-DMA path
    static const int wn2_env = getenv("DKD_NT256_WN2") ? atoi(getenv("DKD_NT256_WN2")) : 0;
    if (wn2_env && (fast == 1 || fast == 2) && g.N % 128 == 0) {
      const int tn128 = g.N / 128;
      int cg2 = (long)g.N * g.K * 2 > (3L << 20) && tn128 % 2 == 0 ? 2 : 1;
      if (cg_env > 0 && tn128 % cg_env == 0 && 8 % cg_env == 0) cg2 = cg_env;
      const int nt2 = (cdiv(g.M, 256) * tn128) | (cg2 << 24);
      const dim3 grid2(2 * n_cu);
      if (fast == 1 && g.ln_stats) hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 0, 2, 1, true>), grid2, dim3(256), 0, as_stream(stream), g, nt2);
      else if (fast == 2 && g.ln_stats) hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 0, 2, 2, true>), grid2, dim3(256), 0, as_stream(stream), g, nt2);
      else if (fast == 1) hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 0, 2, 1>), grid2, dim3(256), 0, as_stream(stream), g, nt2);
      else hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 0, 2, 2>), grid2, dim3(256), 0, as_stream(stream), g, nt2);
      DKD_CHECK_LAUNCH("gemm_nt256 (256 x 128)");
      return DKD_OK;
    }
    if (fast == 1 && g.ln_stats) hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 24, 4, 1, true>), grid, dim3(512), 0, as_stream(stream), g, n_tiles);
    else if (fast == 2 && g.ln_stats) hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 24, 4, 2, true>), grid, dim3(512), 0, as_stream(stream), g, n_tiles);
    else if (g.ln_stats) {
      dkd_set_error("gemm_nt: LayerNorm fold (consumer): unsupported epilogue");
      return DKD_ERR_UNSUPPORTED;
    } else if (fast == 1) hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 16, 4, 1>), grid, dim3(512), 0, as_stream(stream), g, n_tiles);
    else if (fast == 2) hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 16, 4, 2>), grid, dim3(512), 0, as_stream(stream), g, n_tiles);
    else if (plain16) hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 16, 4, 0>), grid, dim3(512), 0, as_stream(stream), g, n_tiles);
    // f32 residual epilogue on the wide path: N = 768 (teacher proj / fc2) when >= 3 batches go through the teacher in one call
    else if (fast3w) hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 0, 4, 3>), grid, dim3(512), 0, as_stream(stream), g, n_tiles);
    else hipLaunchKernelGGL((gemm_nt256_kernel<DKD_NT256_ABL, 0, 4, 0>), grid, dim3(512), 0, as_stream(stream), g, n_tiles);
    DKD_CHECK_LAUNCH("gemm_nt256");
    return DKD_OK;
  }
  // compile-time epilogues for the most frequent plain cases (identity row maps, 16-byte aligned operands): see nt_epilogue
  const bool ident = vec_ok && g.amap.rpg >= 0 && g.cmap.rpg == 0 && !g.rowscale && !g.tap && !g.resid;
  int fast = 0;
  if (ident && g.N % 8 == 0) {
    if (g.epi == DKD_EPI_BIAS && !g.preact) fast = 1;
    else if (g.epi == (DKD_EPI_BIAS | DKD_EPI_GELU) && g.preact) fast = 4;
    else if (g.epi == DKD_EPI_DGELU) fast = 5;
    else if (g.epi == 0 && !g.preact) fast = 6;
  }
#define NT_LAUNCH(BN_, F_) hipLaunchKernelGGL((gemm_nt_kernel<BN_, F_>), dim3(tiles_m * cdiv(g.N, BN_)), dim3(256), 0, as_stream(stream), g, vec_ok)
  const bool fast7 = vec_ok && g.N % 8 == 0 && g.epi == (DKD_EPI_BIAS | DKD_EPI_RESID | DKD_EPI_OUT_F32) && g.rowscale && g.amap.rpg >= 0 &&
                     g.cmap.rpg == 0 && g.rmap.rpg == 0 && !g.preact && !(g.tap && (g.epi & DKD_EPI_TAP_F32));
  if (narrow) {
    if (fast7) NT_LAUNCH(64, 7);
    else if (fast == 1) NT_LAUNCH(64, 1);
    else if (fast == 4) NT_LAUNCH(64, 4);
    else if (fast == 5) NT_LAUNCH(64, 5);
    else if (fast == 6) NT_LAUNCH(64, 6);
    else NT_LAUNCH(64, 0);
  } else {
    const bool fast3 = vec_ok && g.N % 128 == 0 && g.epi == (DKD_EPI_BIAS | DKD_EPI_RESID | DKD_EPI_OUT_F32) && !g.rowscale &&
                       g.cmap.rpg == 0 && g.rmap.rpg == 0 && !g.preact && !(g.tap && (g.epi & DKD_EPI_TAP_F32));
    if (fast3) NT_LAUNCH(128, 3);
    else if (fast7 && g.N % 128 == 0) NT_LAUNCH(128, 7);
    else if (fast == 1) NT_LAUNCH(128, 1);
    else if (fast == 4) NT_LAUNCH(128, 4);
    else if (fast == 5) NT_LAUNCH(128, 5);
    else if (fast == 6) NT_LAUNCH(128, 6);
    else NT_LAUNCH(128, 0);
  }
#undef NT_LAUNCH
  DKD_CHECK_LAUNCH("gemm_nt");
  return DKD_OK;
}

namespace {
// kernel-order description of one weight gradient for the LDS-DMA ring kernel, or false when it must take the other kernels
bool tn192d_plan(const DkdTnProblem& q, TnProb* out, int min_tiles, int target_blocks = 384, int aw = 128) {
  const bool wide_b = q.N2 > 128 && q.N2 <= 192;
  const bool wide_a = !wide_b && q.N1 > 128 && q.N1 <= 192 && q.N2 > 192;
  if (!(wide_b || wide_a) || q.N1 % 8 || q.N2 % 8 || q.M < 32 * 16) return false;
  if ((q.lda % 8) || (q.ldb % 8) || ((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15)) return false;
  const int t1 = cdiv(wide_b ? q.N1 : q.N2, aw);
  if (t1 < min_tiles) return false;
  const int U = cdiv(q.M, 32);
  int sp = cdiv(target_blocks, t1);    // alone: 1.5 blocks per CU -- fewer partial tiles to add atomically than at 2 (measured 256..768)
  if (sp > cdiv(U, 8)) sp = cdiv(U, 8);
  const int per = cdiv(U, sp);
  sp = cdiv(U, per);
  TnProb p;
  p.A = (const bf16_t*)(wide_b ? q.A : q.B);
  p.B = (const bf16_t*)(wide_b ? q.B : q.A);
  p.C = q.C;
  p.colsum = q.a_colsum;
  p.M = q.M;
  p.N1 = wide_b ? q.N1 : q.N2;
  p.N2 = wide_b ? q.N2 : q.N1;
  p.lda = wide_b ? q.lda : q.ldb;
  p.ldb = wide_b ? q.ldb : q.lda;
  p.ldc = q.ldc;
  p.amap = wide_b ? q.amap : q.bmap;
  p.bmap = wide_b ? q.bmap : q.amap;
  p.units_per_split = per;
  p.tiles1 = t1;
  p.swap = wide_b ? 0 : 1;
  p.n_blocks = (t1 * sp + 7) & ~7;
  *out = p;
  return true;
}
}  // namespace

extern "C" int dkd_gemm_nt_lnbwd(const void* A, const void* W, int32_t M, int32_t K, int32_t lda, int32_t ldb, const float* x, int32_t ldx,
                                 const float* gamma, const float* mean, const float* rstd, float* dx, int32_t lddx, float* dgamma,
                                 float* dbeta, float* ws, void* cast_out, const float* rowscale, int32_t rows_per_sample, void* stream) {
  DKD_CHECK_ARG(A && W && x && gamma && mean && rstd && dx && dgamma && dbeta && ws, "gemm_nt_lnbwd: null operand");
  DKD_CHECK_ARG(M > 0 && K > 0 && K % BK == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldx % 4 == 0 && lddx % 4 == 0,
                "gemm_nt_lnbwd: K=%d must be a multiple of %d, row strides 16-byte multiples", K, BK);
  DKD_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dx & 15) == 0 &&
                    ((uintptr_t)gamma & 15) == 0 && (!cast_out || ((uintptr_t)cast_out & 7) == 0),
                "gemm_nt_lnbwd: operands must be 16-byte aligned");
  DKD_CHECK_ARG(!rowscale || rows_per_sample > 0, "gemm_nt_lnbwd: rowscale needs rows_per_sample");
  DKD_CHECK_ARG((long)M * lda < (1L << 31), "gemm_nt_lnbwd: A too large for 32-bit offsets");
  LnBwdEpi ln;
  ln.x = x; ln.gamma = gamma; ln.mean = mean; ln.rstd = rstd; ln.dx = dx; ln.part = ws;
  ln.cast_out = (bf16_t*)cast_out; ln.rowscale = rowscale; ln.ldx = ldx; ln.lddx = lddx; ln.rows_per_sample = rows_per_sample;
  const int nblk = cdiv(M, LNB_BM);
  {
    DkdProbeScope probe(0, 2.0 * M * LNB_D * K, 0.0, as_stream(stream));
    hipLaunchKernelGGL(gemm_nt_lnbwd_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), (const bf16_t*)A, (const bf16_t*)W, M, K, lda, ldb,
                       ln);
    DKD_CHECK_LAUNCH("gemm_nt_lnbwd");
  }
  return dkd_ln_bwd_reduce(ws, nblk, dgamma, dbeta, LNB_D, stream);
}

namespace {
// Tile columns the ring kernel would cut problem q into, or 0 when it does not take the shape (same test as tn192d_plan).
int tn192d_tiles(const DkdTnProblem& q, int aw) {
  const bool wide_b = q.N2 > 128 && q.N2 <= 192;
  const bool wide_a = !wide_b && q.N1 > 128 && q.N1 <= 192 && q.N2 > 192;
  if (!(wide_b || wide_a) || q.N1 % 8 || q.N2 % 8 || q.M < 32 * 16) return 0;
  if ((q.lda % 8) || (q.ldb % 8) || ((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15)) return 0;
  return cdiv(wide_b ? q.N1 : q.N2, aw);
}

int tn_group_launch(const DkdTnProblem* probs, int n, void* stream) {
  // Every tile of every problem gets the SAME number of M splits, chosen so that the launch is one round of the 512 workgroup slots
  // (2 per CU): all blocks then stream the same number of 32-row units and end together.  (Round 2 gave every PROBLEM the same block
  // count: the blocks of a 6-tile gradient ran 3x as long as those of a 2-tile one, and for the second half of the launch a CU held
  // less than one block.)  With the gradients of several transformer blocks in one launch (dkd_block_wgrad_group) the tiles
  // themselves supply the parallelism: 114 tiles -> 4 splits, i.e. a quarter of the atomically added partial tiles per gradient.
  static const int slots_env = getenv("DKD_TN_GROUP_SLOTS") ? atoi(getenv("DKD_TN_GROUP_SLOTS")) : 0;       // (dev: A/B)
  static const int blocks_env = getenv("DKD_TN_GROUP_BLOCKS") ? atoi(getenv("DKD_TN_GROUP_BLOCKS")) : 0;    // (dev: round 2's rule)
  // DKD_TN_GROUP_WIDE=1 (dev, A/B): 256 x 192 tiles, one 8-wave workgroup per CU -- 28 KiB of LDS-DMA per unit for twice the FLOPs of
  // the 128-row tile's 20 KiB.  Measured (tools_dev/wgrad_bench.py, six blocks = 24 problems): 560 us either way at 8 splits, 860 us at
  // 4 (one workgroup per CU): the launch already runs at the rate a CU gathers rows that come from HBM (~24 GB/s per CU,
  // MI355X_MICROARCH.md "Indexed rows"), which the tile shape does not change, so the 128-row tiles (two workgroups per CU) stay.
  static const int wide_env = getenv("DKD_TN_GROUP_WIDE") ? atoi(getenv("DKD_TN_GROUP_WIDE")) : 0;
  const bool wide = wide_env && n >= 4;
  const int aw = wide ? 256 : 128;
  int total_tiles = 0;
  for (int i = 0; i < n; ++i) total_tiles += tn192d_tiles(probs[i], aw);
  int splits = total_tiles > 0 ? (slots_env > 0 ? slots_env : 512) / total_tiles : 1;
  if (splits < 1) splits = 1;
  TnGroup grp;
  grp.n = 0;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const DkdTnProblem& q = probs[i];
    DKD_CHECK_ARG(q.A && q.B && q.C && q.M > 0 && q.N1 > 0 && q.N2 > 0, "gemm_tn_group: bad problem %d", i);
    const int t1 = tn192d_tiles(q, aw);
    if (t1 > 0 && tn192d_plan(q, &grp.p[grp.n], 1, blocks_env > 0 ? blocks_env : t1 * splits, aw)) {
      total += grp.p[grp.n].n_blocks;
      ++grp.n;
    } else {                            // shapes the ring kernel does not take: launched on their own
      int rc = dkd_gemm_tn(q.A, q.B, q.C, q.M, q.N1, q.N2, q.lda, q.ldb, q.ldc, q.amap, q.bmap, q.a_colsum, stream);
      if (rc != DKD_OK) return rc;
    }
  }
  grp.xsplits = 0;
  static const int xsplit_env = getenv("DKD_TN_GROUP_XSPLIT") ? atoi(getenv("DKD_TN_GROUP_XSPLIT")) : 1;      // (0: round 3's mapping, A/B)
  if (grp.n > 0 && !wide && xsplit_env && blocks_env == 0 && (splits == 1 || splits == 2 || splits == 4 || splits == 8)) {
    // one XCD per (problem, M split): possible when every problem really got `splits` splits and the problems divide into 8 / splits
    // groups of about equal tile count
    const int G = 8 / splits;
    bool ok = grp.n >= G;
    int tiles_all = 0;
    for (int i = 0; i < grp.n && ok; ++i) {
      const TnProb& q = grp.p[i];
      ok = cdiv(cdiv(q.M, 32), q.units_per_split) == splits;
      tiles_all += q.tiles1;
    }
    if (ok) {
      int k = 0, acc = 0, worst = 0;
      for (int gi = 0; gi < G; ++gi) {
        grp.gstart[gi] = k;
        const int want = (int)((long)tiles_all * (gi + 1) / G);
        int mine = 0;
        while (k < grp.n && (gi == G - 1 || acc + grp.p[k].tiles1 / 2 < want || mine == 0) && grp.n - k > G - 1 - gi) {
          acc += grp.p[k].tiles1;
          mine += grp.p[k].tiles1;
          ++k;
        }
        worst = mine > worst ? mine : worst;
      }
      grp.gstart[G] = grp.n;
      for (int gi = G + 1; gi < 9; ++gi) grp.gstart[gi] = grp.n;
      if (k == grp.n && worst * 8 <= (slots_env > 0 ? slots_env : 512) + 64) {
        grp.xsplits = splits;
        total = worst * 8;
      }
    }
  }
  if (grp.n > 0) {
    if (wide) hipLaunchKernelGGL(gemm_tn192g_kernel<4>, dim3(total), dim3(512), 0, as_stream(stream), grp);
    else hipLaunchKernelGGL(gemm_tn192g_kernel<2>, dim3(total), dim3(256), 0, as_stream(stream), grp);
    DKD_CHECK_LAUNCH("gemm_tn_group");
  }
  return DKD_OK;
}
}  // namespace

extern "C" int dkd_gemm_tn_group(const DkdTnProblem* probs, int32_t n, void* stream) {
  DKD_CHECK_ARG(probs && n > 0 && n <= TN_GROUP_MAX, "gemm_tn_group: need 1..%d problems (n=%d)", TN_GROUP_MAX, n);
  return tn_group_launch(probs, n, stream);
}

extern "C" int dkd_block_wgrad_group(const DkdTnProblem* probs, int32_t n, void* stream) {
  DKD_CHECK_ARG(probs && n > 0 && n <= TN_GROUP_MAX, "block_wgrad_group: need 1..%d problems (n=%d)", TN_GROUP_MAX, n);
  DkdProbeScope probe(3, 0.0, 0.0, as_stream(stream));      // time of the student block backward; its FLOPs are counted by dkd_block_bwd
  return tn_group_launch(probs, n, stream);
}

// M splits of gemm_tn_kernel: the launch is `blocks_per_split x splits` workgroups on 512 slots (2 per CU); its time is
// rounds x (steps per block + the block's atomic epilogue, worth ~12 steps).  The old rule (ceil(512 / tiles)) put 36 tiles on 540
// blocks = two rounds, the second 5 % full (768 x 768: 211 -> 184 us).
static void tn_pick_splits(const int blocks_per_split, const int KT, int& splits, int& per) {
  const int smax = cdiv(KT, 4) < 64 ? cdiv(KT, 4) : 64;
  long best = -1;
  for (int s = 1; s <= (smax < 1 ? 1 : smax); ++s) {
    const int p = cdiv(KT, s);
    if (cdiv(KT, p) != s) continue;                      // an empty split
    const long rounds = cdiv(blocks_per_split * s, 512);
    const long t = rounds * (p + 12);
    if (best < 0 || t < best) {
      best = t;
      splits = s;
      per = p;
    }
  }
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
    // The LDS-DMA ring kernel needs 16-byte column granules; with fewer than 3 tile columns its blocks are too short (8 units)
    // to amortise the 3-unit ring fill and the register-staged kernel below is faster (192 x 192: 44 vs 54 us) -- alone; in a
    // group launch (dkd_gemm_tn_group) the other problems' blocks cover that.
    DkdTnProblem q = {A, B, C, a_colsum, M, N1, N2, lda, ldb, ldc, amap, bmap};
    TnProb pl;
    if (tn192d_plan(q, &pl, 3)) {
      const dim3 grid(pl.tiles1 * cdiv(cdiv(M, 32), pl.units_per_split));
      if (!pl.swap)
        hipLaunchKernelGGL(gemm_tn192d_kernel<false>, grid, dim3(256), 0, as_stream(stream), pl.A, pl.B, pl.C, pl.M, pl.N1, pl.N2, pl.lda,
                           pl.ldb, pl.ldc, pl.amap, pl.bmap, pl.units_per_split, pl.colsum, pl.tiles1);
      else
        hipLaunchKernelGGL(gemm_tn192d_kernel<true>, grid, dim3(256), 0, as_stream(stream), pl.A, pl.B, pl.C, pl.M, pl.N1, pl.N2, pl.lda,
                           pl.ldb, pl.ldc, pl.amap, pl.bmap, pl.units_per_split, pl.colsum, pl.tiles1);
      DKD_CHECK_LAUNCH("gemm_tn192d");
      return DKD_OK;
    }
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
  int splits = 1, per = KT;
  tn_pick_splits(tiles, KT, splits, per);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), 0, as_stream(stream), (const bf16_t*)A, (const bf16_t*)B, C, M,
                     N1, N2, lda, ldb, ldc, amap, bmap, per, a_colsum, 0);
  DKD_CHECK_LAUNCH("gemm_tn");
  return DKD_OK;
}

extern "C" int dkd_conv3x3_wgrad(const void* dY, const void* X, float* dW, float* dbias, int32_t B, int32_t hw, int32_t Cin, int32_t Cout,
                                 void* stream) {
  DKD_CHECK_ARG(dY && X && dW && B > 0 && hw > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad: bad arguments");
  DKD_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0, "conv3x3_wgrad: channel counts must be multiples of 8 (Cin=%d Cout=%d)", Cin, Cout);
  DKD_CHECK_ARG(((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0, "conv3x3_wgrad: operands must be 16-byte aligned");
  const int M = B * hw * hw, KT = cdiv(M, 64);
  const int tiles = cdiv(Cout, 128) * cdiv(Cin, 128);
  // dW[o][tap][c] += sum_m dY[m][o] * X[m + shift(tap)][c]: one [Cout, Cin] block per tap, the nine taps side by side in ONE launch
  // (grid.z)
  // ~1500 workgroups in all, so few M-splits (= few atomically added partial tiles) per tap.  (Measured: 3 splits = two full rounds
  // instead of 5 = 3.2 rounds changes nothing -- the kernel is bound by the fabric traffic of its operand strips, not by rounds.)
  int splits = cdiv(1536, 9 * tiles);
  if (splits > cdiv(KT, 4)) splits = cdiv(KT, 4);
  if (splits < 1) splits = 1;
  const int per = cdiv(KT, splits);
  splits = cdiv(KT, per);
  const DkdRowMap id = {0, 0, 0};
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits, 9), dim3(256), 0, as_stream(stream), (const bf16_t*)dY, (const bf16_t*)X, dW, M,
                     Cout, Cin, Cout, Cin, 9 * Cin, id, id, per, dbias, 0, hw, 0, 0);
  DKD_CHECK_LAUNCH("conv3x3_wgrad");
  return DKD_OK;
}

extern "C" int dkd_gram(const void* A, float* C, int32_t M, int32_t N, int32_t lda, int32_t ldc, DkdRowMap amap, void* stream) {
  DKD_CHECK_ARG(A && C && M > 0 && N > 0, "gram: bad operand");
  DKD_CHECK_ARG(lda % 8 == 0 && ((uintptr_t)A & 15) == 0, "gram: rows of A must be 16-byte aligned (lda=%d)", lda);
  const int KT = cdiv(M, 64), T = cdiv(N, 128);
  const int tiles = T * (T + 1) / 2;   // tile pairs t1 <= t2
  int splits = 512 / tiles;            // at most ONE round of 2 workgroups per CU (rounding up left a 13-block second round at 768)
  if (splits < 1) splits = 1;
  if (splits > cdiv(KT, 4)) splits = cdiv(KT, 4);
  if (splits < 1) splits = 1;
  const int per = cdiv(KT, splits);
  splits = cdiv(KT, per);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), 0, as_stream(stream), (const bf16_t*)A, (const bf16_t*)A, C, M, N, N,
                     lda, lda, ldc, amap, amap, per, (float*)nullptr, 1);
  DKD_CHECK_LAUNCH("gram");
  return DKD_OK;
}

// L Gram matrices of one shape in ONE launch (the LRKD targets' three teacher taps): A_l = A + l stride_a (elements), C_l = C + l
// stride_c.  The tile pairs of all L problems share the launch's one round of workgroups, so every workgroup walks an L times longer
// K range and the f32 atomics of the partial tiles shrink L-fold (three launches: 3 x 504 partial tiles of 64 KB; one: 504).
extern "C" int dkd_gram_batched(const void* A, int64_t stride_a, float* C, int64_t stride_c, int32_t L, int32_t M, int32_t N, int32_t lda,
                                int32_t ldc, DkdRowMap amap, void* stream) {
  DKD_CHECK_ARG(A && C && M > 0 && N > 0 && L > 0 && L <= 64, "gram_batched: bad operand");
  DKD_CHECK_ARG(lda % 8 == 0 && ((uintptr_t)A & 15) == 0 && stride_a % 8 == 0, "gram_batched: rows of A must be 16-byte aligned (lda=%d)", lda);
  const int KT = cdiv(M, 64), T = cdiv(N, 128);
  const int tiles = T * (T + 1) / 2;
  int splits = 512 / (tiles * L);
  if (splits < 1) splits = 1;
  if (splits > cdiv(KT, 4)) splits = cdiv(KT, 4);
  if (splits < 1) splits = 1;
  const int per = cdiv(KT, splits);
  splits = cdiv(KT, per);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits, L), dim3(256), 0, as_stream(stream), (const bf16_t*)A, (const bf16_t*)A, C, M, N, N,
                     lda, lda, ldc, amap, amap, per, (float*)nullptr, 1, 0, 0, 0, (long)stride_a, (long)stride_c);
  DKD_CHECK_LAUNCH("gram_batched");
  return DKD_OK;
}

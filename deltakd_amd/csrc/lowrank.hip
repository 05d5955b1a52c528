// LRKD target chain on the device (model/loss.py:318-324 of the reference: U_k S_k of the [B*196, Dt] teacher matrix).
//
// The tall matrix is never factorised: U_k S_k = T V_k, V_k = the k leading eigenvectors of the Gram matrix G = T^T T (dkd_gram).
// This file tracks those eigenvectors by block subspace iteration with a block of b = 96 > k vectors, everything in fp32.
// The tracking step run once per batch (mode 1; V = last batch's basis, orthonormal, columns ~ eigenvectors in descending order):
//     Y = G V                             power step                            (sgemm_kernel<2>: G symmetric, upper 128-tiles given)
//     [S ; H] = [Y | V]^T Y               S = Y^T Y,  H = V^T G V               (sgemm_kernel<1>, one launch, 192 x 96)
//     one workgroup per matrix, LDS-resident (lowrank_small_kernel):
//         H = W E W^T      Jacobi, AT MOST `ritz_sweeps` sweeps: rotations are exactly orthogonal, so stopping early costs
//                          Ritz accuracy inside near-degenerate clusters only, never orthonormality; the basis carries over to the
//                          next batch, so the diagonalisation of the slowly changing H is continued there.  Columns sorted by E.
//         S' = W^T S W,  S' = D L L^T D (Cholesky of the column-scaled matrix),  C = W D^-1 L^-T
//     V = Y C              = orth(G V W), Gram-Schmidt in Ritz order (orthogonal iteration with Ritz acceleration); optionally also
//                          V_k^T as a bf16 hi/lo pair for the projection GEMMs       (sgemm_kernel<0>)
// i.e. 4 launches + one strided copy per batch for all layers together, instead of ~45 rocBLAS / ATen launches and two 10-sweep
// Jacobi runs.  A cold start (modes 2, 0 x 16, 2, 3) orthonormalises through the eigen-decomposition of Y^T Y with clamped eigenvalues
// (robust for the arbitrarily conditioned first iterates) and ends with a fully converged Rayleigh-Ritz step.
#include "common.h"

namespace {

constexpr int LB = 96;          // subspace block
constexpr int LLD = LB + 1;     // LDS row stride of the 96 x 96 working matrices (odd: row and column walks conflict-free)

// ------------------------------------------------------------------------------------------------ fp32 GEMM (small problems)
// C[M x N] = op(A)[M x K] * B[K x N] per layer (blockIdx.z); tile TM x 96 (TM = 32 or 16: more workgroups for the tall problems, whose
// grids are otherwise a few dozen blocks), 256 threads, each TM / 8 rows x 3 columns.
//   MODE 0: A row-major [M x K];  MODE 1: A stored [K x M] (C = A^T B);  MODE 2: A = symmetric [M x M] of which only the 128 x 128
//   tiles on and above the diagonal are valid (what dkd_gram writes): element (m, k) of a tile below the diagonal is read as (k, m).
// M, K multiples of 32, N a multiple of 96 (checked on the host).  gridDim.x > N / 96 splits K: slice blockIdx.x / (N / 96) of K is
// ADDED atomically to C, which the caller zeroes first (these problems are tiny -- 18 to 144 output tiles -- and latency-bound on
// their 24-step K loops otherwise).
// hi / lo (MODE 0 only, may be null): bf16 split of the first `rank` output columns, stored TRANSPOSED [rank x M] per layer:
//   hi = bf16(v), lo = bf16(v - hi) -- the B operands of the projection GEMMs T V_k (two bf16 MFMA passes keep ~16 bits of V).
template <int MODE, int TM>
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, long strideA, int lda, const float* __restrict__ B,
                                                    long strideB, int ldb, float* __restrict__ C, long strideC, int ldc, int M, int N,
                                                    int K, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int rank) {
  constexpr int RM = TM / 8;
  __shared__ float As[32][TM + 1];
  __shared__ float Bs[32][96];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int ntile = N / 96, ksplit = gridDim.x / ntile;
  const int n0 = (blockIdx.x % ntile) * 96, m0 = blockIdx.y * TM, layer = blockIdx.z;
  const int kper = K / ksplit, kbeg = (blockIdx.x / ntile) * kper;
  A += (size_t)layer * strideA;
  B += (size_t)layer * strideB;
  C += (size_t)layer * strideC;
  float acc[RM][3];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = 0.f;
  for (int k0 = kbeg; k0 < kbeg + kper; k0 += 32) {
    bool by_rows = MODE == 0;                       // coalesce the A tile along k (row-major source) or along m
    if (MODE == 2) by_rows = (m0 >> 7) <= (k0 >> 7);
#pragma unroll
    for (int e = 0; e < TM / 8; ++e) {
      const int idx = tid + 256 * e;
      if (by_rows) {
        const int kk = idx & 31, mm = idx >> 5;
        As[kk][mm] = A[(size_t)(m0 + mm) * lda + k0 + kk];
      } else {                                       // A^T, or the mirrored tile of the symmetric matrix
        const int mm = idx % TM, kk = idx / TM;
        As[kk][mm] = A[(size_t)(k0 + kk) * lda + m0 + mm];
      }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      const int idx = tid + 256 * e;
      const int kk = idx / 96, nn = idx % 96;
      Bs[kk][nn] = B[(size_t)(k0 + kk) * ldb + n0 + nn];
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      float a[RM], b[3];
#pragma unroll
      for (int i = 0; i < RM; ++i) a[i] = As[kk][ty * RM + i];
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = Bs[kk][tx + 32 * j];
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int m = m0 + ty * RM + i, n = n0 + tx + 32 * j;
      const float v = acc[i][j];
      if (ksplit > 1) {
        atomicAdd(&C[(size_t)m * ldc + n], v);
        continue;
      }
      C[(size_t)m * ldc + n] = v;
      if (MODE == 0 && hi != nullptr && n < rank) {
        const bf16_t h = f2bf(v);
        const size_t o = ((size_t)layer * rank + n) * M + m;
        hi[o] = h;
        lo[o] = f2bf(v - bf2f(h));
      }
    }
}

// ------------------------------------------------------------------------------------------------ LDS-resident Jacobi
// Cyclic two-sided Jacobi on the symmetric ne x ne matrix A (LDS, ODD row stride ld); V receives the eigenvectors (columns), the
// eigenvalues end up on the diagonal of A.  A sweep is ne - 1 rounds of a round-robin tournament; a round rotates ne / 2 disjoint
// index pairs at once in three LDS passes (rows of A; columns of A; columns of V).  A round is bound by LDS instruction issue, so the
// passes are laid out conflict-free: a half-wave of 32 lanes walks 32 consecutive columns of the two rows of a pair (row pass) or 32
// consecutive rows of its two columns (stride ld, odd), and keeps the pair's rotation in registers meanwhile.  (A one-pass variant
// -- one thread per 2 x 2 block (pair i) x (pair j) -- has fewer barriers but reads A[p_i][p_j] at tournament-permuted addresses:
// 3-way bank conflicts on average made it 1.5x slower.)
// A pair is rotated while |a_pq| > max(rel_tol * sqrt|a_pp a_qq|, abs_tol * max|a_ii|); before every sweep the whole matrix is
// tested against that bound in one pass and the iteration stops when nothing is left to rotate.  Returns the sweeps run.
__device__ int jacobi_lds(float* A, float* V, int ne, int ld, float2* cs, int2* pq, int* flag, float* dmaxp, int max_sweeps,
                          float rel_tol, float abs_tol) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int half = ne >> 1;
  const int l32 = tid & 31, grp = tid >> 5, ngrp = nt >> 5;
  for (int i = tid; i < ne * ne; i += nt) V[(i / ne) * ld + (i % ne)] = (i / ne) == (i % ne) ? 1.f : 0.f;
  int sweeps = 0;
  for (int sw = 0; sw < max_sweeps; ++sw) {
    __syncthreads();
    if (tid < 64) {
      float m = 0.f;
      for (int i = tid; i < ne; i += 64) m = fmaxf(m, fabsf(A[i * ld + i]));
      m = wave_max(m);
      if (tid == 0) {
        *dmaxp = m;
        *flag = 0;
      }
    }
    __syncthreads();
    const float floor_abs = abs_tol * *dmaxp;
    bool mine = false;                                   // anything left above the bound?
    for (int w = tid; w < ne * ne; w += nt) {
      const int p = w / ne, q = w % ne;
      if (p < q) {
        const float apq = fabsf(A[p * ld + q]);
        mine = mine || apq > fmaxf(rel_tol * sqrtf(fabsf(A[p * ld + p] * A[q * ld + q])), floor_abs);
      }
    }
    if (mine) *flag = 1;
    __syncthreads();
    if (*flag == 0) break;
    ++sweeps;
    for (int rd = 0; rd < ne - 1; ++rd) {
      if (tid < half) {      // round-robin tournament: position 0 is fixed, the other ne - 1 positions rotate by rd
        const int i = tid;
        const int a = i == 0 ? 0 : 1 + (i - 1 + rd) % (ne - 1);
        const int b = 1 + (ne - 1 - i - 1 + rd) % (ne - 1);
        const int p = a < b ? a : b, q = a < b ? b : a;
        const float app = A[p * ld + p], aqq = A[q * ld + q], apq = A[p * ld + q];
        float c = 1.f, s = 0.f;
        if (fabsf(apq) > fmaxf(rel_tol * sqrtf(fabsf(app * aqq)), floor_abs) && apq != 0.f) {
          const float tau = (aqq - app) / (2.f * apq);
          const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
          c = rsqrtf(1.f + t * t);
          s = t * c;
        }
        cs[i] = make_float2(c, s);
        pq[i] = make_int2(p, q);
      }
      __syncthreads();
      for (int i = grp; i < half; i += ngrp) {             // rows: A <- J^T A
        const float2 r = cs[i];
        if (r.y == 0.f) continue;
        const int2 ii = pq[i];
        float* rp = A + ii.x * ld;
        float* rq = A + ii.y * ld;
        for (int k = l32; k < ne; k += 32) {
          const float x = rp[k], y = rq[k];
          rp[k] = r.x * x - r.y * y;
          rq[k] = r.y * x + r.x * y;
        }
      }
      __syncthreads();
      for (int i = grp; i < half; i += ngrp) {             // columns: A <- A J, V <- V J
        const float2 r = cs[i];
        if (r.y == 0.f) continue;
        const int2 ii = pq[i];
        for (int k = l32; k < ne; k += 32) {
          float* ra = A + k * ld;
          float x = ra[ii.x], y = ra[ii.y];
          ra[ii.x] = r.x * x - r.y * y;
          ra[ii.y] = r.y * x + r.x * y;
          float* rv = V + k * ld;
          x = rv[ii.x];
          y = rv[ii.y];
          rv[ii.x] = r.x * x - r.y * y;
          rv[ii.y] = r.y * x + r.x * y;
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
  return sweeps;
}

// ---- the public batched eigensolver (n <= 128): one workgroup per matrix
constexpr int NMAX = 128;
constexpr int LD = NMAX + 1;
__global__ __launch_bounds__(1024) void jacobi_eigh_kernel(const float* __restrict__ Ain, float* __restrict__ evals,
                                                           float* __restrict__ evecs, int n, int sweeps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* A = sm;                 // [ne][LD]
  float* V = A + NMAX * LD;
  float2* cs = (float2*)(V + NMAX * LD);     // (c, s)[64]
  int2* pq = (int2*)(cs + 64);               // (p, q)[64]
  int* flag = (int*)(pq + 64);
  float* dmaxp = (float*)(flag + 1);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ne = (n + 1) & ~1;   // even working size (a padded index gets a zero row / column and never rotates)
  const float* Ab = Ain + (size_t)blockIdx.x * n * n;
  for (int i = tid; i < ne * ne; i += nt) {
    const int r = i / ne, c = i % ne;
    A[r * LD + c] = (r < n && c < n) ? Ab[r * n + c] : 0.f;
  }
  __syncthreads();
  jacobi_lds(A, V, ne, LD, cs, pq, flag, dmaxp, sweeps, 2.4e-7f, 0.f);     // 4 ulp of sqrt(a_pp a_qq): fp32 roundoff
  for (int i = tid; i < n; i += nt) evals[(size_t)blockIdx.x * n + i] = A[i * LD + i];
  float* Vb = evecs + (size_t)blockIdx.x * n * n;
  for (int i = tid; i < n * n; i += nt) Vb[i] = V[(i / n) * LD + (i % n)];
}

// ------------------------------------------------------------------------------------------------ the 96 x 96 stage
// C = A B or A^T B on 96 x 96 matrices in LDS: 1024 threads, each a 3 x 3 block of rows {ty, ty+32, ty+64} x cols {tx, tx+32, tx+64}.
template <bool TRANS_A, bool TRANS_B = false, int LLD = LB + 1>
__device__ void mm96(const float* A, const float* B, float* C) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float acc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll 4
  for (int k = 0; k < LB; ++k) {
    float a[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = TRANS_A ? A[k * LLD + ty + 32 * i] : A[(ty + 32 * i) * LLD + k];
#pragma unroll
    for (int j = 0; j < 3; ++j) b[j] = TRANS_B ? B[(tx + 32 * j) * LLD + k] : B[k * LLD + tx + 32 * j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[(ty + 32 * i) * LLD + tx + 32 * j] = acc[i][j];
}

// In-place Cholesky of the SPD matrix A (lower triangle used; unit-scaled input): on return the lower triangle holds L with
// A = L L^T.  Right-looking, one barrier per column; a pivot that fp32 cancellation drove below `floor_` is clamped (the column then
// carries no weight instead of NaN).
template <int LLD = LB + 1>
__device__ void cholesky_lds(float* A, float floor_) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int k = 0; k < LB; ++k) {
    __syncthreads();
    const float piv = fmaxf(A[k * LLD + k], floor_);
    const float inv = 1.f / piv;
    const int n = LB - 1 - k;                       // trailing block: rows / cols k+1 .. 95, lower triangle
    for (int w = tid; w < n * n; w += nt) {
      const int i = k + 1 + w / n, j = k + 1 + w % n;
      if (j <= i) A[i * LLD + j] -= A[i * LLD + k] * A[j * LLD + k] * inv;
    }
  }
  __syncthreads();
  for (int w = tid; w < LB * LB; w += nt) {          // scale the columns: L[i][k] = A[i][k] / sqrt(piv_k); the diagonal last (others read it)
    const int i = w / LB, k = w % LB;
    if (i > k) A[i * LLD + k] *= rsqrtf(fmaxf(A[k * LLD + k], floor_));
  }
  __syncthreads();
  if (tid < LB) A[tid * LLD + tid] = sqrtf(fmaxf(A[tid * LLD + tid], floor_));
  __syncthreads();
}

// X = L^-1 (lower triangular) by forward substitution, one thread per column (the row of L being read is a broadcast).
template <int LLD = LB + 1>
__device__ void tri_inverse_lds(const float* Lm, float* X) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int w = tid; w < LB * LB; w += nt) X[(w / LB) * LLD + (w % LB)] = 0.f;
  __syncthreads();
  if (tid < LB) {
    const int j = tid;
    for (int i = j; i < LB; ++i) {
      float acc = i == j ? 1.f : 0.f;
      for (int m = j; m < i; ++m) acc = fmaf(-Lm[i * LLD + m], X[m * LLD + j], acc);
      X[i * LLD + j] = acc / Lm[i * LLD + i];
    }
  }
  __syncthreads();
}

// S: [L][192][96] (mode 1) or [L][96][96 of row stride 192] (modes 0, 3);  Cout: [L][96][96];  evals: [L][96] (Ritz values, descending) or
// null;  info: [L][2] Jacobi sweeps run (orthonormalisation, Rayleigh-Ritz).
// mode 0: S = Y^T Y                        -> C = D^-1 U E^-1/2            orthonormalising transform of Y by eigen-decomposition,
//                                                                           eigenvalues clamped at 1e-12 of the largest (cold start)
// mode 1: S = [Y^T Y ; V^T Y], Y = G V     -> C = W D^-1 L^-T              the tracking step (see the head of this file)
// mode 3: S = V^T G V, V orthonormal       -> C = W sorted                 Rayleigh-Ritz in span(V), run to convergence
__global__ __launch_bounds__(1024) void lowrank_small_kernel(const float* __restrict__ S, float* __restrict__ Cout,
                                                             float* __restrict__ evals, int* __restrict__ info, int mode,
                                                             int ritz_sweeps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* B0 = sm;
  float* B1 = B0 + LB * LLD;
  float* B2 = B1 + LB * LLD;
  float* B3 = B2 + LB * LLD;
  float* d = B3 + LB * LLD;        // [96]
  float* ev = d + LB;              // [96]
  float2* cs = (float2*)(ev + LB); // [64]
  int2* pq = (int2*)(cs + 64);     // [64]
  int* rnk = (int*)(pq + 64);      // [96]
  int* flag = rnk + LB;
  float* dmaxp = (float*)(flag + 1);
  const int tid = threadIdx.x, nt = blockDim.x;
  const float* Sl = S + (size_t)blockIdx.x * LB * 2 * LB;
  float* Cl = Cout + (size_t)blockIdx.x * LB * LB;
  int sw0 = 0, sw1 = 0;

  if (mode == 0) {
    if (tid < LB) d[tid] = rsqrtf(fmaxf(Sl[tid * 2 * LB + tid], 1e-30f));
    __syncthreads();
    for (int i = tid; i < LB * LB; i += nt) {
      const int r = i / LB, c = i % LB;
      B0[r * LLD + c] = 0.5f * (Sl[r * 2 * LB + c] + Sl[c * 2 * LB + r]) * d[r] * d[c];
    }
    __syncthreads();
    sw0 = jacobi_lds(B0, B1, LB, LLD, cs, pq, flag, dmaxp, 12, 2.4e-7f, 0.f);       // Sn = U E U^T
    if (tid < 64) {
      float m = 0.f;
      for (int i = tid; i < LB; i += 64) m = fmaxf(m, B0[i * LLD + i]);
      m = wave_max(m);
      if (tid == 0) *dmaxp = m;
    }
    __syncthreads();
    if (tid < LB) ev[tid] = rsqrtf(fmaxf(B0[tid * LLD + tid], 1e-12f * *dmaxp));
    __syncthreads();
    for (int i = tid; i < LB * LB; i += nt) Cl[i] = B1[(i / LB) * LLD + (i % LB)] * ev[i % LB] * d[i / LB];
    if (info && tid == 0) {
      info[2 * blockIdx.x] = sw0;
      info[2 * blockIdx.x + 1] = 0;
    }
    return;
  }
  // ---- Rayleigh quotient H of the orthonormal basis V: rows 96..191 of S (mode 1, row stride 96) or S itself (mode 3, row stride 192)
  const float* Hs = mode == 1 ? Sl + LB * LB : Sl;
  const int ldh = mode == 1 ? LB : 2 * LB;
  for (int i = tid; i < LB * LB; i += nt) {
    const int r = i / LB, c = i % LB;
    B0[r * LLD + c] = 0.5f * (Hs[r * ldh + c] + Hs[c * ldh + r]);
  }
  __syncthreads();
  // off-diagonals below 1e-7 of the largest Ritz value are fp32 noise of the GEMMs that built H: rotating on them never converges
  sw1 = jacobi_lds(B0, B1, LB, LLD, cs, pq, flag, dmaxp, ritz_sweeps, mode == 1 ? 1e-5f : 2.4e-7f, 1e-7f);      // H = W E W^T
  if (tid < LB) ev[tid] = B0[tid * LLD + tid];
  __syncthreads();
  if (tid < LB) {                              // rank of each Ritz value (descending, index breaks ties)
    const float e = ev[tid];
    int r = 0;
    for (int j = 0; j < LB; ++j) r += (ev[j] > e) || (ev[j] == e && j < tid);
    rnk[tid] = r;
    if (evals) evals[(size_t)blockIdx.x * LB + r] = e;
  }
  __syncthreads();
  if (mode == 3) {
    for (int i = tid; i < LB * LB; i += nt) Cl[(i / LB) * LB + rnk[i % LB]] = B1[(i / LB) * LLD + (i % LB)];
  } else {
    for (int i = tid; i < LB * LB; i += nt) {                                  // B2 = W, columns in Ritz order;  B3 = S (symmetrised)
      const int r = i / LB, c = i % LB;
      B2[r * LLD + rnk[c]] = B1[r * LLD + c];
      B3[r * LLD + c] = 0.5f * (Sl[r * LB + c] + Sl[c * LB + r]);
    }
    __syncthreads();
    mm96<false>(B3, B2, B0);                   // S W
    __syncthreads();
    mm96<true>(B2, B0, B1);                    // S' = W^T S W = (Y W)^T (Y W)
    __syncthreads();
    if (tid < LB) d[tid] = rsqrtf(fmaxf(B1[tid * LLD + tid], 1e-30f));
    __syncthreads();
    for (int i = tid; i < LB * LB; i += nt) {  // unit diagonal, lower triangle symmetrised
      const int r = i / LB, c = i % LB;
      if (c <= r) B0[r * LLD + c] = 0.5f * (B1[r * LLD + c] + B1[c * LLD + r]) * d[r] * d[c];
    }
    __syncthreads();
    cholesky_lds(B0, 1e-6f);                   // S'n = L L^T
    tri_inverse_lds(B0, B3);                   // B3 = L^-1
    for (int i = tid; i < LB * LB; i += nt) B2[(i / LB) * LLD + (i % LB)] *= d[i % LB];      // W D^-1
    __syncthreads();
    mm96<false, true>(B2, B3, B1);             // C = W D^-1 L^-T
    __syncthreads();
    for (int i = tid; i < LB * LB; i += nt) Cl[i] = B1[(i / LB) * LLD + (i % LB)];
  }
  if (info && tid == 0) {
    info[2 * blockIdx.x] = sw0;
    info[2 * blockIdx.x + 1] = sw1;
  }
}


// ================================================================================================ round 5: the converged chain
// dkd_lowrank_chain: n power steps per batch from the previous batch's basis, ending in a converged Rayleigh-Ritz step -- the setting
// that reproduces the reference's per-batch exact svd (model/loss.py:318-326) -- as a chain of SHORT launches (a long-lived 3-CU kernel
// holds every 256-workgroup kernel of the other streams back by its own duration: round 4's 8-step setting cost 12.8 ms of chain):
//     lr_mult_kernel      Y' = (G Y) C   16 rows of Y' per workgroup (K = Dt streamed through LDS, the 16 x Dt row panel of the
//                         symmetric G resident), the previous stage's 96 x 96 transform C applied to the finished tile -- an
//                         orthonormalised V = Y C is never materialised between steps: G (Y C) = (G Y) C -- and the tile's
//                         contribution to S = Y'^T Y' (and P = Y^T Y' in the last stage) added atomically
//     lr_orth_kernel      S -> C = D^-1 L^-T, Cholesky of the column-scaled S (order-preserving Gram-Schmidt)    one workgroup per layer
//     lr_ritz_kernel      last stage: H = C_prev^T P = V^T G V, Jacobi H = W E W^T, S' = W^T S W = D L L^T D, C = W D^-1 L^-T
//     sgemm_kernel<0>     V = Y' C, with the bf16 hi / lo split of V_k^T
// Schedule for n multiplies: 1, 2, 2, ..., 1 multiplies per stage -- two multiplies between orthonormalisations square the
// contamination by the dominant directions ((lambda_1 / lambda_j)^2 ~ 2e3 at the 64th vector of a deit_base tap: 1e5 in the
// condition number of the scaled S, which fp32 Cholesky takes), three do not fit (tools_dev/lowrank_proto_algo.py); the first stage
// stays single because there the basis is the previous batch's.
constexpr int LDJ = LB + 2;        // EVEN row stride: the float2 accesses of an index pair's two adjacent columns stay 8-byte aligned
constexpr int LR_ROWS = 16;
constexpr int LR_APPLY_C = 1, LR_GRAM_S = 2, LR_GRAM_P = 4;

// dev-only (-DDKD_LR_STAMPS, tools_dev/lowrank_stamps.py): s_memrealtime (100 MHz) at phase boundaries of workgroup 0 of each kernel
#ifdef DKD_LR_STAMPS
__device__ unsigned long long lr_stamp_buf[3][64];
#define LR_STAMP(kern_, n_)                                                                         \
  do {                                                                                              \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) lr_stamp_buf[kern_][n_] = wall_clock64(); \
  } while (0)
#else
#define LR_STAMP(kern_, n_) do { } while (0)
#endif

__global__ __launch_bounds__(256) void lr_mult_kernel(const float* __restrict__ G, const float* __restrict__ Bsrc, const float* __restrict__ Cg,
                                                      float* __restrict__ Yout, float* __restrict__ Sg, float* __restrict__ Pg, int Dt,
                                                      int flags) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lda = Dt + 4;
  const int as_floats = LR_ROWS * lda;
  float* As = sm;                   // [16][Dt + 4] row panel of G
  float* Bs = sm + as_floats;       // [2][32][96] K chunks of the right operand; later the tiles Zt | Yt | Pt [16][96]
  float* Cs = Bs + 2 * 32 * LB;     // [96][96] the previous stage's transform
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int m0 = blockIdx.x * LR_ROWS, layer = blockIdx.y;
  G += (size_t)layer * Dt * Dt;
  Bsrc += (size_t)layer * Dt * LB;
  Yout += (size_t)layer * Dt * LB;
  LR_STAMP(0, 0);
  if (flags & LR_APPLY_C) {          // (whole: 9 float4 per thread, consumed in the epilogue)
    const float* Cl = Cg + (size_t)layer * LB * LB;
    f32x4 cv[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) cv[e] = *(const f32x4*)&Cl[4 * (tid + 256 * e)];
#pragma unroll
    for (int e = 0; e < 9; ++e) *(f32x4*)&Cs[4 * (tid + 256 * e)] = cv[e];
  }
  // ---- the row panel: G is symmetric with only its 128 x 128 tiles on and above the diagonal valid (dkd_gram): columns left of this
  // row block's diagonal tile are read as the transposed element.  4 Dt float4 items per workgroup, requested EIGHT per thread at a
  // time before the first is stored (a load-store loop with a run-time trip count exposed one memory round trip per item: ~20 us).
  const int kdir = (m0 >> 7) << 7;
  {
    const int nk4 = (Dt - kdir) >> 2, nmir = kdir * 4, total = 4 * Dt;
    for (int base = 0; base < total; base += 256 * 8) {
      f32x4 v[8];
      int dst[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = base + tid + 256 * e;
        dst[e] = -1;
        if (idx < nmir) {
          const int k = idx >> 2, r4 = idx & 3;
          v[e] = *(const f32x4*)&G[(size_t)k * Dt + m0 + 4 * r4];
          dst[e] = (4 * r4) * lda + k;                     // transposed: four rows, one column
        } else if (idx < total) {
          const int id2 = idx - nmir, r = id2 / nk4, k4 = id2 - r * nk4;
          v[e] = *(const f32x4*)&G[(size_t)(m0 + r) * Dt + kdir + 4 * k4];
          dst[e] = (r * lda + kdir + 4 * k4) | (1 << 30);   // flag: a row segment
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (dst[e] < 0) continue;
        if (dst[e] & (1 << 30)) {
          *(f32x4*)&As[dst[e] & ~(1 << 30)] = v[e];
        } else {
          As[dst[e]] = v[e][0];
          As[dst[e] + lda] = v[e][1];
          As[dst[e] + 2 * lda] = v[e][2];
          As[dst[e] + 3 * lda] = v[e][3];
        }
      }
    }
  }
  // ---- K loop on the fp32 matrix cores.  Round 5's first form (one thread = 2 x 3 outputs, both operands through LDS) spent 30 us
  // here: 5 LDS reads per 6 multiply-adds at one wave per SIMD.  Now wave w multiplies the K range [w Dt / 4, (w + 1) Dt / 4) for the
  // whole 16 x 96 tile with v_mfma_f32_16x16x4_f32 (A = 16 rows x 4 k of the G panel, from LDS; B = 4 k x 16 columns of the right
  // operand, read from global memory straight into registers -- every wave needs different rows of it, so there is nothing to share
  // through LDS); the four partial tiles are added through LDS afterwards.  One 16-byte + one 8-byte load per lane and k row feed six
  // column tiles: tile t < 4 is the columns {4 c + t}, tiles 4, 5 the columns {64 + 2 c + (t - 4)}, c = lane % 16 (a permutation of the
  // columns, undone when the partial tiles are written).  A lane's four k rows of a group of 16 are k0 + 4 q + e (q = lane / 16), e = the
  // MFMA step: any four distinct rows per step will do as long as A and B agree.  Four groups (1 KiB per lane) are in flight.
  const int lane = tid & 63, wv = tid >> 6, q4 = lane >> 4, c16 = lane & 15;
  const int kw0 = wv * (Dt >> 2), ngrp = Dt >> 6;          // groups of 16 k per wave (Dt % 64 == 0: host-checked)
  f32x4 b4[4][4];
  dkd_f32x2 b2[4][4];
  const float* bsrc = Bsrc + (size_t)(kw0 + 4 * q4) * LB;
#define LR_FETCH(slot_, g_)                                                                         \
  if ((g_) < ngrp) {                                                                                \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                 \
      const float* r_ = bsrc + (size_t)((g_) * 16 + e) * LB;                                        \
      b4[slot_][e] = *(const f32x4*)&r_[4 * c16];                                                   \
      b2[slot_][e] = *(const dkd_f32x2*)&r_[64 + 2 * c16];                                          \
    }                                                                                               \
  }
  LR_FETCH(0, 0)
  LR_FETCH(1, 1)
  LR_FETCH(2, 2)
  LR_FETCH(3, 3)
  __syncthreads();                                           // the panel (and C) are in LDS
  LR_STAMP(0, 1);
  f32x4 macc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) macc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* arow = As + c16 * lda + kw0 + 4 * q4;
#define LR_STEP(slot_, g_)                                                                          \
  if ((g_) < ngrp) {                                                                                \
    const f32x4 a4 = *(const f32x4*)&arow[(g_) * 16];                                               \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                 \
      _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                 \
        macc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b4[slot_][e][t], macc[t], 0, 0, 0);  \
      macc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b2[slot_][e][0], macc[4], 0, 0, 0);     \
      macc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b2[slot_][e][1], macc[5], 0, 0, 0);     \
    }                                                                                               \
    LR_FETCH(slot_, (g_) + 4)                                                                       \
  }
  for (int g = 0; g < ngrp; g += 4) {
    LR_STEP(0, g)
    LR_STEP(1, g + 1)
    LR_STEP(2, g + 2)
    LR_STEP(3, g + 3)
  }
#undef LR_STEP
#undef LR_FETCH
  // the four waves' partial tiles -> LDS (column permutation undone), summed into the (ty, tx) layout the epilogue works in
  {
    float* red = Bs + wv * LR_ROWS * LB;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int col = t < 4 ? 4 * c16 + t : 64 + 2 * c16 + (t - 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) red[(4 * q4 + i) * LB + col] = macc[t][i];
    }
  }
  __syncthreads();
  float acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int o = (2 * ty + i) * LB + tx + 32 * j;
      acc[i][j] = (Bs[o] + Bs[LR_ROWS * LB + o]) + (Bs[2 * LR_ROWS * LB + o] + Bs[3 * LR_ROWS * LB + o]);
    }
  __syncthreads();                                           // (the tiles below reuse this LDS)
  LR_STAMP(0, 2);
  // ---- epilogue: the tile through the transform, out, and into the 96 x 96 Gram matrices
  float* Zt = Bs;                  // [16][96]
  float* Yt = Bs + LR_ROWS * LB;
  float* Pt = Yt + LR_ROWS * LB;
  float y[2][3];
  if (flags & LR_APPLY_C) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Zt[(2 * ty + i) * LB + tx + 32 * j] = acc[i][j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) y[i][j] = 0.f;
    const float* z0 = Zt + (2 * ty) * LB;
    const float* z1 = z0 + LB;
#pragma unroll 2
    for (int k = 0; k < LB; k += 4) {
      const f32x4 a0 = *(const f32x4*)&z0[k], a1 = *(const f32x4*)&z1[k];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float b = Cs[(k + u) * LB + tx + 32 * j];
          y[0][j] = fmaf(a0[u], b, y[0][j]);
          y[1][j] = fmaf(a1[u], b, y[1][j]);
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) y[i][j] = acc[i][j];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Yout[(size_t)(m0 + 2 * ty + i) * LB + tx + 32 * j] = y[i][j];
      Yt[(2 * ty + i) * LB + tx + 32 * j] = y[i][j];
    }
  if (!(flags & (LR_GRAM_S | LR_GRAM_P))) return;
  LR_STAMP(0, 3);
  if (flags & LR_GRAM_P)
    for (int idx = tid; idx < LR_ROWS * LB / 4; idx += 256) *(float4*)&Pt[4 * idx] = *(const float4*)&Bsrc[(size_t)m0 * LB + 4 * idx];
  __syncthreads();
  // The tile's contribution to the 96 x 96 Gram accumulators on the fp32 matrix cores: output tile (i0, j0) = Yt[:, i0..]^T Yt[:, j0..]
  // over the block's 16 rows = four 16x16x4 steps (A = the transposed tile: lane supplies Yt[k][i0 + lane % 16], B: Yt[k][j0 + lane % 16],
  // k = 4 step + lane / 16).  S is symmetric: only its tiles on and below the diagonal are formed and added (21 of 36; the consumers
  // read the lower triangle); P = Y_prev^T Y' needs all 36.  Tiles are dealt round-robin to the four waves.  (VALU form: 4.8 us.)
  const bool want_p = flags & LR_GRAM_P;
  LR_STAMP(0, 4);
  float* Sl = Sg + (size_t)layer * LB * LB;
  float* Pl = Pg + (size_t)layer * LB * LB;
  {
    int t = 0;
    for (int ti = 0; ti < 6; ++ti)
      for (int tj = 0; tj < 6; ++tj) {
        const bool do_s = (flags & LR_GRAM_S) && tj <= ti;
        if (!do_s && !want_p) continue;
        if ((t++ & 3) != wv) continue;
        f32x4 as = {0.f, 0.f, 0.f, 0.f}, ap = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int r = 4 * ks + q4;
          const float bj = Yt[r * LB + 16 * tj + c16];
          if (do_s) as = __builtin_amdgcn_mfma_f32_16x16x4f32(Yt[r * LB + 16 * ti + c16], bj, as, 0, 0, 0);
          if (want_p) ap = __builtin_amdgcn_mfma_f32_16x16x4f32(Pt[r * LB + 16 * ti + c16], bj, ap, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int o = (16 * ti + 4 * q4 + i) * LB + 16 * tj + c16;
          if (do_s) atomicAdd(&Sl[o], as[i]);
          if (want_p) atomicAdd(&Pl[o], ap[i]);
        }
      }
  }
  LR_STAMP(0, 5);
}

// ---- Cholesky and triangular inverse for the chain's stages (1024 threads = a 32 x 32 grid of 3 x 3 blocks, row stride LDJ).
// The first forms above spend their time in integer divisions (every work item decodes (i, j) from a flat index by a run-time
// extent: ~100 us for 96 columns) and in a one-thread-per-column substitution whose every step is an LDS round trip (~55 us); a
// fixed 32 x 32 thread grid with one barrier per column still took 48 us -- 96 rounds of read-modify-write passes over the trailing
// matrix in LDS, issue-bound (s_memrealtime stamps, tools_dev/lowrank_stamps.py).
// Cholesky, third form: the trailing matrix lives in REGISTERS -- thread (ty, tx), ty >= tx, owns the 3 x 3 block of rows 3 ty.. and
// columns 3 tx.. for the whole factorisation; a step eliminates one block column: its owners publish their current blocks in a
// two-deep LDS panel buffer (one barrier per step, 32 steps), every thread below / right of the pivot factors the 3 x 3 pivot block
// itself (10 flops), substitutes its three panel rows and its three panel columns through it and subtracts their product from its
// block.  Pivots that fp32 cancellation drove below `floor_` are clamped.  Pb: scratch of 2 x 96 x 4 floats.  On return A holds L
// (upper triangle zero).
__device__ void cholesky96(float* A, float* Pb, float floor_) {
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const bool lower = ty >= tx;
  float a[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) a[r][c] = (lower && (ty > tx || c <= r)) ? A[(3 * ty + r) * LDJ + 3 * tx + c] : 0.f;
  for (int kb = 0; kb < LB / 3; ++kb) {
    float* P = Pb + (kb & 1) * LB * 4;
    if (tx == kb && ty >= kb) {
#pragma unroll
      for (int r = 0; r < 3; ++r) *(f32x4*)&P[(3 * ty + r) * 4] = f32x4{a[r][0], a[r][1], a[r][2], 0.f};
    }
    __syncthreads();
    if (!lower || tx < kb || ty < kb) continue;
    // the pivot block's factor (every thread its own copy)
    // (one 16-byte LDS read per panel row: the 24 scalar reads of a step were most of its time)
    const f32x4 pr0 = *(const f32x4*)&P[(3 * kb) * 4], pr1 = *(const f32x4*)&P[(3 * kb + 1) * 4], pr2 = *(const f32x4*)&P[(3 * kb + 2) * 4];
    const float p00 = pr0[0], p10 = pr1[0], p11 = pr1[1], p20 = pr2[0], p21 = pr2[1], p22 = pr2[2];
    // (v_rsq_f32, 1 ulp: sqrtf and the IEEE division expand to ~25 dependent instructions each, and this chain -- three of each --
    // is on every step's critical path)
    const float d00 = fmaxf(p00, floor_), i00 = __builtin_amdgcn_rsqf(d00), l00 = d00 * i00;
    const float l10 = p10 * i00, l20 = p20 * i00;
    const float d11 = fmaxf(p11 - l10 * l10, floor_), i11 = __builtin_amdgcn_rsqf(d11), l11 = d11 * i11;
    const float l21 = (p21 - l20 * l10) * i11;
    const float d22 = fmaxf(p22 - l20 * l20 - l21 * l21, floor_), i22 = __builtin_amdgcn_rsqf(d22), l22 = d22 * i22;
    if (ty == kb) {                                    // (then tx == kb: the pivot block itself)
      float* d = A + (3 * kb) * LDJ + 3 * kb;
      d[0] = l00;
      d[LDJ] = l10, d[LDJ + 1] = l11;
      d[2 * LDJ] = l20, d[2 * LDJ + 1] = l21, d[2 * LDJ + 2] = l22;
      continue;
    }
    float li[3][3];                                    // my rows of the panel: L[3 ty + r][3 kb + m]
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const f32x4 qv = *(const f32x4*)&P[(3 * ty + r) * 4];
      const float q0 = qv[0], q1 = qv[1], q2 = qv[2];
      li[r][0] = q0 * i00;
      li[r][1] = (q1 - li[r][0] * l10) * i11;
      li[r][2] = (q2 - li[r][0] * l20 - li[r][1] * l21) * i22;
    }
    if (tx == kb) {                                    // panel owner: these are final
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int m = 0; m < 3; ++m) A[(3 * ty + r) * LDJ + 3 * kb + m] = li[r][m];
      continue;
    }
    float lj[3][3];                                    // my columns' rows of the panel: L[3 tx + c][3 kb + m]
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4 qv = *(const f32x4*)&P[(3 * tx + c) * 4];
      const float q0 = qv[0], q1 = qv[1], q2 = qv[2];
      lj[c][0] = q0 * i00;
      lj[c][1] = (q1 - lj[c][0] * l10) * i11;
      lj[c][2] = (q2 - lj[c][0] * l20 - lj[c][1] * l21) * i22;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) a[r][c] -= li[r][0] * lj[c][0] + li[r][1] * lj[c][1] + li[r][2] * lj[c][2];
  }
  __syncthreads();
  if (ty <= tx) {                                      // zero above the diagonal (the inverse below reads full rows)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (ty < tx || c > r) A[(3 * ty + r) * LDJ + 3 * tx + c] = 0.f;
  }
  __syncthreads();
}

// One level of the recursive inverse of a lower-triangular matrix: for every pair of adjacent M x M diagonal blocks whose inverses
// X11, X22 are in place,  X21 = -X22 (L21 X11).  T: scratch of the same shape.  A thread owns a 3 x 3 block of the product (18 LDS
// reads per 27 multiply-adds instead of 2 per 1).
template <int M>
__device__ void tri_inverse_level(const float* Lm, float* X, float* T) {
  const int tid = threadIdx.x;
  constexpr int NPAIR = LB / (2 * M), MB = M / 3, NBLK = NPAIR * MB * MB;
  static_assert(NBLK <= 1024, "one pass of the thread block");
  const bool on = tid < NBLK;
  const int p = tid / (MB * MB), rem = tid - p * MB * MB, bi = rem / MB, bj = rem - bi * MB;
  const int r0 = (2 * p + 1) * M, c0 = 2 * p * M;
  if (on) {                                             // T = L21 X11  (X11 lower triangular: k >= the column)
    float acc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    for (int kb = bj; kb < MB; ++kb) {
      float l[3][3], x[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          l[r][m] = Lm[(r0 + 3 * bi + r) * LDJ + c0 + 3 * kb + m];
          x[r][m] = X[(c0 + 3 * kb + r) * LDJ + c0 + 3 * bj + m];
        }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] += l[r][0] * x[0][c] + l[r][1] * x[1][c] + l[r][2] * x[2][c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) T[(r0 + 3 * bi + r) * LDJ + c0 + 3 * bj + c] = acc[r][c];
  }
  __syncthreads();
  if (on) {                                             // X21 = -X22 T   (X22 lower triangular: k <= the row)
    float acc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    for (int kb = 0; kb <= bi; ++kb) {
      float x[3][3], t[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          x[r][m] = X[(r0 + 3 * bi + r) * LDJ + r0 + 3 * kb + m];
          t[r][m] = T[(r0 + 3 * kb + r) * LDJ + c0 + 3 * bj + m];
        }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] += x[r][0] * t[0][c] + x[r][1] * t[1][c] + x[r][2] * t[2][c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) X[(r0 + 3 * bi + r) * LDJ + c0 + 3 * bj + c] = -acc[r][c];
  }
  __syncthreads();
}

// X = L^-1: the eight 12 x 12 diagonal blocks by substitution (one thread per column, 78 steps), then three levels of pairing
// (12 -> 24 -> 48 -> 96), each two small matrix products over all threads.
__device__ void tri_inverse96(const float* Lm, float* X, float* T) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int w = tid; w < LB * LB; w += nt) X[(w / LB) * LDJ + (w % LB)] = 0.f;
  __syncthreads();
  if (tid < LB) {
    const int b0 = (tid / 12) * 12, j = tid;
    float x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int row = b0 + i;
      float acc = row == j ? 1.f : 0.f;
#pragma unroll
      for (int m = 0; m < i; ++m) acc = fmaf(-Lm[row * LDJ + b0 + m], x[m], acc);      // (x[m] = 0 above this column's diagonal)
      x[i] = row >= j ? acc / Lm[row * LDJ + row] : 0.f;
      X[row * LDJ + j] = x[i];
    }
  }
  __syncthreads();
  tri_inverse_level<12>(Lm, X, T);
  tri_inverse_level<24>(Lm, X, T);
  tri_inverse_level<48>(Lm, X, T);
}

// C = op(A) op(B) on 96 x 96 matrices in LDS (row stride LDJ) on the fp32 matrix cores: 36 output tiles of 16 x 16 dealt to the 16 waves,
// 24 steps of v_mfma_f32_16x16x4_f32 each.  (The VALU form above: ~10 us per product at 1024 threads, LDS-issue-bound.)
template <bool TRANS_A, bool TRANS_B>
__device__ void mm96_mfma(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, q4 = lane >> 4, c16 = lane & 15;
  for (int t = wv; t < 36; t += 16) {
    const int i0 = (t / 6) * 16, j0 = (t % 6) * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k0 = 0; k0 < LB; k0 += 4) {
      const int k = k0 + q4;
      const float a = TRANS_A ? A[k * LDJ + i0 + c16] : A[(i0 + c16) * LDJ + k];
      const float b = TRANS_B ? B[(j0 + c16) * LDJ + k] : B[k * LDJ + j0 + c16];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) C[(i0 + 4 * q4 + i) * LDJ + j0 + c16] = acc[i];
  }
}

// ---- Jacobi, second form: ONE apply pass per round.  The 96 indices sit in 48 adjacent position pairs (2i, 2i+1) and the DATA moves
// with the round-robin tournament (position 0 fixed, the others shift along the ring top[1..47], bottom[47..0]): a thread owns the
// 2 x 2 block (pair i) x (pair j), reads it with two aligned 8-byte loads, applies pair i's rotation to its rows and pair j's to its
// columns and writes the four results to the positions they hold in the NEXT round, into the other buffer -- no tournament-permuted
// read addresses (round 2: 3-way bank conflicts), no separate row and column passes (3 barriers per round -> 2: rotations, apply).
// The eigenvector matrix is kept TRANSPOSED and in place, in original index order (row = eigenvector): its two rows of a pair rotate as
// 48 aligned float2 pairs; which rows those are follows from the closed form of the tournament.  After a whole sweep (95 rounds) every
// index is back at its own position, so the matrix the caller sees is in original order; the result of an odd number of sweeps sits in
// the second buffer (returned).
__device__ __forceinline__ int jr_sigma(int x) {                 // position x's content moves to position sigma(x)
  if (x == 0) return 0;
  if (x & 1) return x == 1 ? 2 : x - 2;                           // bottom[i] -> bottom[i-1];  bottom[0] -> top[1]
  return x < LB - 2 ? x + 2 : LB - 1;                             // top[i] -> top[i+1];  top[47] -> bottom[47]
}
__device__ __forceinline__ int jr_pos2idx(int pos, int r) {      // the original index at position pos after r rounds (0 <= r < 95)
  if (pos == 0) return 0;
  const int sl = (pos & 1) ? (LB - 2) - (pos >> 1) : (pos >> 1) - 1;      // ring slot: top[i] -> i - 1, bottom[i] -> 94 - i
  int s0 = sl - r;
  s0 += s0 < 0 ? LB - 1 : 0;
  return s0 < LB / 2 - 1 ? 2 * (s0 + 1) : 2 * ((LB - 2) - s0) + 1;
}

__device__ float* jacobi96(float* A0, float* A1, float* Vt, float2* cs, int2* pq, int* flag, float* dmaxp, int max_sweeps, float rel_tol,
                           float abs_tol, int* sweeps_out) {
  constexpr int NT = 1024, NP = LB / 2, NITEM = NP * NP, PER = (NITEM + NT - 1) / NT;      // 2304 blocks / row pairs, 3 per thread
  const int tid = threadIdx.x;
  for (int i = tid; i < LB * LB; i += NT) Vt[(i / LB) * LDJ + (i % LB)] = (i / LB) == (i % LB) ? 1.f : 0.f;
  // A thread's work items are the same in every round (the DATA moves, not the roles): offsets decoded once.  A round is bound by
  // instruction issue and LDS bandwidth (258 KB through 128 B / clk), so what is left in the loop is loads, 12 + 8 multiply-adds, stores.
  int a_src[PER], a_d0[PER], a_d1[PER], a_j0[PER], a_j1[PER], a_i[PER], a_j[PER], v_off[PER];
  bool ok[PER];
#pragma unroll
  for (int n = 0; n < PER; ++n) {
    const int b = tid + n * NT;
    ok[n] = b < NITEM;
    const int i = ok[n] ? b / NP : 0, j = ok[n] ? b - i * NP : 0;
    a_i[n] = i;
    a_j[n] = j;
    a_src[n] = (2 * i) * LDJ + 2 * j;
    a_d0[n] = jr_sigma(2 * i) * LDJ;
    a_d1[n] = jr_sigma(2 * i + 1) * LDJ;
    a_j0[n] = jr_sigma(2 * j);
    a_j1[n] = jr_sigma(2 * j + 1);
    v_off[n] = 2 * j;                              // (the same decode serves the eigenvector rows: pair i, column pair j)
  }
  float* cur = A0;
  float* nxt = A1;
  int sweeps = 0;
  for (int sw = 0; sw < max_sweeps; ++sw) {
    __syncthreads();
    if (tid < 64) {
      float m = 0.f;
      for (int i = tid; i < LB; i += 64) m = fmaxf(m, fabsf(cur[i * LDJ + i]));
      m = wave_max(m);
      if (tid == 0) {
        *dmaxp = m;
        *flag = 0;
      }
    }
    __syncthreads();
    const float floor_abs = abs_tol * *dmaxp;
    // anything left above the bound?  The same pass puts the mean of a_pq and a_qp in both places: the two are updated by the same
    // rotations in different operation orders and drift apart by roundoff; a rotation decided on one of them and a convergence test
    // that reads the other can disagree for ever at the threshold (seen: 12 of 12 sweeps on 3 of 14 batches).
    bool mine = false;
    for (int w = tid; w < LB * LB; w += NT) {
      const int p = w / LB, q = w % LB;
      if (p < q) {
        const float m = 0.5f * (cur[p * LDJ + q] + cur[q * LDJ + p]);
        cur[p * LDJ + q] = m;
        cur[q * LDJ + p] = m;
        mine = mine || fabsf(m) > fmaxf(rel_tol * sqrtf(fabsf(cur[p * LDJ + p] * cur[q * LDJ + q])), floor_abs);
      }
    }
    if (mine) *flag = 1;
    __syncthreads();
    if (*flag == 0) break;
    ++sweeps;
    for (int rd = 0; rd < LB - 1; ++rd) {
      if (tid < NP) {
        const int i = tid;
        const float app = cur[(2 * i) * LDJ + 2 * i], aqq = cur[(2 * i + 1) * LDJ + 2 * i + 1], 
                    apq = 0.5f * (cur[(2 * i) * LDJ + 2 * i + 1] + cur[(2 * i + 1) * LDJ + 2 * i]);
        float c = 1.f, s = 0.f;
        if (fabsf(apq) > fmaxf(rel_tol * sqrtf(fabsf(app * aqq)), floor_abs) && apq != 0.f) {
          const float tau = (aqq - app) / (2.f * apq);
          const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
          c = rsqrtf(1.f + t * t);
          s = t * c;
        }
        cs[i] = make_float2(c, s);
        pq[i] = make_int2(jr_pos2idx(2 * i, rd) * LDJ, jr_pos2idx(2 * i + 1, rd) * LDJ);
      }
      __syncthreads();
#pragma unroll
      for (int n = 0; n < PER; ++n) {
        if (!ok[n]) continue;
        const float2 ri = cs[a_i[n]], rj = cs[a_j[n]];
        const float2 r0 = *(const float2*)&cur[a_src[n]], r1 = *(const float2*)&cur[a_src[n] + LDJ];
        const float t0x = ri.x * r0.x - ri.y * r1.x, t0y = ri.x * r0.y - ri.y * r1.y;       // rows: A <- J^T A
        const float t1x = ri.y * r0.x + ri.x * r1.x, t1y = ri.y * r0.y + ri.x * r1.y;
        nxt[a_d0[n] + a_j0[n]] = rj.x * t0x - rj.y * t0y;                                     // columns: A <- A J
        nxt[a_d0[n] + a_j1[n]] = rj.y * t0x + rj.x * t0y;
        nxt[a_d1[n] + a_j0[n]] = rj.x * t1x - rj.y * t1y;
        nxt[a_d1[n] + a_j1[n]] = rj.y * t1x + rj.x * t1y;
        if (ri.y != 0.f) {                                                                    // V <- V J, on the rows of V^T
          const int2 ii = pq[a_i[n]];
          float2* vp = (float2*)&Vt[ii.x + v_off[n]];
          float2* vq = (float2*)&Vt[ii.y + v_off[n]];
          const float2 x = *vp, yv = *vq;
          *vp = make_float2(ri.x * x.x - ri.y * yv.x, ri.x * x.y - ri.y * yv.y);
          *vq = make_float2(ri.y * x.x + ri.x * yv.x, ri.y * x.y + ri.x * yv.y);
        }
      }
      __syncthreads();
      float* t = cur;
      cur = nxt;
      nxt = t;
    }
  }
  __syncthreads();
  *sweeps_out = sweeps;
  return cur;
}

// S f32 [L][96][96] (accumulated by lr_mult_kernel; ZEROED again here for the next stage) -> Cout = D^-1 L^-T with S = D (L L^T) D
__global__ __launch_bounds__(1024) void lr_orth_kernel(float* __restrict__ Sg, float* __restrict__ Cout) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* B0 = sm;
  float* B1 = B0 + LB * LDJ;
  float* B2 = B1 + LB * LDJ;
  float* d = B2 + LB * LDJ;        // [96]
  const int tid = threadIdx.x, nt = blockDim.x;
  float* Sl = Sg + (size_t)blockIdx.x * LB * LB;
  float* Cl = Cout + (size_t)blockIdx.x * LB * LB;
  // (S is symmetric up to the order of its atomic additions: the lower triangle as it is.  All nine elements of a thread are
  // requested before the first is used: a load-use loop would expose nine memory round trips in a kernel that lasts tens of us.)
  LR_STAMP(1, 0);
  float sv[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) sv[e] = Sl[tid + 1024 * e];
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const int i = tid + 1024 * e, r = i / LB, c = i % LB;
    B0[r * LDJ + c] = sv[e];
    Sl[i] = 0.f;
  }
  __syncthreads();
  if (tid < LB) d[tid] = rsqrtf(fmaxf(B0[tid * LDJ + tid], 1e-30f));
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const int i = tid + 1024 * e, r = i / LB, c = i % LB;
    if (c <= r) B0[r * LDJ + c] = sv[e] * d[r] * d[c];
  }
  __syncthreads();
  LR_STAMP(1, 1);
  cholesky96(B0, B1, 1e-6f);
  LR_STAMP(1, 2);
  tri_inverse96(B0, B1, B2);                     // B1 = L^-1
  LR_STAMP(1, 3);
  for (int i = tid; i < LB * LB; i += nt) {
    const int r = i / LB, c = i % LB;
    Cl[i] = c >= r ? d[r] * B1[c * LDJ + r] : 0.f;        // C = D^-1 L^-T (upper triangular)
  }
  LR_STAMP(1, 4);
}

// The last stage: P = Y_prev^T Y, S = Y^T Y (both zeroed again here), Cprev = the transform that orthonormalises Y_prev (null: Y_prev is
// the orthonormal basis itself) -> Cout = W D^-1 L^-T, evals (Ritz values, descending), info[2 l + 1] = Jacobi sweeps run.
__global__ __launch_bounds__(1024) void lr_ritz_kernel(float* __restrict__ Pg, float* __restrict__ Sg, const float* __restrict__ Cprev,
                                                       float* __restrict__ Cout, float* __restrict__ evals, int* __restrict__ info,
                                                       int ritz_sweeps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* B0 = sm;
  float* B1 = B0 + LB * LDJ;
  float* B2 = B1 + LB * LDJ;
  float* B3 = B2 + LB * LDJ;
  float* d = B3 + LB * LDJ;        // [96]
  float* ev = d + LB;              // [96]
  float2* cs = (float2*)(ev + LB); // [64]
  int2* pq = (int2*)(cs + 64);     // [64]
  int* rnk = (int*)(pq + 64);      // [96]
  int* flag = rnk + LB;
  float* dmaxp = (float*)(flag + 1);
  int* swp = (int*)(dmaxp + 1);
  const int tid = threadIdx.x, nt = blockDim.x;
  float* Pl = Pg + (size_t)blockIdx.x * LB * LB;
  float* Sl = Sg + (size_t)blockIdx.x * LB * LB;
  float* Cl = Cout + (size_t)blockIdx.x * LB * LB;
  LR_STAMP(2, 0);
  const bool has_c = Cprev != nullptr;
  {                                              // all 27 elements of a thread requested before the first is used
    float pv[9], sv[9], cv[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      pv[e] = Pl[tid + 1024 * e];
      sv[e] = Sl[tid + 1024 * e];
      cv[e] = has_c ? Cprev[(size_t)blockIdx.x * LB * LB + tid + 1024 * e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const int i = tid + 1024 * e, r = i / LB, c = i % LB;
      (has_c ? B1 : B0)[r * LDJ + c] = pv[e];
      B3[r * LDJ + c] = sv[e];
      if (has_c) B2[r * LDJ + c] = cv[e];
      Pl[i] = 0.f;
      Sl[i] = 0.f;
    }
  }
  __syncthreads();
  if (has_c) {
    mm96_mfma<true, false>(B2, B1, B0);          // H = C_prev^T P = V^T G V
    __syncthreads();
  }
  for (int i = tid; i < LB * LB; i += nt) {      // symmetrise H and S in place: one thread per unordered pair
    const int r = i / LB, c = i % LB;
    if (c < r) {
      const float h = 0.5f * (B0[r * LDJ + c] + B0[c * LDJ + r]);
      B0[r * LDJ + c] = h;
      B0[c * LDJ + r] = h;
      B3[c * LDJ + r] = B3[r * LDJ + c];         // (S arrives as its lower triangle: lr_mult_kernel adds only the tiles on and below the diagonal)
    }
  }
  // off-diagonals below 1e-7 of the largest Ritz value are fp32 noise of the GEMMs that built H: rotating on them never converges
  LR_STAMP(2, 1);
  float* Af = jacobi96(B0, B1, B2, cs, pq, flag, dmaxp, ritz_sweeps, 1e-5f, 1e-7f, swp);      // rows of B2 = eigenvectors
  LR_STAMP(2, 2);
  if (tid < LB) ev[tid] = Af[tid * LDJ + tid];
  __syncthreads();
  if (tid < LB) {                              // rank of each Ritz value (descending, index breaks ties)
    const float e = ev[tid];
    int r = 0;
    for (int j = 0; j < LB; ++j) r += (ev[j] > e) || (ev[j] == e && j < tid);
    rnk[tid] = r;
    if (evals) evals[(size_t)blockIdx.x * LB + r] = e;
  }
  __syncthreads();
  for (int i = tid; i < LB * LB; i += nt) {    // B1 = W, columns in Ritz order (eigenvector c is ROW c of B2)
    const int r = i / LB, c = i % LB;
    B1[r * LDJ + rnk[c]] = B2[c * LDJ + r];
  }
  __syncthreads();
  mm96_mfma<false, false>(B3, B1, B0);         // S W
  __syncthreads();
  mm96_mfma<true, false>(B1, B0, B2);          // S' = W^T S W = (Y W)^T (Y W)
  __syncthreads();
  if (tid < LB) d[tid] = rsqrtf(fmaxf(B2[tid * LDJ + tid], 1e-30f));
  __syncthreads();
  for (int i = tid; i < LB * LB; i += nt) {    // unit diagonal, lower triangle symmetrised
    const int r = i / LB, c = i % LB;
    if (c <= r) B0[r * LDJ + c] = 0.5f * (B2[r * LDJ + c] + B2[c * LDJ + r]) * d[r] * d[c];
  }
  __syncthreads();
  LR_STAMP(2, 3);
  cholesky96(B0, B3, 1e-6f);                   // S'n = L L^T (B3 = S is dead: panel scratch)
  LR_STAMP(2, 4);
  tri_inverse96(B0, B3, B2);                   // B3 = L^-1 (B2 = S' is dead: scratch)
  LR_STAMP(2, 5);
  for (int i = tid; i < LB * LB; i += nt) B1[(i / LB) * LDJ + (i % LB)] *= d[i % LB];      // W D^-1
  __syncthreads();
  mm96_mfma<false, true>(B1, B3, B2);          // C = W D^-1 L^-T
  __syncthreads();
  for (int i = tid; i < LB * LB; i += nt) Cl[i] = B2[(i / LB) * LDJ + (i % LB)];
  if (info && tid == 0) {
    info[2 * blockIdx.x] = 0;
    info[2 * blockIdx.x + 1] = *swp;
  }
  LR_STAMP(2, 6);
}

constexpr int ORTH_SMEM = (3 * LB * LDJ + LB) * 4;
constexpr int RITZ_SMEM = (4 * LB * LDJ + 2 * LB) * 4 + 64 * 8 + 64 * 8 + (LB + 4) * 4;
inline int lr_mult_smem(int Dt) { return (LR_ROWS * (Dt + 4) + 2 * 32 * LB + LB * LB) * 4; }

constexpr int SMALL_SMEM = (4 * LB * LLD + 2 * LB) * 4 + 64 * 8 + 64 * 8 + (LB + 2) * 4;
constexpr int JACOBI_SMEM = 2 * NMAX * LD * 4 + 64 * 8 + 64 * 8 + 16;

template <typename K>
int raise_lds(K kernel, int bytes, const char* what) {
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    dkd_set_error("%s: cannot raise dynamic LDS to %d: %s", what, bytes, hipGetErrorString(e));
    return DKD_ERR_HIP;
  }
  return DKD_OK;
}

inline int64_t al256(int64_t b) { return (b + 255) / 256 * 256; }
}  // namespace

extern "C" int dkd_jacobi_eigh(const float* A, float* evals, float* evecs, int32_t batch, int32_t n, int32_t sweeps, void* stream) {
  DKD_CHECK_ARG(A && evals && evecs, "jacobi_eigh: null operand");
  DKD_CHECK_ARG(batch > 0 && n > 0 && n <= NMAX && sweeps > 0, "jacobi_eigh: need 0 < n <= %d (n=%d)", NMAX, n);
  int rc = raise_lds(jacobi_eigh_kernel, JACOBI_SMEM, "jacobi_eigh");     // idempotent attribute of the code object, cheap
  if (rc != DKD_OK) return rc;
  hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(batch), dim3(1024), JACOBI_SMEM, as_stream(stream), A, evals, evecs, n, sweeps);
  DKD_CHECK_LAUNCH("jacobi_eigh");
  return DKD_OK;
}

extern "C" int64_t dkd_lowrank_workspace_bytes(int32_t L, int32_t Dt) {
  return al256((int64_t)L * Dt * 2 * LB * 4) + al256((int64_t)L * LB * 2 * LB * 4) + al256((int64_t)L * LB * LB * 4) + al256((int64_t)L * 8);
}

extern "C" int dkd_lowrank_step(const float* G, float* V, int32_t L, int32_t Dt, int32_t mode, int32_t ritz_sweeps, int32_t rank, void* v_hi,
                                void* v_lo, float* evals, void* ws, void* stream) {
  DKD_CHECK_ARG(G && V && ws, "lowrank_step: null operand");
  DKD_CHECK_ARG(L > 0 && Dt >= 128 && Dt % 32 == 0, "lowrank_step: Dt=%d must be a multiple of 32, >= 128", Dt);
  DKD_CHECK_ARG(mode >= 0 && mode <= 3, "lowrank_step: mode %d", mode);
  DKD_CHECK_ARG(ritz_sweeps >= 0 && ritz_sweeps <= 32, "lowrank_step: ritz_sweeps %d", ritz_sweeps);
  DKD_CHECK_ARG(!v_hi || (v_lo && rank > 0 && rank <= LB && (mode == 1 || mode == 3)),
                "lowrank_step: hi/lo output needs mode 1 or 3 and 0 < rank <= %d", LB);
  DKD_CHECK_ARG(((uintptr_t)ws & 255) == 0, "lowrank_step: workspace must be 256-byte aligned");
  hipStream_t st = as_stream(stream);
  float* YZ = (float*)ws;                                                     // [L][Dt][192]: Y | copy of V
  float* S = (float*)((char*)ws + al256((int64_t)L * Dt * 2 * LB * 4));       // [L][96 x 192 floats]
  float* C = (float*)((char*)S + al256((int64_t)L * LB * 2 * LB * 4));        // [L][96][96]
  int* info = (int*)((char*)C + al256((int64_t)L * LB * LB * 4));             // [L][2] Jacobi sweeps of the last step (diagnostics)
  const long sG = (long)Dt * Dt, sV = (long)Dt * LB, sYZ = (long)Dt * 2 * LB, sS = (long)LB * 2 * LB, sC = (long)LB * LB;
  const dim3 tall(1, Dt / 16, L);     // V = Y C: K = 96, no split
  if (mode != 0) {            // Y (mode 2) or the second half of the rows (modes 1, 3) <- V
    hipError_t e = hipMemcpy2DAsync(mode == 2 ? YZ : YZ + LB, 2 * LB * 4, V, LB * 4, LB * 4, (size_t)L * Dt, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
      dkd_set_error("lowrank_step: copy failed: %s", hipGetErrorString(e));
      return DKD_ERR_HIP;
    }
  }
  const int ks_gv = Dt % 128 == 0 ? 4 : 1, ks_gram = Dt % 256 == 0 ? 8 : 1;      // K slices (each a multiple of 32)
  if (mode != 2) {
    if (ks_gv > 1 && hipMemset2DAsync(YZ, 2 * LB * 4, 0, LB * 4, (size_t)L * Dt, st) != hipSuccess) {
      dkd_set_error("lowrank_step: memset failed");
      return DKD_ERR_HIP;
    }
    hipLaunchKernelGGL((sgemm_kernel<2, 16>), dim3(ks_gv, Dt / 16, L), dim3(256), 0, st, G, sG, Dt, V, sV, LB, YZ, sYZ, 2 * LB, Dt, LB, Dt,
                       nullptr, nullptr, 0);
    DKD_CHECK_LAUNCH("lowrank Y = G V");
  }
  if (ks_gram > 1 && hipMemsetAsync(S, 0, (size_t)L * sS * 4, st) != hipSuccess) {
    dkd_set_error("lowrank_step: memset failed");
    return DKD_ERR_HIP;
  }
  if (mode == 1)              // [Y | V]^T Y -> S [192][96]: rows 0..95 = Y^T Y, rows 96..191 = V^T G V
    hipLaunchKernelGGL((sgemm_kernel<1, 32>), dim3(ks_gram, 2 * LB / 32, L), dim3(256), 0, st, YZ, sYZ, 2 * LB, YZ, sYZ, 2 * LB, S, sS, LB,
                       2 * LB, LB, Dt, nullptr, nullptr, 0);
  else                        // modes 0, 2: Y^T Y;  mode 3: V^T (G V)   -> S [96][row stride 192]
    hipLaunchKernelGGL((sgemm_kernel<1, 32>), dim3(ks_gram, LB / 32, L), dim3(256), 0, st, mode == 3 ? YZ + LB : YZ, sYZ, 2 * LB, YZ, sYZ, 2 * LB,
                       S, sS, 2 * LB, LB, LB, Dt, nullptr, nullptr, 0);
  DKD_CHECK_LAUNCH("lowrank Gram 96");
  int rc = raise_lds(lowrank_small_kernel, SMALL_SMEM, "lowrank_step");
  if (rc != DKD_OK) return rc;
  const bool ritz = mode == 1 || mode == 3;
  hipLaunchKernelGGL(lowrank_small_kernel, dim3(L), dim3(1024), SMALL_SMEM, st, S, C, ritz ? evals : nullptr, info, mode == 2 ? 0 : mode,
                     mode == 1 ? ritz_sweeps : 12);
  DKD_CHECK_LAUNCH("lowrank 96 x 96 stage");
  // V <- (Y or, mode 3, the copy of V) C
  hipLaunchKernelGGL((sgemm_kernel<0, 16>), tall, dim3(256), 0, st, mode == 3 ? YZ + LB : YZ, sYZ, 2 * LB, C, sC, LB, V, sV, LB, Dt, LB, LB,
                     (bf16_t*)v_hi, (bf16_t*)v_lo, rank);
  DKD_CHECK_LAUNCH("lowrank V = Y C");
  return DKD_OK;
}

// ---- the chain (see the "round 5" block above)
namespace {
struct ChainWs {
  float *YA, *YB, *S, *P, *C0, *C1;
  int* info;
};
inline int64_t chain_carve(void* ws, int32_t L, int32_t Dt, ChainWs* o) {
  char* p = (char*)ws;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* q = p ? p + off : nullptr;
    off += al256(bytes);
    return q;
  };
  const int64_t y = (int64_t)L * Dt * LB * 4, m = (int64_t)L * LB * LB * 4;
  char* s0 = take(m);                // S and P first: the part of the workspace that must be zero on entry
  char* p0 = take(m);
  char* ya = take(y);
  char* yb = take(y);
  char* c0 = take(m);
  char* c1 = take(m);
  char* inf = take((int64_t)L * 8);
  if (o) *o = ChainWs{(float*)ya, (float*)yb, (float*)s0, (float*)p0, (float*)c0, (float*)c1, (int*)inf};
  return off;
}
}  // namespace

extern "C" int64_t dkd_lowrank_chain_workspace_bytes(int32_t L, int32_t Dt) { return chain_carve(nullptr, L, Dt, nullptr); }
extern "C" int64_t dkd_lowrank_chain_zero_bytes(int32_t L, int32_t Dt) { return 2 * al256((int64_t)L * LB * LB * 4); }

extern "C" int dkd_lowrank_chain(const float* G, float* V, int32_t L, int32_t Dt, int32_t n_mult, int32_t ritz_sweeps, int32_t rank, void* v_hi,
                                 void* v_lo, float* evals, void* ws, void* stream) {
  DKD_CHECK_ARG(G && V && ws, "lowrank_chain: null operand");
  DKD_CHECK_ARG(L > 0 && Dt >= 128 && Dt % 64 == 0 && Dt <= 2048, "lowrank_chain: Dt=%d must be a multiple of 64 in [128, 2048]", Dt);
  DKD_CHECK_ARG(n_mult >= 1 && n_mult <= 64, "lowrank_chain: n_mult %d", n_mult);
  DKD_CHECK_ARG(ritz_sweeps >= 0 && ritz_sweeps <= 32, "lowrank_chain: ritz_sweeps %d", ritz_sweeps);
  DKD_CHECK_ARG(!v_hi || (v_lo && rank > 0 && rank <= LB), "lowrank_chain: hi/lo output needs 0 < rank <= %d", LB);
  DKD_CHECK_ARG(((uintptr_t)ws & 255) == 0, "lowrank_chain: workspace must be 256-byte aligned");
  hipStream_t st = as_stream(stream);
  ChainWs w;
  chain_carve(ws, L, Dt, &w);
  const int msm = lr_mult_smem(Dt);
  int rc = raise_lds(lr_mult_kernel, msm, "lowrank_chain (mult)");
  if (rc == DKD_OK) rc = raise_lds(lr_orth_kernel, ORTH_SMEM, "lowrank_chain (orth)");
  if (rc == DKD_OK) rc = raise_lds(lr_ritz_kernel, RITZ_SMEM, "lowrank_chain (ritz)");
  if (rc != DKD_OK) return rc;
  // multiplies per stage: 1, 2, 2, ..., (1,) 1
  int stages[64], ns = 0;
  if (n_mult == 1) {
    stages[ns++] = 1;
  } else {
    stages[ns++] = 1;
    int mid = n_mult - 2;
    while (mid >= 2) {
      stages[ns++] = 2;
      mid -= 2;
    }
    if (mid) stages[ns++] = 1;
    stages[ns++] = 1;
  }
  const float* src = V;
  const float* cprev = nullptr;
  float* cnext = w.C0;
  const dim3 grid(Dt / LR_ROWS, L);
  for (int s = 0; s < ns; ++s) {
    const bool last = s == ns - 1;
    for (int m = 0; m < stages[s]; ++m) {
      float* dst = src == w.YA ? w.YB : w.YA;
      const int flags = ((m == 0 && cprev) ? LR_APPLY_C : 0) | (m == stages[s] - 1 ? LR_GRAM_S : 0) | (last ? LR_GRAM_P : 0);
      hipLaunchKernelGGL(lr_mult_kernel, grid, dim3(256), msm, st, G, src, cprev, dst, w.S, w.P, Dt, flags);
      DKD_CHECK_LAUNCH("lowrank_chain Y = (G Y) C");
      src = dst;
    }
    if (!last) {
      hipLaunchKernelGGL(lr_orth_kernel, dim3(L), dim3(1024), ORTH_SMEM, st, w.S, cnext);
      DKD_CHECK_LAUNCH("lowrank_chain orthonormalising transform");
    } else {
      hipLaunchKernelGGL(lr_ritz_kernel, dim3(L), dim3(1024), RITZ_SMEM, st, w.P, w.S, cprev, cnext, evals, w.info, ritz_sweeps);
      DKD_CHECK_LAUNCH("lowrank_chain Rayleigh-Ritz stage");
    }
    cprev = cnext;
    cnext = cnext == w.C0 ? w.C1 : w.C0;
  }
  const long sV = (long)Dt * LB, sC = (long)LB * LB;
  hipLaunchKernelGGL((sgemm_kernel<0, 16>), dim3(1, Dt / 16, L), dim3(256), 0, st, src, sV, LB, cprev, sC, LB, V, sV, LB, Dt, LB, LB,
                     (bf16_t*)v_hi, (bf16_t*)v_lo, rank);
  DKD_CHECK_LAUNCH("lowrank_chain V = Y C");
  return DKD_OK;
}

#ifdef DKD_LR_STAMPS
extern "C" int dkd_lr_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(lr_stamp_buf), sizeof(unsigned long long) * 3 * 64) == hipSuccess ? 0 : -1;
}
#endif

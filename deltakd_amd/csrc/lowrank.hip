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
template <bool TRANS_A, bool TRANS_B = false>
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

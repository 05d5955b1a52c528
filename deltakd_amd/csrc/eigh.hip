// Batched symmetric eigensolver for small matrices (n <= 128), one workgroup per matrix, everything LDS-resident.
//
// Used by the LRKD target (model/loss.py:318-324 of the reference: truncated SVD of the [B*196, Dt] teacher matrix).
// The MI355X path never factorises the tall matrix: Gram (MFMA) -> block subspace iteration (GEMMs) -> this kernel for
// the b x b Rayleigh-Ritz / orthonormalisation problems.  rocSOLVER's syevd spends ~6000 micro-launches per 768^2 matrix
// (profiles/r01_a_*); a cyclic two-sided Jacobi on a <=128^2 matrix is one launch, ~1 ms, and is accurate to fp32 roundoff
// relative to each eigenvalue for the scaled-SPD matrices it is fed.
#include "common.h"

namespace {

constexpr int NMAX = 128;
constexpr int LD = NMAX + 1;

// A: [batch, n, n] symmetric f32 (row stride n).  Outputs: evals [batch, n], evecs [batch, n, n] (column j = eigenvector j).
__global__ __launch_bounds__(1024) void jacobi_eigh_kernel(const float* __restrict__ Ain, float* __restrict__ evals,
                                                           float* __restrict__ evecs, int n, int sweeps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* A = sm;                 // [ne][LD]
  float* V = A + NMAX * LD;      // [ne][LD]
  float* cs = V + NMAX * LD;     // c[64], s[64]
  int* pq = (int*)(cs + 2 * (NMAX / 2));  // p[64], q[64]
  int* rotated = pq + 2 * (NMAX / 2);     // rotations applied in the current sweep (convergence test)
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ne = (n + 1) & ~1;   // even working size (a padded index gets a zero row/col and never rotates)
  const int half = ne >> 1;
  const float* Ab = Ain + (size_t)blockIdx.x * n * n;
  for (int i = tid; i < ne * ne; i += nt) {
    const int r = i / ne, c = i % ne;
    A[r * LD + c] = (r < n && c < n) ? Ab[r * n + c] : 0.f;
    V[r * LD + c] = r == c ? 1.f : 0.f;
  }
  __syncthreads();
  for (int sw = 0; sw < sweeps; ++sw) {
    if (tid == 0) *rotated = 0;
    __syncthreads();
    for (int rd = 0; rd < ne - 1; ++rd) {
      // round-robin tournament: position 0 is fixed, the other ne-1 positions rotate by rd
      if (tid < half) {
        const int i = tid;
        int a = i == 0 ? 0 : 1 + (i - 1 + rd) % (ne - 1);
        int b = 1 + (ne - 1 - i - 1 + rd) % (ne - 1);
        const int p = a < b ? a : b, q = a < b ? b : a;
        const float app = A[p * LD + p], aqq = A[q * LD + q], apq = A[p * LD + q];
        float c = 1.f, s = 0.f;
        // threshold Jacobi: an off-diagonal below fp32 roundoff of sqrt(app aqq) cannot change either eigenvalue
        if (fabsf(apq) > 6e-8f * sqrtf(fabsf(app * aqq)) && apq != 0.f) {
          atomicOr(rotated, 1);
          const float tau = (aqq - app) / (2.f * apq);
          const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
          c = rsqrtf(1.f + t * t);
          s = t * c;
        }
        cs[i] = c;
        cs[NMAX / 2 + i] = s;
        pq[i] = p;
        pq[NMAX / 2 + i] = q;
      }
      __syncthreads();
      // rows: A <- J^T A
      for (int w = tid; w < half * ne; w += nt) {
        const int i = w / ne, k = w % ne;
        const int p = pq[i], q = pq[NMAX / 2 + i];
        const float c = cs[i], s = cs[NMAX / 2 + i];
        const float x = A[p * LD + k], y = A[q * LD + k];
        A[p * LD + k] = c * x - s * y;
        A[q * LD + k] = s * x + c * y;
      }
      __syncthreads();
      // columns: A <- A J, V <- V J
      for (int w = tid; w < half * ne; w += nt) {
        const int i = w / ne, k = w % ne;
        const int p = pq[i], q = pq[NMAX / 2 + i];
        const float c = cs[i], s = cs[NMAX / 2 + i];
        float x = A[k * LD + p], y = A[k * LD + q];
        A[k * LD + p] = c * x - s * y;
        A[k * LD + q] = s * x + c * y;
        x = V[k * LD + p];
        y = V[k * LD + q];
        V[k * LD + p] = c * x - s * y;
        V[k * LD + q] = s * x + c * y;
      }
      __syncthreads();
    }
    if (*rotated == 0) break;     // a full sweep without a rotation: converged (uniform across the workgroup)
    __syncthreads();
  }
  for (int i = tid; i < n; i += nt) evals[(size_t)blockIdx.x * n + i] = A[i * LD + i];
  float* Vb = evecs + (size_t)blockIdx.x * n * n;
  for (int i = tid; i < n * n; i += nt) Vb[i] = V[(i / n) * LD + (i % n)];
}

}  // namespace

extern "C" int dkd_jacobi_eigh(const float* A, float* evals, float* evecs, int32_t batch, int32_t n, int32_t sweeps, void* stream) {
  DKD_CHECK_ARG(A && evals && evecs, "jacobi_eigh: null operand");
  DKD_CHECK_ARG(batch > 0 && n > 0 && n <= NMAX && sweeps > 0, "jacobi_eigh: need 0 < n <= %d (n=%d)", NMAX, n);
  const int smem = (2 * NMAX * LD + 2 * (NMAX / 2)) * 4 + 2 * (NMAX / 2) * 4 + 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)jacobi_eigh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      dkd_set_error("jacobi_eigh: cannot raise dynamic LDS to %d: %s", smem, hipGetErrorString(e));
      return DKD_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(batch), dim3(1024), smem, as_stream(stream), A, evals, evecs, n, sweeps);
  DKD_CHECK_LAUNCH("jacobi_eigh");
  return DKD_OK;
}

// The token scores of the saliency_mgd branch (model/misc.py:38-165 of the reference, with the scorers of model/models.py:14-56): per
// sample, softmax attention weights of a small auxiliary attention (8 heads) averaged over the heads --
//   method 1  SimpleAttention on the patch tokens, the DIAGONAL of the [L, L] weights:   score[i] = mean_h softmax_j(q_i . k_j / sqrt(hd))[i]
//   method 2  the same projection on [CLS | patches], the CLS row, patch columns:        score[j] = mean_h softmax_{CLS, patches}(q_cls . k)[j]
//   method 3  SimpleCrossAttention, CLS as the query, the patches as keys:               score[j] = mean_h softmax_j(q_cls . k_j)[j]
// The scores are only RANKED (argsort; the lowest are kept), no gradient reaches the scorer, so everything here is fp32 -- the
// projections that feed it run as bf16 MFMA GEMMs on a hi / lo split of the fp32 weights (deltakd_amd.models), i.e. at fp32 accuracy
// too: rounding the projection weights to bf16 flipped near-tied tokens against the reference's fp32 nn.Linear.
// Replaces torch matmul + ATen softmax (rocBLAS bmm) in round 2's scorer.
#include "common.h"

namespace {

// method 1: one workgroup per sample; K of one head resident in LDS ([L][HD] f32, 75 KB at L = 196, HD = 96), thread i owns query i.
template <int HD>
__global__ __launch_bounds__(256) void saliency_diag_kernel(const float* __restrict__ q, const float* __restrict__ k, float* __restrict__ out,
                                                            const int L, const int H, const int ldq, const int ldk, const long q_sb,
                                                            const long k_sb, const int q_first, const int k_first, const float scale) {
  extern __shared__ __attribute__((aligned(16))) float ks[];
  const int b = blockIdx.x, i = threadIdx.x;
  const float* qb = q + ((size_t)b * q_sb + q_first) * ldq;
  const float* kb = k + ((size_t)b * k_sb + k_first) * ldk;
  float acc = 0.f;
  for (int h = 0; h < H; ++h) {
    __syncthreads();
    for (int e = threadIdx.x; e < L * (HD / 4); e += 256) {
      const int j = e / (HD / 4), c = e % (HD / 4);
      *(f32x4*)&ks[j * HD + 4 * c] = *(const f32x4*)(kb + (size_t)j * ldk + h * HD + 4 * c);
    }
    __syncthreads();
    if (i < L) {
      f32x4 qv[HD / 4];
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) qv[c] = scale * *(const f32x4*)(qb + (size_t)i * ldq + h * HD + 4 * c);
      float m = -INFINITY, l = 0.f, sii = 0.f;
      for (int j = 0; j < L; ++j) {
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) p += qv[c] * *(const f32x4*)&ks[j * HD + 4 * c];      // (all lanes read the same address: broadcast)
        const float s = (p[0] + p[1]) + (p[2] + p[3]);
        if (j == i) sii = s;
        const float mn = fmaxf(m, s);
        l = l * __expf(m - mn) + __expf(s - mn);
        m = mn;
      }
      acc += __expf(sii - m) / l;
    }
  }
  if (i < L) out[(size_t)b * L + i] = acc / H;
}

// methods 2 / 3: one query per sample and head against L keys (+ optionally one extra key that takes part in the softmax but gets no
// score: the CLS token itself in method 2).  One workgroup per sample, thread j owns key j (thread L: the extra key).
template <int HD>
__global__ __launch_bounds__(256) void saliency_row_kernel(const float* __restrict__ q, const float* __restrict__ k, float* __restrict__ out,
                                                           const int L, const int H, const int ldq, const int ldk, const long q_sb,
                                                           const long k_sb, const int q_first, const int k_first, const int extra_row,
                                                           const float scale) {
  __shared__ float red[8];
  const int b = blockIdx.x, j = threadIdx.x, lane = j & 63, w = j >> 6;
  const int n_keys = L + (extra_row >= 0 ? 1 : 0);
  const float* qb = q + ((size_t)b * q_sb + q_first) * ldq;
  const float* kr = nullptr;
  if (j < L) kr = k + ((size_t)b * k_sb + k_first + j) * ldk;
  else if (j < n_keys) kr = k + ((size_t)b * k_sb + extra_row) * ldk;       // (a row of the sample, counted like k_first)
  float acc = 0.f;
  for (int h = 0; h < H; ++h) {
    float s = -INFINITY;
    if (kr) {
      f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) p += *(const f32x4*)(qb + h * HD + 4 * c) * *(const f32x4*)(kr + h * HD + 4 * c);
      s = scale * ((p[0] + p[1]) + (p[2] + p[3]));
    }
    float m = wave_max(s);
    __syncthreads();
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float e = kr ? __expf(s - m) : 0.f;
    float l = wave_sum(e);
    __syncthreads();
    if (lane == 0) red[4 + w] = l;
    __syncthreads();
    l = (red[4] + red[5]) + (red[6] + red[7]);
    acc += e / l;
  }
  if (j < L) out[(size_t)b * L + j] = acc / H;
}

template <typename K>
int raise_lds_(K kernel, int bytes) {
  return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? DKD_OK : DKD_ERR_HIP;
}

}  // namespace

extern "C" int dkd_saliency_scores(const float* q, const float* k, float* scores, int32_t B, int32_t L, int32_t H, int32_t head_dim,
                                   int32_t ldq, int32_t ldk, int64_t q_rows_per_sample, int64_t k_rows_per_sample, int32_t q_first,
                                   int32_t k_first, int32_t diagonal, int32_t extra_key_row, void* stream) {
  DKD_CHECK_ARG(q && k && scores, "saliency_scores: null operand");
  DKD_CHECK_ARG(B > 0 && L > 0 && L <= 255 && H > 0, "saliency_scores: need 0 < L <= 255 tokens (L=%d)", L);
  DKD_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0, "saliency_scores: rows must be 16-byte aligned");
  DKD_CHECK_ARG(!diagonal || extra_key_row < 0, "saliency_scores: the diagonal form takes no extra key");
  const float scale = 1.0f / sqrtf((float)head_dim);
  hipStream_t st = as_stream(stream);
#define SAL_LAUNCH(HD_)                                                                                                               \
  do {                                                                                                                                \
    if (diagonal) {                                                                                                                   \
      const int smem = L * HD_ * 4;                                                                                                   \
      if (raise_lds_(saliency_diag_kernel<HD_>, smem) != DKD_OK) {                                                                    \
        dkd_set_error("saliency_scores: cannot raise dynamic LDS to %d", smem);                                                       \
        return DKD_ERR_HIP;                                                                                                           \
      }                                                                                                                               \
      hipLaunchKernelGGL(saliency_diag_kernel<HD_>, dim3(B), dim3(256), smem, st, q, k, scores, L, H, ldq, ldk, (long)q_rows_per_sample, \
                         (long)k_rows_per_sample, q_first, k_first, scale);                                                                             \
    } else {                                                                                                                          \
      hipLaunchKernelGGL(saliency_row_kernel<HD_>, dim3(B), dim3(256), 0, st, q, k, scores, L, H, ldq, ldk, (long)q_rows_per_sample,   \
                         (long)k_rows_per_sample, q_first, k_first, extra_key_row, scale);                                                        \
    }                                                                                                                                 \
  } while (0)
  switch (head_dim) {
    case 16: SAL_LAUNCH(16); break;
    case 32: SAL_LAUNCH(32); break;
    case 48: SAL_LAUNCH(48); break;
    case 64: SAL_LAUNCH(64); break;
    case 96: SAL_LAUNCH(96); break;
    case 128: SAL_LAUNCH(128); break;
    default:
      dkd_set_error("saliency_scores: head_dim %d not built (16, 32, 48, 64, 96, 128)", head_dim);
      return DKD_ERR_UNSUPPORTED;
  }
#undef SAL_LAUNCH
  DKD_CHECK_LAUNCH("saliency_scores");
  return DKD_OK;
}

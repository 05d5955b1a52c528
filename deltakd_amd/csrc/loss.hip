// Fused loss value + gradient kernels (HBM-bound: one pass over the operands, fp32 accumulation).
//   logit_loss : base criterion (soft-target CE / label-smoothing CE) + soft (KL) or hard (argmax CE) logit distillation
//                model/loss.py:35,57-67,241 and timm.loss [3P]; closed-form gradients of SURVEY.md Appendix C.
//   mse_loss   : (masked) mean-squared feature matching, model/loss.py:326 (LRKD), :449-451 (MGD), :145,149 (DiffKD)
//   mask_select: where(mask, mask_token, x) == gather/cat/gather of model/loss.py:433-440
#include "common.h"

namespace {

// one wave per row
__global__ __launch_bounds__(256) void logit_loss_kernel(const float* __restrict__ z, const float* __restrict__ soft_target,
                                                         const int64_t* __restrict__ labels, float smoothing, int kd_mode,
                                                         const float* __restrict__ z_kd, const float* __restrict__ z_t, float tau,
                                                         float w_base, float w_kd, float* __restrict__ losses, float* __restrict__ dz,
                                                         float* __restrict__ dz_kd, int B, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float invB = 1.f / B;
  const float* zr = z + (size_t)row * C;

  // ---- base criterion
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, zr[c]);
  mx = wave_max(mx);
  float se = 0.f, sz = 0.f, sy = 0.f, syz = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = zr[c];
    se += __expf(v - mx);
    sz += v;
    if (soft_target) {
      const float y = soft_target[(size_t)row * C + c];
      sy += y;
      syz += y * v;
    }
  }
  se = wave_sum(se);
  const float lse = mx + __logf(se);
  float lb;
  int lab = -1;
  if (soft_target) {
    sy = wave_sum(sy);
    syz = wave_sum(syz);
    lb = lse * sy - syz;  // -sum y (z - lse)
  } else {
    sz = wave_sum(sz);
    lab = (int)labels[row];
    lb = (1.f - smoothing) * (lse - zr[lab]) + smoothing * (lse - sz / C);
  }
  for (int c = lane; c < C; c += 64) {
    const float p = __expf(zr[c] - lse);
    float gr;
    if (soft_target) gr = p * sy - soft_target[(size_t)row * C + c];
    else gr = p - ((c == lab ? 1.f - smoothing : 0.f) + smoothing / C);
    dz[(size_t)row * C + c] = w_base * gr * invB;
  }
  if (lane == 0) {
    atomicAdd(&losses[0], lb * invB);
    atomicAdd(&losses[2], w_base * lb * invB);     // [2] the weighted total, [3] / [4] its two addends: no scalar kernels downstream
    atomicAdd(&losses[3], w_base * lb * invB);
  }

  // ---- logit distillation
  if (kd_mode == 0) return;
  const float* sr = z_kd + (size_t)row * C;
  const float* tr = z_t + (size_t)row * C;
  if (kd_mode == 1) {
    const float it = 1.f / tau;
    float ms = -INFINITY, mt = -INFINITY;
    for (int c = lane; c < C; c += 64) {
      ms = fmaxf(ms, sr[c] * it);
      mt = fmaxf(mt, tr[c] * it);
    }
    ms = wave_max(ms);
    mt = wave_max(mt);
    float es = 0.f, et = 0.f;
    for (int c = lane; c < C; c += 64) {
      es += __expf(sr[c] * it - ms);
      et += __expf(tr[c] * it - mt);
    }
    const float ls = ms + __logf(wave_sum(es)), lt = mt + __logf(wave_sum(et));
    float kl = 0.f;
    const float gsc = w_kd * tau / ((float)B * C);
    for (int c = lane; c < C; c += 64) {
      const float lps = sr[c] * it - ls, lpt = tr[c] * it - lt;
      const float pt = __expf(lpt), ps = __expf(lps);
      kl += pt * (lpt - lps);
      dz_kd[(size_t)row * C + c] = gsc * (ps - pt);
    }
    kl = wave_sum(kl);
    if (lane == 0) {
      const float v = kl * tau * tau / ((float)B * C);
      atomicAdd(&losses[1], v);
      atomicAdd(&losses[2], w_kd * v);
      atomicAdd(&losses[4], w_kd * v);
    }
  } else {
    // hard: CE(z_kd, argmax z_t); first maximal index like torch.argmax
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = tr[c];
      if (v > bv) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    float ms = -INFINITY;
    for (int c = lane; c < C; c += 64) ms = fmaxf(ms, sr[c]);
    ms = wave_max(ms);
    float es = 0.f;
    for (int c = lane; c < C; c += 64) es += __expf(sr[c] - ms);
    const float ls = ms + __logf(wave_sum(es));
    for (int c = lane; c < C; c += 64) dz_kd[(size_t)row * C + c] = w_kd * invB * (__expf(sr[c] - ls) - (c == bi ? 1.f : 0.f));
    if (lane == 0) {
      const float v = (ls - sr[bi]) * invB;
      atomicAdd(&losses[1], v);
      atomicAdd(&losses[2], w_kd * v);
      atomicAdd(&losses[4], w_kd * v);
    }
  }
}

// top-k accuracy (timm.utils.accuracy [3P], tools/engine.py:54-56): one wave per row counts the logits that beat the label's
// (larger, or equal at a lower index: the order torch.topk's sorted output lists them in); the row is correct at k if fewer than k do.
// out[i] += 100 / B for every row correct at ks[i].
__global__ __launch_bounds__(256) void topk_correct_kernel(const float* __restrict__ z, const int64_t* __restrict__ labels, int B, int C,
                                                           int k0, int k1, int k2, int k3, int nk, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* zr = z + (size_t)row * C;
  const int64_t lab64 = labels[row];
  if (lab64 < 0 || lab64 >= C) return;   // ignore_index / a class the head does not have: never among the top k (timm counts it wrong)
  const int lab = (int)lab64;
  const float t = zr[lab];
  int cnt = 0;
  for (int c = lane; c < C; c += 64) {
    const float v = zr[c];
    cnt += (v > t) || (v == t && c < lab);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) {
    const float w = 100.f / B;
    const int ks[4] = {k0, k1, k2, k3};
    for (int i = 0; i < nk; ++i)
      if (cnt < ks[i]) atomicAdd(&out[i], w);
  }
}

__device__ __forceinline__ f32x4 load4(const void* p, bool is_f32, size_t off) {
  if (is_f32) return *(const f32x4*)((const float*)p + off);
  const uint2 pk = *(const uint2*)((const bf16_t*)p + off);
  return f32x4{__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xffff0000u)};
}
__device__ __forceinline__ void store4(void* p, bool is_f32, size_t off, const f32x4& v) {
  if (is_f32) *(f32x4*)((float*)p + off) = v;
  else {
    uint2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *(uint2*)((bf16_t*)p + off) = pk;
  }
}

__global__ __launch_bounds__(256) void mse_loss_kernel(const void* __restrict__ a, int a_f32, int lda, const void* __restrict__ t, int t_f32,
                                                       int ldt, DkdRowMap tmap, const float* __restrict__ mask, float wod,
                                                       float* __restrict__ loss, void* __restrict__ da, int da_f32, int ldda, int M, int D) {
  __shared__ float red[4];
  const int nv = D >> 2;
  const uint32_t total = (uint32_t)M * (uint32_t)nv;        // < 2^31: checked on the host (64-bit division per element is slow)
  float acc = 0.f;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int m = (int)(i / (uint32_t)nv), c = (int)(i % (uint32_t)nv) * 4;
    const f32x4 av = load4(a, a_f32, (size_t)m * lda + c);
    const f32x4 tv = load4(t, t_f32, (size_t)map_row(tmap, m) * ldt + c);
    const float mk = mask ? mask[m] : 1.f;
    f32x4 d = (av - tv) * mk;
    acc += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    if (da) store4(da, da_f32, (size_t)m * ldda + c, d * (2.f * wod * mk));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * wod);
}

__global__ void mask_select_kernel(const bf16_t* __restrict__ x, const float* __restrict__ tok, const float* __restrict__ mask,
                                   bf16_t* __restrict__ out, int M, int D) {
  const int nv = D >> 2;
  const long total = (long)M * nv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / nv), c = (int)(i % nv) * 4;
    if (mask[m] != 0.f) {
      const f32x4 tv = *(const f32x4*)(tok + c);
      uint2 pk = {pack2bf(tv[0], tv[1]), pack2bf(tv[2], tv[3])};
      *(uint2*)(out + (size_t)m * D + c) = pk;
    } else {
      *(uint2*)(out + (size_t)m * D + c) = *(const uint2*)(x + (size_t)m * D + c);
    }
  }
}

// dx = (1 - mask) * dout ; dtok[d] += sum over masked rows.  grid = (ceil(D/64), row splits)
__global__ void mask_select_bwd_kernel(const bf16_t* __restrict__ dout, const float* __restrict__ mask, bf16_t* __restrict__ dx,
                                       float* __restrict__ dtok, int M, int D, int rows_per_block) {
  __shared__ float red[4][64];
  const int d = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  if (d < D)
    for (int m = m0 + rl; m < m1; m += 4) {
      const bf16_t g = dout[(size_t)m * D + d];
      if (mask[m] != 0.f) {
        s += bf2f(g);
        dx[(size_t)m * D + d] = 0;
      } else {
        dx[(size_t)m * D + d] = g;
      }
    }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && d < D) atomicAdd(&dtok[d], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

inline int grid_for(long work, int block = 256, int cap = 4096) {
  long g = (work + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int dkd_logit_loss(const float* z, const float* soft_target, const int64_t* labels, float smoothing, int32_t kd_mode,
                              const float* z_kd, const float* z_t, float tau, float w_base, float w_kd, float* losses, float* dz,
                              float* dz_kd, int32_t B, int32_t C, void* stream) {
  DKD_CHECK_ARG(z && losses && dz && B > 0 && C > 0, "logit_loss: null operand");
  DKD_CHECK_ARG((soft_target != nullptr) != (labels != nullptr), "logit_loss: exactly one of soft_target / labels");
  DKD_CHECK_ARG(kd_mode >= 0 && kd_mode <= 2, "logit_loss: kd_mode %d", kd_mode);
  DKD_CHECK_ARG(kd_mode == 0 || (z_kd && z_t && dz_kd), "logit_loss: distillation needs z_kd, z_t, dz_kd");
  // algorithmic bytes: logits + targets read, gradients written, per head
  DkdProbeScope probe(4, 0.0, (double)B * C * 4.0 * (kd_mode ? 6.0 : (soft_target ? 3.0 : 2.0)), as_stream(stream));
  hipLaunchKernelGGL(logit_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), z, soft_target, labels, smoothing, kd_mode, z_kd,
                     z_t, tau, w_base, w_kd, losses, dz, dz_kd, B, C);
  DKD_CHECK_LAUNCH("logit_loss");
  return DKD_OK;
}

extern "C" int dkd_topk_correct(const float* z, const int64_t* labels, int32_t B, int32_t C, const int32_t* ks, int32_t nk, float* out,
                                void* stream) {
  DKD_CHECK_ARG(z && labels && ks && out && B > 0 && C > 0, "topk_correct: null operand");
  DKD_CHECK_ARG(nk >= 1 && nk <= 4, "topk_correct: 1..4 values of k (nk=%d)", nk);
  int k[4] = {0, 0, 0, 0};
  for (int i = 0; i < nk; ++i) {
    DKD_CHECK_ARG(ks[i] >= 1, "topk_correct: k must be >= 1");
    k[i] = ks[i];
  }
  hipLaunchKernelGGL(topk_correct_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), z, labels, B, C, k[0], k[1], k[2], k[3], nk, out);
  DKD_CHECK_LAUNCH("topk_correct");
  return DKD_OK;
}

extern "C" int dkd_mse_loss(const void* a, int32_t a_is_f32, int32_t lda, const void* t, int32_t t_is_f32, int32_t ldt, DkdRowMap tmap,
                            const float* mask, float w_over_denom, float* loss, void* da, int32_t da_is_f32, int32_t ldda, int32_t M,
                            int32_t D, void* stream) {
  DKD_CHECK_ARG(a && t && loss && M > 0 && D > 0, "mse_loss: null operand");
  DKD_CHECK_ARG(D % 4 == 0 && lda % 4 == 0 && ldt % 4 == 0 && (!da || ldda % 4 == 0), "mse_loss: D/ld must be multiples of 4");
  DKD_CHECK_ARG((long)M * (D / 4) < (1L << 31), "mse_loss: M * D / 4 must be below 2^31");
  // algorithmic bytes: read a, read t, write da
  DkdProbeScope probe(4, 0.0, (double)M * D * ((a_is_f32 ? 4.0 : 2.0) + (t_is_f32 ? 4.0 : 2.0) + (da ? (da_is_f32 ? 4.0 : 2.0) : 0.0)),
                      as_stream(stream));
  // every block ends with one atomic on the SAME address (~23 ns each, serialised): 512 blocks, not 2048
  hipLaunchKernelGGL(mse_loss_kernel, dim3(grid_for((long)M * D / 4, 256, 512)), dim3(256), 0, as_stream(stream), a, a_is_f32, lda, t,
                     t_is_f32, ldt, tmap, mask, w_over_denom, loss, da, da_is_f32, ldda, M, D);
  DKD_CHECK_LAUNCH("mse_loss");
  return DKD_OK;
}

extern "C" int dkd_mask_select(const void* x, const float* mask_token, const float* mask, void* out, int32_t M, int32_t D, void* stream) {
  DKD_CHECK_ARG(x && mask_token && mask && out && D % 4 == 0, "mask_select: bad arguments");
  hipLaunchKernelGGL(mask_select_kernel, dim3(grid_for((long)M * D / 4)), dim3(256), 0, as_stream(stream), (const bf16_t*)x, mask_token, mask,
                     (bf16_t*)out, M, D);
  DKD_CHECK_LAUNCH("mask_select");
  return DKD_OK;
}

extern "C" int dkd_mask_select_bwd(const void* dout, const float* mask, void* dx, float* dmask_token, int32_t M, int32_t D, void* stream) {
  DKD_CHECK_ARG(dout && mask && dx && dmask_token, "mask_select_bwd: null operand");
  const int col_blocks = cdiv(D, 64);
  int splits = cdiv(1024, col_blocks);
  if (splits > cdiv(M, 64)) splits = cdiv(M, 64);
  const int rpb = cdiv(M, splits);
  splits = cdiv(M, rpb);
  hipLaunchKernelGGL(mask_select_bwd_kernel, dim3(col_blocks, splits), dim3(256), 0, as_stream(stream), (const bf16_t*)dout, mask, (bf16_t*)dx,
                     dmask_token, M, D, rpb);
  DKD_CHECK_LAUNCH("mask_select_bwd");
  return DKD_OK;
}

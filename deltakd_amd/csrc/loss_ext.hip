// Kernels for the feature-distillation branches beyond plain MSE:
//   im2col3x3 / col2im3x3 : MGD's generation block Conv3x3-ReLU-Conv3x3 on the 14x14 token grid (model/loss.py:443-446,
//                            model/models.py:148-151) as (gather -> MFMA GEMM); the gather keeps the channel dim contiguous
//                            (k = (ky*3+kx)*C + c), the conv weights are permuted once to match.
//   sort_l1               : WassKD-L1 (model/loss.py:187-199): per (sample, channel) sort over tokens, mean |sorted diff|.
//                            Rank-by-counting in LDS: no data-dependent control flow, 64-column tiles for coalesced rows.
//   normalize_mse         : DiffKD feature match (model/loss.py:138-139,149): mse(s/|s|, t/|t|) with the normalisation
//                            backward folded in.
//   diffkd_prepare        : DiffKD noising (model/loss.py:138,141-142 + models.py:119-120): t^ = t/|t|, nz = noise*sigma_b,
//                            x = t^ + nz + t_emb[b]  (bf16 GEMM operand), in one pass.
#include "common.h"

namespace {

// x bf16 [B, hw, hw, C] (token-major) -> cols bf16 [B*hw*hw, 9*C];  one thread per 8 channels of one (row, tap)
__global__ void im2col3x3_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ cols, int B, int hw, int C) {
  const int cv = C >> 3;
  const long total = (long)B * hw * hw * 9 * cv;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(t % cv);
    const int tap = (int)((t / cv) % 9);
    const long row = t / ((long)cv * 9);
    const int px = (int)(row % hw), py = (int)((row / hw) % hw);
    const long b = row / ((long)hw * hw);
    const int sy = py + tap / 3 - 1, sx = px + tap % 3 - 1;
    s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (sy >= 0 && sy < hw && sx >= 0 && sx < hw) v = *(const s16x8*)(x + ((b * hw + sy) * hw + sx) * C + c8 * 8);
    *(s16x8*)(cols + row * 9 * C + tap * C + c8 * 8) = v;
  }
}

// dx[b, y, x, c] = sum_taps dcols[(b, y - dy, x - dx)][tap*C + c]  (* (gate > 0) if gate: ReLU backward); bf16 in/out
__global__ void col2im3x3_kernel(const bf16_t* __restrict__ dcols, const bf16_t* __restrict__ gate, bf16_t* __restrict__ dx, int B, int hw,
                                 int C) {
  const int cv = C >> 2;
  const long total = (long)B * hw * hw * cv;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(t % cv) * 4;
    const long row = t / cv;
    const int px = (int)(row % hw), py = (int)((row / hw) % hw);
    const long b = row / ((long)hw * hw);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // output pixel (oy, ox) read input (oy + ky - 1, ox + kx - 1): this pixel feeds (py - ky + 1, px - kx + 1)
      const int oy = py - (tap / 3 - 1), ox = px - (tap % 3 - 1);
      if (oy >= 0 && oy < hw && ox >= 0 && ox < hw) {
        const uint2 pk = *(const uint2*)(dcols + ((b * hw + oy) * hw + ox) * 9 * C + tap * C + c4);
        acc[0] += __uint_as_float(pk.x << 16);
        acc[1] += __uint_as_float(pk.x & 0xffff0000u);
        acc[2] += __uint_as_float(pk.y << 16);
        acc[3] += __uint_as_float(pk.y & 0xffff0000u);
      }
    }
    if (gate) {
      const uint2 gk = *(const uint2*)(gate + row * C + c4);
      if (!(__uint_as_float(gk.x << 16) > 0.f)) acc[0] = 0.f;
      if (!(__uint_as_float(gk.x & 0xffff0000u) > 0.f)) acc[1] = 0.f;
      if (!(__uint_as_float(gk.y << 16) > 0.f)) acc[2] = 0.f;
      if (!(__uint_as_float(gk.y & 0xffff0000u) > 0.f)) acc[3] = 0.f;
    }
    uint2 o = {pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3])};
    *(uint2*)(dx + row * C + c4) = o;
  }
}

// ------------------------------------------------------------------------------------------------ WassKD-L1
constexpr int SL_COLS = 64, SL_PMAX = 256, SL_LD = SL_COLS + 1;
template <bool S_F32, bool T_F32, bool DS_F32>
__global__ __launch_bounds__(256) void sort_l1_kernel(const void* __restrict__ s, const void* __restrict__ t, int ldt, DkdRowMap tmap, float w,
                                                      float* __restrict__ loss, void* __restrict__ ds, int P, int D) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* ss = sm;                       // [P][SL_LD] student tile
  float* ts = ss + SL_PMAX * SL_LD;     // [P][SL_LD] teacher tile
  float* tsorted = ts + SL_PMAX * SL_LD;  // [4 waves][SL_PMAX]
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y, c0 = blockIdx.x * SL_COLS;
  const int ncol = min(SL_COLS, D - c0);
  for (int i = tid; i < P * SL_COLS; i += 256) {
    const int r = i / SL_COLS, c = i % SL_COLS;
    float sv = 0.f, tv = 0.f;
    if (c < ncol) {
      const size_t so = ((size_t)b * P + r) * D + c0 + c;
      const size_t to = (size_t)map_row(tmap, b * P + r) * ldt + c0 + c;
      sv = S_F32 ? ((const float*)s)[so] : bf2f(((const bf16_t*)s)[so]);
      tv = T_F32 ? ((const float*)t)[to] : bf2f(((const bf16_t*)t)[to]);
    }
    ss[r * SL_LD + c] = sv;
    ts[r * SL_LD + c] = tv;
  }
  __syncthreads();
  float acc = 0.f;
  float* tsw = tsorted + wv * SL_PMAX;
  for (int c = wv; c < ncol; c += 4) {
    // rank of each element among its column (ties broken by index): lane owns elements lane + 64 e
    float sv[4], tv[4];
    int rs[4], rt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = lane + 64 * e;
      sv[e] = i < P ? ss[i * SL_LD + c] : 0.f;
      tv[e] = i < P ? ts[i * SL_LD + c] : 0.f;
      rs[e] = rt[e] = 0;
    }
    for (int j = 0; j < P; ++j) {
      const float sj = ss[j * SL_LD + c], tj = ts[j * SL_LD + c];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = lane + 64 * e;
        rs[e] += (sj < sv[e]) || (sj == sv[e] && j < i);
        rt[e] += (tj < tv[e]) || (tj == tv[e] && j < i);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (lane + 64 * e < P) tsw[rt[e]] = tv[e];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes landed (single-wave producer/consumer)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = lane + 64 * e;
      if (i < P) {
        const float dlt = sv[e] - tsw[rs[e]];
        acc += fabsf(dlt);
        const float g = dlt > 0.f ? w : (dlt < 0.f ? -w : 0.f);
        const size_t so = ((size_t)b * P + i) * D + c0 + c;
        if (DS_F32) ((float*)ds)[so] = g;
        else ((bf16_t*)ds)[so] = f2bf(g);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (tid == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * w);
}

// ------------------------------------------------------------------------------------------------ DiffKD helpers
constexpr int MAXV = 4;
// one wave per row: s f32 [M, D] (student aligned feature), that bf16 [M, D] (already normalised teacher feature)
__global__ __launch_bounds__(256) void normalize_mse_kernel(const float* __restrict__ s, const bf16_t* __restrict__ that, const float* wscalar,
                                                            float wod, float* __restrict__ loss, bf16_t* __restrict__ ds, int ldds, int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float wt = (wscalar ? *wscalar : 1.f) * wod;
  const int nv = D >> 2;
  f32x4 sv[MAXV], tv[MAXV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    sv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    tv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nv) {
      sv[i] = *(const f32x4*)(s + (size_t)row * D + 4 * c);
      const uint2 pk = *(const uint2*)(that + (size_t)row * D + 4 * c);
      tv[i] = f32x4{__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16),
                    __uint_as_float(pk.y & 0xffff0000u)};
    }
    ss += sv[i][0] * sv[i][0] + sv[i][1] * sv[i][1] + sv[i][2] * sv[i][2] + sv[i][3] * sv[i][3];
  }
  const float inv = rsqrtf(wave_sum(ss));
  float l = 0.f, dot = 0.f;
  f32x4 g[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sh = sv[i][e] * inv;
      const float d = sh - tv[i][e];
      l += d * d;
      g[i][e] = 2.f * wt * d;      // dL/d s^
      dot += g[i][e] * sh;
      sv[i][e] = sh;
    }
  l = wave_sum(l);
  dot = wave_sum(dot);
  if (lane == 0) atomicAdd(loss, l * wt);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (g[i][e] - sv[i][e] * dot) * inv;   // d s^/d s = (I - s^ s^T)/|s|
      uint2 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
      *(uint2*)(ds + (size_t)row * ldds + 4 * c) = pk;
    }
  }
}

// one wave per row: t bf16 [*, D] rows through tmap; noise f32 [M, D]; sigma f32 [B]; temb f32 [B, D]
__global__ __launch_bounds__(256) void diffkd_prepare_kernel(const bf16_t* __restrict__ t, int ldt, DkdRowMap tmap, const float* __restrict__ noise,
                                                             const float* __restrict__ sigma, const float* __restrict__ temb, int rows_per_sample,
                                                             bf16_t* __restrict__ that, float* __restrict__ nz, bf16_t* __restrict__ xin, int M,
                                                             int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int b = row / rows_per_sample;
  const float sg = sigma[b];
  const int nv = D >> 2;
  const bf16_t* tr = t + (size_t)map_row(tmap, row) * ldt;
  f32x4 tv[MAXV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    tv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nv) {
      const uint2 pk = *(const uint2*)(tr + 4 * c);
      tv[i] = f32x4{__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16),
                    __uint_as_float(pk.y & 0xffff0000u)};
    }
    ss += tv[i][0] * tv[i][0] + tv[i][1] * tv[i][1] + tv[i][2] * tv[i][2] + tv[i][3] * tv[i][3];
  }
  const float inv = rsqrtf(wave_sum(ss));
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const f32x4 n4 = *(const f32x4*)(noise + (size_t)row * D + 4 * c) * sg;
      const f32x4 e4 = *(const f32x4*)(temb + (size_t)b * D + 4 * c);
      const f32x4 th = tv[i] * inv;
      const f32x4 x = th + n4 + e4;
      *(f32x4*)(nz + (size_t)row * D + 4 * c) = n4;
      uint2 p1 = {pack2bf(th[0], th[1]), pack2bf(th[2], th[3])};
      *(uint2*)(that + (size_t)row * D + 4 * c) = p1;
      uint2 p2 = {pack2bf(x[0], x[1]), pack2bf(x[2], x[3])};
      *(uint2*)(xin + (size_t)row * D + 4 * c) = p2;
    }
  }
}

// loss += wod * sum (a*keep*ks - t)^2 ; da = 2 wod (a*keep*ks - t) keep*ks   (Dropout folded into the noise-prediction MSE)
__global__ __launch_bounds__(256) void dropout_mse_kernel(const float* __restrict__ a, const float* __restrict__ t, const float* __restrict__ keep,
                                                          float ks, float wod, float* __restrict__ loss, bf16_t* __restrict__ da, long n4) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 av = *(const f32x4*)(a + 4 * i), tv = *(const f32x4*)(t + 4 * i);
    f32x4 kv = {ks, ks, ks, ks};
    if (keep) kv = *(const f32x4*)(keep + 4 * i) * ks;
    const f32x4 d = av * kv - tv;
    acc += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    const f32x4 g = d * kv * (2.f * wod);
    uint2 pk = {pack2bf(g[0], g[1]), pack2bf(g[2], g[3])};
    *(uint2*)(da + 4 * i) = pk;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * wod);
}

inline int grid_for(long work, int block = 256, int cap = 8192) {
  long g = (work + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <bool A, bool B>
int launch_sort(int ds_f32, dim3 grid, int smem, hipStream_t st, const void* s, const void* t, int ldt, DkdRowMap tmap, float w, float* loss,
                void* ds, int P, int D) {
  auto set = [&](const void* k) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, smem); };
  if (ds_f32) {
    if (set((const void*)sort_l1_kernel<A, B, true>) != hipSuccess) return -1;
    hipLaunchKernelGGL((sort_l1_kernel<A, B, true>), grid, dim3(256), smem, st, s, t, ldt, tmap, w, loss, ds, P, D);
  } else {
    if (set((const void*)sort_l1_kernel<A, B, false>) != hipSuccess) return -1;
    hipLaunchKernelGGL((sort_l1_kernel<A, B, false>), grid, dim3(256), smem, st, s, t, ldt, tmap, w, loss, ds, P, D);
  }
  return 0;
}

}  // namespace

extern "C" int dkd_im2col3x3(const void* x, void* cols, int32_t B, int32_t hw, int32_t C, void* stream) {
  DKD_CHECK_ARG(x && cols && B > 0 && hw > 0 && C % 8 == 0, "im2col3x3: C=%d must be a multiple of 8", C);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for((long)B * hw * hw * 9 * (C / 8))), dim3(256), 0, as_stream(stream), (const bf16_t*)x,
                     (bf16_t*)cols, B, hw, C);
  DKD_CHECK_LAUNCH("im2col3x3");
  return DKD_OK;
}

extern "C" int dkd_col2im3x3(const void* dcols, const void* relu_gate, void* dx, int32_t B, int32_t hw, int32_t C, void* stream) {
  DKD_CHECK_ARG(dcols && dx && B > 0 && hw > 0 && C % 4 == 0, "col2im3x3: bad arguments");
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for((long)B * hw * hw * (C / 4))), dim3(256), 0, as_stream(stream), (const bf16_t*)dcols,
                     (const bf16_t*)relu_gate, (bf16_t*)dx, B, hw, C);
  DKD_CHECK_LAUNCH("col2im3x3");
  return DKD_OK;
}

extern "C" int dkd_sort_l1_loss(const void* s, int32_t s_is_f32, const void* t, int32_t t_is_f32, int32_t ldt, DkdRowMap tmap, float w,
                                float* loss, void* ds, int32_t ds_is_f32, int32_t B, int32_t P, int32_t D, void* stream) {
  DKD_CHECK_ARG(s && t && loss && ds, "sort_l1_loss: null operand");
  DKD_CHECK_ARG(B > 0 && P > 0 && P <= SL_PMAX && D > 0, "sort_l1_loss: need 0 < P <= %d (P=%d)", SL_PMAX, P);
  const int smem = (2 * SL_PMAX * SL_LD + 4 * SL_PMAX) * 4;
  dim3 grid(cdiv(D, SL_COLS), B);
  int rc;
  hipStream_t st = as_stream(stream);
  DkdProbeScope probe(4, 0.0, (double)B * P * D * ((s_is_f32 ? 4.0 : 2.0) + (t_is_f32 ? 4.0 : 2.0) + (ds_is_f32 ? 4.0 : 2.0)), st);
  if (s_is_f32)
    rc = t_is_f32 ? launch_sort<true, true>(ds_is_f32, grid, smem, st, s, t, ldt, tmap, w, loss, ds, P, D)
                  : launch_sort<true, false>(ds_is_f32, grid, smem, st, s, t, ldt, tmap, w, loss, ds, P, D);
  else
    rc = t_is_f32 ? launch_sort<false, true>(ds_is_f32, grid, smem, st, s, t, ldt, tmap, w, loss, ds, P, D)
                  : launch_sort<false, false>(ds_is_f32, grid, smem, st, s, t, ldt, tmap, w, loss, ds, P, D);
  if (rc) {
    dkd_set_error("sort_l1_loss: cannot raise dynamic LDS to %d bytes", smem);
    return DKD_ERR_HIP;
  }
  DKD_CHECK_LAUNCH("sort_l1_loss");
  return DKD_OK;
}

extern "C" int dkd_normalize_mse(const float* s, const void* t_hat, const float* w_scalar, float w_over_denom, float* loss, void* ds,
                                 int32_t ldds, int32_t M, int32_t D, void* stream) {
  DKD_CHECK_ARG(s && t_hat && loss && ds, "normalize_mse: null operand");
  DKD_CHECK_ARG(M > 0 && D % 4 == 0 && D <= 1024 && ldds % 4 == 0, "normalize_mse: need D %% 4 == 0, D <= 1024 (D=%d)", D);
  DkdProbeScope probe(4, 0.0, (double)M * D * (4.0 + 2.0 + 2.0), as_stream(stream));
  hipLaunchKernelGGL(normalize_mse_kernel, dim3(cdiv(M, 4)), dim3(256), 0, as_stream(stream), s, (const bf16_t*)t_hat, w_scalar, w_over_denom,
                     loss, (bf16_t*)ds, ldds, M, D);
  DKD_CHECK_LAUNCH("normalize_mse");
  return DKD_OK;
}

extern "C" int dkd_diffkd_prepare(const void* t, int32_t ldt, DkdRowMap tmap, const float* noise, const float* sigma, const float* temb,
                                  int32_t rows_per_sample, void* t_hat, float* nz, void* x_in, int32_t M, int32_t D, void* stream) {
  DKD_CHECK_ARG(t && noise && sigma && temb && t_hat && nz && x_in, "diffkd_prepare: null operand");
  DKD_CHECK_ARG(M > 0 && rows_per_sample > 0 && D % 4 == 0 && D <= 1024 && ldt % 4 == 0, "diffkd_prepare: bad D=%d", D);
  hipLaunchKernelGGL(diffkd_prepare_kernel, dim3(cdiv(M, 4)), dim3(256), 0, as_stream(stream), (const bf16_t*)t, ldt, tmap, noise, sigma, temb,
                     rows_per_sample, (bf16_t*)t_hat, nz, (bf16_t*)x_in, M, D);
  DKD_CHECK_LAUNCH("diffkd_prepare");
  return DKD_OK;
}

extern "C" int dkd_dropout_mse(const float* a, const float* t, const float* keep, float keep_scale, float w_over_denom, float* loss, void* da,
                               int64_t n, void* stream) {
  DKD_CHECK_ARG(a && t && loss && da && n > 0 && n % 4 == 0, "dropout_mse: bad arguments");
  DkdProbeScope probe(4, 0.0, (double)n * (4.0 + 4.0 + (keep ? 4.0 : 0.0) + 2.0), as_stream(stream));
  hipLaunchKernelGGL(dropout_mse_kernel, dim3(grid_for(n / 4, 256, 2048)), dim3(256), 0, as_stream(stream), a, t, keep, keep_scale,
                     w_over_denom, loss, (bf16_t*)da, (long)(n / 4));
  DKD_CHECK_LAUNCH("dropout_mse");
  return DKD_OK;
}

// LayerNorm forward / backward (HBM-bound; 16-B accesses).
// Replaces nn.LayerNorm(eps=1e-6) inside timm's Block ([3P], reached from model/models.py:195 of the reference).
//
// A row is owned by LPR lanes: 64 (one row per wave, D <= 1024) or, for narrow models (D <= 256: DeiT-tiny's 192), 16 lanes so
// that a wave works on 4 rows at once -- with one row per wave a 192-wide row keeps 48 lanes busy with a single float4 each
// and the two dependent wave reductions per row dominate (the backward ran at half of its HBM bound).
#include "common.h"

namespace {

constexpr int MAXV = 4;  // float4 vectors per lane: D <= LPR * 4 * MAXV

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ f32x4 load_dy(const void* dy, bool f32, size_t off) {
  if (f32) return *(const f32x4*)((const float*)dy + off);
  const uint2 pk = *(const uint2*)((const bf16_t*)dy + off);
  return f32x4{__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xffff0000u)};
}

template <bool OUT_F32, int LPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx, DkdRowMap xmap, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int D, float eps) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sl = lane % LPR, gq = lane / LPR;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + gq;
  const bool live = row < M;
  const float* xr = x + (size_t)map_row(xmap, live ? row : 0) * ldx;
  const int nv = D >> 2;
  f32x4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = sl + LPR * i;
    v[i] = (live && c < nv) ? *(const f32x4*)(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mu = group_sum<LPR>(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = sl + LPR * i;
    if (c < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mu;
        q += d * d;
      }
    }
  }
  const float rs = rsqrtf(group_sum<LPR>(q) / D + eps);
  if (!live) return;
  if (sl == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = sl + LPR * i;
    if (c < nv) {
      const f32x4 g = *(const f32x4*)(gamma + 4 * c), bb = *(const f32x4*)(beta + 4 * c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + bb[e];
      if (OUT_F32) *(f32x4*)((float*)y + (size_t)row * D + 4 * c) = o;
      else {
        uint2 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        *(uint2*)((bf16_t*)y + (size_t)row * D + 4 * c) = pk;
      }
    }
  }
}

// Each block walks LNB_ROWS rows, keeps per-column dgamma/dbeta partials in registers, combines them through LDS and issues one
// f32 atomic per column per block.
constexpr int LNB_ROWS = 64;
template <bool DY_F32, int LPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x, int ldx, DkdRowMap xmap,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx, int lddx, DkdRowMap dxmap,
                                                     int accumulate, float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int D,
                                                     float* __restrict__ part) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[2][4][LPR * 4 * MAXV];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sl = lane % LPR, gq = lane / LPR;
  const int nv = D >> 2;
  f32x4 g[MAXV], ag[MAXV], ab[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = sl + LPR * i;
    g[i] = c < nv ? *(const f32x4*)(gamma + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    ag[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int r0 = blockIdx.x * LNB_ROWS;
  for (int rr = w * RPW; rr < LNB_ROWS; rr += 4 * RPW) {
    const int row = r0 + rr + gq;
    const bool live = row < M;          // lanes of a dead row still take part in the group shuffles
    const float* xr = x + (size_t)map_row(xmap, live ? row : 0) * ldx;
    const float mu = live ? mean[row] : 0.f, rs = live ? rstd[row] : 0.f;
    f32x4 xh[MAXV], gy[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = sl + LPR * i;
      xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      gy[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live && c < nv) {
        const f32x4 xv = *(const f32x4*)(xr + 4 * c);
        const f32x4 d = load_dy(dy, DY_F32, (size_t)row * D + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - mu) * rs;
          ab[i][e] += d[e];
          ag[i][e] += d[e] * xh[i][e];
          gy[i][e] = d[e] * g[i][e];
          s1 += gy[i][e];
          s2 += gy[i][e] * xh[i][e];
        }
      }
    }
    s1 = group_sum<LPR>(s1) / D;
    s2 = group_sum<LPR>(s2) / D;
    if (live) {
      float* dr = dx + (size_t)map_row(dxmap, row) * lddx;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = sl + LPR * i;
        if (c < nv) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs * (gy[i][e] - s1 - xh[i][e] * s2);
          if (accumulate) o += *(const f32x4*)(dr + 4 * c);
          *(f32x4*)(dr + 4 * c) = o;
        }
      }
    }
  }
  // combine the RPW row groups of a wave (lanes sl, sl+LPR, ...), then the 4 waves through LDS
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        ag[i][e] += __shfl_xor(ag[i][e], o, 64);
        ab[i][e] += __shfl_xor(ab[i][e], o, 64);
      }
  if (gq == 0) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = sl + LPR * i;
      if (c < nv) {
        *(f32x4*)&red[0][w][4 * c] = ag[i];
        *(f32x4*)&red[1][w][4 * c] = ab[i];
      }
    }
  }
  __syncthreads();
  // Every block adds to the same 2 D addresses (a dozen cache lines, i.e. a dozen L2 channels): with ~800 blocks the atomics
  // serialise and were half of the kernel's time.  With a workspace the block writes its partial row instead and
  // ln_bwd_reduce_kernel adds the rows up.
  for (int c = threadIdx.x; c < D; c += 256) {
    const float pg = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float pb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    if (part) {
      part[(size_t)blockIdx.x * 2 * D + c] = pg;
      part[(size_t)blockIdx.x * 2 * D + D + c] = pb;
    } else {
      atomicAdd(&dgamma[c], pg);
      atomicAdd(&dbeta[c], pb);
    }
  }
}

// dgamma[c] += sum_b part[b][c], dbeta[c] += sum_b part[b][D + c].  grid (column groups of 64, LNR_CHUNKS row chunks): each block
// sums its chunk of partial rows (4 waves, unrolled by 4 for loads in flight) and adds ONE value per column atomically -- 32
// adds per address instead of one per ln_bwd block.
constexpr int LNR_CHUNKS = 32;
__device__ __forceinline__ void ln_bwd_reduce_body(const float* __restrict__ part, int nblk, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, int D) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;          // column in [0, 2 D)
  const int per = (nblk + LNR_CHUNKS - 1) / LNR_CHUNKS;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float s = 0.f;
  if (c < 2 * D) {
    int b = b0 + w;
    for (; b + 12 < b1; b += 16) {
      const float v0 = part[(size_t)b * 2 * D + c], v1 = part[(size_t)(b + 4) * 2 * D + c], v2 = part[(size_t)(b + 8) * 2 * D + c],
                  v3 = part[(size_t)(b + 12) * 2 * D + c];
      s += (v0 + v1) + (v2 + v3);
    }
    for (; b < b1; b += 4) s += part[(size_t)b * 2 * D + c];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < 2 * D && b0 < b1) {
    const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    atomicAdd(c < D ? &dgamma[c] : &dbeta[c - D], t);
  }
}
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int D) {
  ln_bwd_reduce_body(part, nblk, dgamma, dbeta, D);
}

}  // namespace

// While a capture list is installed (dkd_block_bwd with DkdBlockGrads.ln_defer), reductions are recorded instead of launched: the
// caller runs them later, several blocks' worth in one launch (dkd_ln_bwd_reduce_group).
static thread_local DkdLnReduce* ln_capture = nullptr;
static thread_local int ln_capture_n = 0, ln_capture_cap = 0;
void dkd_ln_capture_begin(DkdLnReduce* items, int cap) { ln_capture = items; ln_capture_n = 0; ln_capture_cap = cap; }
int dkd_ln_capture_end() { const int n = ln_capture_n; ln_capture = nullptr; ln_capture_n = ln_capture_cap = 0; return n; }

int dkd_ln_bwd_reduce(const float* part, int nblk, float* dgamma, float* dbeta, int D, void* stream) {
  if (ln_capture && ln_capture_n < ln_capture_cap) {
    DkdLnReduce& it = ln_capture[ln_capture_n++];
    it.part = part; it.nblk = nblk; it.D = D; it.dgamma = dgamma; it.dbeta = dbeta;
    return DKD_OK;
  }
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cdiv(2 * D, 64), LNR_CHUNKS), dim3(256), 0, as_stream(stream), part, nblk, dgamma, dbeta, D);
  DKD_CHECK_LAUNCH("layernorm_bwd_reduce");
  return DKD_OK;
}

namespace {
constexpr int LNR_GROUP_MAX = 12;
struct LnReduceGroup { DkdLnReduce it[LNR_GROUP_MAX]; };
__global__ __launch_bounds__(256) void ln_bwd_reduce_group_kernel(const LnReduceGroup grp) {
  const DkdLnReduce& q = grp.it[blockIdx.z];
  if ((int)blockIdx.x * 64 >= 2 * q.D) return;
  ln_bwd_reduce_body(q.part, q.nblk, q.dgamma, q.dbeta, q.D);
}
}  // namespace

extern "C" int dkd_ln_bwd_reduce_group(const DkdLnReduce* items, int32_t n, void* stream) {
  DKD_CHECK_ARG(items && n > 0 && n <= LNR_GROUP_MAX, "ln_bwd_reduce_group: need 1..%d reductions (n=%d)", LNR_GROUP_MAX, n);
  LnReduceGroup grp;
  int dmax = 0;
  for (int i = 0; i < n; ++i) {
    DKD_CHECK_ARG(items[i].part && items[i].dgamma && items[i].dbeta && items[i].nblk > 0 && items[i].D > 0, "ln_bwd_reduce_group: bad item %d", i);
    grp.it[i] = items[i];
    if (items[i].D > dmax) dmax = items[i].D;
  }
  DkdProbeScope probe(3, 0.0, 0.0, as_stream(stream));      // part of the student block backward's time (FLOPs / bytes counted there)
  hipLaunchKernelGGL(ln_bwd_reduce_group_kernel, dim3(cdiv(2 * dmax, 64), LNR_CHUNKS, n), dim3(256), 0, as_stream(stream), grp);
  DKD_CHECK_LAUNCH("layernorm_bwd_reduce_group");
  return DKD_OK;
}

extern "C" int dkd_layernorm_fwd(const float* x, int32_t ldx, DkdRowMap xmap, const float* gamma, const float* beta, void* y,
                                 float* mean, float* rstd, int32_t M, int32_t D, float eps, int32_t y_is_f32, void* stream) {
  DKD_CHECK_ARG(x && gamma && beta && y, "layernorm_fwd: null operand");
  DKD_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 1024 && ldx % 4 == 0, "layernorm_fwd: need D %% 4 == 0, D <= 1024 (D=%d ldx=%d)", D, ldx);
  hipStream_t st = as_stream(stream);
  if (D <= 256) {
    const dim3 grid(cdiv(M, 16));
    if (y_is_f32) hipLaunchKernelGGL((ln_fwd_kernel<true, 16>), grid, dim3(256), 0, st, x, ldx, xmap, gamma, beta, y, mean, rstd, M, D, eps);
    else hipLaunchKernelGGL((ln_fwd_kernel<false, 16>), grid, dim3(256), 0, st, x, ldx, xmap, gamma, beta, y, mean, rstd, M, D, eps);
  } else {
    const dim3 grid(cdiv(M, 4));
    if (y_is_f32) hipLaunchKernelGGL((ln_fwd_kernel<true, 64>), grid, dim3(256), 0, st, x, ldx, xmap, gamma, beta, y, mean, rstd, M, D, eps);
    else hipLaunchKernelGGL((ln_fwd_kernel<false, 64>), grid, dim3(256), 0, st, x, ldx, xmap, gamma, beta, y, mean, rstd, M, D, eps);
  }
  DKD_CHECK_LAUNCH("layernorm_fwd");
  return DKD_OK;
}

extern "C" int dkd_layernorm_bwd(const void* dy, int32_t dy_is_f32, const float* x, int32_t ldx, DkdRowMap xmap, const float* gamma,
                                 const float* mean, const float* rstd, float* dx, int32_t lddx, DkdRowMap dxmap, int32_t accumulate,
                                 float* dgamma, float* dbeta, int32_t M, int32_t D, float* ws, void* stream) {
  DKD_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "layernorm_bwd: null operand");
  DKD_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 1024 && ldx % 4 == 0 && lddx % 4 == 0, "layernorm_bwd: bad D=%d", D);
  hipStream_t st = as_stream(stream);
  const dim3 grid(cdiv(M, LNB_ROWS));
#define LNB_ARGS dy, x, ldx, xmap, gamma, mean, rstd, dx, lddx, dxmap, accumulate, dgamma, dbeta, M, D, ws
  if (D <= 256) {
    if (dy_is_f32) hipLaunchKernelGGL((ln_bwd_kernel<true, 16>), grid, dim3(256), 0, st, LNB_ARGS);
    else hipLaunchKernelGGL((ln_bwd_kernel<false, 16>), grid, dim3(256), 0, st, LNB_ARGS);
  } else {
    if (dy_is_f32) hipLaunchKernelGGL((ln_bwd_kernel<true, 64>), grid, dim3(256), 0, st, LNB_ARGS);
    else hipLaunchKernelGGL((ln_bwd_kernel<false, 64>), grid, dim3(256), 0, st, LNB_ARGS);
  }
#undef LNB_ARGS
  DKD_CHECK_LAUNCH("layernorm_bwd");
  if (ws) return dkd_ln_bwd_reduce(ws, (int)grid.x, dgamma, dbeta, D, stream);
  return DKD_OK;
}

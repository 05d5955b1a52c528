// HBM-bound data-movement kernels around the GEMMs: patch gather, token assembly, casts, column sums, AdamW.
// All use 16-byte accesses per lane where the layout allows (cdna_hip_programming Guideline 13).
#include "common.h"

namespace {

// img f32 [B, C, H, W] -> patches bf16 [B*gh*gw, C*p*p], k = c*p*p + i*p + j (Conv2d weight order).
// One thread handles 4 consecutive j (16-B read, 8-B write).
__global__ void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int C, int H, int W, int p, long total4) {
  const int gw = W / p, gh = H / p, K = C * p * p;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total4; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 4;
    const int k = (int)(e % K);
    const long row = e / K;
    const int j = k % p, i = (k / p) % p, c = k / (p * p);
    const int px = (int)(row % gw), py = (int)((row / gw) % gh), b = (int)(row / ((long)gw * gh));
    const float* src = img + (((size_t)b * C + c) * H + (py * p + i)) * W + px * p + j;
    const f32x4 v = *(const f32x4*)src;
    uint2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *(uint2*)(out + e) = pk;
  }
}

__global__ void prefix_tokens_kernel(float* __restrict__ x, const float* __restrict__ tok, const float* __restrict__ pos, int B, int N, int D, int npre) {
  const int total = B * npre * D;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int d = t % D, tk = (t / D) % npre, b = t / (D * npre);
    x[((size_t)b * N + tk) * D + d] = tok[tk * D + d] + pos[tk * D + d];
  }
}

// dpos[t, d] += sum_b dx[b, t, d]; dtok[t, d] += same for t < npre.  grid = (N, splits over B)
__global__ void embed_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dtok, float* __restrict__ dpos, int B, int N, int D, int npre) {
  const int t = blockIdx.x;
  const int b0 = blockIdx.y * 32, b1 = min(B, b0 + 32);
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float s = 0.f;
    for (int b = b0; b < b1; ++b) s += dx[((size_t)b * N + t) * D + d];
    atomicAdd(&dpos[t * D + d], s);
    if (t < npre) atomicAdd(&dtok[t * D + d], s);
  }
}

__global__ void scale_cast_kernel(const float* __restrict__ x, int ldx, DkdRowMap xmap, const float* __restrict__ rowscale, int rps,
                                  const void* __restrict__ add, int add_f32, int ldadd, bf16_t* __restrict__ y, int ldy, int M, int D) {
  const int nv = D >> 2;
  const long total = (long)M * nv;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int m = (int)(t / nv), c = (int)(t % nv) * 4;
    f32x4 v = *(const f32x4*)(x + (size_t)map_row(xmap, m) * ldx + c);
    if (rowscale) v *= rowscale[m / rps];
    if (add) {
      if (add_f32) v += *(const f32x4*)((const float*)add + (size_t)m * ldadd + c);
      else {
        const uint2 pa = *(const uint2*)((const bf16_t*)add + (size_t)m * ldadd + c);
        v += f32x4{__uint_as_float(pa.x << 16), __uint_as_float(pa.x & 0xffff0000u), __uint_as_float(pa.y << 16),
                   __uint_as_float(pa.y & 0xffff0000u)};
      }
    }
    uint2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *(uint2*)(y + (size_t)m * ldy + c) = pk;
  }
}

// w f32 [rows, cols] -> bf16 [rows, cols] and (optional) bf16 [cols, rows]; 32x32 tiles through LDS.
__global__ void cast_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ wb, bf16_t* __restrict__ wt, int rows, int cols) {
  __shared__ bf16_t tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    if (r < rows && c < cols) {
      const bf16_t h = f2bf(w[(size_t)r * cols + c]);
      if (wb) wb[(size_t)r * cols + c] = h;
      tile[i][tx] = h;
    }
  }
  if (!wt) return;
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) wt[(size_t)c * rows + r] = tile[tx][i];
  }
}

// The same for a whole table of matrices in one launch (the transposed shadows of every student weight after an optimizer step:
// ~50 matrices of 37k-150k elements, each too small to fill the GPU and ~5 us as a launch of its own).
// items (device memory): n records {w, wt, rows, cols, first_tile}; first_tile[i] = sum of ceil(rows/32)*ceil(cols/32) before i.
__global__ void cast_weight_group_kernel(const DkdCastItem* __restrict__ items, int n) {
  __shared__ bf16_t tile[32][33];
  __shared__ int which;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n - 1;                       // last item whose first_tile <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].first_tile <= (int)blockIdx.x) lo = mid;
      else hi = mid - 1;
    }
    which = lo;
  }
  __syncthreads();
  const DkdCastItem it = items[which];
  const int t = blockIdx.x - it.first_tile, tcols = (it.cols + 31) >> 5;
  const int r0 = (t / tcols) * 32, c0 = (t % tcols) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* w = it.w;
  bf16_t* wt = (bf16_t*)it.wt;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    if (r < it.rows && c < it.cols) tile[i][tx] = f2bf(w[(size_t)r * it.cols + c]);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < it.rows && c < it.cols) wt[(size_t)c * it.rows + r] = tile[tx][i];
  }
}

// out[n] += sum_m x[xmap(m), n]   grid = (ceil(N/64), row splits); block 256 = 4 row-lanes x 64 columns
template <bool X_F32>
__global__ void colsum_kernel(const void* __restrict__ x, int ldx, DkdRowMap xmap, float* __restrict__ out, int M, int N, int rows_per_block) {
  __shared__ float red[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  if (n < N)
    for (int m = m0 + rl; m < m1; m += 4) {
      const size_t off = (size_t)map_row(xmap, m) * ldx + n;
      s += X_F32 ? ((const float*)x)[off] : bf2f(((const bf16_t*)x)[off]);
    }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && n < N) atomicAdd(&out[n], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <bool X_F32>
__global__ void add_rows_kernel(const void* __restrict__ x, int ldx, float* __restrict__ y, int ldy, DkdRowMap ymap, int M, int D, int accumulate) {
  const long total = (long)M * D;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int m = (int)(t / D), d = (int)(t % D);
    const float v = X_F32 ? ((const float*)x)[(size_t)m * ldx + d] : bf2f(((const bf16_t*)x)[(size_t)m * ldx + d]);
    float* yp = y + (size_t)map_row(ymap, m) * ldy + d;
    *yp = accumulate ? *yp + v : v;
  }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             bf16_t* __restrict__ pb, long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                             float gscale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
    if (pb) pb[i] = f2bf(pi);
  }
}

// The same update for parameters that may receive NO gradient in a step (torch.optim.AdamW skips ``p.grad is None`` entirely: no
// moment decay, no weight decay, no step count): state[0] = largest |g| over the unit this segment belongs to (0: nothing wrote a
// gradient -> skip), state[1] = the unit's own step count, already advanced by the caller for this step.
__global__ void adamw_gated_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                   bf16_t* __restrict__ pb, long n, float lr, float b1, float b2, float eps, float wd,
                                   const float* __restrict__ state) {
  if (state[0] == 0.f) return;
  const float step = state[1];
  const float bc1 = 1.f - powf(b1, step), bc2_sqrt = sqrtf(1.f - powf(b2, step));
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
    if (pb) pb[i] = f2bf(pi);
  }
}

// Mixup / CutMix on a batch resident in HBM (timm Mixup mode='batch' [3P], reached from tools/engine.py:16-18): sample b is
// mixed with sample B-1-b.  The pair is processed by ONE thread per element so the update may be in place (src == x) and is race-free.
//   mixup : x_b <- lam s_b + (1-lam) s_{B-1-b}        cutmix: the box [yl,yh) x [xl,xh) of x_b <- that of s_{B-1-b}, the rest s_b
// src != x (dkd_mixup_to): the same bytes moved as in place, and the caller's batch survives -- a batch that stays resident in HBM across
// steps needs no copy per step to be mixed again.
// PATCHES (dkd_mixup_to_patches): the mix is ALSO written as the bf16 patch matrix [B * gh * gw, C * p * p] the patch-embedding GEMMs of
// student and teacher read (im2col_kernel's layout): the batch is read once, no separate gather pass over the mixed images.
template <bool PATCHES>
__global__ void mixup_kernel(const float* src, float* x, int B, int C, int H, int W, float lam, int cutmix, int yl, int yh, int xl, int xh,
                             bf16_t* __restrict__ patches, int p) {
  const long per = (long)C * H * W;
  const long total4 = (long)(B / 2) * per / 4;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total4; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 4;
    const int b = (int)(e / per);
    const long off = e % per;
    const long oa = (long)b * per + off, ob = (long)(B - 1 - b) * per + off;
    float* pa = x + oa;
    float* pb = x + ob;
    const f32x4 a = *(const f32x4*)(src + oa), c = *(const f32x4*)(src + ob);
    f32x4 na, nc;
    if (!cutmix) {
      na = a * lam + c * (1.f - lam);
      nc = c * lam + a * (1.f - lam);
    } else {
      const int px = (int)(off % W), py = (int)((off / W) % H);
      na = a;
      nc = c;
      if (py >= yl && py < yh) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (px + k >= xl && px + k < xh) {
            na[k] = c[k];
            nc[k] = a[k];
          }
      }
    }
    *(f32x4*)pa = na;
    *(f32x4*)pb = nc;
    if (PATCHES) {
      const int px0 = (int)(off % W), py0 = (int)((off / W) % H), ch = (int)(off / ((long)W * H));
      const int gw = W / p, gh = H / p, K = C * p * p;
      const long rowa = ((long)b * gh + py0 / p) * gw + px0 / p, rowb = ((long)(B - 1 - b) * gh + py0 / p) * gw + px0 / p;
      const int k = ch * p * p + (py0 % p) * p + (px0 % p);
      *(uint2*)(patches + rowa * K + k) = uint2{pack2bf(na[0], na[1]), pack2bf(na[2], na[3])};
      *(uint2*)(patches + rowb * K + k) = uint2{pack2bf(nc[0], nc[1]), pack2bf(nc[2], nc[3])};
    }
  }
}

// soft targets of timm's mixup_target: lam * smooth_onehot(y_b) + (1-lam) * smooth_onehot(y_{B-1-b})
__global__ void mixup_target_kernel(const int64_t* __restrict__ y, float* __restrict__ out, int B, int C, float lam, float on, float off) {
  const long total = (long)B * C;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int b = (int)(t / C), c = (int)(t % C);
    const float v1 = (int)y[b] == c ? on : off, v2 = (int)y[B - 1 - b] == c ? on : off;
    out[t] = v1 * lam + v2 * (1.f - lam);
  }
}

// EMA of the flat parameter buffer (timm ModelEma [3P], tools/engine.py:68-69): ema <- d ema + (1-d) p
__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float d) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) ema[i] = d * ema[i] + (1.f - d) * p[i];
}

// ---- the glue that used to be ATen launches on the step
// x (bf16, n8 groups of 8) *= *s, the product formed in f32: the upstream scalar of an align term's backward arrives as a device tensor.
__global__ void scale_bf16_kernel(bf16_t* __restrict__ x, const float* __restrict__ s, long n8) {
  const float f = *s;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
    uint4 v = *(const uint4*)(x + t * 8);
    uint32_t* w = (uint32_t*)&v;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack2bf(__uint_as_float(w[i] << 16) * f, __uint_as_float(w[i] & 0xffff0000u) * f);
    *(uint4*)(x + t * 8) = v;
  }
}
// dst bf16 [rows, Cp] = bf16(src f32 [rows, C] (row stride lds) * (*scalar or 1)), columns C .. Cp zero (the K padding of the head's dgrad)
__global__ void cast_pad_kernel(const float* __restrict__ src, int lds, const float* __restrict__ scalar, bf16_t* __restrict__ dst, int rows,
                                int C, int Cp) {
  const float f = scalar ? *scalar : 1.f;
  const long total = (long)rows * Cp;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int r = (int)(t / Cp), c = (int)(t % Cp);
    dst[t] = c < C ? f2bf(src[(size_t)r * lds + c] * f) : (bf16_t)0;
  }
}
// DropPath: out[i, b] = (u < keep[i]) / keep[i] with u ~ U[0, 1) from a counter-based generator (splitmix64 of seed and element index):
// timm's x.new_empty(shape).bernoulli_(keep_prob) / keep_prob for all 2 x depth branches of a step in one launch.
__global__ void droppath_scales_kernel(float* __restrict__ out, const float* __restrict__ keep, int n, int B, unsigned long long seed) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * B) return;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(t + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.f / 16777216.f);          // 24 bits
  const float k = keep[t / B];
  out[t] = u < k ? 1.f / k : 0.f;
}

inline int grid_for(long work, int block = 256, int cap = 4096) {
  long g = (work + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int dkd_im2col_patches(const float* img, void* patches, int32_t B, int32_t C, int32_t H, int32_t W, int32_t p, void* stream) {
  DKD_CHECK_ARG(img && patches, "im2col: null operand");
  DKD_CHECK_ARG(p % 4 == 0 && H % p == 0 && W % p == 0, "im2col: patch %d must divide %dx%d and be a multiple of 4", p, H, W);
  const long total4 = (long)B * C * H * W / 4;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total4)), dim3(256), 0, as_stream(stream), img, (bf16_t*)patches, B, C, H, W, p, total4);
  DKD_CHECK_LAUNCH("im2col");
  return DKD_OK;
}

extern "C" int dkd_prefix_tokens_fwd(float* x, const float* tok, const float* pos, int32_t B, int32_t N, int32_t D, int32_t npre, void* stream) {
  DKD_CHECK_ARG(x && tok && pos && npre > 0 && npre <= N, "prefix_tokens: bad arguments");
  hipLaunchKernelGGL(prefix_tokens_kernel, dim3(grid_for((long)B * npre * D)), dim3(256), 0, as_stream(stream), x, tok, pos, B, N, D, npre);
  DKD_CHECK_LAUNCH("prefix_tokens");
  return DKD_OK;
}

extern "C" int dkd_embed_bwd(const float* dx, float* dtok, float* dpos, int32_t B, int32_t N, int32_t D, int32_t npre, void* stream) {
  DKD_CHECK_ARG(dx && dtok && dpos, "embed_bwd: null operand");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(N, cdiv(B, 32)), dim3(256), 0, as_stream(stream), dx, dtok, dpos, B, N, D, npre);
  DKD_CHECK_LAUNCH("embed_bwd");
  return DKD_OK;
}

extern "C" int dkd_scale_cast_bf16(const float* x, int32_t ldx, DkdRowMap xmap, const float* rowscale, int32_t rows_per_sample,
                                   const void* add, int32_t add_is_f32, int32_t ldadd, void* y, int32_t ldy, int32_t M, int32_t D,
                                   void* stream) {
  DKD_CHECK_ARG(x && y, "scale_cast: null operand");
  DKD_CHECK_ARG(D % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (!add || ldadd % 4 == 0), "scale_cast: D/ld must be multiples of 4");
  DKD_CHECK_ARG(!rowscale || rows_per_sample > 0, "scale_cast: rowscale needs rows_per_sample");
  hipLaunchKernelGGL(scale_cast_kernel, dim3(grid_for((long)M * D / 4)), dim3(256), 0, as_stream(stream), x, ldx, xmap, rowscale,
                     rows_per_sample, add, add_is_f32, ldadd, (bf16_t*)y, ldy, M, D);
  DKD_CHECK_LAUNCH("scale_cast");
  return DKD_OK;
}

extern "C" int dkd_cast_weight(const float* w, void* w_bf16, void* w_t_bf16, int32_t rows, int32_t cols, void* stream) {
  DKD_CHECK_ARG(w && (w_bf16 || w_t_bf16) && rows > 0 && cols > 0, "cast_weight: bad arguments");
  hipLaunchKernelGGL(cast_weight_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, as_stream(stream), w, (bf16_t*)w_bf16,
                     (bf16_t*)w_t_bf16, rows, cols);
  DKD_CHECK_LAUNCH("cast_weight");
  return DKD_OK;
}

extern "C" int dkd_cast_weight_group(const DkdCastItem* items_dev, int32_t n, int32_t total_tiles, void* stream) {
  DKD_CHECK_ARG(items_dev && n > 0 && total_tiles > 0, "cast_weight_group: bad arguments");
  hipLaunchKernelGGL(cast_weight_group_kernel, dim3(total_tiles), dim3(256), 0, as_stream(stream), items_dev, n);
  DKD_CHECK_LAUNCH("cast_weight_group");
  return DKD_OK;
}

extern "C" int dkd_colsum(const void* x, int32_t x_is_f32, int32_t ldx, DkdRowMap xmap, float* out, int32_t M, int32_t N, void* stream) {
  DKD_CHECK_ARG(x && out && M > 0 && N > 0, "colsum: bad arguments");
  const int col_blocks = cdiv(N, 64);
  int splits = cdiv(1024, col_blocks);
  if (splits > cdiv(M, 64)) splits = cdiv(M, 64);
  const int rpb = cdiv(M, splits);
  splits = cdiv(M, rpb);
  if (x_is_f32)
    hipLaunchKernelGGL(colsum_kernel<true>, dim3(col_blocks, splits), dim3(256), 0, as_stream(stream), x, ldx, xmap, out, M, N, rpb);
  else
    hipLaunchKernelGGL(colsum_kernel<false>, dim3(col_blocks, splits), dim3(256), 0, as_stream(stream), x, ldx, xmap, out, M, N, rpb);
  DKD_CHECK_LAUNCH("colsum");
  return DKD_OK;
}

extern "C" int dkd_add_rows(const void* x, int32_t x_is_f32, int32_t ldx, float* y, int32_t ldy, DkdRowMap ymap, int32_t M, int32_t D,
                            int32_t accumulate, void* stream) {
  DKD_CHECK_ARG(x && y && M > 0 && D > 0, "add_rows: bad arguments");
  if (x_is_f32)
    hipLaunchKernelGGL(add_rows_kernel<true>, dim3(grid_for((long)M * D)), dim3(256), 0, as_stream(stream), x, ldx, y, ldy, ymap, M, D, accumulate);
  else
    hipLaunchKernelGGL(add_rows_kernel<false>, dim3(grid_for((long)M * D)), dim3(256), 0, as_stream(stream), x, ldx, y, ldy, ymap, M, D, accumulate);
  DKD_CHECK_LAUNCH("add_rows");
  return DKD_OK;
}

extern "C" int dkd_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int32_t step, float grad_scale, void* stream) {
  DKD_CHECK_ARG(p && g && m && v && n > 0 && step > 0, "adamw: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, as_stream(stream), p, g, m, v, (bf16_t*)p_bf16, (long)n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2s, grad_scale);
  DKD_CHECK_LAUNCH("adamw");
  return DKD_OK;
}

extern "C" int dkd_adamw_step_gated(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                                    float eps, float weight_decay, const float* state, void* stream) {
  DKD_CHECK_ARG(p && g && m && v && state && n > 0, "adamw_gated: bad arguments");
  hipLaunchKernelGGL(adamw_gated_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, as_stream(stream), p, g, m, v, (bf16_t*)p_bf16, (long)n, lr,
                     beta1, beta2, eps, weight_decay, state);
  DKD_CHECK_LAUNCH("adamw_gated");
  return DKD_OK;
}

extern "C" int dkd_mixup(float* x, int32_t B, int32_t C, int32_t H, int32_t W, float lam, int32_t cutmix, int32_t yl, int32_t yh, int32_t xl,
                         int32_t xh, void* stream) {
  DKD_CHECK_ARG(x && B > 0 && B % 2 == 0, "mixup: batch size should be even (B=%d)", B);
  DKD_CHECK_ARG(W % 4 == 0, "mixup: W=%d must be a multiple of 4", W);
  hipLaunchKernelGGL(mixup_kernel<false>, dim3(grid_for((long)(B / 2) * C * H * W / 4)), dim3(256), 0, as_stream(stream), x, x, B, C, H, W, lam, cutmix, yl,
                     yh, xl, xh, (bf16_t*)nullptr, 1);
  DKD_CHECK_LAUNCH("mixup");
  return DKD_OK;
}

extern "C" int dkd_mixup_to(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W, float lam, int32_t cutmix, int32_t yl,
                            int32_t yh, int32_t xl, int32_t xh, void* stream) {
  DKD_CHECK_ARG(src && dst && B > 0 && B % 2 == 0, "mixup_to: batch size should be even (B=%d)", B);
  DKD_CHECK_ARG(W % 4 == 0, "mixup_to: W=%d must be a multiple of 4", W);
  const size_t n = (size_t)B * C * H * W;
  DKD_CHECK_ARG(src == dst || src + n <= dst || dst + n <= src, "mixup_to: src and dst overlap without being the same batch");
  hipLaunchKernelGGL(mixup_kernel<false>, dim3(grid_for((long)(B / 2) * C * H * W / 4)), dim3(256), 0, as_stream(stream), src, dst, B, C, H, W, lam,
                     cutmix, yl, yh, xl, xh, (bf16_t*)nullptr, 1);
  DKD_CHECK_LAUNCH("mixup_to");
  return DKD_OK;
}

extern "C" int dkd_mixup_to_patches(const float* src, float* dst, void* patches, int32_t p, int32_t B, int32_t C, int32_t H, int32_t W,
                                    float lam, int32_t cutmix, int32_t yl, int32_t yh, int32_t xl, int32_t xh, void* stream) {
  DKD_CHECK_ARG(src && dst && patches && B > 0 && B % 2 == 0, "mixup_to_patches: batch size should be even (B=%d)", B);
  DKD_CHECK_ARG(p > 0 && p % 4 == 0 && H % p == 0 && W % p == 0, "mixup_to_patches: patch size %d must be a multiple of 4 dividing H, W", p);
  const size_t n = (size_t)B * C * H * W;
  DKD_CHECK_ARG(src + n <= dst || dst + n <= src, "mixup_to_patches: src and dst must not overlap");
  hipLaunchKernelGGL(mixup_kernel<true>, dim3(grid_for((long)(B / 2) * C * H * W / 4)), dim3(256), 0, as_stream(stream), src, dst, B, C, H, W, lam,
                     cutmix, yl, yh, xl, xh, (bf16_t*)patches, p);
  DKD_CHECK_LAUNCH("mixup_to_patches");
  return DKD_OK;
}

extern "C" int dkd_mixup_targets(const int64_t* labels, float* out, int32_t B, int32_t C, float lam, float smoothing, void* stream) {
  DKD_CHECK_ARG(labels && out && B > 0 && C > 0, "mixup_targets: bad arguments");
  const float off = smoothing / C, on = 1.f - smoothing + off;
  hipLaunchKernelGGL(mixup_target_kernel, dim3(grid_for((long)B * C)), dim3(256), 0, as_stream(stream), labels, out, B, C, lam, on, off);
  DKD_CHECK_LAUNCH("mixup_targets");
  return DKD_OK;
}

extern "C" int dkd_ema_update(float* ema, const float* p, int64_t n, float decay, void* stream) {
  DKD_CHECK_ARG(ema && p && n > 0, "ema_update: bad arguments");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, as_stream(stream), ema, p, (long)n, decay);
  DKD_CHECK_LAUNCH("ema_update");
  return DKD_OK;
}

extern "C" int dkd_scale_bf16(void* x, const float* scalar_dev, int64_t n, void* stream) {
  DKD_CHECK_ARG(x && scalar_dev && n >= 0 && n % 8 == 0 && ((uintptr_t)x & 15) == 0, "scale_bf16: n must be a multiple of 8, x 16-byte aligned");
  if (n == 0) return DKD_OK;
  hipLaunchKernelGGL(scale_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, as_stream(stream), (bf16_t*)x, scalar_dev, (long)(n / 8));
  DKD_CHECK_LAUNCH("scale_bf16");
  return DKD_OK;
}

extern "C" int dkd_cast_pad_bf16(const float* src, int32_t ld_src, const float* scalar_dev, void* dst, int32_t rows, int32_t C, int32_t Cp,
                                 void* stream) {
  DKD_CHECK_ARG(src && dst && rows > 0 && C > 0 && Cp >= C && ld_src >= C, "cast_pad_bf16: bad shape");
  hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_for((long)rows * Cp)), dim3(256), 0, as_stream(stream), src, ld_src, scalar_dev, (bf16_t*)dst,
                     rows, C, Cp);
  DKD_CHECK_LAUNCH("cast_pad_bf16");
  return DKD_OK;
}

extern "C" int dkd_droppath_scales(float* out, const float* keep_prob, int32_t n, int32_t B, uint64_t seed, void* stream) {
  DKD_CHECK_ARG(out && keep_prob && n > 0 && B > 0, "droppath_scales: bad shape");
  hipLaunchKernelGGL(droppath_scales_kernel, dim3(cdiv(n * B, 256)), dim3(256), 0, as_stream(stream), out, keep_prob, n, B,
                     (unsigned long long)seed);
  DKD_CHECK_LAUNCH("droppath_scales");
  return DKD_OK;
}

// Shared device/host helpers for libdkd (gfx950 only: wave64, MFMA, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dkd.h"

typedef uint16_t bf16_t;  // raw bf16 payload
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

void dkd_set_error(const char* fmt, ...);

#define DKD_CHECK_ARG(cond, ...)                \
  do {                                          \
    if (!(cond)) {                              \
      dkd_set_error(__VA_ARGS__);               \
      return DKD_ERR_ARG;                       \
    }                                           \
  } while (0)

#define DKD_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      dkd_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return DKD_ERR_HIP;                                                        \
    }                                                                            \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even f32 -> bf16 (plain cast: v_cvt_pk_bf16_f32, NaN stays NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
// one v_cvt_pk_bf16_f32 (two scalar casts + shift/or make the compiler pair the wrong elements and re-shuffle them with SDWA ops)
typedef float dkd_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 dkd_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(dkd_f32x2{lo, hi}, dkd_bf16x2));
}

__device__ __forceinline__ int map_row(const DkdRowMap& m, int r) {
  return m.rpg > 0 ? (r / m.rpg) * m.gstride + (r % m.rpg) + m.off : r;
}

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. at fp32 roundoff of the GELU it feeds): one v_exp + one v_rcp
// instead of libm's branchy erff -- the GELU epilogue of the fc1 GEMM was VALU-bound on erff.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32 (1 ulp): __frcp_rn expands to a full IEEE division
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
// The GELU of the bf16-output epilogues, ONE transcendental per element:  gelu(x) = max(x, 0) - |x| Q(|x|),  Q(a) = 1 - Phi(a) =
// erfc(a / sqrt 2) / 2  (x >= 0: x - x Q;  x < 0: x (1 - Phi(|x|)) = -|x| Q), with  Q(a) = exp2(P6(a))  on [0, 6.5] -- log2 Q is smooth
// (it runs from -1 to -34), a degree-6 polynomial (Chebyshev fit, tools_dev/fit_gelu.py) reproduces Q to 7.8e-5 RELATIVE, i.e. gelu to
// 1.1e-5 absolute and < 8e-5 relative everywhere: 25x below the bf16 rounding of the value it feeds.  Beyond 6.5 the argument is
// clamped (Q < 5e-11).  The previous form (Abramowitz-Stegun 7.1.26: v_rcp + v_exp, both quarter rate) made the teacher's fc1
// epilogue transcendental-bound: 2 x 128 elements x 16 issue cycles per wave and tile = 3.8 us of its 4.5 us.
// The clamp is a compare + select, not v_min_f32: IEEE minNum(NaN, 6.5) = 6.5 would turn a NaN pre-activation (an overflow inside the
// fc1 GEMM) into a finite activation and hide a divergence from the loss; with the select NaN stays NaN through the polynomial.
__device__ __forceinline__ float clamp_abs65(float x) {
  const float ax = fabsf(x);
  return ax > 6.5f ? 6.5f : ax;
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float ax = clamp_abs65(x);
  float p = fmaf(1.9976321103e-05f, ax, -5.5957009936e-04f);
  p = fmaf(p, ax, 6.8683694644e-03f);
  p = fmaf(p, ax, -5.0240056942e-02f);
  p = fmaf(p, ax, -4.6248400028e-01f);
  p = fmaf(p, ax, -1.1496221801e+00f);
  p = fmaf(p, ax, -1.0001030679e+00f);
  return fmaf(-ax, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}
// The same on two values with the polynomial on packed f32 FMAs (v_pk_fma_f32: two lanes' worth per issue slot).  For epilogues in which
// no MFMA is in flight -- there the VALU is the only busy pipe and the six Horner steps are 60 % of its instructions.
__device__ __forceinline__ dkd_f32x2 gelu_erf_fast2(dkd_f32x2 x) {
  const dkd_f32x2 ax = {clamp_abs65(x[0]), clamp_abs65(x[1])};
  auto k = [](float c) { return dkd_f32x2{c, c}; };
  dkd_f32x2 p = __builtin_elementwise_fma(k(1.9976321103e-05f), ax, k(-5.5957009936e-04f));
  p = __builtin_elementwise_fma(p, ax, k(6.8683694644e-03f));
  p = __builtin_elementwise_fma(p, ax, k(-5.0240056942e-02f));
  p = __builtin_elementwise_fma(p, ax, k(-4.6248400028e-01f));
  p = __builtin_elementwise_fma(p, ax, k(-1.1496221801e+00f));
  p = __builtin_elementwise_fma(p, ax, k(-1.0001030679e+00f));
  return dkd_f32x2{fmaf(-ax[0], __builtin_amdgcn_exp2f(p[0]), fmaxf(x[0], 0.f)), fmaf(-ax[1], __builtin_amdgcn_exp2f(p[1]), fmaxf(x[1], 0.f))};
}
// gelu'(x) = Phi(x) + x phi(x) with ONE transcendental.  With a = |x|, e = exp(-a^2 / 2) and Q = 1 - Phi(a):
//     gelu'(x) = 1/2 + sign(x) (1/2 - e U(a)),     U(a) = Q(a) / e - a / sqrt(2 pi)
// (gelu'(-x) = 1 - gelu'(x)).  U is smooth (a Mills ratio minus a line), so a degree-6 polynomial weighted by e fits e U to 1.6e-5
// absolute on [0, 6.5] (tools_dev/fit_gelu.py), two orders below the bf16 rounding of the product it feeds; beyond 6.5 the argument is
// clamped (e < 7e-10).  The previous form (A&S 7.1.26) needed v_rcp and v_exp plus the case split: 16 VALU + 2 transcendentals per
// element, which made the dGELU epilogue (64 elements per thread and tile, three K steps of MFMA) VALU-bound.  Two values at a time:
// the Horner steps are packed f32 FMAs.
__device__ __forceinline__ dkd_f32x2 dgelu_erf_fast2(dkd_f32x2 x) {
  const dkd_f32x2 ax = {clamp_abs65(x[0]), clamp_abs65(x[1])};
  auto k = [](float c) { return dkd_f32x2{c, c}; };
  dkd_f32x2 u = __builtin_elementwise_fma(k(7.0407724585e-04f), ax, k(-8.0406090368e-03f));
  u = __builtin_elementwise_fma(u, ax, k(3.9668294313e-02f));
  u = __builtin_elementwise_fma(u, ax, k(-1.1693547981e-01f));
  u = __builtin_elementwise_fma(u, ax, k(2.4441981382e-01f));
  u = __builtin_elementwise_fma(u, ax, k(-7.9715079225e-01f));
  u = __builtin_elementwise_fma(u, ax, k(4.9998423263e-01f));
  const dkd_f32x2 a2 = ax * ax;
  const dkd_f32x2 e = {__builtin_amdgcn_exp2f(a2[0] * -0.7213475204f), __builtin_amdgcn_exp2f(a2[1] * -0.7213475204f)};
  const dkd_f32x2 t = __builtin_elementwise_fma(-e, u, k(0.5f));            // 1/2 - e U
  dkd_f32x2 r;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    r[i] = 0.5f + __uint_as_float(__float_as_uint(t[i]) ^ (__float_as_uint(x[i]) & 0x80000000u));
  return r;
}
__device__ __forceinline__ float dgelu_erf_fast(float x) { return dgelu_erf_fast2(dkd_f32x2{x, x})[0]; }
__device__ __forceinline__ float dgelu_erf(float x) {
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
// norm.hip: dgamma[c] += sum_b part[b][c], dbeta[c] += sum_b part[b][D + c] over nblk partial rows of 2 D floats (the second launch of
// the LayerNorm backward; also closes the fused dgrad + LayerNorm-backward GEMM of gemm.hip)
int dkd_ln_bwd_reduce(const float* part, int nblk, float* dgamma, float* dbeta, int D, void* stream);
void dkd_ln_capture_begin(DkdLnReduce* items, int cap);      // dkd_ln_bwd_reduce records into `items` instead of launching ...
int dkd_ln_capture_end();                                    // ... until here; returns how many were recorded

// Launch probe of bench.py (api.hip): while dkd_probe_begin() .. dkd_probe_end*() is active, the scope brackets the launches made
// inside it with HIP events recorded on THEIR stream and files them under `sym` with their algorithmic FLOPs / bytes.
//   0 gemm_nt_kernel<128>   1 gemm_nt_kernel<64>   2 gemm_nt256_kernel   3 student block backward (all launches of dkd_block_bwd)
//   4 fused loss kernels    5 student block forward (training)
constexpr int DKD_PROBE_SYMS = 6;
struct DkdProbeScope {
  bool on;
  int sym;
  double flops, bytes;
  hipEvent_t e0, e1;
  hipStream_t st;
  DkdProbeScope(int sym, double flops, double bytes, hipStream_t s);
  ~DkdProbeScope();
};
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

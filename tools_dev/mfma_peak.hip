// Dev microbenchmark: sustained v_mfma_f32_16x16x32_bf16 rate with register-resident operands (no memory traffic), to calibrate
// what "MFMA-bound" means on the box under its power-managed clock.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o bin/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f + i); b[i] = (__bf16)(seed - i); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wpc = 1; wpc <= 4; wpc *= 2) {       // workgroups of 4 waves per CU: 1, 2, 4 waves per SIMD
    const int grid = 256 * wpc;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(spin<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 4 * iters * 16 * 2.0 * 16 * 16 * 32;
      printf("waves/SIMD %d rep %d: %.3f ms  %.1f TFLOP/s\n", wpc, rep, ms, flop / ms * 1e-9);
    }
  }
  return 0;
}

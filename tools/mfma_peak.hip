// Sustained bf16 MFMA rate of the chip with nothing else going on: 8 waves per CU, 16 independent accumulators per wave,
// v_mfma_f32_32x32x16_bf16 back to back.  Context for the roofline fraction (the 2.5 PF dense peak assumes 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    const int iters = 20000 << rep;
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = 256.0 * 8 * iters * 8 * 2.0 * 32 * 32 * 16;
    printf("%d iterations: %.2f ms  %.0f TF/s  (per-CU MFMA clock if 8 passes each: %.2f GHz)\n", iters, ms, flop / ms / 1e9,
           8.0 * iters * 8 * 32 / 4 / (ms * 1e-3) / 1e9);
  }
  return 0;
}

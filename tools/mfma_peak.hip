// Sustained bf16 MFMA rate of the chip with nothing else going on: 8 waves per CU, 16 independent accumulators per wave,
// v_mfma_f32_32x32x16_bf16 back to back.  Context for the roofline fraction (the 2.5 PF dense peak assumes 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC>
void run(const char* what, int threads, float* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 60000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = 256.0 * (threads / 64) * (double)iters * NACC * 2.0 * 32 * 32 * 16;
    if (rep == 1) printf("%-44s %.2f ms  %.0f TF/s\n", what, ms, flop / ms / 1e9);
  }
}
int main() {
  float* out; hipMalloc(&out, 4);
  run<8>("8 waves/CU, 8 independent accumulators:", 512, out);
  run<8>("4 waves/CU (one per SIMD), 8 accumulators:", 256, out);
  run<16>("4 waves/CU (one per SIMD), 16 accumulators:", 256, out);
  run<2>("8 waves/CU, 2 accumulators (dependent chains):", 512, out);
  run<4>("4 waves/CU, 4 accumulators:", 256, out);
  return 0;
}

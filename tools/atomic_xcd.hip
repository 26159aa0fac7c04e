// Do f32 atomics of workgroups that share an output tile get cheaper when those workgroups sit on the same XCD?
// 252 workgroups (36 tiles x 7 splits), each adds a 256x256 f32 tile (65536 atomics) -- the gemm_tn split reduction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(float* out, int tiles, int splits, int mode) {
  int id = blockIdx.x, tile;
  if (mode == 0) tile = id % tiles;                       // as gemm_tn: consecutive ids = consecutive tiles -> splits spread over XCDs
  else { int xcd = id % 8, j = id / 8; tile = xcd + 8 * (j / splits); if (tile >= tiles) return; }   // all splits of a tile on one XCD
  float* t = out + (long long)tile * 65536;
  for (int i = threadIdx.x; i < 65536; i += blockDim.x) atomicAdd(t + i, 1.0f);
}
__global__ void kstore(float* out, int tiles) {            // plain stores for scale
  float* t = out + (long long)(blockIdx.x % tiles) * 65536;
  for (int i = threadIdx.x; i < 65536; i += blockDim.x) t[i] = 1.0f;
}
int main() {
  const int tiles = 40, splits = 6;                        // 40 tiles: 5 per XCD, 30 workgroups per XCD
  float* out; CK(hipMalloc(&out, (size_t)tiles * 65536 * 4)); CK(hipMemset(out, 0, (size_t)tiles * 65536 * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int it = 0; it < 12; ++it) {
      CK(hipEventRecord(a));
      if (mode < 2) hipLaunchKernelGGL(k, dim3(tiles * splits), dim3(512), 0, 0, out, tiles, splits, mode);
      else hipLaunchKernelGGL(kstore, dim3(tiles * splits), dim3(512), 0, 0, out, tiles);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it > 1 && ms < best) best = ms;
    }
    printf("%s: %.1f us for %d x 65536 %s\n", mode == 0 ? "splits spread over XCDs" : mode == 1 ? "splits on one XCD      " : "plain stores           ",
           best * 1e3f, tiles * splits, mode < 2 ? "atomic adds" : "stores");
  }
  return 0;
}

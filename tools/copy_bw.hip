// HBM streaming ceilings on one MI355X for the access mixes of the LayerNorm kernels: read-only, copy (1 read + 1 write stream),
// "ln_bwd-like" (bf16 + 2 x f32 in, f32 + bf16 out), with plain / nontemporal accesses and several grid sizes.
//   hipcc --offload-arch=gfx950 -O3 tools/copy_bw.hip -o tools/copy_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ f32x4 ld(const f32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f32x4* p, f32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT, int U> __global__ __launch_bounds__(256) void read_k(const f32x4* a, float* out, size_t n) {
  f32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) acc += ld<NT>(a + i + u * stride);
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}
template <bool NT, int U> __global__ __launch_bounds__(256) void copy_k(const f32x4* a, f32x4* b, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) v[u] = ld<NT>(a + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) st<NT>(b + i + u * stride, v[u]);
  }
}
// ln_bwd-like: per 4 elements read 8 B (bf16x4 as f32x2) + 16 + 16, write 16 + 8
template <bool NT> __global__ __launch_bounds__(256) void mix_k(const f32x2* dy, const f32x4* x, const f32x4* add, f32x4* o32, f32x2* o16, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    f32x2 d = NT ? __builtin_nontemporal_load(dy + i) : dy[i];
    f32x4 a = ld<NT>(x + i), b = ld<NT>(add + i);
    f32x4 r = a * d[0] + b;
    st<NT>(o32 + i, r);
    f32x2 h = {r[0] + r[1], r[2] + r[3]};
    if (NT) __builtin_nontemporal_store(h, o16 + i); else o16[i] = h;
  }
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) f();
  std::vector<float> ts;
  for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); ts.push_back(t / 10); }
  std::sort(ts.begin(), ts.end()); return ts[2];
}
int main() {
  const size_t n4 = (size_t)100608 * 768 / 4;           // one vision activation in f32x4 units (309 MB)
  f32x4 *a, *b, *c, *d; f32x2 *e, *g; float* out;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&c, n4 * 16)); CK(hipMalloc(&d, n4 * 16)); CK(hipMalloc(&e, n4 * 8)); CK(hipMalloc(&g, n4 * 8)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16)); CK(hipMemset(c, 0, n4 * 16)); CK(hipMemset(e, 0, n4 * 8));
  const double MB = n4 * 16 / 1e6;
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    float t;
    t = timeit([&] { read_k<false, 1><<<grid, 256>>>(a, out, n4); });  printf("grid %5d  read  plain U1: %.2f TB/s\n", grid, MB / t / 1e3);
    t = timeit([&] { read_k<true, 4><<<grid, 256>>>(a, out, n4); });   printf("grid %5d  read  nt    U4: %.2f TB/s\n", grid, MB / t / 1e3);
    t = timeit([&] { copy_k<false, 1><<<grid, 256>>>(a, b, n4); });    printf("grid %5d  copy  plain U1: %.2f TB/s (r+w)\n", grid, 2 * MB / t / 1e3);
    t = timeit([&] { copy_k<true, 1><<<grid, 256>>>(a, b, n4); });     printf("grid %5d  copy  nt    U1: %.2f TB/s (r+w)\n", grid, 2 * MB / t / 1e3);
    t = timeit([&] { copy_k<true, 4><<<grid, 256>>>(a, b, n4); });     printf("grid %5d  copy  nt    U4: %.2f TB/s (r+w)\n", grid, 2 * MB / t / 1e3);
    t = timeit([&] { mix_k<false><<<grid, 256>>>(e, a, c, d, g, n4); });  printf("grid %5d  ln_bwd-like plain: %.2f TB/s\n", grid, 4 * MB / t / 1e3);
    t = timeit([&] { mix_k<true><<<grid, 256>>>(e, a, c, d, g, n4); });   printf("grid %5d  ln_bwd-like nt   : %.2f TB/s\n", grid, 4 * MB / t / 1e3);
  }
  CK(hipMemcpy(b, a, n4 * 16, hipMemcpyDeviceToDevice));
  float t = timeit([&] { CK(hipMemcpyAsync(b, a, n4 * 16, hipMemcpyDeviceToDevice, 0)); }); printf("hipMemcpy D2D: %.2f TB/s (r+w)\n", 2 * MB / t / 1e3);
  return 0;
}

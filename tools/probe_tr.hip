// Diagnostic: prints what ds_read_b64_tr_b16 delivers per lane for a linear per-lane address (lane*8 bytes)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + threadIdx.x * 8));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = t[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) { printf(" %4d", h[l * 4 + j]); if (h[l*4+j] != (l & 15) + j * 16 + (l >> 4) * 64) ok = 0; }
    printf("\n");
  }
  printf("tr semantic matches (l&15)+16j+64(l>>4): %s\n", ok ? "YES" : "NO");
  return 0;
}

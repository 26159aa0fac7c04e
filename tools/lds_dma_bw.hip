// Micro-benchmark: sustained global->LDS (LDS-DMA, 16 B/lane) bandwidth of one 512-thread workgroup per CU, for the access
// patterns of the NT GEMM stage (diagnostic for docs/LAB_r01-r05.md section 4.1).   usage: lds_dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned short u16;

// mode 0: every WG streams its own contiguous region (coalesced 1 KB per wave-instruction)
// mode 1: GEMM-like: thread -> (row, 16-B chunk) of a [256 x 64] tile, row stride ld elements; tile origin per WG
template <int PER_ITER, int SLOTS>
__global__ __launch_bounds__(512) void k(const u16* __restrict__ src, long long wg_stride, int ld, int iters, int mode, int share, long long span, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = share > 0 ? blockIdx.x / share : blockIdx.x;        // WGs sharing the same source (L2 reuse)
  const u16* base = src + ((long long)g * wg_stride) % span;
  unsigned off[PER_ITER];
#pragma unroll
  for (int r = 0; r < PER_ITER; ++r) {
    int c = r * 512 + tid;
    if (mode == 0) off[r] = c * 8;
    else { int row = c >> 3, pc = c & 7, lc = pc ^ ((row >> 1) & 7); off[r] = (unsigned)row * ld + lc * 8; }
  }
  for (int t = 0; t < iters; ++t) {
    char* slot = smem + (t % SLOTS) * (PER_ITER * 8192);
    const long long adv = mode == 0 ? (long long)t * PER_ITER * 4096 : (long long)t * 64;   // next k-slice
#pragma unroll
    for (int r = 0; r < PER_ITER; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (adv % (mode == 0 ? (1 << 20) : ld - 63)) + off[r]),
                                       (__attribute__((address_space(3))) void*)(slot + (r * 512 + wave * 64) * 16), 16, 0, 0);
    if (SLOTS == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (t >= SLOTS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_ITER * (SLOTS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && tid == 0) sink[blockIdx.x] = ((float*)smem)[blockIdx.x & 1023];
}

template <int PER_ITER, int SLOTS>
void run(const char* name, const u16* d, long long wg_stride, int ld, int mode, int share, long long span, float* sink, int nwg) {
  const int iters = 2000;
  const int lds = PER_ITER * 8192 * SLOTS;
  CK(hipFuncSetAttribute((const void*)k<PER_ITER, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds < 131072 ? 131072 : lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int L = lds < 131072 ? 131072 : lds;      // >= 128 KB forces one workgroup per CU
  hipLaunchKernelGGL((k<PER_ITER, SLOTS>), dim3(nwg), dim3(512), L, 0, d, wg_stride, ld, 50, mode, share, span, sink);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<PER_ITER, SLOTS>), dim3(nwg), dim3(512), L, 0, d, wg_stride, ld, iters, mode, share, span, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double bytes = (double)nwg * iters * PER_ITER * 8192.0;
  printf("%-58s per-iter %3d KB slots %d : %7.2f TB/s  (%.1f B/clk/CU @2.0GHz)\n", name, PER_ITER * 8, SLOTS, bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.0e3);
}

int main() {
  const long long N = 512LL << 20;       // 1 GiB of bf16
  u16* d; CK(hipMalloc(&d, N * 2)); CK(hipMemset(d, 0, N * 2));
  float* sink; CK(hipMalloc(&sink, 4096 * 4));
  const int nwg = 256;
  // contiguous private regions, working set 256 x 1 MB = 256 MB (HBM/MALL) vs shared 2 MB (L2)
  run<8, 2>("contiguous, private 2 MB per WG (HBM/MALL)", d, 1 << 20, 0, 0, 0, N, sink, nwg);
  run<8, 2>("contiguous, all WGs same 2 MB (L2 hits)", d, 0, 0, 0, 0, N, sink, nwg);
  run<8, 2>("contiguous, 32 groups of 8 WGs share (L2)", d, 1 << 20, 0, 0, 8, N, sink, nwg);
  run<4, 2>("contiguous, all WGs same 2 MB (L2 hits)", d, 0, 0, 0, 0, N, sink, nwg);
  run<4, 4>("contiguous, all WGs same 2 MB (L2 hits)", d, 0, 0, 0, 0, N, sink, nwg);
  run<8, 1>("contiguous, all WGs same 2 MB (L2 hits), drain each iter", d, 0, 0, 0, 0, N, sink, nwg);
  // GEMM-like tiles: 256 rows x 64 cols per 32 KB; two operands per iteration = 8 chunks/thread
  run<8, 2>("gemm-like ld=768, every WG its own 256-row panel", d, 256LL * 768, 768, 1, 0, N, sink, nwg);
  run<8, 2>("gemm-like ld=768, 9 WGs share a panel", d, 256LL * 768, 768, 1, 9, N, sink, nwg);
  run<8, 2>("gemm-like ld=768, all WGs same panel", d, 0, 768, 1, 0, N, sink, nwg);
  run<8, 2>("gemm-like ld=3072, 3 WGs share a panel", d, 256LL * 3072, 3072, 1, 3, N, sink, nwg);
  run<4, 4>("gemm-like ld=768, 9 WGs share a panel", d, 256LL * 768, 768, 1, 9, N, sink, nwg);
  return 0;
}

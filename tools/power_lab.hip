// Power lab (round 5): what a 256x256x64 NT K loop's ingredients cost in WATTS, added one at a time, each variant alone on the chip for a few
// seconds while tools/power_lab.sh samples rocm-smi.  256 workgroups x 8 waves (the GEMM's occupancy), 128 accumulator registers per wave.
//   0  MFMA only, all-zero operands                         (the instruction stream without data activity)
//   1  MFMA only, random operands held in registers         (matrix-pipe data activity, no memory of any kind)
//   2  1 + the loop's fragment reads: 24 ds_read_b128 per 32 MFMAs from a random 64 KB LDS image (operands come from the reads)
//   3  2 + the loop's LDS-DMA: 8 global_load_lds_dwordx4 per wave and 32 MFMAs, from a per-workgroup 64 KB window that stays in L2
//   4  3 with the DMA source walking a 600 MB operand (fabric / HBM side live)
//   5  MFMA 16x16x32 only, random operands (same flops per instruction pair)
// usage: power_lab <variant> <seconds>      prints TF/s
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

extern "C" void mart_set_error(const char*) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4_;

template <int V>
__global__ __launch_bounds__(512, 1) void k(const bf16* __restrict__ src, float* out, int iters, long long src_elems) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // random LDS image (64 KB) / random register operands
  unsigned x = (unsigned)(blockIdx.x * 512 + tid) * 2654435761u + 12345u;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
  constexpr bool M16 = (V == 5 || V == 6 || V == 7 || V == 9);
  constexpr bool RD = (V >= 2 && V <= 4) || V == 6 || V == 7;
  constexpr bool DMA = V == 3 || V == 4 || V == 6;
  constexpr bool WALK = V == 4 || V == 6;
  if constexpr (RD) {
    for (int c = tid; c < 65536 / 4; c += 512) {
      unsigned r = rnd();
      // two bf16 in [-1, 1): sign + exponent 0x3f00..0x3f7f region keeps magnitudes sane, mantissas random
      ((unsigned*)smem)[c] = (r & 0x807f807fu) | 0x3f003f00u;
    }
    __syncthreads();
  }
  bf16x8 fa[4][4], fb[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      u32x4 w = {0, 0, 0, 0};
      if constexpr (V != 0) w = u32x4{(rnd() & 0x807f807fu) | 0x3f003f00u, (rnd() & 0x807f807fu) | 0x3f003f00u, (rnd() & 0x807f807fu) | 0x3f003f00u, (rnd() & 0x807f807fu) | 0x3f003f00u};
      fa[i][ks] = __builtin_bit_cast(bf16x8, w);
      if (i < 2) {
        u32x4 v = {0, 0, 0, 0};
        if constexpr (V != 0) v = u32x4{(rnd() & 0x807f807fu) | 0x3f003f00u, (rnd() & 0x807f807fu) | 0x3f003f00u, (rnd() & 0x807f807fu) | 0x3f003f00u, (rnd() & 0x807f807fu) | 0x3f003f00u};
        fb[i][ks] = __builtin_bit_cast(bf16x8, v);
      }
    }
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)LDS_PTR(smem);
  const int l31 = lane & 31, h = lane >> 5;
  // per-lane DMA source offsets (bytes): 8 rows x 128 B per wave-instruction, row stride 1536 B (ld = 768 bf16)
  unsigned voff[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { const int c = r * 512 + tid, row = c >> 3, pc = c & 7; voff[r] = (unsigned)row * 1536u + pc * 16u; }
  const char* base = (const char*)src + (long long)blockIdx.x * (512 * 1536);
  for (int it = 0; it < iters; ++it) {
    if constexpr (DMA) {
      const char* b = base;
      if constexpr (WALK) b = (const char*)src + (long long)(((unsigned)it * 256u + blockIdx.x) % 760u) * 786432;      // a fresh 512-row window every iteration
#pragma unroll
      for (int r = 0; r < 8; ++r)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds0 + 65536 + (r * 512 + wave * 64) * 16), "v"(voff[r]), "s"(b) : "memory");
    }
    if constexpr (RD) {
      asm volatile("" ::: "memory");                           // (keeps the loop-invariant reads inside the loop)
      // the loop's fragment reads: A 4 blocks x 4 k-steps, B 2 x 4 (addresses as in the GEMM: row * 128 + swizzled chunk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[i][ks] = *(const bf16x8*)(smem + ((wave >> 2) * 128 + i * 32 + l31) * 128 + (((ks * 2 + h) ^ ((l31 >> 1) & 7)) << 4));
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[j][ks] = *(const bf16x8*)(smem + 32768 + ((wave & 3) * 64 + j * 32 + l31) * 128 + (((ks * 2 + h) ^ ((l31 >> 1) & 7)) << 4));
    }
    if constexpr (M16) {
      // 16x16x32: the wave tile as 8 x 4 blocks of 16 x 16 (4 accumulator registers each), two 32-deep k-steps per 64-deep K-tile: 64 MFMAs of
      // 16 cycles for the flops of 32 MFMAs 32x32x16.  A fragments: fa[i][2 * kk + half] = 16-row block 2 i + half; B fragments: fb[j][2 * kk + half]
      typedef __attribute__((ext_vector_type(4))) float f4;
      auto one = [&](int i, int j, int q, bf16x8 b, bf16x8 a) {
        f4 c = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
        acc[i][j][4 * q] = c[0]; acc[i][j][4 * q + 1] = c[1]; acc[i][j][4 * q + 2] = c[2]; acc[i][j][4 * q + 3] = c[3];
      };
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if constexpr (V == 9) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ha = 0; ha < 2; ++ha) one(i, j, 2 * ha + hb, fb[j][2 * kk + hb], fa[i][2 * kk + ha]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) one(i, j, 2 * ha + hb, fb[j][2 * kk + hb], fa[i][2 * kk + ha]);
        }
      }
    } else if constexpr (V == 8) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = mfma32(fb[j][ks], fa[i][ks], acc[i][j]);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fb[j][ks], fa[i][ks], acc[i][j]);
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[tid] = s;
}

__global__ void fill_rand(unsigned* p, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ 0x9e3779b9u; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    p[i] = (x & 0x807f807fu) | 0x3f003f00u;
  }
}

typedef void (*kern_t)(const bf16*, float*, int, long long);

int main(int argc, char** argv) {
  const int v = argc > 1 ? atoi(argv[1]) : 1;
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  kern_t ks[10] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>};
  const char* names[10] = {"MFMA only, zero operands", "MFMA only, random register operands", "+ fragment reads (24 ds_read_b128 / 32 MFMA)",
                          "+ LDS-DMA from an L2-resident window (8 / 32 MFMA)", "+ LDS-DMA walking 600 MB (fabric / HBM live)", "MFMA 16x16x32 only, random operands",
                          "16x16x32 + fragment reads + LDS-DMA walking 600 MB", "16x16x32 + fragment reads", "MFMA 32x32x16 only, B-stationary issue order",
                          "MFMA 16x16x32 only, B-stationary issue order"};
  const long long elems = 300LL * 1024 * 1024;
  bf16* src; CK(hipMalloc(&src, elems * 2));
  fill_rand<<<2048, 256>>>((unsigned*)src, elems / 2);
  float* out; CK(hipMalloc(&out, 4096));
  CK(hipFuncSetAttribute((const void*)ks[v], hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  const int iters = 4000;                                   // 4000 x 32 MFMAs per wave ~ 0.25 ms at 2 PF
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(ks[v], dim3(256), dim3(512), 131072, 0, src, out, iters, elems);
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  double ms_tot = 0; long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(ks[v], dim3(256), dim3(512), 131072, 0, src, out, iters, elems);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_tot += ms; n += 40;
  }
  const double flop = 256.0 * 8 * (double)iters * 32 * 32768.0;
  printf("variant %d  %-52s %.4f ms / launch  %7.1f TF/s\n", v, names[v], ms_tot / n, flop / (ms_tot / n) / 1e9);
  return 0;
}

// K-loop lab (round 5): the 256x256x64 NT tile loop of gemm_nt.hip taken out of the product kernel so that loop STRUCTURES can be
// compared in one process, on the step's shapes, with rotating A operands -- and so that the shipped loop can be read cycle by cycle.
//   V0  the shipped 8-phase loop (two wave-rows one barrier apart, 8 MFMAs per phase, 8 barriers per K-tile and wave)
//   V1  4-phase loop: the same row ownership, 16 MFMAs per phase (4 barriers per K-tile and wave); every counted vmcnt wait sits in a
//       READ segment, before a barrier that precedes the first read of the half-tiles it retires by ANY wave (formally race-free)
//   V2  no ping-pong: 8 symmetric waves, fragment reads of k-step u+1 interleaved with the MFMAs of k-step u inside each wave, 4-slot ring
//       of 32-deep steps, ONE barrier per step placed mid-step
// STAMP builds record s_memtime at the segment boundaries of one workgroup (sum over the K loop) -> where a phase's cycles go.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imkg_analogy_amd/csrc tools/kloop_lab.hip -o tools/kloop_lab
//   tools/kloop_lab [check] [time] [stamp]
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <type_traits>

extern "C" void mart_set_error(const char*) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct P {
  const bf16* A; const bf16* B; bf16* C;
  int M, N, K, lda, ldb, ldc;
  unsigned* stamps; int stamp_wg;
  int flags;            // 1: no epilogue stores (one checksum store per lane instead)
};

#define SB() __builtin_amdgcn_sched_barrier(0)

// LDS-DMA, 16 B per lane, SADDR form: address = 64-bit scalar base + zero-extended 32-bit lane byte offset (no per-issue VALU)
__device__ __forceinline__ void dma16(unsigned lds_wave_base, unsigned voff, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  // the result lands asynchronously (SMEM): wait inside the block, or the compiler reuses the destination SGPRs while the write is in flight
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int V, bool STAMP>
__global__ __launch_bounds__(512, 1) void lab_kernel(P p) {
  constexpr int BM = 256, BN = 256, NT = 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  const int nk = p.K / 64;
  const bool stamp_me = STAMP && (int)blockIdx.x == p.stamp_wg;
  const int grp = wave >> 2;
  constexpr bool ILV = (V != 2);
  constexpr bool M16 = (V == 3 || V == 5 || V == 6);            // v_mfma_f32_16x16x32_bf16: 8 x 4 blocks of 16 x 16 per wave, 64 MFMAs of 16 cycles per K-tile
  constexpr bool RSEG = (V == 6);                     // the LAGGING wave-row issues its LDS-DMA from the read segments (legal for it one segment earlier; its clusters are bare MFMAs)
  constexpr bool FINE = (V == 4 || V == 5);           // no blanket lgkmcnt(0) behind the barrier: the compiler's per-MFMA counted waits only
  auto ro = [](int i) constexpr { return ILV ? (i >> 1) * 128 + (i & 1) * 32 : i * 32; };
  const int wm0 = (wave / 4) * (ILV ? 64 : 128), wn0 = (wave % 4) * 64;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto bar = [&]() { SB(); __builtin_amdgcn_s_barrier(); SB(); };
  auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); };

  f32x4 acc16[2][4][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc16[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (V != 2) {
    // ---- 64-deep K-tiles, two 64 KB buffers of four 16 KB half-tiles: A0 (rows 0..127) A1 B0 B1; 128-byte rows, 16-byte chunk XOR swizzle
    constexpr int STAGE = 65536, A_BYTES = 32768;
    unsigned offA[4], offB[4];                           // per-lane source BYTE offsets of the 4 chunks of A / B this thread stages
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = r * NT + tid, row = c >> 3, pc = c & 7, lc = pc ^ ((row >> 1) & 7);
      offA[r] = ((unsigned)min(m0 + row, p.M - 1) * (unsigned)p.lda + lc * 8) * 2u;
      offB[r] = ((unsigned)min(n0 + row, p.N - 1) * (unsigned)p.ldb + lc * 8) * 2u;
    }
    int rowA[4], rowB[2], keyA[4], keyB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { rowA[i] = wm0 + ro(i) + l31; keyA[i] = h ^ ((rowA[i] >> 1) & 7); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { rowB[j] = wn0 + j * 32 + l31; keyB[j] = h ^ ((rowB[j] >> 1) & 7); }
    bf16x8 af[2][4], bfr[2][4];
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)LDS_PTR(smem);
    // issue half-tile hf of A / B of K-tile t (2 DMA instructions per wave)
    auto issueA = [&](int t, auto HALF) {
      constexpr int hf = decltype(HALF)::value;
      const char* base = (const char*)p.A + (long long)t * 128;
      const unsigned l = lds0 + (t & 1) * STAGE + wave * 1024;
#pragma unroll
      for (int r = 2 * hf; r < 2 * hf + 2; ++r) dma16(l + r * NT * 16, offA[r], base);
    };
    auto issueB = [&](int t, auto HALF) {
      constexpr int hf = decltype(HALF)::value;
      const char* base = (const char*)p.B + (long long)t * 128;
      const unsigned l = lds0 + (t & 1) * STAGE + A_BYTES + wave * 1024;
#pragma unroll
      for (int r = 2 * hf; r < 2 * hf + 2; ++r) dma16(l + r * NT * 16, offB[r], base);
    };
    auto readA = [&](const char* sA_, auto IH) {
      constexpr int ih = decltype(IH)::value;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[ii][ks] = *(const bf16x8*)(sA_ + rowA[2 * ih + ii] * 128 + (((ks * 2) ^ keyA[2 * ih + ii]) << 4));
    };
    auto readB = [&](const char* sB_, auto J) {
      constexpr int j = decltype(J)::value;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bfr[j][ks] = *(const bf16x8*)(sB_ + rowB[j] * 128 + (((ks * 2) ^ keyB[j]) << 4));
    };
    // stamps: [phase][0 reads issued, 1 MFMA start, 2 MFMA issued, 3 after the closing barrier]; sums of the four segment lengths per phase
    unsigned long long ts[4][4];
    unsigned sum[4][4];
    unsigned long long tprev = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) { ts[a][b] = 0; sum[a][b] = 0; }
#define ST(ph, k) do { if constexpr (STAMP) { ts[ph][k] = memtime(); SB(); } } while (0)

    if constexpr (V == 0) {
      auto mma = [&](auto IH, auto J, auto&& dma) {
        constexpr int ih = decltype(IH)::value, j = decltype(J)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) acc[2 * ih + ii][j] = mfma32(bfr[j][0], af[ii][0], acc[2 * ih + ii][j]);
        SB(); dma(); SB();
#pragma unroll
        for (int ks = 1; ks < 4; ++ks)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[2 * ih + ii][j] = mfma32(bfr[j][ks], af[ii][ks], acc[2 * ih + ii][j]);
        __builtin_amdgcn_s_setprio(0);
      };
      issueA(0, I0{}); issueB(0, I0{}); issueB(0, I1{}); issueA(0, I1{});
      if (nk > 1) { issueA(1, I0{}); issueB(1, I0{}); issueB(1, I1{}); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bar();
      if (grp == 1) bar();
      if constexpr (STAMP) { tprev = memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      for (int t = 0; t < nk; ++t) {
        const char* sA = smem + (t & 1) * STAGE;
        const char* sB = sA + A_BYTES;
        const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
        readA(sA, I0{}); SB(); readB(sB, I0{}); ST(0, 0);
        bar(); lgkm0(); ST(0, 1);
        mma(I0{}, I0{}, [&] { if (more1) issueA(t + 1, I1{}); }); ST(0, 2);
        bar(); ST(0, 3);
        readB(sB, I1{}); ST(1, 0);
        bar(); lgkm0(); ST(1, 1);
        mma(I0{}, I1{}, [&] { if (more2) issueA(t + 2, I0{}); }); ST(1, 2);
        bar(); ST(1, 3);
        readA(sA, I1{}); ST(2, 0);
        bar(); lgkm0(); ST(2, 1);
        mma(I1{}, I1{}, [&] { if (more2) issueB(t + 2, I0{}); }); ST(2, 2);
        bar(); ST(2, 3);
        ST(3, 0);
        bar(); ST(3, 1);
        mma(I1{}, I0{}, [&] { if (more2) issueB(t + 2, I1{}); });
        if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ST(3, 2);
        bar(); ST(3, 3);
        if constexpr (STAMP) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int ph = 0; ph < 4; ++ph) {
            sum[ph][0] += (unsigned)(ts[ph][0] - (ph ? ts[ph - 1][3] : tprev));
            sum[ph][1] += (unsigned)(ts[ph][1] - ts[ph][0]);
            sum[ph][2] += (unsigned)(ts[ph][2] - ts[ph][1]);
            sum[ph][3] += (unsigned)(ts[ph][3] - ts[ph][2]);
          }
          tprev = ts[3][3];
        }
      }
      if (grp == 0) bar();
    } else {
      // ---- V1: S1 = [reads A(ih0) + B(j0) + B(j1) | bar | 16 MFMAs (ih0,j0) (ih0,j1) | bar], S2 = [reads A(ih1) | bar | 16 MFMAs (ih1,j1) (ih1,j0) | bar]
      // issue order per wave:  M-S1(t): X(t) = A1(t+1) [2]      M-S2(t): Y(t) = A0, B0, B1 of t+2 [6]
      // waits (all waves, in the read segments):  end of R-S1(t): vmcnt(6) -> X(t-1) = A1(t) landed (read in R-S2(t), two barriers later for the leading row)
      //                                           end of R-S2(t): vmcnt(2) -> Y(t-1) = A0,B0,B1(t+1) landed (read in R-S1(t+1))
      // WAR: A0,B0,B1(t) last read in R-S1(t) by the lagging row (interval 2), re-staged from M-S2(t) (interval 4 / 5); A1(t) last read in R-S2(t)
      //      (interval 4), re-staged from M-S1(t+1) (interval 6 / 7): at least one full barrier interval after the reads retired.
      // 16x16x32 fragments: lane reads row (l & 15) of a 16-row block, 16-byte chunk kk * 4 + (l >> 4); the image's swizzle key (row >> 1) & 7 is
      // ((l & 15) >> 1) for every block, and the second k-step is the same address ^ 64
      bf16x8 a16[4][2], b16[4][2];
      const int l15 = lane & 15;
      const int colsw = ((lane >> 4) ^ (l15 >> 1)) << 4;
      auto readA16 = [&](const char* sA_, auto IH) {
        constexpr int ih = decltype(IH)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) a16[mb][kk] = *(const bf16x8*)(sA_ + (ih * 128 + wm0 + mb * 16 + l15) * 128 + (colsw ^ (kk << 6)));
      };
      auto readAB16 = [&](const char* sA_, const char* sB_) {            // S1: k-step 0 of B and A first (the order the MFMAs consume them in)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) b16[nb][kk] = *(const bf16x8*)(sB_ + (wn0 + nb * 16 + l15) * 128 + (colsw ^ (kk << 6)));
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) a16[mb][kk] = *(const bf16x8*)(sA_ + (wm0 + mb * 16 + l15) * 128 + (colsw ^ (kk << 6)));
        }
      };
      auto readB16 = [&](const char* sB_) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) b16[nb][kk] = *(const bf16x8*)(sB_ + (wn0 + nb * 16 + l15) * 128 + (colsw ^ (kk << 6)));
      };
      auto mma32x = [&](auto IH, auto&& d0, auto&& d1, auto&& d2) {
        constexpr int ih = decltype(IH)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc16[ih][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b16[nb][kk], a16[mb][kk], acc16[ih][mb][nb], 0, 0, 0);
            SB();
            if (kk == 0 && mb == 0) d0();
            if (kk == 0 && mb == 1) d1();
            if (kk == 0 && mb == 2) d2();
            SB();
          }
        __builtin_amdgcn_s_setprio(0);
      };
      auto mma16 = [&](auto IH, auto JA, auto&& d0, auto&& d1, auto&& d2) {
        constexpr int ih = decltype(IH)::value, ja = decltype(JA)::value, jb = 1 - ja;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[2 * ih + ii][ja] = mfma32(bfr[ja][ks], af[ii][ks], acc[2 * ih + ii][ja]);
          SB();
          if (ks == 0) d0();
          if (ks == 1) d1();
          if (ks == 2) d2();
          SB();
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) acc[2 * ih + ii][jb] = mfma32(bfr[jb][ks], af[ii][ks], acc[2 * ih + ii][jb]);
        __builtin_amdgcn_s_setprio(0);
      };
      // prologue: K-tile 0 complete; of K-tile 1: A0, B0, B1 (= Y(-1)); X(0) = A1(1) comes from M-S1(0)
      issueA(0, I0{}); issueB(0, I0{}); issueB(0, I1{}); issueA(0, I1{});
      if (nk > 1) { issueA(1, I0{}); issueB(1, I0{}); issueB(1, I1{}); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bar();
      if (grp == 1) bar();
      if constexpr (STAMP) { tprev = memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      auto nop = [] {};
      for (int t = 0; t < nk; ++t) {
        const char* sA = smem + (t & 1) * STAGE;
        const char* sB = sA + A_BYTES;
        const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
        // R-S1
        if constexpr (M16) { readAB16(sA, sB); } else { readA(sA, I0{}); SB(); readB(sB, I0{}); SB(); readB(sB, I1{}); }
        // outstanding here (newest first): Y(t-1) [6 if it was issued] | X(t-1) [2] -> X(t-1) landed.  (t = 0: prologue waited already)
        if (more1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool rs = RSEG && grp == 1;
        if (rs && more1) issueA(t + 1, I1{});
        ST(0, 0);
        bar(); if constexpr (!FINE) lgkm0(); ST(0, 1);
        if constexpr (M16) mma32x(I0{}, [&] { if (more1 && !rs) issueA(t + 1, I1{}); }, nop, nop);
        else mma16(I0{}, I0{}, [&] { if (more1) issueA(t + 1, I1{}); }, nop, nop);
        ST(0, 2);
        bar(); ST(0, 3);
        // R-S2
        if constexpr (M16) readA16(sA, I1{}); else readA(sA, I1{});
        // outstanding: X(t) [2 if issued] | Y(t-1) -> Y(t-1) landed
        if (more1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (rs && more2) { issueA(t + 2, I0{}); issueB(t + 2, I0{}); issueB(t + 2, I1{}); }
        ST(1, 0);
        bar(); if constexpr (!FINE) lgkm0(); ST(1, 1);
        if constexpr (M16) mma32x(I1{}, [&] { if (more2 && !rs) issueA(t + 2, I0{}); }, [&] { if (more2 && !rs) issueB(t + 2, I0{}); }, [&] { if (more2 && !rs) issueB(t + 2, I1{}); });
        else mma16(I1{}, I1{}, [&] { if (more2) issueA(t + 2, I0{}); }, [&] { if (more2) issueB(t + 2, I0{}); }, [&] { if (more2) issueB(t + 2, I1{}); });
        ST(1, 2);
        bar(); ST(1, 3);
        if constexpr (STAMP) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {
            sum[ph][0] += (unsigned)(ts[ph][0] - (ph ? ts[ph - 1][3] : tprev));
            sum[ph][1] += (unsigned)(ts[ph][1] - ts[ph][0]);
            sum[ph][2] += (unsigned)(ts[ph][2] - ts[ph][1]);
            sum[ph][3] += (unsigned)(ts[ph][3] - ts[ph][2]);
          }
          tprev = ts[1][3];
        }
      }
      if (grp == 0) bar();
    }
    if constexpr (STAMP) {
      if (stamp_me && lane == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) p.stamps[wave * 16 + a * 4 + b] = sum[a][b];
      }
    }
#undef ST
  } else {
    // ---- V2: 4-slot ring of 32-deep steps.  slot: A [256 rows][64 B] (16 KB) then B (16 KB); 16-byte chunk c' = c ^ ((row >> 2) & 3)
    constexpr int SLOT = 32768, SA = 16384;
    unsigned offA[2], offB[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int c = r * NT + tid, row = c >> 2, pc = c & 3, lc = pc ^ ((row >> 2) & 3);
      offA[r] = ((unsigned)min(m0 + row, p.M - 1) * (unsigned)p.lda + lc * 8) * 2u;
      offB[r] = ((unsigned)min(n0 + row, p.N - 1) * (unsigned)p.ldb + lc * 8) * 2u;
    }
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)LDS_PTR(smem);
    const int ns = p.K / 32;
    auto issue = [&](int s) {
      const char* ba = (const char*)p.A + (long long)s * 64;
      const char* bb = (const char*)p.B + (long long)s * 64;
      const unsigned l = lds0 + (s & 3) * SLOT + wave * 1024;
      dma16(l, offA[0], ba); dma16(l + NT * 16, offA[1], ba);
      dma16(l + SA, offB[0], bb); dma16(l + SA + NT * 16, offB[1], bb);
    };
    // fragment read addresses: row*64 + ((ks*2 + h) ^ key) * 16, key = (row >> 2) & 3 = (l31 >> 2) & 3 for every block
    const int key = (l31 >> 2) & 3;
    const int cA0 = (wm0 + l31) * 64 + ((h ^ key) << 4);          // ks = 0; ks = 1 is the same address ^ 32
    const int cB0 = SA + (wn0 + l31) * 64 + ((h ^ key) << 4);
    bf16x8 fa[2][4], fb[2][2];
    auto rdA = [&](int s, int ks, int buf, int i) { fa[buf][i] = *(const bf16x8*)(smem + (s & 3) * SLOT + ((cA0 + i * 2048) ^ (ks << 5))); };
    auto rdB = [&](int s, int ks, int buf, int j) { fb[buf][j] = *(const bf16x8*)(smem + (s & 3) * SLOT + ((cB0 + j * 2048) ^ (ks << 5))); };
    unsigned long long t0 = 0, t1 = 0, t2 = 0; unsigned sbar = 0, stot = 0;
    // prologue: steps 0, 1, 2 in flight; step 0 landed and visible; fragments of (0, ks 0) requested
    issue(0); if (ns > 1) issue(1); if (ns > 2) issue(2);
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
#pragma unroll
    for (int i = 0; i < 4; ++i) rdA(0, 0, 0, i);
#pragma unroll
    for (int j = 0; j < 2; ++j) rdB(0, 0, 0, j);
    if constexpr (STAMP) { t0 = memtime(); }
    for (int s = 0; s < ns; ++s) {
      // sub-step 0: MFMAs on buffer 0, reads of (s, ks 1) into buffer 1 in between
      SB();
      __builtin_amdgcn_s_setprio(1);
      acc[0][0] = mfma32(fb[0][0], fa[0][0], acc[0][0]); SB(); rdB(s, 1, 1, 0); rdA(s, 1, 1, 0); SB();
      acc[1][0] = mfma32(fb[0][0], fa[0][1], acc[1][0]); SB(); rdA(s, 1, 1, 1); rdB(s, 1, 1, 1); SB();
      acc[0][1] = mfma32(fb[0][1], fa[0][0], acc[0][1]); SB(); rdA(s, 1, 1, 2); rdA(s, 1, 1, 3); SB();
      acc[1][1] = mfma32(fb[0][1], fa[0][1], acc[1][1]);
      acc[2][0] = mfma32(fb[0][0], fa[0][2], acc[2][0]);
      acc[3][0] = mfma32(fb[0][0], fa[0][3], acc[3][0]);
      acc[2][1] = mfma32(fb[0][1], fa[0][2], acc[2][1]);
      acc[3][1] = mfma32(fb[0][1], fa[0][3], acc[3][1]);
      __builtin_amdgcn_s_setprio(0);
      SB();
      // step s+1 landed (own share), then the barrier: everybody's share landed, and everybody is done with step s-1
      if (s + 2 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (STAMP) { t1 = memtime(); }
      bar();
      if constexpr (STAMP) { t2 = memtime(); }
      if (s + 3 < ns) issue(s + 3);
      SB();
      __builtin_amdgcn_s_setprio(1);
      const int sn = s + 1;                                  // past the end: the slot of a step that was never staged -- stale bytes, never used
      acc[0][0] = mfma32(fb[1][0], fa[1][0], acc[0][0]); SB(); rdB(sn, 0, 0, 0); rdA(sn, 0, 0, 0); SB();
      acc[1][0] = mfma32(fb[1][0], fa[1][1], acc[1][0]); SB(); rdA(sn, 0, 0, 1); rdB(sn, 0, 0, 1); SB();
      acc[0][1] = mfma32(fb[1][1], fa[1][0], acc[0][1]); SB(); rdA(sn, 0, 0, 2); rdA(sn, 0, 0, 3); SB();
      acc[1][1] = mfma32(fb[1][1], fa[1][1], acc[1][1]);
      acc[2][0] = mfma32(fb[1][0], fa[1][2], acc[2][0]);
      acc[3][0] = mfma32(fb[1][0], fa[1][3], acc[3][0]);
      acc[2][1] = mfma32(fb[1][1], fa[1][2], acc[2][1]);
      acc[3][1] = mfma32(fb[1][1], fa[1][3], acc[3][1]);
      __builtin_amdgcn_s_setprio(0);
      if constexpr (STAMP) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sbar += (unsigned)(t2 - t1);
      }
    }
    if constexpr (STAMP) {
      unsigned long long te = memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stot = (unsigned)(te - t0);
      if (stamp_me && lane == 0) { p.stamps[wave * 16] = stot; p.stamps[wave * 16 + 1] = sbar; }
    }
  }

  // ---- epilogue (the same for every variant): lane owns row m, 4 consecutive n per register quad -> 8-byte bf16 stores
  if constexpr (M16) {
    if (p.flags & 1) {
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int c = 0; c < 4; ++c) s += acc16[a][b][c][0] + acc16[a][b][c][1] + acc16[a][b][c][2] + acc16[a][b][c][3];
      if (s == 12345.678f) p.C[tid] = (bf16)s;
      return;
    }
#pragma unroll
    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int m = m0 + ih * 128 + wm0 + mb * 16 + (lane & 15);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int n = n0 + wn0 + nb * 16 + 4 * (lane >> 4);
          if (m < p.M && n < p.N) *(bf16x4*)(p.C + (size_t)m * p.ldc + n) = f4_to_bf4(acc16[ih][mb][nb]);
        }
      }
    return;
  }
  if (p.flags & 1) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) p.C[tid] = (bf16)s;
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm0 + ro(i) + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn0 + j * 32 + 8 * q + 4 * h;
        if (m < p.M && n < p.N)
          *(bf16x4*)(p.C + (size_t)m * p.ldc + n) = f4_to_bf4(f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]});
      }
  }
}

// ---- V7: ONE wave per SIMD.  4 waves (2 x 2), wave tile 128 x 128 as 8 x 8 blocks of v_mfma_f32_16x16x32 (256 accumulator registers: the other half
// of the register file that two waves per SIMD cannot use), so that a K-tile costs 128 KB of LDS fragment reads per CU instead of 192 KB.  No partner
// wave hides anything: the fragment reads of k-step kk + 1 are interleaved with the MFMAs of k-step kk inside the wave (one ds_read_b128 per four
// MFMAs), the LDS-DMA of K-tile t + 1 (16 instructions per wave) is spread over the MFMAs of K-tile t, ONE barrier per K-tile (double buffer: the wait
// for K-tile t + 1 and the barrier sit at the end of K-tile t; the first fragments of t + 1 are requested behind it -- that bubble is not hidden).
template <bool STAMP>
__global__ __launch_bounds__(256, 1) void lab7_kernel(P p) {
  constexpr int BM = 256, BN = 256, NT = 256, STAGE = 65536, A_BYTES = 32768;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  const int nk = p.K / 64;
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned offA[8], offB[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int c = r * NT + tid, row = c >> 3, pc = c & 7, lc = pc ^ ((row >> 1) & 7);
    offA[r] = ((unsigned)min(m0 + row, p.M - 1) * (unsigned)p.lda + lc * 8) * 2u;
    offB[r] = ((unsigned)min(n0 + row, p.N - 1) * (unsigned)p.ldb + lc * 8) * 2u;
  }
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)LDS_PTR(smem) + wave * 1024;
  auto issue1 = [&](int t, int i) {                      // DMA instruction i (0..15) of K-tile t: 0..7 = A chunks, 8..15 = B chunks
    const unsigned l = lds0 + (t & 1) * STAGE;
    if (i < 8) dma16(l + i * NT * 16, offA[i], (const char*)p.A + (long long)t * 128);
    else dma16(l + A_BYTES + (i - 8) * NT * 16, offB[i - 8], (const char*)p.B + (long long)t * 128);
  };
  const int colsw = (g4 ^ (l15 >> 1)) << 4;
  const int rdA = (wr * 128 + l15) * 128 + colsw, rdB = A_BYTES + (wc * 128 + l15) * 128 + colsw;
  // fragments are SINGLE-buffered and refreshed in place: A block mb is dead after its row of 8 MFMAs, B block nb after the last row -- the read of the
  // next k-step's fragment goes out right behind the last MFMA that uses the old one (64 fragment registers instead of 128: with 256 accumulators
  // the double-buffered form spilled through v_accvgpr moves, 900 of them per kernel)
  bf16x8 fa[8], fb[8];
  auto rA = [&](const char* s_, int kk, int mb) { fa[mb] = *(const bf16x8*)(s_ + ((rdA + mb * 2048) ^ (kk << 6))); };
  auto rB = [&](const char* s_, int kk, int nb) { fb[nb] = *(const bf16x8*)(s_ + ((rdB + nb * 2048) ^ (kk << 6))); };
#pragma unroll
  for (int i = 0; i < 16; ++i) issue1(0, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SB(); __builtin_amdgcn_s_barrier(); SB();
#pragma unroll
  for (int x = 0; x < 8; ++x) { rB(smem, 0, x); rA(smem, 0, x); }
  for (int t = 0; t < nk; ++t) {
    const char* sT = smem + (t & 1) * STAGE;
    const bool more = t + 1 < nk;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          // accumulator placement by hand: rows 0-5 in AGPRs (192), rows 6-7 in VGPRs (64) -- with all 256 in AGPRs the compiler's allocator
          // shuffled ~150 registers per K-tile between the two files (v_accvgpr moves); the constraint letters pin them
          if (mb < 6) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mb][nb]) : "v"(fb[nb]), "v"(fa[mb]));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[mb][nb]) : "v"(fb[nb]), "v"(fa[mb]));
          if (kk == 0 && mb == 7) { SB(); rB(sT, 1, nb); SB(); }          // last row of k-step 0: B block nb is dead, fetch its k-step 1 fragment
        }
        SB();
        if (kk == 0) rA(sT, 1, mb);                                       // row done: A block mb of k-step 1
        if (more) issue1(t + 1, kk * 8 + mb);                             // one of the 16 DMA instructions of K-tile t + 1 per row
        SB();
      }
    }
    if (more) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // K-tile t + 1 (own share); its buffer was free: everybody finished K-tile t - 1 before the previous barrier
      SB(); __builtin_amdgcn_s_barrier(); SB();
      const char* sN = smem + ((t + 1) & 1) * STAGE;
#pragma unroll
      for (int x = 0; x < 8; ++x) { rB(sN, 0, x); rA(sN, 0, x); }
    }
  }
  if (p.flags & 1) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (s == 12345.678f) p.C[tid] = (bf16)s;
    return;
  }
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    const int m = m0 + wr * 128 + mb * 16 + l15;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const int n = n0 + wc * 128 + nb * 16 + 4 * g4;
      if (m < p.M && n < p.N) *(bf16x4*)(p.C + (size_t)m * p.ldc + n) = f4_to_bf4(acc[mb][nb]);
    }
  }
}

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    float f = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
    uint32_t u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); p[i] = (uint16_t)(u >> 16);
  }
}
__global__ void ref_nt(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K, int lda, int ldb) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= M) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += __uint_as_float((uint32_t)A[(size_t)m * lda + k] << 16) * __uint_as_float((uint32_t)B[(size_t)n * ldb + k] << 16);
  C[(size_t)m * N + n] = acc;
}
static float bf2f_h(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

typedef void (*kern_t)(P);
struct Var { const char* name; kern_t k, ks; int same_as; int threads = 512; };     // same_as: variant whose results this one must equal bit for bit (-1: naive check only)

static void launch(kern_t k, const P& p, hipStream_t st, int threads = 512) {
  static bool set[64] = {};
  static kern_t seen[64];
  int idx = -1;
  for (int i = 0; i < 64; ++i) { if (set[i] && seen[i] == k) { idx = i; break; } if (!set[i]) { idx = i; break; } }
  if (!set[idx]) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); set[idx] = true; seen[idx] = k; }
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  hipLaunchKernelGGL(k, dim3(tiles), dim3(threads), 131072, st, p);
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  bool do_check = false, do_time = false, do_stamp = false;
  for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "check")) do_check = true; if (!strcmp(argv[i], "time")) do_time = true; if (!strcmp(argv[i], "stamp")) do_stamp = true; }
  if (argc == 1) do_check = do_time = do_stamp = true;
  std::vector<Var> vars = {
    {"V0 8-phase (shipped)", lab_kernel<0, false>, lab_kernel<0, true>, -1},
    {"V1 4-phase 16-MFMA", lab_kernel<1, false>, lab_kernel<1, true>, 0},
    {"V2 free-running ring", lab_kernel<2, false>, lab_kernel<2, true>, 0},
    {"V3 4-phase 16x16x32", lab_kernel<3, false>, lab_kernel<3, true>, -1},
    {"V4 = V1, counted lgkm", lab_kernel<4, false>, lab_kernel<4, true>, 0},
    {"V5 = V3, counted lgkm", lab_kernel<5, false>, lab_kernel<5, true>, 3},
    {"V6 = V3, row 1 DMA in R", lab_kernel<6, false>, lab_kernel<6, true>, 3},
    {"V7 one wave per SIMD", lab7_kernel<false>, lab7_kernel<true>, 3, 256},
  };
  const int MV = 256 * 393;
  const size_t maxA = (size_t)MV * 3072;
  uint16_t* A[12]; for (int i = 0; i < 12; ++i) { CK(hipMalloc(&A[i], maxA * 2)); fill_bf16<<<1024, 256>>>(A[i], maxA, 17 + i, 1.0f); }
  uint16_t* B; CK(hipMalloc(&B, (size_t)3072 * 3072 * 2)); fill_bf16<<<1024, 256>>>(B, (size_t)3072 * 3072, 101, 0.05f);
  uint16_t* C[2]; CK(hipMalloc(&C[0], maxA * 2)); CK(hipMalloc(&C[1], maxA * 2));
  float* refC; CK(hipMalloc(&refC, (size_t)1024 * 4096 * 4));
  unsigned* stamps; CK(hipMalloc(&stamps, 8 * 16 * 4));
  CK(hipDeviceSynchronize());
  hipStream_t st; CK(hipStreamCreate(&st));
  int bad = 0;

  auto mkp = [&](int M, int N, int K, int rot, int which, int flags, bool hot) {
    P p; p.A = (const bf16*)A[rot % 12]; p.B = (const bf16*)B; p.C = (bf16*)C[which]; p.M = M; p.N = N; p.K = K; p.lda = hot ? 0 : K; p.ldb = hot ? 0 : K; p.ldc = N;
    p.stamps = stamps; p.stamp_wg = 300; p.flags = flags; return p;
  };

  if (do_check) {
    struct { int M, N, K; } cs[] = {{300, 256, 64}, {777, 512, 128}, {1000, 768, 192}, {12576 + 77, 768, 768}, {9000, 768, 2304}, {5000, 3072, 768}};
    for (auto& c : cs) {
      const size_t ob = (size_t)c.M * c.N * 2;
      std::vector<uint16_t> h0(ob / 2), h1(ob / 2), h2(ob / 2);
      CK(hipMemsetAsync(C[0], 0xff, ob, st));
      launch(vars[0].k, mkp(c.M, c.N, c.K, 0, 0, 0, false), st, vars[0].threads);
      CK(hipStreamSynchronize(st)); CK(hipGetLastError());
      CK(hipMemcpy(h0.data(), C[0], ob, hipMemcpyDeviceToHost));
      // V0 vs naive
      int Mr = std::min(c.M, 1024);
      ref_nt<<<dim3((c.N + 255) / 256, Mr), 256, 0, st>>>(A[0], B, refC, Mr, c.N, c.K, c.K, c.K);
      CK(hipStreamSynchronize(st));
      std::vector<float> hr((size_t)Mr * c.N); CK(hipMemcpy(hr.data(), refC, hr.size() * 4, hipMemcpyDeviceToHost));
      double maxerr = 0; for (int m = 0; m < Mr; ++m) for (int n = 0; n < c.N; ++n) { double r = hr[(size_t)m * c.N + n]; maxerr = std::max(maxerr, fabs(bf2f_h(h0[(size_t)m * c.N + n]) - r) / (1.0 + fabs(r))); }
      printf("check M=%6d N=%5d K=%4d: V0 vs naive max rel err %.3e\n", c.M, c.N, c.K, maxerr);
      if (maxerr > 2e-2) ++bad;
      std::vector<std::vector<uint16_t>> hv(vars.size());
      hv[0] = h0;
      for (size_t v = 1; v < vars.size(); ++v) {
        int races = 0;
        for (int r = 0; r < 10; ++r) {
          CK(hipMemsetAsync(C[1], 0xff, ob, st));
          launch(vars[v].k, mkp(c.M, c.N, c.K, 0, 1, 0, false), st, vars[v].threads);
          CK(hipStreamSynchronize(st)); CK(hipGetLastError());
          CK(hipMemcpy(r ? h2.data() : h1.data(), C[1], ob, hipMemcpyDeviceToHost));
          if (r && memcmp(h1.data(), h2.data(), ob)) ++races;
        }
        hv[v] = h1;
        double me = 0; for (int m = 0; m < Mr; ++m) for (int n = 0; n < c.N; ++n) { double r = hr[(size_t)m * c.N + n]; me = std::max(me, fabs(bf2f_h(h1[(size_t)m * c.N + n]) - r) / (1.0 + fabs(r))); }
        const int sa = vars[v].same_as;
        const bool same = sa < 0 || !memcmp(hv[sa].data(), h1.data(), ob);
        printf("   %-24s vs naive %.3e, %s, run-to-run differences %d/9\n", vars[v].name, me, sa < 0 ? "(own summation order)" : same ? "bitwise == its reference variant" : "DIFFERS from its reference variant", races);
        if (!same || races || me > 2e-2) ++bad;
      }
      // stamp builds compute the same numbers
      for (size_t v = 0; v < vars.size(); ++v) {
        CK(hipMemsetAsync(C[1], 0xff, ob, st));
        launch(vars[v].ks, mkp(c.M, c.N, c.K, 0, 1, 0, false), st, vars[v].threads);
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        CK(hipMemcpy(h1.data(), C[1], ob, hipMemcpyDeviceToHost));
        if (memcmp(hv[v].data(), h1.data(), ob)) { printf("   %-24s STAMP build differs\n", vars[v].name); ++bad; }
      }
    }
  }

  if (do_time) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* name; int M, N, K; } sh[] = {{"qkv  [M,2304,768]", MV, 2304, 768}, {"fc2  [M,768,3072]", MV, 768, 3072}, {"dqkv [M,768,2304]", MV, 768, 2304}, {"fc1  [M,3072,768]", MV, 3072, 768}};
    const int NIT = 12, rounds = 5;
    for (int hot = 0; hot < 2; ++hot)
      for (int noepi = 0; noepi < 2; ++noepi)
        for (auto& s : sh) {
          std::vector<std::vector<float>> ms(vars.size());
          for (size_t v = 0; v < vars.size(); ++v) for (int w = 0; w < 2; ++w) launch(vars[v].k, mkp(s.M, s.N, s.K, w, 0, noepi, hot), st, vars[v].threads);
          for (int r = 0; r < rounds; ++r)
            for (size_t v = 0; v < vars.size(); ++v) {
              CK(hipEventRecord(e0, st));
              for (int it = 0; it < NIT; ++it) launch(vars[v].k, mkp(s.M, s.N, s.K, it, 0, noepi, hot), st, vars[v].threads);
              CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
              float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[v].push_back(t / NIT);
            }
          const double fl = 2.0 * s.M * s.N * (double)s.K;
          printf("time %-18s %s %s:", s.name, hot ? "HOT " : "cold", noepi ? "no-epi" : "epi   ");
          for (size_t v = 0; v < vars.size(); ++v) { std::sort(ms[v].begin(), ms[v].end()); double m = ms[v][ms[v].size() / 2]; printf("  V%zu %.4f ms %6.1f TF/s", v, m, fl / m * 1e-9); }
          printf("\n");
        }
  }

  if (do_stamp) {
    struct { const char* name; int M, N, K; } sh[] = {{"qkv  [M,2304,768]", MV, 2304, 768}, {"fc2  [M,768,3072]", MV, 768, 3072}};
    for (int hot = 0; hot < 2; ++hot)
      for (auto& s : sh) {
        for (size_t v = 0; v < vars.size(); ++v) {
          std::vector<unsigned> hs(128);
          const int nk = s.K / 64;
          for (int wg : {300}) {
            CK(hipMemsetAsync(stamps, 0, 512, st));
            for (int w = 0; w < 3; ++w) { P p = mkp(s.M, s.N, s.K, w, 0, 1, hot); p.stamp_wg = wg; launch(vars[v].ks, p, st, vars[v].threads); }
            CK(hipStreamSynchronize(st)); CK(hipGetLastError());
            CK(hipMemcpy(hs.data(), stamps, 512, hipMemcpyDeviceToHost));
            printf("stamp %-18s %s %-22s wg %3d (cycles per K-tile, s_memtime units)\n", s.name, hot ? "HOT " : "cold", vars[v].name, wg);
            if (v != 2) {
              const int nph = v == 0 ? 4 : 2;
              for (int w : {0, 2, 4, 6}) {
                printf("   wave %d:", w);
                double tot = 0;
                for (int ph = 0; ph < nph; ++ph) {
                  printf("  P%d[rd %5.0f | wait %5.0f | mfma %5.0f | wait %5.0f]", ph + 1, hs[w * 16 + ph * 4] / (double)nk, hs[w * 16 + ph * 4 + 1] / (double)nk, hs[w * 16 + ph * 4 + 2] / (double)nk, hs[w * 16 + ph * 4 + 3] / (double)nk);
                  for (int b = 0; b < 4; ++b) tot += hs[w * 16 + ph * 4 + b] / (double)nk;
                }
                printf("  total %6.0f\n", tot);
              }
            } else {
              for (int w = 0; w < 8; w += 2) printf("   wave %d: total %6.0f per 64-deep K-tile, of it at the barrier %5.0f\n", w, hs[w * 16] / (double)nk, hs[w * 16 + 1] / (double)nk);
            }
          }
        }
      }
  }
  printf(bad ? "LAB: %d FAILURES\n" : "LAB: ok\n", bad);
  return bad ? 1 : 0;
}

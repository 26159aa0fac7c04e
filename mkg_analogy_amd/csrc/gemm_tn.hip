// TN GEMM on gfx950 MFMA:  out[NX,NY] (f32, atomic +=) = alpha * X[M,NX]^T Y[M,NY]   (contraction over rows)
//
// Weight gradients dW = dC^T A of every linear layer, the tied-embedding gradient rows of the scoring head
// (out_rows scatter) and the fusion op's d(vis).  Both operands are contraction-major in memory, so the MFMA
// fragments are fetched with the gfx950 LDS transpose read (ds_read_b64_tr_b16): tiles are staged row-major
// [64 m][256 n] by LDS-DMA and each 16-lane group pulls a 4(m) x 16(n) block already transposed.
// The contraction (M) is split across workgroups; partial tiles are combined with f32 atomics.
// Optional fused column sums of X (bias gradients): VALU sums of the fragments the MFMAs consume anyway.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include "mart_hip.h"

namespace {

struct Args {
  const bf16* X; const bf16* Y; int ldx, ldy;
  int M, NX, NY;
  float* out; int ldo; const int* out_rows;
  float* colsum; int colsum_by_row;
  long long sX, sY, sO;
  int splits, rows_per_split;
  float alpha;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int BNX = 256, BNY = 256, BKM = 64, NT = 512;
constexpr int X_BYTES = BKM * BNX * 2, Y_BYTES = BKM * BNY * 2, STAGE = X_BYTES + Y_BYTES;

__global__ __launch_bounds__(NT) void gemm_tn_kernel(Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31, g1 = (lane >> 4) & 1, pp = lane & 15;

  const int tiles_y = (p.NY + BNY - 1) / BNY;
  // XCD-aware order: consecutive logical ids (= the tiles of ONE split, which share its X / Y row slabs) land on the same
  // XCD and therefore in the same L2; with the natural order every XCD saw a few tiles of every split (+6-10 % per launch)
  const int ntile = gridDim.x;
  const int lid = xcd_remap((int)(blockIdx.x + ntile * blockIdx.y), (int)(ntile * gridDim.y));
  const int tile = lid % ntile, split = lid / ntile;
  const int tx = tile / tiles_y, ty = tile % tiles_y;
  const int nx0 = tx * BNX, ny0 = ty * BNY;
  const long long bz = blockIdx.z;
  const int ms = split * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);
  if (ms >= me) return;
  const bf16* X = p.X + bz * p.sX;
  const bf16* Y = p.Y + bz * p.sY;

  // staging: 2048 16-byte chunks per operand tile, 4 per thread; chunk c -> (row c>>5, physical chunk c&31)
  // logical chunk = physical ^ ((row&3)<<2)  (spreads the 4 rows of a transpose-read block over all 64 banks)
  unsigned colX[4], colY[4]; int rowS[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int c = r * NT + tid, row = c >> 5, pc = c & 31, lc = pc ^ ((row & 3) << 2);
    rowS[r] = row;
    colX[r] = (unsigned)min(nx0 + lc * 8, p.ldx - 8);
    colY[r] = (unsigned)min(ny0 + lc * 8, p.ldy - 8);
  }
  auto stage = [&](int m_base, int buf) {
    char* sX = smem + buf * STAGE;
    char* sY = sX + X_BYTES;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m_base + rowS[r];
      char* dX = sX + (r * NT + wave * 64) * 16;
      char* dY = sY + (r * NT + wave * 64) * 16;
      if (m < me) {
        glds16_raw(X + (long long)m * p.ldx + colX[r], dX);
        glds16_raw(Y + (long long)m * p.ldy + colY[r], dY);
      } else {                                   // contraction tail: zero rows
        *(f32x4*)(dX + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(dY + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  // wave tile: 128 (nx) x 64 (ny) -> 4 x 2 MFMA tiles
  const int wnx0 = (wave >> 2) * 128, wny0 = (wave & 3) * 64;
  f32x16 acc[4][2];
  float cs[4] = {0.f, 0.f, 0.f, 0.f};              // column sums of X (bias gradients), VALU side
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }
  }
  const bool do_colsum = p.colsum && ty == 0;

  // transpose-read addressing.  lane (p = lane&15, g1, h): supplies row (p>>2) of the 4-row block, 8-byte piece p&3
  // of the 16-column half g1; receives column l31 of the 32-wide tile, rows 8h + 4rd + 0..3 of the 16-row k-step.
  const int prow = pp >> 2, key = prow << 2;
  auto frag = [&](const char* tile, int col0, int a) -> bf16x8 {
    const int col = col0 + g1 * 16 + (pp & 3) * 4;               // column of this lane's 8-byte piece
    const int chunk = (col >> 3) ^ key;
    const int boff = chunk * 16 + (col & 7) * 2;
    const int r0 = a * 16 + 8 * h + prow;
    s16x4 lo = lds_tr_read(tile + r0 * 512 + boff);
    s16x4 hi = lds_tr_read(tile + (r0 + 4) * 512 + boff);
    return join_tr(lo, hi);
  };

  const int nsteps = (me - ms + BKM - 1) / BKM;
  stage(ms, 0);
  for (int t = 0; t < nsteps; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own share of tile t landed (DMA is opaque asm: no compiler wait)
    __syncthreads();
    if (t + 1 < nsteps) stage(ms + (t + 1) * BKM, (t + 1) & 1);
    const char* sX = smem + (t & 1) * STAGE;
    const char* sY = sX + X_BYTES;
    // fragments of k-step a+1 are fetched while the MFMAs of k-step a run (the wave's own LDS latency is hidden by its
    // own matrix work, not only by the second wave of the SIMD)
    bf16x8 xf[2][4], yf[2][2];
    auto fetch = [&](int a, int slot) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[slot][i] = frag(sX, wnx0 + i * 32, a);
#pragma unroll
      for (int j = 0; j < 2; ++j) yf[slot][j] = frag(sY, wny0 + j * 32, a);
    };
    fetch(0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int cur = a & 1;
      if (a + 1 < 4) fetch(a + 1, cur ^ 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][0] = mfma32(xf[cur][i], yf[cur][0], acc[i][0]);
        acc[i][1] = mfma32(xf[cur][i], yf[cur][1], acc[i][1]);
      }
      if (do_colsum && (wave & 3) == a) {                         // k-step a of every tile belongs to wave column a
        // each lane holds 8 contraction rows of column nx = ... + 32 i + l31: sum them on the VALU, which idles next to
        // the matrix pipe (the first version spent 4 extra MFMAs per k-step against a ones operand here: +50 % MFMAs on
        // the two waves that set the pace of the ty == 0 workgroups, 7-22 % of the kernel, and 64 accumulator VGPRs)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 w = __builtin_bit_cast(u32x4, xf[cur][i]);
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 += __builtin_bit_cast(float, w[e] << 16);
            s1 += __builtin_bit_cast(float, w[e] & 0xffff0000u);
          }
          cs[i] += s0 + s1;
        }
      }
    }
  }

  if (do_colsum) {                                  // workgroup-uniform
    // lanes l and l+32 saw the two halves of every 16-row k-step, the four wave columns one k-step each: combine through
    // LDS so that the workgroup issues ONE atomic per column (contended cross-XCD atomics are what is expensive here)
    __syncthreads();                                // K-loop buffers are free
    float* red = (float*)smem;                      // [4 wave columns][256 nx]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float tot = cs[i] + __shfl_xor(cs[i], 32);
      if (h == 0) red[(wave & 3) * BNX + wnx0 + i * 32 + l31] = tot;
    }
    __syncthreads();
    if (tid < BNX) {
      const int nx = nx0 + tid;
      const float tot = red[tid] + red[BNX + tid] + red[2 * BNX + tid] + red[3 * BNX + tid];
      if (nx < p.NX) atomicAdd(p.colsum + (p.colsum_by_row ? (long long)p.out_rows[nx] : (long long)nx), tot * p.alpha);
    }
  }
  // epilogue: lane holds column ny = l31 of each tile and 16 rows nx
  float* out = p.out + bz * p.sO;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nx = nx0 + wnx0 + i * 32 + mfma_row(r, h);
      if (nx >= p.NX) continue;
      const long long orow = p.out_rows ? p.out_rows[nx] : nx;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ny = ny0 + wny0 + j * 32 + l31;
        if (ny < p.NY) atomicAdd(out + orow * p.ldo + ny, acc[i][j][r] * p.alpha);
      }
    }
  }
}

// ============================================================================================================================
// Deterministic variant: 8-phase K loop + two-stage split reduction (no atomics on the output).
//   * K loop: the structure of gemm_nt's 8-phase loop (gemm_nt.hip) on transposed fragment reads.  A 64-row contraction step is
//     staged as FOUR half-tiles of 16 KB -- X columns 0..127 (Xl), 128..255 (Xr), Y columns 0..127 (Yl), 128..255 (Yr) --
//     each [64 m][256 B] with the 16-byte-chunk swizzle chunk ^ ((m & 3) << 2) (the four rows of a transposed 4x16 read land in
//     four different 64-byte bank groups); two steps fit the 128 KB ring.  Four phases per step, each = [ transposed fragment
//     reads (+ ONE half-tile of LDS-DMA) | s_barrier | 8 MFMAs | s_barrier ], the two wave-rows one barrier apart.  Wave-row wr
//     owns the X columns {ih*128 + wr*64 + [0,64)}: Xl is dead after phase 1, Yl/Yr after phase 2, Xr after phase 3, so the
//     half-tiles of step t+2 are issued while step t computes, behind ONE counted s_waitcnt vmcnt(6) per step.
//   * split reduction: every (tile, split) workgroup writes its 256x256 f32 partial to a workspace slab in FRAGMENT-MAJOR order
//     (16 bytes per lane, 1 KB per store instruction); tn_reduce_k sums the slabs of a tile in split order and adds the result
//     into the output (and the column sums into the bias gradient): run-to-run identical gradients, no contended atomics.
constexpr int HALF_BYTES = 64 * 256;               // one half-tile
constexpr int SLAB = 256 * 256;                    // floats per (tile, split) partial

struct Args2 {
  const bf16* X; const bf16* Y; int ldx, ldy;
  int M, NX, NY;
  float* out; int ldo; const int* out_rows;
  float* colsum; int colsum_by_row;
  int splits, rows_per_split, tiles_y, ntile;
  float alpha;
  float* ws;                                       // [split][tile][SLAB] partial tiles, then [split][tiles_x][256] column sums
  float* ws_col;
};

__global__ __launch_bounds__(NT) void gemm_tn8_kernel(Args2 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31, g1 = (lane >> 4) & 1, pp = lane & 15;
  const int wr = wave >> 2, wq = wave & 3;
  const int ntile = p.ntile;
  const int lid = xcd_remap((int)(blockIdx.x + ntile * blockIdx.y), (int)(ntile * gridDim.y));
  const int tile = lid % ntile, split = lid / ntile;
  const int tx = tile / p.tiles_y, ty = tile % p.tiles_y;
  const int nx0 = tx * BNX, ny0 = ty * BNY;
  const int ms = split * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);
  const int nsteps = (me - ms) / BKM;                // host guarantees whole steps (M % 64 == 0) and >= 1 step per split

  // staging: a half-tile = 1024 chunks of 16 B, two per thread: chunk c -> row c>>4, physical chunk c&15,
  // logical chunk = physical ^ ((row & 3) << 2)
  long long srcX[2][2], srcY[2][2];                  // [half][r]: element offset of this thread's chunk at contraction row 0
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int c = r * NT + tid, row = c >> 4, pc = c & 15, lc = pc ^ ((row & 3) << 2);
      srcX[hf][r] = (long long)row * p.ldx + min(nx0 + hf * 128 + lc * 8, p.ldx - 8);
      srcY[hf][r] = (long long)row * p.ldy + min(ny0 + hf * 128 + lc * 8, p.ldy - 8);
    }
  auto issueX = [&](int t, auto HALF) {
    constexpr int hf = decltype(HALF)::value;
    const bf16* base = p.X + (long long)(ms + t * BKM) * p.ldx;
    char* dst = smem + (t & 1) * STAGE + hf * HALF_BYTES;
#pragma unroll
    for (int r = 0; r < 2; ++r) glds16_raw(base + srcX[hf][r], dst + (r * NT + wave * 64) * 16);
  };
  auto issueY = [&](int t, auto HALF) {
    constexpr int hf = decltype(HALF)::value;
    const bf16* base = p.Y + (long long)(ms + t * BKM) * p.ldy;
    char* dst = smem + (t & 1) * STAGE + X_BYTES + hf * HALF_BYTES;
#pragma unroll
    for (int r = 0; r < 2; ++r) glds16_raw(base + srcY[hf][r], dst + (r * NT + wave * 64) * 16);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = p.colsum && ty == 0;

  // transposed fragment of a half-tile: 32 columns starting at col0 (inside the half), contraction rows 16a + 8h + {0..7}
  const int prow = pp >> 2, key = prow << 2;
  auto frag = [&](const char* half, int col0, int a) -> bf16x8 {
    const int col = col0 + g1 * 16 + (pp & 3) * 4;
    const int boff = (((col >> 3) ^ key) << 4) + (col & 7) * 2;
    const int r0 = a * 16 + 8 * h + prow;
    s16x4 lo = lds_tr_read(half + r0 * 256 + boff);
    s16x4 hi = lds_tr_read(half + (r0 + 4) * 256 + boff);
    return join_tr(lo, hi);
  };
  // Harness-only knock-outs (tools/tn_one_transposed.sh, -DMART_EXPERIMENTS -DTN_KO_PLAIN=1|2|3): the fragments of Y (1), X (2) or both (3) come from ONE
  // plain ds_read_b128 each instead of two transposed reads -- WRONG RESULTS, same LDS-DMA traffic, same MFMAs: the upper bound of what an operand
  // that arrives M-contiguous (a transposed copy written by its producer) could save in the loop.
#if defined(MART_EXPERIMENTS) && defined(TN_KO_PLAIN)
  constexpr int KO = TN_KO_PLAIN;
#else
  constexpr int KO = 0;
#endif
  auto frag_plain = [&](const char* half, int col0, int a) -> bf16x8 {      // (knock-out) a conflict-free 16-byte read per lane: row l31 of the 64 x 256 B image
    return *(const bf16x8*)(half + (a * 16 + (l31 >> 1)) * 256 + ((((col0 >> 3) + (l31 & 1) * 2 + h) ^ ((l31 >> 1) & 3) << 2) & 15) * 16);
  };
  bf16x8 xf[2][4], yf[2][4];                         // X fragments [block of the pair][k-step], Y fragments [j][k-step]
  auto readX = [&](const char* st, auto IH) {
    constexpr int ih = decltype(IH)::value;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int a = 0; a < 4; ++a) xf[ii][a] = (KO & 2) ? frag_plain(st + ih * HALF_BYTES, wr * 64 + ii * 32, a) : frag(st + ih * HALF_BYTES, wr * 64 + ii * 32, a);
  };
  auto readY = [&](const char* st, auto J) {
    constexpr int j = decltype(J)::value;
#pragma unroll
    for (int a = 0; a < 4; ++a)
      yf[j][a] = (KO & 1) ? frag_plain(st + X_BYTES + (wq >> 1) * HALF_BYTES, (wq & 1) * 64 + j * 32, a) : frag(st + X_BYTES + (wq >> 1) * HALF_BYTES, (wq & 1) * 64 + j * 32, a);
  };
  auto colsum_step = [&](auto IH) {                   // k-step a == wave column: VALU sums of fragments the MFMAs consume anyway
    constexpr int ih = decltype(IH)::value;
    if (do_colsum) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (wq == a) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const u32x4 w = __builtin_bit_cast(u32x4, xf[ii][a]);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s0 += __builtin_bit_cast(float, w[e] << 16);
              s1 += __builtin_bit_cast(float, w[e] & 0xffff0000u);
            }
            cs[2 * ih + ii] += s0 + s1;
          }
        }
    }
  };
  auto mma = [&](auto IH, auto J) {
    constexpr int ih = decltype(IH)::value, j = decltype(J)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) acc[2 * ih + ii][j] = mfma32(xf[ii][a], yf[j][a], acc[2 * ih + ii][j]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto lgkm0 = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // prologue: step 0 complete, three half-tiles of step 1 in flight
  issueX(0, I0{}); issueY(0, I0{}); issueY(0, I1{}); issueX(0, I1{});
  if (nsteps > 1) {
    issueX(1, I0{}); issueY(1, I0{}); issueY(1, I1{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  bar();
  if (wr == 1) bar();                                 // the lower wave-row runs one barrier behind
  for (int t = 0; t < nsteps; ++t) {
    const char* st = smem + (t & 1) * STAGE;
    const bool more1 = t + 1 < nsteps, more2 = t + 2 < nsteps;
    // P1
    readX(st, I0{});
    __builtin_amdgcn_sched_barrier(0);
    readY(st, I0{});
    if (more1) issueX(t + 1, I1{});
    if (KO) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); // the 16 X reads (issued first) are retired: Xl may be re-staged in P2
    bar();
    lgkm0();
    mma(I0{}, I0{});
    colsum_step(I0{});
    bar();
    // P2
    readY(st, I1{});
    if (more2) issueX(t + 2, I0{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // all Y reads retired: Yl / Yr may be re-staged in P3 / P4
    bar();
    mma(I0{}, I1{});
    bar();
    // P3
    readX(st, I1{});
    if (more2) issueY(t + 2, I0{});
    bar();
    lgkm0();
    mma(I1{}, I1{});
    colsum_step(I1{});
    bar();
    // P4
    if (more2) {
      issueY(t + 2, I1{});
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // everything but the three half-tiles of step t+2: step t+1 has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    mma(I1{}, I0{});
    bar();
  }
  if (wr == 0) bar();                                 // balance the barrier count

  // ---- partial tile -> workspace slab, fragment-major: ((((wave*4 + i)*2 + j)*4 + q)*64 + lane)*4 + e  <->  acc[i][j][4q + e]
  float* slab = p.ws + ((long long)split * ntile + tile) * SLAB;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4*)(slab + ((((wave * 4 + i) * 2 + j) * 4 + q) * 64 + lane) * 4) =
            f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
  if (do_colsum) {                                   // workgroup-uniform
    __syncthreads();                                 // K-loop buffers are free
    float* red = (float*)smem;                       // [4 wave columns][256 nx]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float tot = cs[i] + __shfl_xor(cs[i], 32);
      if (h == 0) red[wq * BNX + (i >> 1) * 128 + wr * 64 + (i & 1) * 32 + l31] = tot;
    }
    __syncthreads();
    if (tid < BNX) p.ws_col[((long long)split * (ntile / p.tiles_y) + tx) * BNX + tid] = red[tid] + red[BNX + tid] + red[2 * BNX + tid] + red[3 * BNX + tid];
  }
}


#ifdef MART_EXPERIMENTS
// ============================================================================================================================
// Round 5 EXPERIMENT (harness builds only, -DMART_EXPERIMENTS + MART_TN_LOOP=4): the same product on v_mfma_f32_16x16x32 with the 4-phase loop of
// gemm_nt.hip (PIPE 4).  Measured (tools/tn_harness, profiles/r05_tn_harness_4phase16.txt): 0.482 / 0.471 / 0.358 / 0.130 ms on the four vision
// weight-gradient shapes against 0.468 / 0.466 / 0.351 / 0.132 for the 8-phase loop on 32x32x16 -- 1-3 % SLOWER where gemm_nt gained 3-6 %: with
// two transposed reads per fragment the 32-read segment S1 is longer than the partner's 32-MFMA cluster.  Not shipped.
//   * half-tiles as above ([64 m][256 B]: Xl Xr Yl Yr), 16-byte-chunk swizzle chunk ^ ((m & 3) << 2) ^ (((m >> 3) & 1) << 1): the two 4 x 16
//     blocks one 32-lane group of a transposed read covers now sit 8 contraction rows apart (lane group g = l >> 4 holds k = 8 g + 0..7 of a
//     32-deep k-step), and the extra key bit puts them in different 32-byte halves of a 64-byte bank group (conflict-free).
//   * per 64-row step and wave two segments of 32 MFMAs:  S1 = [reads X(ih0) + Y | bar | MFMAs | bar],  S2 = [reads X(ih1) | bar | MFMAs | bar];
//     LDS-DMA (SADDR form) from inside the MFMA clusters: M-S1(t): Xr(t+1);  M-S2(t): Xl, Yl, Yr of step t+2;  counted waits in the READ
//     segments (end of R-S1: vmcnt(6) -> Xr(t) landed; end of R-S2: vmcnt(2) -> Xl, Yl, Yr of t+1 landed): RAW / WAR argument as in gemm_nt.hip.
//   * accumulator quads: acc[i][j][4 q ..] = block (nx: ih = i >> 1, 16-column block xb = 2 (i & 1) + j; ny: 16-column block q); lane l holds
//     ny = l & 15 and nx = 4 (l >> 4) + 0..3 of it.  Slab layout unchanged (fragment-major); tn_reduce4_k decodes this mapping.
template <int PH>          // 4: two segments of 32 MFMAs per step (DMA from the clusters); 8: the shipped kernel's four quadrant phases of 16 MFMAs (DMA from the read segments)
__global__ __launch_bounds__(NT) void gemm_tn4_kernel(Args2 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int wr = wave >> 2, wq = wave & 3;
  const int ntile = p.ntile;
  const int lid = xcd_remap((int)(blockIdx.x + ntile * blockIdx.y), (int)(ntile * gridDim.y));
  const int tile = lid % ntile, split = lid / ntile;
  const int tx = tile / p.tiles_y, ty = tile % p.tiles_y;
  const int nx0 = tx * BNX, ny0 = ty * BNY;
  const int ms = split * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);
  const int nsteps = (me - ms) / BKM;

  // staging: half-tile = 1024 chunks of 16 B, two per thread: chunk c -> row c >> 4, physical chunk c & 15
  unsigned srcX[2][2], srcY[2][2];                   // [half][r]: BYTE offset of this thread's chunk at contraction row 0 of a step (< 64 rows * ld * 2)
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int c = r * NT + tid, row = c >> 4, pc = c & 15, lc = pc ^ ((row & 3) << 2) ^ (((row >> 3) & 1) << 1);
      srcX[hf][r] = ((unsigned)row * (unsigned)p.ldx + (unsigned)min(nx0 + hf * 128 + lc * 8, p.ldx - 8)) * 2u;
      srcY[hf][r] = ((unsigned)row * (unsigned)p.ldy + (unsigned)min(ny0 + hf * 128 + lc * 8, p.ldy - 8)) * 2u;
    }
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)LDS_PTR(smem) + wave * 1024;
  auto dma = [&](unsigned lds, unsigned voff, const void* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory");
  };
  auto issueX = [&](int t, auto HALF) {
    constexpr int hf = decltype(HALF)::value;
    const char* base = (const char*)(p.X + (long long)(ms + t * BKM) * p.ldx);
    const unsigned l = lds0 + (t & 1) * STAGE + hf * HALF_BYTES;
#pragma unroll
    for (int r = 0; r < 2; ++r) dma(l + r * NT * 16, srcX[hf][r], base);
  };
  auto issueY = [&](int t, auto HALF) {
    constexpr int hf = decltype(HALF)::value;
    const char* base = (const char*)(p.Y + (long long)(ms + t * BKM) * p.ldy);
    const unsigned l = lds0 + (t & 1) * STAGE + X_BYTES + hf * HALF_BYTES;
#pragma unroll
    for (int r = 0; r < 2; ++r) dma(l + r * NT * 16, srcY[hf][r], base);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }
  float cs[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  const bool do_colsum = p.colsum && ty == 0;

  // transposed fragment: 16 columns starting at col0 (inside the half-tile), contraction rows 32 kk + 8 g4 + {0..7}
  const int prow = l15 >> 2;
  const int fo = (prow * 256) + ((l15 & 3) & 1) * 8;          // row part + byte inside the 16-byte chunk
  auto frag = [&](const char* half, int col0, int kk) -> bf16x8 {
    const int r0 = kk * 32 + 8 * g4;                           // (+ prow, + 4): (row & 3) = prow, (row >> 3) & 1 = g4 & 1
    const int chunk = ((col0 >> 3) + ((l15 & 3) >> 1)) ^ (prow << 2) ^ ((g4 & 1) << 1);
    const char* a = half + r0 * 256 + fo + (chunk << 4);
    s16x4 lo = lds_tr_read(a);
    s16x4 hi = lds_tr_read(a + 4 * 256);
    return join_tr(lo, hi);
  };
  bf16x8 xf[4][2], yf[4][2];                         // [16-column block][k-step]
  auto readX = [&](const char* st, auto IH) {
    constexpr int ih = decltype(IH)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int xb = 0; xb < 4; ++xb) xf[xb][kk] = frag(st + ih * HALF_BYTES, wr * 64 + xb * 16, kk);
  };
  auto readXY = [&](const char* st) {                // S1: k-step 0 of Y and X first (the order the MFMAs consume them in)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int yb = 0; yb < 4; ++yb) yf[yb][kk] = frag(st + X_BYTES + (wq >> 1) * HALF_BYTES, (wq & 1) * 64 + yb * 16, kk);
#pragma unroll
      for (int xb = 0; xb < 4; ++xb) xf[xb][kk] = frag(st, wr * 64 + xb * 16, kk);
    }
  };
  // bias gradients: wave column wq sums k-step wq >> 1 of the blocks xb = (wq & 1), (wq & 1) + 2 -- fragments the MFMAs consume anyway.  One
  // literal-index arm per wave column (a loop over "if (wq == ...)" made the compiler index xf[][] through scratch, whose loads and stores
  // then drained the LDS-DMA queue every step: 3.4 x slower)
  auto colsum_arm = [&](auto IH, auto KK, auto PAR) {
    constexpr int ih = decltype(IH)::value, kk = decltype(KK)::value, par = decltype(PAR)::value;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x4 w = __builtin_bit_cast(u32x4, xf[par + 2 * u][kk]);
      asm volatile("" : "+v"(w));                      // (a register copy: keeps the fragment array out of memory)
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0 += __builtin_bit_cast(float, w[e] << 16);
        s1 += __builtin_bit_cast(float, w[e] & 0xffff0000u);
      }
      cs[ih][u] += s0 + s1;
    }
  };
  auto colsum_step = [&](auto IH) {
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    if (do_colsum) {
      if (wq == 0) colsum_arm(IH, J0{}, J0{});
      else if (wq == 1) colsum_arm(IH, J0{}, J1{});
      else if (wq == 2) colsum_arm(IH, J1{}, J0{});
      else colsum_arm(IH, J1{}, J1{});
    }
  };
  auto mma = [&](auto IH, auto&& d0, auto&& d1, auto&& d2) {
    constexpr int ih = decltype(IH)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int xb = 0; xb < 4; ++xb) {
        f32x16& C = acc[2 * ih + (xb >> 1)][xb & 1];
#pragma unroll
        for (int yb = 0; yb < 4; ++yb) {
          f32x4 c = {C[4 * yb], C[4 * yb + 1], C[4 * yb + 2], C[4 * yb + 3]};
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[xb][kk], yf[yb][kk], c, 0, 0, 0);
          C[4 * yb] = c[0]; C[4 * yb + 1] = c[1]; C[4 * yb + 2] = c[2]; C[4 * yb + 3] = c[3];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 0 && xb == 0) d0();
        if (kk == 0 && xb == 1) d1();
        if (kk == 0 && xb == 2) d2();
        __builtin_amdgcn_sched_barrier(0);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto lgkm0 = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto nop = [] {};
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // prologue: step 0 complete; Xl, Yl, Yr of step 1 in flight (Xr(1) comes from M-S1(0))
  issueX(0, I0{}); issueY(0, I0{}); issueY(0, I1{}); issueX(0, I1{});
  if (nsteps > 1) {
    issueX(1, I0{}); issueY(1, I0{}); issueY(1, I1{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  bar();
  if (wr == 1) bar();                                 // the lower wave-row runs one barrier behind
  auto readYj = [&](const char* st, auto J) {          // Y blocks 2 j, 2 j + 1 (the 32 columns of quadrant j)
    constexpr int j = decltype(J)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int yb = 2 * j; yb < 2 * j + 2; ++yb) yf[yb][kk] = frag(st + X_BYTES + (wq >> 1) * HALF_BYTES, (wq & 1) * 64 + yb * 16, kk);
  };
  auto mmaq = [&](auto IH, auto J) {                   // quadrant (ih, j): 4 x 2 blocks x 2 k-steps = 16 MFMAs
    constexpr int ih = decltype(IH)::value, j = decltype(J)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int xb = 0; xb < 4; ++xb) {
        f32x16& C = acc[2 * ih + (xb >> 1)][xb & 1];
#pragma unroll
        for (int yb = 2 * j; yb < 2 * j + 2; ++yb) {
          f32x4 c = {C[4 * yb], C[4 * yb + 1], C[4 * yb + 2], C[4 * yb + 3]};
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[xb][kk], yf[yb][kk], c, 0, 0, 0);
          C[4 * yb] = c[0]; C[4 * yb + 1] = c[1]; C[4 * yb + 2] = c[2]; C[4 * yb + 3] = c[3];
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };
  for (int t = 0; t < nsteps; ++t) {
    const char* st = smem + (t & 1) * STAGE;
    const bool more1 = t + 1 < nsteps, more2 = t + 2 < nsteps;
    if constexpr (PH == 8) {
      // the shipped kernel's schedule (gemm_tn8_kernel) on 16x16x32 fragments
      readX(st, I0{});
      __builtin_amdgcn_sched_barrier(0);
      readYj(st, I0{});
      if (more1) issueX(t + 1, I1{});
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the 16 X reads (issued first) are retired: Xl may be re-staged in P2
      bar();
      lgkm0();
      mmaq(I0{}, I0{});
      colsum_step(I0{});
      bar();
      readYj(st, I1{});
      if (more2) issueX(t + 2, I0{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bar();
      mmaq(I0{}, I1{});
      bar();
      readX(st, I1{});
      if (more2) issueY(t + 2, I0{});
      bar();
      lgkm0();
      mmaq(I1{}, I1{});
      colsum_step(I1{});
      bar();
      if (more2) {
        issueY(t + 2, I1{});
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      bar();
      mmaq(I1{}, I0{});
      bar();
      continue;
    }
    // R-S1
    readXY(st);
    if (more1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    lgkm0();
    mma(I0{}, [&] { if (more1) issueX(t + 1, I1{}); }, nop, nop);
    colsum_step(I0{});
    bar();
    // R-S2
    readX(st, I1{});
    if (more1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    lgkm0();
    mma(I1{}, [&] { if (more2) issueX(t + 2, I0{}); }, [&] { if (more2) issueY(t + 2, I0{}); }, [&] { if (more2) issueY(t + 2, I1{}); });
    colsum_step(I1{});
    bar();
  }
  if (wr == 0) bar();                                 // balance the barrier count

  // ---- partial tile -> workspace slab, fragment-major (same index as gemm_tn8_kernel; the quads mean different elements: see tn_reduce4_k)
  float* slab = p.ws + ((long long)split * ntile + tile) * SLAB;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4*)(slab + ((((wave * 4 + i) * 2 + j) * 4 + q) * 64 + lane) * 4) =
            f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
  if (do_colsum) {                                   // workgroup-uniform
    __syncthreads();                                 // K-loop buffers are free
    float* red = (float*)smem;                       // [2 k-steps][256 nx]: wave column wq holds k-step wq >> 1 of the blocks xb = (wq & 1) + 2 u
#pragma unroll
    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float tot = cs[ih][u];
        tot += __shfl_xor(tot, 16);
        tot += __shfl_xor(tot, 32);
        if (g4 == 0) red[(wq >> 1) * BNX + ih * 128 + wr * 64 + ((wq & 1) + 2 * u) * 16 + l15] = tot;
      }
    __syncthreads();
    if (tid < BNX) p.ws_col[((long long)split * (ntile / p.tiles_y) + tx) * BNX + tid] = red[tid] + red[BNX + tid];
  }
}

#endif  // MART_EXPERIMENTS

// out[nx][ny] += alpha * sum_{split} slab[split][tile](nx, ny), splits in order; colsum likewise.  One thread = one register quad
// of one fragment (4 consecutive nx at one ny); the 32 lanes of a half-wave cover 32 consecutive ny (128-byte segments).
template <bool M16>                                  // M16: the slabs were written by gemm_tn4_kernel (16x16x32 quads)
__global__ __launch_bounds__(256) void tn_reduce_k(Args2 p) {
  const int ntile = p.ntile, tiles_x = ntile / p.tiles_y;
  const int nblk_tiles = ntile * (SLAB / 4 / 256);
  if ((int)blockIdx.x >= nblk_tiles) {               // bias-gradient part: one thread per nx
    const int idx = ((int)blockIdx.x - nblk_tiles) * 256 + threadIdx.x;
    if (!p.colsum || idx >= tiles_x * BNX) return;
    const int tx = idx / BNX, nx = tx * BNX + (idx % BNX);
    if (nx >= p.NX) return;
    float tot = 0.f;
    for (int s = 0; s < p.splits; ++s) tot += p.ws_col[((long long)s * tiles_x + tx) * BNX + (idx % BNX)];
    float* dst = p.colsum + (p.colsum_by_row ? (long long)p.out_rows[nx] : (long long)nx);
    *dst += tot * p.alpha;
    return;
  }
  const int g = blockIdx.x * 256 + threadIdx.x;      // quad index over all tiles
  const int tile = g / (SLAB / 4), qi = g % (SLAB / 4);
  const int lane = qi & 63, q = (qi >> 6) & 3, j = (qi >> 8) & 1, i = (qi >> 9) & 3, wave = qi >> 11;
  const int h = lane >> 5, l31 = lane & 31, wr = wave >> 2, wq = wave & 3;
  f32x4 tot = {0.f, 0.f, 0.f, 0.f};
  const float* src = p.ws + (long long)tile * SLAB + (long long)qi * 4;
  for (int s = 0; s < p.splits; ++s) tot += *(const f32x4*)(src + (long long)s * ntile * SLAB);
  const int tx = tile / p.tiles_y, ty = tile % p.tiles_y;
  const int ny = M16 ? ty * BNY + wq * 64 + q * 16 + (lane & 15) : ty * BNY + wq * 64 + j * 32 + l31;
  const int nxb = M16 ? tx * BNX + (i >> 1) * 128 + wr * 64 + (2 * (i & 1) + j) * 16 + 4 * (lane >> 4)
                      : tx * BNX + (i >> 1) * 128 + wr * 64 + (i & 1) * 32 + 8 * q + 4 * h;
  if (ny >= p.NY) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int nx = nxb + e;
    if (nx < p.NX) {
      const long long orow = p.out_rows ? p.out_rows[nx] : nx;
      float* dst = p.out + orow * p.ldo + ny;
      *dst += tot[e] * p.alpha;
    }
  }
}

}  // namespace

extern "C" int mart_gemm_tn(const mart_gemm_tn_desc* d, void* stream) {
  MART_CHECK(d != nullptr, "gemm_tn: null descriptor");
  MART_CHECK(d->M > 0 && d->NX > 0 && d->NY > 0, "gemm_tn: M,NX,NY must be positive");
  MART_CHECK(d->ldx % 8 == 0 && d->ldy % 8 == 0, "gemm_tn: ldx/ldy must be multiples of 8");
  MART_CHECK(d->ldx >= ((d->NX + 7) / 8) * 8 && d->ldy >= ((d->NY + 7) / 8) * 8, "gemm_tn: operands must be padded to 8 columns");
  MART_CHECK(((uintptr_t)d->X & 15) == 0 && ((uintptr_t)d->Y & 15) == 0, "gemm_tn: X/Y must be 16-byte aligned");
  MART_CHECK(d->out != nullptr && d->ldo >= d->NY, "gemm_tn: bad out/ldo");
  MART_CHECK(!d->colsum_by_row || d->out_rows, "gemm_tn: colsum_by_row needs out_rows");
  static MartAttrOnce once;
  bool* attr_set = once.slot();
  constexpr int LDS = 2 * STAGE;
  if (!*attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      mart_set_error("gemm_tn: hipFuncSetAttribute failed");
      return -2;
    }
    *attr_set = true;
  }
  Args a;
  a.X = (const bf16*)d->X; a.Y = (const bf16*)d->Y; a.ldx = d->ldx; a.ldy = d->ldy;
  a.M = d->M; a.NX = d->NX; a.NY = d->NY; a.out = d->out; a.ldo = d->ldo; a.out_rows = d->out_rows;
  a.colsum = d->colsum; a.colsum_by_row = d->colsum_by_row;
  a.sX = d->stride_x; a.sY = d->stride_y; a.sO = d->stride_o;
  a.alpha = d->alpha;
  const int batch = d->batch > 0 ? d->batch : 1;
  const int tiles = ((d->NX + BNX - 1) / BNX) * ((d->NY + BNY - 1) / BNY);
  int splits = d->splits;
  const int max_splits = (d->M + 4 * BKM - 1) / (4 * BKM);        // keep >= 256 rows per split
  // one workgroup per CU (256 CUs): measured best on MI355X -- a partial second round of workgroups costs more than
  // the shorter contraction saves (profiles/: 27x9, 36x7, 9x28 splits win by 20-40 % over 3 rounds)
  if (splits <= 0) splits = 256 / (tiles * batch);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rps = (d->M + splits - 1) / splits;
  rps = ((rps + BKM - 1) / BKM) * BKM;
  splits = (d->M + rps - 1) / rps;
  a.splits = splits; a.rows_per_split = rps;
  if (d->workspace && batch == 1) {
    // deterministic path: 8-phase loop over the whole 64-row steps, partial tiles to the caller's workspace, ordered reduction
    // kernel; a contraction tail (M % 64 rows) is added by ONE workgroup per tile of the single-split kernel below (one
    // contribution per output element, stream-ordered behind the reduction: still run-to-run identical)
    const int Mmain = d->M - d->M % BKM;
    if (Mmain > 0) {
      const int tiles_x = (d->NX + BNX - 1) / BNX, tiles_y = (d->NY + BNY - 1) / BNY;
      int sp = d->splits;
      const int max_sp = (Mmain + 4 * BKM - 1) / (4 * BKM);
      if (sp <= 0) sp = 256 / tiles;
      if (sp > max_sp) sp = max_sp;
      if (sp < 1) sp = 1;
      int r = (Mmain + sp - 1) / sp;
      r = ((r + BKM - 1) / BKM) * BKM;
      sp = (Mmain + r - 1) / r;
      const size_t need = ((size_t)sp * tiles * SLAB + (size_t)sp * tiles_x * BNX) * sizeof(float);
      MART_CHECK(d->workspace_bytes >= (long long)need, "gemm_tn: workspace too small (mart_gemm_tn_workspace_bytes)");
      MART_CHECK(((uintptr_t)d->workspace & 15) == 0, "gemm_tn: workspace must be 16-byte aligned");
      static MartAttrOnce once8;
      bool* set8 = once8.slot();
      if (!*set8) {
        if (hipFuncSetAttribute((const void*)gemm_tn8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess
#ifdef MART_EXPERIMENTS
            || hipFuncSetAttribute((const void*)gemm_tn4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess
            || hipFuncSetAttribute((const void*)gemm_tn4_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess
#endif
        ) {
          mart_set_error("gemm_tn: hipFuncSetAttribute failed");
          return -2;
        }
        *set8 = true;
      }
#ifdef MART_EXPERIMENTS
      static const int loop_env = getenv("MART_TN_LOOP") ? atoi(getenv("MART_TN_LOOP")) : 8;
      const bool loop4 = (loop_env == 4 || loop_env == 816) && (long long)64 * a.ldx * 2 < (1LL << 31) && (long long)64 * a.ldy * 2 < (1LL << 31);   // 816: 8 phases on 16x16x32
#else
      constexpr bool loop4 = false;
#endif
      Args2 b;
      b.X = a.X; b.Y = a.Y; b.ldx = a.ldx; b.ldy = a.ldy; b.M = Mmain; b.NX = a.NX; b.NY = a.NY; b.out = a.out; b.ldo = a.ldo;
      b.out_rows = a.out_rows; b.colsum = a.colsum; b.colsum_by_row = a.colsum_by_row; b.splits = sp; b.rows_per_split = r;
      b.tiles_y = tiles_y; b.ntile = tiles; b.alpha = a.alpha;
      b.ws = (float*)d->workspace; b.ws_col = b.ws + (size_t)sp * tiles * SLAB;
#ifdef MART_EXPERIMENTS
      if (loop4 && loop_env == 816) hipLaunchKernelGGL(gemm_tn4_kernel<8>, dim3(tiles, sp, 1), dim3(NT), LDS, (hipStream_t)stream, b);
      else if (loop4) hipLaunchKernelGGL(gemm_tn4_kernel<4>, dim3(tiles, sp, 1), dim3(NT), LDS, (hipStream_t)stream, b);
      else
#endif
      hipLaunchKernelGGL(gemm_tn8_kernel, dim3(tiles, sp, 1), dim3(NT), LDS, (hipStream_t)stream, b);
      MART_LAUNCH_CHECK();
      const int nblk = tiles * (SLAB / 4 / 256) + (d->colsum ? (tiles_x * BNX + 255) / 256 : 0);
      if (loop4) hipLaunchKernelGGL(tn_reduce_k<true>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, b);
      else hipLaunchKernelGGL(tn_reduce_k<false>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, b);
      MART_LAUNCH_CHECK();
    }
    if (d->M > Mmain) {
      Args t = a;
      t.X = a.X + (long long)Mmain * a.ldx; t.Y = a.Y + (long long)Mmain * a.ldy; t.M = d->M - Mmain;
      t.splits = 1; t.rows_per_split = BKM;
      hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, 1, 1), dim3(NT), LDS, (hipStream_t)stream, t);
      MART_LAUNCH_CHECK();
    }
    return 0;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits, batch), dim3(NT), LDS, (hipStream_t)stream, a);
  MART_LAUNCH_CHECK();
  return 0;
}

/* bytes of workspace the deterministic path of mart_gemm_tn needs for this problem (same split rule as the launcher) */
extern "C" long long mart_gemm_tn_workspace_bytes(int M, int NX, int NY, int splits_req) {
  M -= M % BKM;                                        // the tail rows go through the single-split kernel, no workspace
  if (M <= 0) return 16;
  const int tiles_x = (NX + BNX - 1) / BNX, tiles = tiles_x * ((NY + BNY - 1) / BNY);
  int splits = splits_req;
  const int max_splits = (M + 4 * BKM - 1) / (4 * BKM);
  if (splits <= 0) splits = 256 / tiles;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rps = (M + splits - 1) / splits;
  rps = ((rps + BKM - 1) / BKM) * BKM;
  splits = (M + rps - 1) / rps;
  return (long long)(((size_t)splits * tiles * SLAB + (size_t)splits * tiles_x * BNX) * sizeof(float));
}

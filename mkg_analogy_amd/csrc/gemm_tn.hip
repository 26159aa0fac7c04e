// TN GEMM on gfx950 MFMA:  out[NX,NY] (f32, atomic +=) = alpha * X[M,NX]^T Y[M,NY]   (contraction over rows)
//
// Weight gradients dW = dC^T A of every linear layer, the tied-embedding gradient rows of the scoring head
// (out_rows scatter) and the fusion op's d(vis).  Both operands are contraction-major in memory, so the MFMA
// fragments are fetched with the gfx950 LDS transpose read (ds_read_b64_tr_b16): tiles are staged row-major
// [64 m][256 n] by LDS-DMA and each 16-lane group pulls a 4(m) x 16(n) block already transposed.
// The contraction (M) is split across workgroups; partial tiles are combined with f32 atomics.
// Optional fused column sums of X (bias gradients): VALU sums of the fragments the MFMAs consume anyway.
#include "common.h"
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
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits, batch), dim3(NT), LDS, (hipStream_t)stream, a);
  MART_LAUNCH_CHECK();
  return 0;
}

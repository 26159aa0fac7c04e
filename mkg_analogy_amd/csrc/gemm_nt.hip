// NT GEMM on gfx950 MFMA:  C[M,N] = epilogue( A[M,K] * B[N,K]^T (+ A2[M,K2] * B2[N,K2]^T) )
//
// Every dense contraction of the MKGformer step is an instance (SURVEY 2.2 K1,K3,K5-K7,K10-K12 and
// their data-gradients, which use the transposed bf16 weight shadows so they are NT as well).
//   * bf16 operands, fp32 accumulate, v_mfma_f32_32x32x16_bf16
//   * tiles BMxBNx64, LDS double buffer filled by LDS-DMA (global_load_lds, 16 B/lane); the LDS image is
//     lane-linear, the 16-byte-chunk XOR swizzle is applied on the per-lane SOURCE address and on the
//     ds_read_b128 address (conflict-free for the 4x16-lane b128 service groups)
//   * operands swapped into the MFMA so each lane owns one output row and 4 consecutive columns per
//     register quad: bias / activation / residual / dual-dtype stores are 8-16 B vector accesses
//   * optional row gathers on A and B (scoring head: mask rows x word_emb[entity ids]) folded into the
//     per-lane source address -- the gather costs nothing extra
//   * workgroup ids remapped so that neighbouring tiles (same A panel) share an XCD's L2
#include "common.h"
#include <type_traits>
#include "mart_hip.h"

namespace {

// Outputs are written once and not re-read by this kernel: nontemporal stores keep them from displacing the A/B panels
// in L2 (streamed operands of the epilogue -- residual, z -- are read nontemporally for the same reason).
#ifndef MART_NT_EPILOGUE
#define MART_NT_EPILOGUE 1
#endif
// The pointers are cast to the global address space explicitly: where address-space inference fails (pointers carried
// around the persistent tile loop) the compiler emits FLAT accesses, which have no scalar-base addressing form and count
// on lgkmcnt as well -- every LDS wait of the epilogue then also waited for the stores before it.
// Harness-only (tools/nt_store_policy.sh, -DMART_EXPERIMENTS -DMART_ST_POLICY=k): the cache policy of the epilogue's 16-byte stores -- 1 sc1 (write-through,
// the line is dropped from the XCD's L2), 2 sc0 sc1, 3 sc1 nt -- against the shipped nt stores (which keep the line in L2 until evicted).
#if defined(MART_EXPERIMENTS) && defined(MART_ST_POLICY)
template <typename T> __device__ __forceinline__ void st_policy16(T* p, T v) {
  static_assert(sizeof(T) == 16, "16-byte stores only");
  const f32x4 d = __builtin_bit_cast(f32x4, v);
#if MART_ST_POLICY == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
#elif MART_ST_POLICY == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(d) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(d) : "memory");
#endif
}
#endif
template <typename T> __device__ __forceinline__ void st_stream(T* p, T v) {
#if defined(MART_EXPERIMENTS) && defined(MART_ST_POLICY)
  if constexpr (sizeof(T) == 16) { st_policy16(p, v); return; }
#endif
  auto* g = (__attribute__((address_space(1))) T*)p;
#if MART_NT_EPILOGUE
  __builtin_nontemporal_store(v, g);
#else
  *g = v;
#endif
}
template <typename T> __device__ __forceinline__ T ld_stream(const T* p) {
  auto* g = (const __attribute__((address_space(1))) T*)p;
#if MART_NT_EPILOGUE
  return __builtin_nontemporal_load(g);
#else
  return *g;
#endif
}

// Timing / A-B experiments (DMA knock-outs, the round-1 K loops, start-stagger sweeps) exist only in harness builds
// (-DMART_EXPERIMENTS: tools/nt_harness, tools/build_variant.sh); the product library carries none of them.
#ifdef MART_EXPERIMENTS
#define NT_DBG(p) ((p).dbg)
#else
#define NT_DBG(p) 0
#endif

#ifdef MART_EXPERIMENTS
// cycle stamps of ONE workgroup (tile_cfg 2567; tools/nt_harness stamp): [wave][0 loop start, 1 loop end, 2 epilogue start, 3..6 after block 0..3, 7 end]
__device__ unsigned long long g_nt_stamps[3 * 8 * 9];        // [tile iteration of the workgroup (persistent loop), < 3][wave][stamp]; stamp 8 = kernel entry / tile start
__device__ __forceinline__ unsigned long long nt_memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define NT_STAMP(k) do { if ((NT_DBG(p) & 8) && (int)blockIdx.x == 200 && lane == 0 && nt_tile_it < 3) g_nt_stamps[(nt_tile_it * 8 + wave) * 9 + (k)] = nt_memtime(); } while (0)
#else
#define NT_STAMP(k) do { } while (0)
#endif

struct Args {
  const bf16* A; const bf16* B; const bf16* A2; const bf16* B2;
  int lda, ldb;
  int M, N, K, K2;
  const int* a_rows; const int* b_rows;
  long long sA, sB, sC, sAux;       // batch strides in elements (A/A2, B/B2, C/C2/preact, residual/mulz)
  const float* bias; const float* bias2; int bias_by_brow;
  int act;
  bf16* preact; int preact_grad;
  const bf16* mulz; int mul_act;
  const float* res_f32; const bf16* res_bf16; int ldres;
  float alpha;
  void* C; int ldc; int c_f32;
  bf16* C2; int ldc2;
  int b_blocked;
  float* row_stats;                  // F_STATS: [M][N / 64][2] partial (sum, sum of squares) of the f32 rows this epilogue writes
  const float* ln_mean; const float* ln_rstd; const float* ln_colsum;   // F_LNFOLD: row statistics [M] and s[n] = sum_k B'[n][k] of the folded weight
  int stagger_ticks;                 // >0: first-wave workgroups start g*ticks (100 MHz) late, g = 0..7
  int dbg;                           // 1: skip the LDS-DMA after the first tile (timing experiment only)
};

// Epilogue feature mask of the specialised ("fast") instantiations.  EPI < 0 = the general epilogue (tails, gathers,
// bf16 residual, unaligned rows).  Every large product of the step maps to one of the fast masks; their epilogues are a few
// hundred bytes of straight-line vector code (the general one made the kernel 200 KB and instruction-fetch bound).
enum { F_RES = 1, F_MULZ = 2, F_PREACT = 4, F_ACT = 8, F_CF32 = 16, F_C2 = 32, F_PGRAD = 64, F_SPLIT3 = 128, F_STATS = 256, F_LNFOLD = 512 };   // F_PGRAD: preact holds act'(z)
// LayerNorm folded into the product that consumes it (modeling_unimo.py:509 -> 223-225, :518 -> 284-286; forward passes that keep nothing for a
// backward pass):  LN(x) W^T + b = rstd (x (gamma o W)^T - mean s) + b',  s[n] = sum_k (gamma o W)[n][k],  b' = b + W beta.
// F_STATS (f32 residual epilogue: the PRODUCER of x): per row and 64-column wave slice, (sum, sum of squares) of the f32 values written -> row_stats;
// mart_ln_stats_finalize turns the N / 64 partials of a row into mean / rstd.  F_LNFOLD (packed 16-bit epilogue: the CONSUMER): A is the bf16 copy of x
// (the producer's C2), B the folded weight gamma o W, bias b'; the epilogue applies (acc - mean[m] s[n]) rstd[m] + b'[n] in MFMA layout.
// F_SPLIT3 (with F_PREACT | F_C2, bf16): the three 16-bit outputs are the two-term split [hi | lo | hi] of the f32 result (mart_gemm_nt_desc.c_split3)

// PERSIST: one workgroup per CU slot walks over its tiles; the first K-tile of the NEXT tile is put in flight before the
// epilogue of the current one, so the ~2-3 us of launch + first-DMA latency per tile hide behind the epilogue.
// DT: operand / output element type.  0 = bf16 operands, C bf16 | f32 (every product of the vision stream and of the backward pass);
// 1 = fp16 operands (v_mfma_f32_32x32x16_f16: same rate, 11-bit significands), C bf16 | f32; 2 = fp16 operands AND C in fp16 (the GELU output that
// is the next product's A operand), with C2 as its bf16 copy for the backward pass.  The 16-bit lanes move through LDS untyped.
// cache-policy bits of the 8-phase loop's LDS-DMA (gfx940+: 1 = sc0, 2 = nt, 16 = sc1); experiments: -DGLDS_AUX_A=.. -DGLDS_AUX_B=..
#ifndef GLDS_AUX_A
#define GLDS_AUX_A 0
#endif
#ifndef GLDS_AUX_B
#define GLDS_AUX_B 0
#endif
// sum over the 16 lanes of a DPP row (quad swaps, then the two mirrors), result in every lane
__device__ __forceinline__ float row16_sum(float x) {
  auto sh = [](float v, auto CTRL) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(CTRL)::value, 0xF, 0xF, true)); };
  x += sh(x, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
  x += sh(x, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
  x += sh(x, std::integral_constant<int, 0x141>{});     // row_half_mirror
  x += sh(x, std::integral_constant<int, 0x140>{});     // row_mirror
  return x;
}
template <int DT> __device__ __forceinline__ f32x16 mm(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (DT == 0) return mfma32(a, b, c); else return mfma32h(a, b, c);
}
// v_mfma_f32_16x16x32: D[i][j] += sum_k A[i][k] B[k][j]; lane l holds A[i = l & 15][k = 8 (l >> 4) + 0..7], B[k = 8 (l >> 4) + 0..7][j = l & 15];
// result: lane l, reg r holds D[row = 4 (l >> 4) + r][col = l & 15]
template <int DT> __device__ __forceinline__ f32x4 mm16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (DT == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
// LDS-DMA, 16 B per lane, SADDR form: source = 64-bit scalar base + zero-extended 32-bit lane byte offset -- no per-issue vector address
// arithmetic, and opaque to the compiler's waitcnt insertion (the loop that uses it counts vmcnt itself).  The block overwrites M0 without telling
// the compiler (hipcc rejects M0 in a clobber list as reserved): nothing else in a kernel that calls this may depend on M0 -- the PIPE 4 instantiations
// issue no builtin LDS-DMA, no s_movrel, no GWS / interpolation instruction; LDS instructions do not read M0 on gfx9+
__device__ __forceinline__ void dma16(unsigned lds_wave_base, unsigned voff, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
template <int DT> __device__ __forceinline__ bf16x4 cvt_c(f32x4 v) {            // primary 16-bit output
  if constexpr (DT == 2) return f4_to_h4raw(v); else return f4_to_bf4(v);
}
template <int DT> __device__ __forceinline__ bf16x4 cvt_c2(f32x2 lo, f32x2 hi) {
  if constexpr (DT == 2) return f2x2_to_h4raw(lo, hi); else return f2x2_to_bf4(lo, hi);
}
constexpr bool epi_packed(int EPI, int DT) {     // 16-bit outputs only, no streamed operands: staged through LDS as packed 16-bit rows
  return EPI >= 0 && (EPI & (1 | 2 | 16)) == 0 && (DT == 2 || (EPI & 32) == 0 || (EPI & 128) != 0);
}
constexpr int epi_outputs(int EPI) { return 1 + ((EPI & 4) != 0 ? 1 : 0) + ((EPI & 32) != 0 ? 1 : 0); }

template <int BM, int BN, int WAVES_M, int WAVES_N, int PIPE = 0, int EPI = -1, int ACTK = 0, bool PERSIST = false, int DT = 0>   // ACTK: activation kind of the fast masks (literal: no erf code in the quick-GELU kernels)
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, PIPE == 3 ? 2 : 1) void gemm_nt_kernel(Args p) {   // PIPE 3: two 4-wave workgroups per CU (<= 256 VGPRs)
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int BK = 64;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int RA = (BM * 8) / NT, RB = (BN * 8) / NT;   // 16-byte chunks per thread per stage
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile/threads mismatch");
  static_assert(PIPE != 4 || (BN / WAVES_N == 64), "the 16x16x32 quad mapping of the epilogues is written for 64-column wave tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;

  if (p.stagger_ticks > 0 && blockIdx.x < 256 && blockIdx.y == 0) {
    // De-phase the CUs: all first-wave workgroups start together, run the same K loop and would all reach their
    // epilogue at the same instant -- a 33 MB write burst that the memory side absorbs at ~4.4 TB/s while every
    // matrix core idles.  Starting eighths of the CUs an eighth of a tile period apart spreads the bursts.
    const long long t0 = wall_clock64();
    // phase group: tiles that share an A panel (tiles_n consecutive logical ids of one XCD) start together, so that the second and
    // third reader of the panel still find it in L2
    const int tn_ = (p.N + BN - 1) / BN;
    const long long d = (long long)((((int)blockIdx.x >> 3) / tn_) & 7) * p.stagger_ticks;
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(32);
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int total_tiles = tiles_m * tiles_n;
  const long long bz = blockIdx.y;

  const bf16* A = p.A + bz * p.sA;
  const bf16* B = p.B + bz * p.sB;
  const bf16* A2 = p.A2 ? p.A2 + bz * p.sA : nullptr;
  const bf16* B2 = p.B2 ? p.B2 + bz * p.sB : nullptr;

  // per-thread source offsets (elements) of the chunks this thread stages; constant over the K loop of one tile
  int vb = blockIdx.x, m0 = 0, n0 = 0;
  unsigned offA[RA], offB[RB];
  long long blkB1 = 0, blkB2 = 0;
  auto set_tile = [&](int vbid) {
    const int lid = xcd_remap(vbid, total_tiles);
    m0 = (lid / tiles_n) * BM; n0 = (lid % tiles_n) * BN;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      int c = r * NT + tid, row = c >> 3, pc = c & 7, lc = pc ^ ((row >> 1) & 7);
      int gm = min(m0 + row, p.M - 1);
      if (p.a_rows) gm = p.a_rows[gm];
      offA[r] = ((unsigned)gm * (unsigned)p.lda + lc * 8) * (PIPE == 4 ? 2u : 1u);      // PIPE 4: BYTE offsets (SADDR-form DMA; < 2^32 by dispatch)
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      int c = r * NT + tid, row = c >> 3, pc = c & 7, lc = pc ^ ((row >> 1) & 7);
      int gn = min(n0 + row, p.N - 1);
      if (p.b_rows) gn = p.b_rows[gn];
      offB[r] = ((unsigned)gn * (unsigned)p.ldb + lc * 8) * (PIPE == 4 ? 2u : 1u);
      if (p.b_blocked) offB[r] = (unsigned)(row & 255) * 64 + lc * 8;       // inside the 256x64 block
    }
    // tile-blocked weights: block (n0/256, kt) of operand with K' columns starts at ((n0/256)*(K'/64) + kt) * 16384 elements
    blkB1 = p.b_blocked ? (long long)(n0 >> 8) * (p.K >> 6) * 16384 : 0;
    blkB2 = p.b_blocked ? (long long)(n0 >> 8) * (p.K2 >> 6) * 16384 : 0;
  };
  set_tile(vb);

  const int nk1 = p.K / BK, nk = nk1 + p.K2 / BK;

  auto stage = [&](int t, int buf) {
    if constexpr (PIPE == 4) {                               // (the persistent loop's prefetch of the next tile's K-tile 0)
      const char* Ab = (const char*)(t >= p.K / BK ? A2 + (t - p.K / BK) * BK : A + t * BK);
      const char* Bb = (const char*)(t >= p.K / BK ? B2 + (t - p.K / BK) * BK : B + t * BK);
      const unsigned l = (unsigned)(__UINTPTR_TYPE__)LDS_PTR(smem) + buf * STAGE + wave * 1024;
#pragma unroll
      for (int r = 0; r < RA; ++r) dma16(l + r * NT * 16, offA[r], Ab);
#pragma unroll
      for (int r = 0; r < RB; ++r) dma16(l + A_BYTES + r * NT * 16, offB[r], Bb);
      return;
    }
    const bf16* Ap = A; const bf16* Bp = B + blkB1; int k0 = t * BK;
    if (t >= nk1) { Ap = A2; Bp = B2 + blkB2; k0 = (t - nk1) * BK; }
    const long long kb = p.b_blocked ? (long long)(k0 >> 6) * 16384 : k0;      // blocked: whole 256x64 blocks per K-tile
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_BYTES;
#pragma unroll
    for (int r = 0; r < RA; ++r) glds16(Ap + offA[r] + k0, sA + (r * NT + wave * 64) * 16);
#pragma unroll
    for (int r = 0; r < RB; ++r) glds16(Bp + offB[r] + kb, sB + (r * NT + wave * 64) * 16);
  };

  // PIPE == 2 (8-phase loop): the two wave-rows interleave in 64-row slabs -- wave-row wr owns rows {ih*128 + wr*64 + [0,64)},
  // ih = 0,1 -- so that the first 32-row block pair of EVERY wave lies in the upper half-tile of A (rows 0..127) and the
  // second pair in the lower one: a half-tile is dead (and can be re-staged) as soon as its phase has been read.
  constexpr bool ILV = (PIPE == 2 || PIPE == 4);
  // PIPE 4 computes on v_mfma_f32_16x16x32: a 32-row block of the wave tile is 2 x 4 blocks of 16 x 16, and register quad q of acc[i][j] is the
  // block (rows j * 16 .. + 15, columns q * 16 .. + 15) of 32-row block i: lane l holds row (l & 15) and columns 4 (l >> 4) + 0..3 of it.
  // (32x32x16, every other PIPE: quad q of acc[i][j] = row (l & 31), columns j * 32 + 8 q + 4 (l >> 5) + 0..3.)  The epilogues address quads
  // through erow / ecol and are otherwise the same code.
  constexpr bool M16 = (PIPE == 4);
  const int l15 = lane & 15, g4 = lane >> 4;
  auto erow = [&](int j) { return M16 ? j * 16 + l15 : l31; };
  auto ecol = [&](int j, int q) { return M16 ? q * 16 + 4 * g4 : j * 32 + 8 * q + 4 * h; };
  auto ro = [](int i) constexpr { return ILV ? (i >> 1) * 128 + (i & 1) * 32 : i * 32; };   // row offset of 32-row block i inside the wave's rows
  const int wm0 = (wave / WAVES_N) * (ILV ? 64 : WM), wn0 = (wave % WAVES_N) * WN;
  // LDS read addressing: row -> byte base and swizzle key
  int rowA[TM], rowB[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) rowA[i] = wm0 + ro(i) + l31;
#pragma unroll
  for (int j = 0; j < TN; ++j) rowB[j] = wn0 + j * 32 + l31;

  bool have0 = false;                  // K-tile 0 of the current tile is already in flight (persistent variant)
#ifdef MART_EXPERIMENTS
  int nt_tile_it = 0;
#endif
  for (;;) {
  NT_STAMP(8);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (PIPE == 1 || PIPE == 3) {
    // ---- 4-slot ring of 32-deep K-steps (32 KB per slot), LDS-DMA three steps ahead behind COUNTED vmcnt waits: the
    // memory pipe always has 2-3 steps (64-96 KB per CU) in flight instead of one 64 KB burst per barrier.
    //   slot image: 64-byte rows, 16-byte chunk c' = c ^ ((row>>2)&3)  (conflict-free for the b128 service groups)
    constexpr int SA = BM * 64, SB = BN * 64, SLOT = SA + SB;          // bytes
    constexpr int QA = (BM * 4) / NT, QB = (BN * 4) / NT;              // chunks per thread per step
    static_assert(QA >= 1 && QB >= 1, "tile too small for the ring");
    unsigned oA[QA], oB[QB];
#pragma unroll
    for (int r = 0; r < QA; ++r) {
      int c = r * NT + tid, row = c >> 2, pc = c & 3, lc = pc ^ ((row >> 2) & 3);
      int gm = min(m0 + row, p.M - 1);
      if (p.a_rows) gm = p.a_rows[gm];
      oA[r] = (unsigned)gm * (unsigned)p.lda + lc * 8;
    }
#pragma unroll
    for (int r = 0; r < QB; ++r) {
      int c = r * NT + tid, row = c >> 2, pc = c & 3, lc = pc ^ ((row >> 2) & 3);
      int gn = min(n0 + row, p.N - 1);
      if (p.b_rows) gn = p.b_rows[gn];
      oB[r] = (unsigned)gn * (unsigned)p.ldb + lc * 8;
    }
    const int ns1 = p.K / 32, ns = ns1 + p.K2 / 32;
    // PIPE 1: 4 slots, three steps ahead.  PIPE 3 (experiment: 128x256 tile, 4 waves, TWO workgroups per CU so that one workgroup's epilogue runs under
    // the other's K loop): 3 slots of (BM + BN) x 64 B = 24 KB -> 72 KB per workgroup, two steps ahead.
    constexpr int NSLOT = PIPE == 3 ? 3 : 4, AHEAD = NSLOT - 1;
    auto issue = [&](int t) {
      const bf16* Ap = A; const bf16* Bp = B; int k0 = t * 32;
      if (t >= ns1) { Ap = A2; Bp = B2; k0 = (t - ns1) * 32; }
      char* sA = smem + (t % NSLOT) * SLOT;
      char* sB = sA + SA;
#pragma unroll
      for (int r = 0; r < QA; ++r) glds16(Ap + oA[r] + k0, sA + (r * NT + wave * 64) * 16);
#pragma unroll
      for (int r = 0; r < QB; ++r) glds16(Bp + oB[r] + k0, sB + (r * NT + wave * 64) * 16);
    };
    int keyA[TM], keyB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) keyA[i] = (rowA[i] >> 2) & 3;
#pragma unroll
    for (int j = 0; j < TN; ++j) keyB[j] = (rowB[j] >> 2) & 3;
    issue(0);
    if (ns > 1) issue(1);
    if (AHEAD > 2 && ns > 2) issue(2);
    for (int t = 0; t < ns; ++t) {
      const int ahead = min(ns - 1 - t, AHEAD - 1);         // newer steps that may stay in flight
      if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (QA + QB)) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA + QB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                          // step t visible to all; slot of step t-1 free
      __builtin_amdgcn_sched_barrier(0);
      if (t + AHEAD < ns) issue(t + AHEAD);
      const char* sA = smem + (t % NSLOT) * SLOT;
      const char* sB = sA + SA;
      bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int lc = u * 2 + h;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[u][i] = *(const bf16x8*)(sA + rowA[i] * 64 + ((lc ^ keyA[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[u][j] = *(const bf16x8*)(sB + rowB[j] * 64 + ((lc ^ keyB[j]) << 4));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mm<DT>(bfr[u][j], af[u][i], acc[i][j]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // all fragment reads of this slot retired before the next barrier
    }
  } else if constexpr (PIPE == 2) {
    // ---- 8-phase main loop (four phases per 64-deep K-tile, two K-tiles per LDS ring turn).
    //   LDS ring: 2 K-tile buffers x 4 half-tiles of 16 KB -- A rows 0..127 (A0), A rows 128..255 (A1), B rows 0..127 (B0),
    //   B rows 128..255 (B1).  One half-tile = 2 LDS-DMA instructions per thread: the unit of issue and of the vmcnt count.
    //   A phase = [ fragment reads (+ ONE half-tile of LDS-DMA) | s_barrier | 8 MFMAs = one 64x32 quadrant x K=64 | s_barrier ].
    //   The two wave-rows run ONE barrier apart (the lower one executes an extra s_barrier up front): while one wave of a
    //   SIMD issues MFMAs its partner reads fragments / issues DMA, so the matrix pipe never waits for LDS.
    //   Quadrant order (ih0,j0) (ih0,j1) (ih1,j1) (ih1,j0): P1 reads A(ih0) + B(j0), P2 reads B(j1), P3 reads A(ih1), P4 reads
    //   nothing; both B fragments stay in registers.  With the interleaved row ownership (ILV) every wave's ih0 rows lie in
    //   A0 and its ih1 rows in A1, so A0 is dead after P1, B0/B1 after P2, A1 after P3, and the half-tiles of K-tile t+2 are
    //   issued into the buffer of K-tile t while t is still being computed:
    //        P1(t): A1(t+1)      P2(t): A0(t+2)      P3(t): B0(t+2)      P4(t): B1(t+2), then s_waitcnt vmcnt(6)
    //   (each issued from inside the phase's MFMA cluster, see mma())
    //   i.e. ONE counted wait per K-tile that leaves the three newest half-tiles (48 KB per CU) in flight across the
    //   barriers and never drains the queue; every half-tile has at least three phases to land.
    //   RAW: the wait sits in front of P4's FIRST barrier (vmcnt(4): B0 / A0 of K-tile t+2 stay in flight, B1(t+2) is issued behind it); K-tile
    //        t+1 is first read in P1(t+1), two barriers later for the leading wave-row and one for the lagging one: every wave has waited
    //        before a barrier that precedes the first read by any wave.
    //   WAR: a half-tile is re-staged from the MFMA cluster of the phase AFTER its last read: every reader has retired those reads
    //        (lgkmcnt(0) behind the reading phase's first barrier) at least one barrier before any wave reaches that cluster.
    static_assert(BM == 256 && BN == 256 && WAVES_M == 2 && WAVES_N == 4, "the 8-phase loop is laid out for 256x256 tiles, 2x4 waves");
    const int grp = wave >> 2;
    int keyA[TM], keyB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) keyA[i] = h ^ ((rowA[i] >> 1) & 7);
#pragma unroll
    for (int j = 0; j < TN; ++j) keyB[j] = h ^ ((rowB[j] >> 1) & 7);
    bf16x8 af[2][4], bfr[2][4];                            // A fragments [block of the pair][k-step], B fragments [j][k-step]
    // operand pointers of a K-tile (dual-K switch, tile-blocked weights) are resolved ONCE per loop iteration on the scalar unit and
    // handed to the issue lambdas: computed inside every issue they were ~20 SALU instructions per phase in the DMA-issuing wave
    auto tileA = [&](int t) -> const bf16* { return t >= nk1 ? A2 + (t - nk1) * BK : A + t * BK; };
    auto tileB = [&](int t) -> const bf16* {
      const bf16* Bp = B + blkB1; int k0 = t * BK;
      if (t >= nk1) { Bp = B2 + blkB2; k0 = (t - nk1) * BK; }
      return Bp + (p.b_blocked ? (long long)(k0 >> 6) * 16384 : (long long)k0);
    };
    auto issueA = [&](const bf16* Ap, int t, auto HALF) {
      constexpr int hf = decltype(HALF)::value;
      char* sA = smem + (t & 1) * STAGE;
#pragma unroll
      for (int r = 2 * hf; r < 2 * hf + 2; ++r) __builtin_amdgcn_global_load_lds(GLB_PTR(Ap + offA[r]), LDS_PTR(sA + (r * NT + wave * 64) * 16), 16, 0, GLDS_AUX_A);
    };
    auto issueB = [&](const bf16* Bp, int t, auto HALF) {
      constexpr int hf = decltype(HALF)::value;
      char* sB = smem + (t & 1) * STAGE + A_BYTES;
#pragma unroll
      for (int r = 2 * hf; r < 2 * hf + 2; ++r) __builtin_amdgcn_global_load_lds(GLB_PTR(Bp + offB[r]), LDS_PTR(sB + (r * NT + wave * 64) * 16), 16, 0, GLDS_AUX_B);
    };
    auto readA = [&](const char* sA_, auto IH) {
      constexpr int ih = decltype(IH)::value;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          af[ii][ks] = *(const bf16x8*)(sA_ + rowA[2 * ih + ii] * 128 + (((ks * 2) ^ keyA[2 * ih + ii]) << 4));
    };
    auto readB = [&](const char* sB_, auto J) {
      constexpr int j = decltype(J)::value;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bfr[j][ks] = *(const bf16x8*)(sB_ + rowB[j] * 128 + (((ks * 2) ^ keyB[j]) << 4));
    };
    // the phase's half-tile of LDS-DMA is issued from INSIDE the MFMA cluster (after the first two MFMAs): next to matrix
    // instructions a DMA piece costs the wave ~60 cycles of issue, inside the fragment-read interval (LDS queue busy) 100-185
    auto mma = [&](auto IH, auto J, auto&& dma) {
      constexpr int ih = decltype(IH)::value, j = decltype(J)::value;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) acc[2 * ih + ii][j] = mm<DT>(bfr[j][0], af[ii][0], acc[2 * ih + ii][j]);
      __builtin_amdgcn_sched_barrier(0);
      dma();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) acc[2 * ih + ii][j] = mm<DT>(bfr[j][ks], af[ii][ks], acc[2 * ih + ii][j]);
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
    // ---- prologue: K-tile 0 complete, three half-tiles of K-tile 1 in flight
    if (!have0) { issueA(tileA(0), 0, I0{}); issueB(tileB(0), 0, I0{}); issueB(tileB(0), 0, I1{}); issueA(tileA(0), 0, I1{}); }
    else bar();                                            // persistent loop: the epilogue staging region overlaps buffer 1
    if (nk > 1 && !(NT_DBG(p) & 1)) {
      issueA(tileA(1), 1, I0{}); issueB(tileB(1), 1, I0{}); issueB(tileB(1), 1, I1{});
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    if (grp == 1) bar();                                   // the lower wave-row runs one barrier behind
    const bf16* A_t1 = tileA(min(1, nk - 1));              // operands of the K-tiles the coming iteration will issue; refreshed in P4,
    const bf16* A_t2 = tileA(min(2, nk - 1));              // the phase without fragment reads
    const bf16* B_t2 = tileB(min(2, nk - 1));
    for (int t = 0; t < nk; ++t) {
      const char* sA = smem + (t & 1) * STAGE;
      const char* sB = sA + A_BYTES;
      const bool more1 = t + 1 < nk && !(NT_DBG(p) & 1), more2 = t + 2 < nk && !(NT_DBG(p) & 1);
      // P1
      readA(sA, I0{});
      __builtin_amdgcn_sched_barrier(0);
      readB(sB, I0{});
      bar();
      lgkm0();
      mma(I0{}, I0{}, [&] { if (more1) issueA(A_t1, t + 1, I1{}); });
      bar();
      // P2
      readB(sB, I1{});
      bar();
      lgkm0();
      mma(I0{}, I1{}, [&] { if (more2) issueA(A_t2, t + 2, I0{}); });
      bar();
      // P3
      readA(sA, I1{});
      bar();
      lgkm0();
      mma(I1{}, I1{}, [&] { if (more2) issueB(B_t2, t + 2, I0{}); });
      bar();
      // P4 (no fragment reads).  The counted wait sits HERE, in front of P4's first barrier (round 5; rounds 2-4 had it at the end of P4's MFMA
      // cluster, where the lagging wave-row executes it in the same interval in which the leading row already reads K-tile t+1 -- eight phases of
      // slack, but not ordered: tests/test_kloop_schedule_cpu.py).  Outstanding, newest first: B0(t+2), A0(t+2) | A1(t+1), B1(t+1), ... -> vmcnt(4)
      // retires K-tile t+1; B1(t+2) is issued behind it from the cluster.  Same sums, bit-identical results.
      const bf16* nA1 = tileA(min(t + 2, nk - 1));
      const bf16* nA2 = tileA(min(t + 3, nk - 1));
      const bf16* nB2 = tileB(min(t + 3, nk - 1));
      if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bar();
      mma(I1{}, I0{}, [&] { if (more2) issueB(B_t2, t + 2, I1{}); });
      A_t1 = nA1; A_t2 = nA2; B_t2 = nB2;
      bar();
    }
    if (grp == 0) bar();                                   // balance the barrier count
  } else if constexpr (PIPE == 4) {
    // ---- 4-phase main loop on v_mfma_f32_16x16x32 (round 5; tools/kloop_lab.hip V3, tools/power_lab.hip).
    //   Why 16x16x32: on random operands the matrix pipe alone is POWER-limited -- 1.95 PFLOP/s at the 1.4 kW package cap with 32x32x16,
    //   2.2 PFLOP/s with 16x16x32 (a quarter of the accumulator traffic per flop): the same flops for ~11 % fewer joules, which is what a
    //   loop that runs at the power cap is paid in (profiles/r05_power_lab.txt).
    //   Same LDS image, same LDS-DMA half-tiles (A0 A1 B0 B1, 2 instructions per wave each) and the same row ownership as the 8-phase loop; the
    //   two wave-rows run ONE barrier apart.  Per K-tile and wave two segments of 32 MFMAs (4 barriers instead of 8):
    //        S1 = [ reads A(ih0) + B, 16 x ds_read_b128 | bar | 32 MFMAs: rows ih0 x all 64 columns x K=64 | bar ]
    //        S2 = [ reads A(ih1),      8 x ds_read_b128 | bar | 32 MFMAs: rows ih1                          | bar ]
    //   LDS-DMA issue (from inside the MFMA clusters, SADDR form: no vector address arithmetic):
    //        M-S1(t): X(t) = A1(t+1)                  M-S2(t): Y(t) = A0(t+2), B0(t+2), B1(t+2)
    //   Counted waits, both in the READ segments (every wave executes them, whichever wave-row it is in):
    //        end of R-S1(t): vmcnt(6) -> X(t-1) = A1(t) has landed           end of R-S2(t): vmcnt(2) -> Y(t-1) = A0, B0, B1 of t+1 have landed
    //   RAW: a half-tile is first read (by the leading wave-row) in the segment AFTER a barrier that every wave passed behind its own wait:
    //        A1(t) is waited for in R-S1(t) (intervals 1 / 2 of K-tile t) and read from R-S2(t) on (interval 3); A0, B0, B1(t+1) are waited for in
    //        R-S2(t) (intervals 3 / 4) and read from R-S1(t+1) on (interval 5).  (The 8-phase loop's single wait at the end of P4 was one
    //        barrier short of this for the lagging wave-row's share -- 8 phases of slack in practice, but not ordered.)
    //   WAR: A0, B0, B1(t) are last read in R-S1(t) by the lagging row (interval 2, retired by its lgkmcnt(0) at the start of interval 3) and
    //        re-staged from M-S2(t) (intervals 4 / 5); A1(t) is last read in interval 4 and re-staged from M-S1(t+1) (intervals 6 / 7).
    static_assert(BM == 256 && BN == 256 && WAVES_M == 2 && WAVES_N == 4, "the 4-phase loop is laid out for 256x256 tiles, 2x4 waves");
    const int grp = wave >> 2;
    const int colsw = (g4 ^ (l15 >> 1)) << 4;             // swizzled 16-byte chunk of k-step 0 (key (row >> 1) & 7 = (l & 15) >> 1 for every block); k-step 1: ^ 64
    const int rdA = (wm0 + l15) * 128 + colsw, rdB = A_BYTES + (wn0 + l15) * 128 + colsw;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)LDS_PTR(smem) + wave * 1024;
    bf16x8 a16[4][2], b16[4][2];
    auto tileA = [&](int t) -> const char* { return (const char*)(t >= nk1 ? A2 + (t - nk1) * BK : A + t * BK); };
    auto tileB = [&](int t) -> const char* { return (const char*)(t >= nk1 ? B2 + (t - nk1) * BK : B + t * BK); };
    auto issueA = [&](const char* base, int t, auto HALF) {
      constexpr int hf = decltype(HALF)::value;
      const unsigned l = lds0 + (t & 1) * STAGE;
#pragma unroll
      for (int r = 2 * hf; r < 2 * hf + 2; ++r) dma16(l + r * NT * 16, offA[r], base);
    };
    auto issueB = [&](const char* base, int t, auto HALF) {
      constexpr int hf = decltype(HALF)::value;
      const unsigned l = lds0 + (t & 1) * STAGE + A_BYTES;
#pragma unroll
      for (int r = 2 * hf; r < 2 * hf + 2; ++r) dma16(l + r * NT * 16, offB[r], base);
    };
    auto readS1 = [&](const char* s_) {                    // k-step 0 of B and A first: the order the MFMAs consume them in
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) b16[nb][kk] = *(const bf16x8*)(s_ + ((rdB + nb * 2048) ^ (kk << 6)));
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a16[mb][kk] = *(const bf16x8*)(s_ + ((rdA + mb * 2048) ^ (kk << 6)));
      }
    };
    auto readS2 = [&](const char* s_) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a16[mb][kk] = *(const bf16x8*)(s_ + ((rdA + 16384 + mb * 2048) ^ (kk << 6)));
    };
    auto mma = [&](auto IH, auto&& d0, auto&& d1, auto&& d2) {
      constexpr int ih = decltype(IH)::value;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          f32x16& C = acc[2 * ih + (mb >> 1)][mb & 1];
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            f32x4 c = {C[4 * nb], C[4 * nb + 1], C[4 * nb + 2], C[4 * nb + 3]};
            c = mm16<DT>(b16[nb][kk], a16[mb][kk], c);
            C[4 * nb] = c[0]; C[4 * nb + 1] = c[1]; C[4 * nb + 2] = c[2]; C[4 * nb + 3] = c[3];
          }
          __builtin_amdgcn_sched_barrier(0);
          if (kk == 0 && mb == 0) d0();
          if (kk == 0 && mb == 1) d1();
          if (kk == 0 && mb == 2) d2();
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
    // ---- prologue: K-tile 0 complete, A0 / B0 / B1 of K-tile 1 in flight (= Y(-1); X(0) = A1(1) comes from M-S1(0))
    if (!have0) { issueA(tileA(0), 0, I0{}); issueB(tileB(0), 0, I0{}); issueB(tileB(0), 0, I1{}); issueA(tileA(0), 0, I1{}); }
    else bar();                                            // persistent loop: the epilogue staging region overlaps buffer 1
    if (nk > 1) {
      issueA(tileA(1), 1, I0{}); issueB(tileB(1), 1, I0{}); issueB(tileB(1), 1, I1{});
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    if (grp == 1) bar();                                   // the lower wave-row runs one barrier behind
    NT_STAMP(0);
    for (int t = 0; t < nk; ++t) {
      const char* sT = smem + (t & 1) * STAGE;
      const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
      const char* pA1 = tileA(min(t + 1, nk - 1));        // operand bases of the half-tiles this K-tile's clusters issue (scalar unit)
      const char* pA2 = tileA(min(t + 2, nk - 1));
      const char* pB2 = tileB(min(t + 2, nk - 1));
      // R-S1.  Outstanding behind the reads, newest first: Y(t-1) [6, if issued] | X(t-1) [2]  ->  X(t-1) = A1(t) landed
      readS1(sT);
      if (more1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bar();
      lgkm0();
      mma(I0{}, [&] { if (more1) issueA(pA1, t + 1, I1{}); }, nop, nop);
      bar();
      // R-S2.  Outstanding: X(t) [2, if issued] | Y(t-1)  ->  Y(t-1) = A0, B0, B1 of K-tile t+1 landed
      readS2(sT);
      if (more1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bar();
      lgkm0();
      mma(I1{}, [&] { if (more2) issueA(pA2, t + 2, I0{}); }, [&] { if (more2) issueB(pB2, t + 2, I0{}); }, [&] { if (more2) issueB(pB2, t + 2, I1{}); });
      bar();
    }
    if (grp == 0) bar();                                   // balance the barrier count
    NT_STAMP(1);
  } else if constexpr (WAVES_M == 2 && WAVES_N == 4) {
    // ---- ping-pong main loop (8 waves).  The two wave-rows of the tile (waves 0-3 / 4-7; waves w and w+4 share a
    // SIMD) run the same 4-phase K-tile sequence  R0 | M0 | R1 | M1  (R = fragment reads of two k-steps (+ LDS-DMA
    // of the next tile in R0), M = 32 MFMAs), every phase ending in s_barrier -- but the lower wave-row executes ONE
    // extra barrier up front, so it always runs one phase behind: while one wave of a SIMD issues MFMAs the other
    // one reads LDS / issues DMA, and the matrix pipe never waits for fragment reads.
    //   RAW: a wave waits vmcnt(0) for its own DMA at the end of R1(t); every wave passes >= 1 further barrier
    //        before anyone reads tile t+1.
    //   WAR: DMA for tile t+1 (buffer of tile t-1) is issued in R0(t), after a barrier that every wave passed with
    //        lgkmcnt(0) behind its last read of tile t-1.
    const int grp = wave >> 2;
    bf16x8 af[2][TM], bfr[2][TN];
    auto read2 = [&](const char* sA_, const char* sB_, int ks0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int lc = (ks0 + u) * 2 + h;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[u][i] = *(const bf16x8*)(sA_ + rowA[i] * 128 + ((lc ^ ((rowA[i] >> 1) & 7)) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[u][j] = *(const bf16x8*)(sB_ + rowB[j] * 128 + ((lc ^ ((rowB[j] >> 1) & 7)) << 4));
      }
    };
    auto mma2 = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mm<DT>(bfr[u][j], af[u][i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    };
    auto phase_end = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    if (!have0) stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // one-phase lag of the lower wave-row
    for (int t = 0; t < nk; ++t) {
      const char* sA = smem + (t & 1) * STAGE;
      const char* sB = sA + A_BYTES;
      read2(sA, sB, 0);                                   // R0
      if (t + 1 < nk && !(NT_DBG(p) & 1)) stage(t + 1, (t + 1) & 1);
      phase_end();
      mma2();                                             // M0
      phase_end();
      read2(sA, sB, 2);                                   // R1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      phase_end();
      mma2();                                             // M1
      phase_end();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // balance the barrier count
  } else {
  if (!have0) stage(0, 0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own share of tile t landed -- explicit: never rely on the compiler's placement
    __syncthreads();                    // tile t visible to all waves; buffer (t+1)&1 is free
    if (t + 1 < nk) stage(t + 1, (t + 1) & 1);
    const char* sA = smem + (t & 1) * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[TM], bfr[TN];
      const int lc = ks * 2 + h;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *(const bf16x8*)(sA + rowA[i] * 128 + ((lc ^ ((rowA[i] >> 1) & 7)) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *(const bf16x8*)(sB + rowB[j] * 128 + ((lc ^ ((rowB[j] >> 1) & 7)) << 4));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mm<DT>(bfr[j], af[i], acc[i][j]);   // D[n][m]: lane = row m
    }
  }
  }

  // ---------------- epilogue
  // Accumulators -> LDS -> registers, WAVE-PRIVATE: each wave transposes its own WM x WN sub-tile through a private
  // [32][WN+4] f32 region, one 32-row block at a time (16-byte writes in MFMA layout: conflict-free with the +4 pad;
  // 16-byte reads with WN/4 lanes covering one row), so that ALL global epilogue traffic (bias, residual,
  // pre-activation, output) is coalesced: a row segment of WN columns = 128 B of bf16 / 256 B of f32 per request.
  // No block-level barrier after the first one, and all waves (both SIMD halves of the LDS store path) stay busy.
  if (NT_DBG(p) & 2) return;                                         // timing experiment: no epilogue (non-persistent only)
  constexpr int EP_LD = WN + 4;
  constexpr int LPR = WN / 4, RPI = 64 / LPR;                    // lanes per row, rows per wave-instruction
  constexpr int NIT = 32 / RPI;
  __syncthreads();                                               // every wave is done with the K-loop buffers
  NT_STAMP(2);
  const int em0 = m0, en0 = n0;                                  // the tile being written out
  bool more = false;
  if constexpr (PERSIST) {
    const int nvb = vb + (int)gridDim.x;
    more = nvb < total_tiles;
    if (more) { vb = nvb; set_tile(vb); stage(0, 0); }            // next tile's first K-tile lands in buffer 0 during the epilogue
  }
  float* ep = (float*)(smem + (PERSIST ? STAGE : 0)) + wave * (32 * EP_LD);   // persistent: staging lives above buffer 0
  const int er = lane / LPR, ec = (lane % LPR) * 4;
  auto stage_block = [&](const f32x16 (&ai)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4*)(ep + erow(j) * EP_LD + ecol(j, q)) =
            f32x4{ai[j][4 * q], ai[j][4 * q + 1], ai[j][4 * q + 2], ai[j][4 * q + 3]};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };

  static_assert(EPI < 0 || (EPI & F_PGRAD) == 0 || (EPI & (F_PREACT | F_ACT | F_RES | F_MULZ | F_CF32)) == (F_PREACT | F_ACT),
                "act'(z) output is a fast-lane option of the 16-bit pre-activation + activation epilogue only");
  static_assert(EPI < 0 || (EPI & F_C2) == 0 || DT == 2 || (EPI & F_CF32) != 0 || (EPI & F_SPLIT3) != 0, "C2 is the bf16 copy of an f32 C, or of an fp16 C (DT 2)");
  static_assert(EPI < 0 || (EPI & F_SPLIT3) == 0 || ((EPI & (F_PREACT | F_C2 | F_PGRAD | F_RES | F_MULZ | F_CF32)) == (F_PREACT | F_C2) && DT == 0),
                "the split output is the three-output 16-bit epilogue on bf16 operands");
  if constexpr (epi_packed(EPI, DT)) {
    // ---- bf16-only outputs without streamed operands (plain / bias, and fc1's pre-activation + activation): bias and
    // activation are applied in MFMA layout, the results go through LDS as packed bf16 (half the staging traffic of the
    // f32 round trip) and leave as 16-byte stores, 8 lanes per 128-byte row segment.
    constexpr int NO = epi_outputs(EPI);                          // outputs: C, [preact], [C2 = bf16 copy of an fp16 C]
    constexpr int S2 = ((EPI & F_PREACT) != 0) ? 2 : 1;           // staging slot of C2
    constexpr int RS = WN * 2 + 16;                               // staging row stride in bytes (16-byte aligned rows)
    static_assert(WN == 64, "bf16 staging is laid out for 64-column wave tiles");
    char* epb = smem + (PERSIST ? STAGE : 0) + wave * (NO * 32 * RS);
    const long long cb = bz * p.sC;
    f32x4 bq[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bq[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = en0 + wn0 + ecol(j, q);
        if (p.bias) bq[j][q] = *(const f32x4*)(p.bias + n);
        if (p.bias2) bq[j][q] += *(const f32x4*)(p.bias2 + n);
      }
    const int rr = lane >> 3, rc = (lane & 7) * 8;                // read-back: row within 8, first of 8 columns
    // Store addresses = wave-uniform sub-tile base (SGPRs, advanced per row group by the scalar unit) + ONE 32-bit lane
    // offset: the stores take the saddr form and the epilogue carries no per-row 64-bit VALU address arithmetic (it was
    // a third of this lane's VALU instructions; the epilogue is VALU-bound).
    const long long ub = cb + (long long)(em0 + wm0) * p.ldc + en0 + wn0;
    char* const cbase = (char*)p.C + ub * 2;
    char* const pbase = (char*)p.preact + ub * 2;
    char* const c2base = (char*)p.C2 + (cb + (long long)(em0 + wm0) * p.ldc2 + en0 + wn0) * 2;
    const unsigned loff = ((unsigned)rr * (unsigned)p.ldc + (unsigned)rc) * 2u;
    const unsigned loff2 = ((unsigned)rr * (unsigned)p.ldc2 + (unsigned)rc) * 2u;
    const int mrem = p.M - (em0 + wm0);                            // rows of this sub-tile inside the matrix (wave-uniform)
    f32x4 sq[TN][4];                                               // F_LNFOLD: s[n] of this lane's column quads
    if constexpr ((EPI & F_LNFOLD) != 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) sq[j][q] = *(const f32x4*)(p.ln_colsum + en0 + wn0 + ecol(j, q));
    }
    // F_LNFOLD: statistics of the rows this lane holds in acc[i][j] (rows past M: those of row M - 1, never stored), all requested before the first block
    float lmean[(EPI & F_LNFOLD) != 0 ? TM : 1][TN], lrstd[(EPI & F_LNFOLD) != 0 ? TM : 1][TN];
    if constexpr ((EPI & F_LNFOLD) != 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int m = em0 + wm0 + min(ro(i) + erow(j), mrem - 1);
          lmean[i][j] = p.ln_mean[m]; lrstd[i][j] = p.ln_rstd[m];
        }
    }
    auto run = [&](auto guard) {                                  // two straight-line arms, one wave-uniform row-guard test per sub-tile
    constexpr bool GUARD = decltype(guard)::value;
    auto block = [&](const int i, const f32x16 (&ai)[TN]) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = f32x4{ai[j][4 * q], ai[j][4 * q + 1], ai[j][4 * q + 2], ai[j][4 * q + 3]} * p.alpha;
          if constexpr ((EPI & F_LNFOLD) != 0) v = (v - sq[j][q] * lmean[i][j]) * lrstd[i][j] + bq[j][q];
          else v += bq[j][q];
          char* dst = epb + erow(j) * RS + ecol(j, q) * 2;
          if constexpr ((EPI & F_PGRAD) != 0) {
            f32x2 a0, a1, g0, g1;
#ifdef KO_EPI_ACT
            a0 = f32x2{v[0], v[1]}; a1 = f32x2{v[2], v[3]}; g0 = a0 + a0; g1 = a1 + a1;
#else
            act_fwd_grad2(f32x2{v[0], v[1]}, ACTK, a0, g0);
            act_fwd_grad2(f32x2{v[2], v[3]}, ACTK, a1, g1);
#endif
            *(bf16x4*)(dst + 32 * RS) = f2x2_to_bf4(g0, g1);
            *(bf16x4*)dst = cvt_c2<DT>(a0, a1);
            if constexpr ((EPI & F_C2) != 0) *(bf16x4*)(dst + S2 * 32 * RS) = f2x2_to_bf4(a0, a1);
            continue;
          } else {
            if constexpr ((EPI & F_PREACT) != 0 && (EPI & F_SPLIT3) == 0) *(bf16x4*)(dst + 32 * RS) = f4_to_bf4(v);
            if constexpr ((EPI & F_ACT) != 0) {
              // bf16 outputs: the same packed fast forms as the act + act' epilogue (bit-identical activations whether or
              // not the derivative is kept); the f32-output lanes keep the accurate erff / division forms
              f32x2 a0, a1, g0, g1;
              act_fwd_grad2(f32x2{v[0], v[1]}, ACTK, a0, g0);
              act_fwd_grad2(f32x2{v[2], v[3]}, ACTK, a1, g1);
              v = f32x4{a0[0], a0[1], a1[0], a1[1]};
            }
          }
          if constexpr ((EPI & F_SPLIT3) != 0) {                  // [hi | lo | hi] of the f32 result: the A operand of the next split GEMM
            const bf16x4 hi = f4_to_bf4(v);
            *(bf16x4*)(dst + 32 * RS) = f4_to_bf4(v - bf4_to_f4(hi));
            *(bf16x4*)dst = hi;
            *(bf16x4*)(dst + S2 * 32 * RS) = hi;
            continue;
          }
          *(bf16x4*)dst = cvt_c<DT>(v);
          if constexpr ((EPI & F_C2) != 0) *(bf16x4*)(dst + S2 * 32 * RS) = f4_to_bf4(v);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rr;
        const bf16x8 o = *(const bf16x8*)(epb + row * RS + rc * 2);
        const long long uo = (long long)(ro(i) + it * 8) * p.ldc * 2;          // wave-uniform byte offset of the row group
#ifdef KO_EPI_ST
        if (p.alpha == 12345.f) {
#else
        if (!GUARD || ro(i) + row < mrem) {
#endif
          st_stream((bf16x8*)(cbase + uo + loff), o);
          if constexpr ((EPI & F_PREACT) != 0) st_stream((bf16x8*)(pbase + uo + loff), *(const bf16x8*)(epb + 32 * RS + row * RS + rc * 2));
          if constexpr ((EPI & F_C2) != 0)
            st_stream((bf16x8*)(c2base + (long long)(ro(i) + it * 8) * p.ldc2 * 2 + loff2), *(const bf16x8*)(epb + S2 * 32 * RS + row * RS + rc * 2));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    };
    block(0, acc[0]); NT_STAMP(3);
    if constexpr (TM > 1) { block(1, acc[1]); NT_STAMP(4); }
    if constexpr (TM > 2) { block(2, acc[2]); NT_STAMP(5); }
    if constexpr (TM > 3) { block(3, acc[3]); NT_STAMP(6); }
    };
    if (mrem >= ro(TM - 1) + 32) run(std::false_type{}); else run(std::true_type{});
    NT_STAMP(7);
  } else if constexpr (EPI >= 0) {
    // ---- fast lane: full-width tiles, 16-byte aligned rows (checked by the host dispatcher).  The streamed operands of a
    // 32-row block (fp32 residual, z of the activation derivative) are fetched BEFORE the LDS round trip of the
    // accumulators: NIT independent loads in flight per lane instead of one load -> compute -> store chain per quad.
    const int n = en0 + wn0 + ec;
    const long long cb = bz * p.sC, ab = bz * p.sAux;
    // addresses = wave-uniform sub-tile base + one 32-bit lane offset per stream (saddr form, see the bf16 lane above)
    const long long ubc = cb + (long long)(em0 + wm0) * p.ldc + en0 + wn0, ubr = ab + (long long)(em0 + wm0) * p.ldres + en0 + wn0;
    const long long ub2 = cb + (long long)(em0 + wm0) * p.ldc2 + en0 + wn0;
    constexpr int CS = (EPI & F_CF32) != 0 ? 4 : 2;               // bytes per output element
    char* const cbase = (char*)p.C + ubc * CS;
    char* const pbase = (char*)p.preact + ubc * 2;
    char* const c2base = (char*)p.C2 + ub2 * 2;
    const char* const rbase = (const char*)p.res_f32 + ubr * 4;
    const char* const zbase = (const char*)p.mulz + ubr * 2;
    const int mrem = p.M - (em0 + wm0);                            // rows of this sub-tile inside the matrix (wave-uniform)
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *(const f32x4*)(p.bias + n);
    if (p.bias2) bv += *(const f32x4*)(p.bias2 + n);
    // Straight-line code per arm: one wave-uniform row-guard test for the whole sub-tile (with a per-row branch around
    // the stores -- or even a per-block one -- the compiler cannot count the outstanding stores at the join and waits
    // vmcnt(0) before every use of a prefetched operand: each row then waited for the previous row's store to be
    // acknowledged by memory).  The streamed operands run NPRE blocks ahead of their use and are requested before the
    // stores of the blocks in between, so the in-order vmcnt wait for them never covers a store of this tile.
    constexpr bool PF = (EPI & (F_RES | F_MULZ)) != 0;
    constexpr int NPRE = !PF ? 1 : (EPI & F_RES) != 0 ? (TM < 2 ? TM : 2) : (PERSIST && TM > 3 ? 2 : TM);   // 32 / 16 VGPRs per block in flight (persistent: next to the following tile's addressing)
    auto run = [&](auto guard) {
      constexpr bool GUARD = decltype(guard)::value;
      f32x4 pr[NPRE][NIT];
      bf16x4 pz[NPRE][NIT];
      // address of (row group g = i*32 + it*RPI, this lane) in a stream with row stride ld and esz-byte elements.
      // Unguarded arm: uniform base + g*ld (scalar unit) + a zero-extended 32-bit lane constant -> saddr form.
      // Guarded arm: the row is clamped per lane and the offset is SIGNED (a sub-tile that starts past the last row
      // clamps to row M-1, which lies before its base).
      const unsigned lo_c = (unsigned)(er * p.ldc + ec), lo_r = (unsigned)(er * p.ldres + ec), lo_2 = (unsigned)(er * p.ldc2 + ec);
      auto adr = [&](const char* base, int g, int ld, int esz, unsigned lo) -> char* {
        if constexpr (GUARD) return (char*)base + ((long long)min(g + er, mrem - 1) * ld + ec) * esz;
        else return (char*)base + (long long)g * ld * esz + lo * (unsigned)esz;
      };
      auto fetch = [&](const int i) {
        if constexpr (PF) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int g = ro(i) + it * RPI;
#ifdef KO_EPI_LD
            if constexpr ((EPI & F_RES) != 0) pr[i % NPRE][it] = f32x4{p.alpha, p.alpha, p.alpha, p.alpha};
            if constexpr ((EPI & F_MULZ) != 0) pz[i % NPRE][it] = f4_to_bf4(f32x4{p.alpha, p.alpha, p.alpha, p.alpha});
#else
            if constexpr ((EPI & F_RES) != 0) pr[i % NPRE][it] = ld_stream((const f32x4*)adr(rbase, g, p.ldres, 4, lo_r));
            if constexpr ((EPI & F_MULZ) != 0) pz[i % NPRE][it] = ld_stream((const bf16x4*)adr(zbase, g, p.ldres, 2, lo_r));
#endif
          }
        }
      };
      auto block = [&](const int i, const f32x16 (&ai)[TN]) {
        stage_block(ai);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int row = it * RPI + er, g = ro(i) + it * RPI;
          f32x4 v = *(const f32x4*)(ep + row * EP_LD + ec);
          v = v * p.alpha + bv;
#ifdef KO_EPI_ST
          const bool ok = p.alpha == 12345.f;
#else
          const bool ok = !GUARD || g + er < mrem;
#endif
          if constexpr ((EPI & F_PREACT) != 0) { if (ok) st_stream((bf16x4*)adr(pbase, g, p.ldc, 2, lo_c), f4_to_bf4(v)); }
          if constexpr ((EPI & F_ACT) != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], ACTK);
          }
          if constexpr ((EPI & F_MULZ) != 0) {
            const f32x4 z = bf4_to_f4(pz[i % NPRE][it]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= act_grad(z[e], ACTK);
          }
          if constexpr ((EPI & F_RES) != 0) v += pr[i % NPRE][it];
          if constexpr ((EPI & F_STATS) != 0) {
            static_assert(LPR == 16, "the row reduction below runs over the 16 lanes of a DPP row");
            const float s1 = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
            const float s2 = row16_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
            // lanes 0 / 1 of the row's 16 write sum / sum of squares: one predicated 4-byte store per quad of rows
            const int mrow = GUARD ? min(g + er, mrem - 1) : g + er;
            float* sp = p.row_stats + ((long long)(em0 + wm0 + mrow) * (p.N >> 6) + ((en0 + wn0) >> 6)) * 2 + (lane & 1);
            if ((lane & 14) == 0 && ok) *sp = (lane & 1) ? s2 : s1;
          }
          if (ok) {
            if constexpr ((EPI & F_CF32) != 0) st_stream((f32x4*)adr(cbase, g, p.ldc, 4, lo_c), v);
            else st_stream((bf16x4*)adr(cbase, g, p.ldc, 2, lo_c), cvt_c<DT>(v));
            if constexpr ((EPI & F_C2) != 0) st_stream((bf16x4*)adr(c2base, g, p.ldc2, 2, lo_2), f4_to_bf4(v));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      };
      fetch(0);
      if constexpr (NPRE > 1) fetch(1);
      if constexpr (NPRE > 2) fetch(2);
      if constexpr (NPRE > 3) fetch(3);
      block(0, acc[0]); NT_STAMP(3);
      if constexpr (NPRE < TM) fetch(NPRE);                        // refill the slot block 0 just freed
      if constexpr (TM > 1) { block(1, acc[1]); NT_STAMP(4); }
      if constexpr (NPRE + 1 < TM) fetch(NPRE + 1);
      if constexpr (TM > 2) { block(2, acc[2]); NT_STAMP(5); }
      if constexpr (NPRE + 2 < TM) fetch(NPRE + 2);
      if constexpr (TM > 3) { block(3, acc[3]); NT_STAMP(6); }
    };
    if (em0 + wm0 + ro(TM - 1) + 32 <= p.M) run(std::false_type{}); else run(std::true_type{});
    NT_STAMP(7);
    static_assert(TM <= 4, "epilogue blocks are written out for TM <= 4");
  } else {
  // ---- general lane: lane owns row m, 4 consecutive n per register quad
  const bool vec = ((p.ldc & 3) == 0) && (!p.res_f32 && !p.res_bf16 && !p.mulz || (p.ldres & 3) == 0) &&
                   (!p.C2 || (p.ldc2 & 3) == 0);
  float* Cf = p.c_f32 ? (float*)p.C + bz * p.sC : nullptr;
  bf16* Cb = p.c_f32 ? nullptr : (bf16*)p.C + bz * p.sC;
  bf16* C2 = p.C2 ? p.C2 + bz * p.sC : nullptr;
  bf16* PA = p.preact ? p.preact + bz * p.sC : nullptr;
  const float* Rf = p.res_f32 ? p.res_f32 + bz * p.sAux : nullptr;
  const bf16* Rb = p.res_bf16 ? p.res_bf16 + bz * p.sAux : nullptr;
  const bf16* MZ = p.mulz ? p.mulz + bz * p.sAux : nullptr;

  // one register quad = 4 consecutive columns of one row; every loop below has a compile-time trip count so the
  // accumulators stay in registers (a runtime-indexed accumulator array would be demoted to scratch)
  auto emit = [&](int m, int n, float v0, float v1, float v2, float v3) {
    float v[4] = {v0 * p.alpha, v1 * p.alpha, v2 * p.alpha, v3 * p.alpha};
    const int nv = p.N - n;                                  // >= 1
    const bool full = vec && nv >= 4;
    if (p.bias && full && !p.bias_by_brow) {
      const f32x4 b = *(const f32x4*)(p.bias + n);
      v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
      if (p.bias2) { const f32x4 c = *(const f32x4*)(p.bias2 + n); v[0] += c[0]; v[1] += c[1]; v[2] += c[2]; v[3] += c[3]; }
    } else if (p.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nv) {
          const int bi = p.bias_by_brow ? p.b_rows[n + e] : n + e;
          v[e] += p.bias[bi];
          if (p.bias2) v[e] += p.bias2[bi];
        }
    }
    const long long oc = (long long)m * p.ldc + n;
    if (PA) {
      float pa[4] = {v[0], v[1], v[2], v[3]};
      if (p.preact_grad) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pa[e] = act_grad(v[e], p.act);
      }
      if (full) st_stream((bf16x4*)(PA + oc), f4_to_bf4(f32x4{pa[0], pa[1], pa[2], pa[3]}));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) PA[oc + e] = f2bf(pa[e]);
      }
    }
    if (p.act != ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], p.act);
    }
    const long long orr = (long long)m * p.ldres + n;
    if (MZ) {
      float z[4] = {0.f, 0.f, 0.f, 0.f};
      if (full) { f32x4 t = bf4_to_f4(ld_stream((const bf16x4*)(MZ + orr))); z[0] = t[0]; z[1] = t[1]; z[2] = t[2]; z[3] = t[3]; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) z[e] = bf2f(MZ[orr + e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= act_grad(z[e], p.mul_act);
    }
    if (Rf) {
      if (full) { f32x4 t = ld_stream((const f32x4*)(Rf + orr)); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) v[e] += Rf[orr + e];
      }
    }
    if (Rb) {
      if (full) { f32x4 t = bf4_to_f4(ld_stream((const bf16x4*)(Rb + orr))); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) v[e] += bf2f(Rb[orr + e]);
      }
    }
    if (Cf) {
      if (full) st_stream((f32x4*)(Cf + oc), f32x4{v[0], v[1], v[2], v[3]});
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) Cf[oc + e] = v[e];
      }
    } else {
      if (full) st_stream((bf16x4*)(Cb + oc), cvt_c<DT>(f32x4{v[0], v[1], v[2], v[3]}));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) {
          if constexpr (DT == 2) ((h16*)Cb)[oc + e] = (h16)v[e]; else Cb[oc + e] = f2bf(v[e]);
        }
      }
    }
    if (C2) {
      const long long o2 = (long long)m * p.ldc2 + n;
      if (full) st_stream((bf16x4*)(C2 + o2), f4_to_bf4(f32x4{v[0], v[1], v[2], v[3]}));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) C2[o2 + e] = f2bf(v[e]);
      }
    }
  };
  auto block = [&](const int i, const f32x16 (&ai)[TN]) {
    stage_block(ai);
#pragma unroll 2
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + er;
      const int m = em0 + wm0 + ro(i) + row, n = en0 + wn0 + ec;
      const f32x4 a = *(const f32x4*)(ep + row * EP_LD + ec);
      if (m < p.M && n < p.N) emit(m, n, a[0], a[1], a[2], a[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  block(0, acc[0]);
  if constexpr (TM > 1) block(1, acc[1]);
  if constexpr (TM > 2) block(2, acc[2]);
  if constexpr (TM > 3) block(3, acc[3]);
  }
  if (!more) break;
  have0 = true;
#ifdef MART_EXPERIMENTS
  ++nt_tile_it;
#endif
  }   // tile loop
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int PIPE = 0, int EPI = -1, int ACTK = 0, bool PERSIST = false, int DT = 0>
int launch(const Args& a, int batch, hipStream_t st) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int LDS_LOOP = PIPE == 3 ? 3 * (BM + BN) * 64 : 2 * (BM + BN) * 64 * 2;
  // epilogue staging per wave: packed bf16 rows (one or two outputs) in the bf16-only lane, one f32 block otherwise
  constexpr bool PACKED = epi_packed(EPI, DT);
  constexpr int EPI_WAVE = PACKED ? epi_outputs(EPI) * 32 * (BN / WAVES_N * 2 + 16) : 32 * (BN / WAVES_N + 4) * 4;
  constexpr int LDS_EPI = WAVES_M * WAVES_N * EPI_WAVE + (PERSIST ? (BM + BN) * 64 * 2 : 0);   // persistent: staging above buffer 0
  static_assert(LDS_EPI <= 160 * 1024, "epilogue staging does not fit next to the first K-tile buffer");
  constexpr int LDS = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;
  static MartAttrOnce once;
  bool* attr_set = once.slot();
  auto kern = gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N, PIPE, EPI, ACTK, PERSIST, DT>;
  if (!*attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      mart_set_error("gemm_nt: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      return -2;
    }
    *attr_set = true;
  }
  int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  if (PERSIST) {                                                 // one workgroup per CU slot (256 CUs; LDS allows 1 / 2 per CU)
    const int slots = 256 * (LDS <= 80 * 1024 ? 2 : 1);
    if (tiles > slots) tiles = slots;
  }
  hipLaunchKernelGGL(kern, dim3(tiles, batch), dim3(NT), LDS, st, a);
  MART_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifdef MART_EXPERIMENTS
extern "C" int mart_debug_nt_stamps(unsigned long long* host_out) {        // 8 waves x 8 stamps of the last tile_cfg 2567 launch
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nt_stamps), sizeof(unsigned long long) * 3 * 8 * 9) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int mart_gemm_nt(const mart_gemm_nt_desc* d, void* stream) {
  MART_CHECK(d != nullptr, "gemm_nt: null descriptor");
  MART_CHECK(d->M > 0 && d->N > 0 && d->K > 0, "gemm_nt: M,N,K must be positive");
  MART_CHECK(d->K % 64 == 0 && d->K2 % 64 == 0, "gemm_nt: K and K2 must be multiples of 64");
  MART_CHECK(d->lda % 8 == 0 && d->ldb % 8 == 0, "gemm_nt: lda/ldb must be multiples of 8 (16-byte rows)");
  MART_CHECK(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0, "gemm_nt: A/B must be 16-byte aligned");
  MART_CHECK((d->K2 == 0) == (d->A2 == nullptr) && (d->K2 == 0) == (d->B2 == nullptr), "gemm_nt: A2/B2/K2 inconsistent");
  MART_CHECK(d->C != nullptr && d->ldc >= d->N, "gemm_nt: bad C/ldc");
  MART_CHECK(!d->bias_by_brow || d->b_rows, "gemm_nt: bias_by_brow needs b_rows");
  MART_CHECK(d->act >= ACT_NONE && d->act <= ACT_QGELU, "gemm_nt: act must be NONE, GELU or QGELU");
  MART_CHECK(!d->preact_grad || (d->preact && d->act != ACT_NONE), "gemm_nt: preact_grad needs preact and an activation");
  MART_CHECK(d->mul_act >= ACT_NONE && d->mul_act <= ACT_STORED && (d->mul_act == ACT_NONE || d->mulz), "gemm_nt: bad mul_act");
  MART_CHECK(!d->c_f16 || (d->in_f16 && !d->c_f32), "gemm_nt: c_f16 needs in_f16 and a 16-bit C");
  MART_CHECK(!d->in_f16 || (!d->mulz && !d->res_bf16), "gemm_nt: fp16 operands are a forward-pass option (no mulz / bf16 residual)");
  MART_CHECK(!d->b_blocked || (d->N % 256 == 0 && !d->b_rows && (d->batch <= 1 || d->stride_b == 0)), "gemm_nt: b_blocked needs N % 256 == 0, no b_rows, shared B");
  MART_CHECK((long long)(d->a_rows ? d->a_src_rows : d->M) * d->lda < (1LL << 32) && (long long)(d->b_rows ? d->b_src_rows : d->N) * d->ldb < (1LL << 32),
             "gemm_nt: operand (or the table a row gather indexes: a_src_rows / b_src_rows) too large for 32-bit element offsets");
#if defined(MART_EXPERIMENTS) && defined(NT_SPLIT_ROUNDS) && NT_SPLIT_ROUNDS
  // Harness / variant builds only (-DMART_EXPERIMENTS -DNT_SPLIT_ROUNDS=1; valid for the plain, residual and activation epilogues of the benchmarked shapes:
  // A2 / bias_by_brow / a_rows semantics are NOT carried into the second launch).  Tile quantisation experiment (round 5): a product of 1.0-1.6 rounds of 256x256 tiles on the 256 CUs -- the
  // five N = 768 products per layer at the reference's default geometry, M = 25 344: 297 tiles -- pays two rounds when it runs alone.  Here it runs as
  // ONE full round of large tiles over the first rows and 128x128 tiles (two workgroups per CU) over the rest: two launches on the same stream, every
  // row-indexed operand advanced by the rows of the first part.  Alone: the eight products of a layer 0.888 -> 0.824 ms (-7 %, tools/bench_nt_p49.py).
  // In the step: 34.61 / 34.63 ms against 34.55 / 34.64 without it (same box, alternating) -- NOTHING: the weight-gradient and text queues already
  // run in the CUs a second round leaves idle.  Not shipped (docs/LAB_r01-r05.md section 6).
  if (d->tile_cfg == 0 && d->batch <= 1 && !d->a_rows && !d->A2 && !d->bias_by_brow && !d->b_blocked && !d->c_split3 && d->N % 256 == 0 && d->M > 256) {
    const int tn = d->N / 256, t256s = ((d->M + 255) / 256) * tn;
    const int rows_big = (256 / tn) * 256;
    if (t256s > 256 && t256s <= 410 && rows_big > 0 && rows_big < d->M) {
      if (d->K + d->K2 <= 1024) {                      // short contractions (12 K-tiles): the small tile alone fills the chip better than one round + a tail
        mart_gemm_nt_desc ds = *d;                     // (out-proj forward at M = 25 344: 0.053 ms against 0.060 split and 0.069 on 256x256 tiles)
        ds.tile_cfg = 128;
        return mart_gemm_nt(&ds, stream);
      }
      mart_gemm_nt_desc d1 = *d, d2 = *d;
      d1.M = rows_big; d1.tile_cfg = 256;
      const long long r = rows_big;
      const int ldres = d->ldres ? d->ldres : d->ldc, ldc2 = d->ldc2 ? d->ldc2 : d->ldc;
      auto adv = [](const void* ptr, long long bytes) -> const void* { return ptr ? (const void*)((const char*)ptr + bytes) : nullptr; };
      d2.M = d->M - rows_big; d2.tile_cfg = 128;
      d2.A = adv(d->A, r * d->lda * 2); d2.A2 = adv(d->A2, r * d->lda * 2);
      d2.preact = (void*)adv(d->preact, r * d->ldc * 2);
      d2.mulz = adv(d->mulz, r * ldres * 2);
      d2.res_f32 = (const float*)adv(d->res_f32, r * ldres * 4); d2.res_bf16 = adv(d->res_bf16, r * ldres * 2);
      d2.C = (void*)adv(d->C, r * d->ldc * (d->c_f32 ? 4 : 2));
      d2.C2 = (void*)adv(d->C2, r * ldc2 * 2);
      d2.row_stats = (float*)adv(d->row_stats, r * (d->N / 64) * 8);
      d2.ln_mean = (const float*)adv(d->ln_mean, r * 4); d2.ln_rstd = (const float*)adv(d->ln_rstd, r * 4);
      const int rc = mart_gemm_nt(&d1, stream);
      return rc ? rc : mart_gemm_nt(&d2, stream);
    }
  }
#endif
  Args a;
  a.A = (const bf16*)d->A; a.B = (const bf16*)d->B; a.A2 = (const bf16*)d->A2; a.B2 = (const bf16*)d->B2;
  a.lda = d->lda; a.ldb = d->ldb; a.M = d->M; a.N = d->N; a.K = d->K; a.K2 = d->K2;
  a.a_rows = d->a_rows; a.b_rows = d->b_rows;
  a.sA = d->stride_a; a.sB = d->stride_b; a.sC = d->stride_c; a.sAux = d->stride_aux;
  a.bias = d->bias; a.bias2 = d->bias2; a.bias_by_brow = d->bias_by_brow;
  a.act = d->act; a.preact = (bf16*)d->preact; a.preact_grad = d->preact_grad; a.mulz = (const bf16*)d->mulz; a.mul_act = d->mul_act;
  a.res_f32 = d->res_f32; a.res_bf16 = (const bf16*)d->res_bf16; a.ldres = d->ldres ? d->ldres : d->ldc;
  a.alpha = d->alpha; a.C = d->C; a.ldc = d->ldc; a.c_f32 = d->c_f32; a.C2 = (bf16*)d->C2;
  a.ldc2 = d->ldc2 ? d->ldc2 : d->ldc;
  const int batch = d->batch > 0 ? d->batch : 1;
  hipStream_t st = (hipStream_t)stream;
  long long t256 = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256) * batch;
  int cfg = d->tile_cfg;
  bool old_loop = false;
  // K loop of the 256x256 kernels: 4 = the 4-phase loop on 16x16x32 MFMAs (default); 2 = the 8-phase loop on 32x32x16 (rounds 2-4), kept for the
  // tile-blocked weight layout, for operands whose BYTE offsets do not fit the 32-bit lane offsets of the SADDR-form DMA, and (harness builds) for A/B runs
  const bool big_off = (long long)(d->a_rows ? d->a_src_rows : d->M) * d->lda >= (1LL << 31) || (long long)(d->b_rows ? d->b_src_rows : d->N) * d->ldb >= (1LL << 31);
  bool loop8 = d->b_blocked || big_off;
  a.dbg = 0;
  a.stagger_ticks = 0;
  a.b_blocked = d->b_blocked;
  a.row_stats = d->row_stats; a.ln_mean = d->ln_mean; a.ln_rstd = d->ln_rstd; a.ln_colsum = d->ln_colsum;
#ifdef MART_EXPERIMENTS
  if (cfg == 999) { a.dbg = 1; cfg = 256; }
  if (cfg >= 70000 && cfg < 80000) { a.stagger_ticks = cfg - 70000; cfg = 256; }
  if (cfg == 9992 || cfg == 9993) { a.dbg = cfg - 9990; cfg = 256; }           // 9992: no epilogue, 9993: no LDS-DMA and no epilogue
  if (cfg >= 99900 && cfg <= 99903) { a.dbg = cfg - 99900; old_loop = true; cfg = 256; }   // the same experiments on the round-1 loop
#else
  MART_CHECK(cfg == 0 || cfg == 128 || cfg == 256 || cfg == 2561, "gemm_nt: tile_cfg must be 0 (auto), 128, 256 or 2561 (256 tile, general epilogue); "
             "experiment codes need a -DMART_EXPERIMENTS build");
#endif
  // 256x256 tiles from half a round of workgroups up (128 tiles): the 192-tile text products (16384 x 768) run 7-15 % faster on
  // the large tile than on four times as many 128x128 tiles (step -0.6 %); below that the small tile fills the CUs better
  if (cfg == 0) cfg = ((t256 >= 128 && d->M > 128) || d->b_blocked) ? 256 : 128;   // short-M batched products (fusion, M = L = 64): 128-row tiles waste less
#ifdef MART_EXPERIMENTS
  // 2563 / 2564: the round-1 ping-pong K loop (one vmcnt(0) per K-tile), fast / general epilogue (A/B + bit-identity tests)
  if (cfg == 2563 || cfg == 2564) { old_loop = true; cfg = cfg == 2563 ? 256 : 2561; }
  if (cfg == 2562) cfg = 256;                       // 2562: fast epilogue without the persistent loop (A/B)
  if (cfg == 2565 || cfg == 2566) { loop8 = true; cfg = cfg == 2565 ? 256 : 2561; }   // the 8-phase loop, fast / general epilogue (A/B against the 4-phase loop)
  bool stamp_persist = false;
  if (cfg == 2567 || cfg == 2568) { a.dbg = 8; stamp_persist = cfg == 2568; cfg = 256; }   // cycle stamps of workgroup 200 (mart_debug_nt_stamps); 2568: persistent loop where the product uses it
  if (a.dbg & 7) loop8 = true;                      // the knock-out experiments are written into the older loops
#endif
  MART_CHECK(!d->b_blocked || cfg == 256, "gemm_nt: b_blocked requires the 256x256 tile");
#ifdef MART_EXPERIMENTS
  if (cfg == 2560) return launch<256, 256, 2, 4, 1>(a, batch, st);
  // 1282: 128x256 tiles, 4 waves, 3-slot ring of 32-deep steps, two workgroups per CU (fast epilogues where the shape allows)
  if (cfg == 1282) {
    const bool al = (d->ldc % (d->c_f32 ? 4 : 8) == 0) && (d->N % 256 == 0) && !d->res_bf16 && !d->bias_by_brow && !d->a_rows && !d->b_rows && d->K2 == 0 && !d->in_f16;
    const int mk = (d->res_f32 ? F_RES : 0) | (d->mulz ? F_MULZ : 0) | (d->preact ? F_PREACT : 0) | (d->preact_grad ? F_PGRAD : 0) | (d->act != ACT_NONE ? F_ACT : 0) |
                   (d->c_f32 ? F_CF32 : 0) | (d->C2 ? F_C2 : 0);
    const int kd = d->mulz ? d->mul_act : d->act;
    if (al && mk == 0) return launch<128, 256, 1, 4, 3, 0, ACT_NONE>(a, batch, st);
    if (al && mk == (F_CF32 | F_RES)) return launch<128, 256, 1, 4, 3, F_CF32 | F_RES, ACT_NONE>(a, batch, st);
    if (al && mk == F_MULZ && kd == ACT_STORED) return launch<128, 256, 1, 4, 3, F_MULZ, ACT_STORED>(a, batch, st);
    if (al && mk == (F_PREACT | F_ACT | F_PGRAD) && kd == ACT_QGELU) return launch<128, 256, 1, 4, 3, F_PREACT | F_ACT | F_PGRAD, ACT_QGELU>(a, batch, st);
    return launch<128, 256, 1, 4, 3>(a, batch, st);
  }
#endif
  // fast epilogue instantiations: full-width tiles, 16-byte aligned rows, no gathers / bf16 residual / debug modes
  const int tile = cfg == 256 ? 256 : 128;
  const bool aligned = (d->ldc % (d->c_f32 ? 4 : 8) == 0) && ((uintptr_t)d->preact % 16 == 0) && (d->stride_c % 8 == 0 || d->c_f32) &&
                       (d->ldc % 4 == 0) && (a.ldres % 4 == 0) && (a.ldc2 % 4 == 0) && (d->N % tile == 0) && !d->res_bf16 &&
                       !d->bias_by_brow && d->tile_cfg != 2561 && d->tile_cfg != 2564 &&
                      
                       ((uintptr_t)d->C % 16 == 0) && ((uintptr_t)d->res_f32 % 16 == 0) && ((uintptr_t)d->mulz % 8 == 0) &&
                       ((uintptr_t)d->preact % 8 == 0) && ((uintptr_t)d->C2 % 8 == 0) && ((uintptr_t)d->bias % 16 == 0) &&
                       ((uintptr_t)d->bias2 % 16 == 0) && (d->stride_c % 4 == 0) && (d->stride_aux % 4 == 0);
  const int dt = d->in_f16 ? (d->c_f16 ? 2 : 1) : 0;
  // fast-epilogue instantiations exist on the 4-phase loop only (and, in harness builds, on the 8-phase loop for A/B); a product call that needs the
  // 8-phase loop (loop8) takes its general-epilogue kernel
#ifdef MART_EXPERIMENTS
#define L256(...) (loop8 ? launch<256, 256, 2, 4, 2, __VA_ARGS__>(a, batch, st) : launch<256, 256, 2, 4, 4, __VA_ARGS__>(a, batch, st))
  const bool fast_ok = true;
#else
#define L256(...) launch<256, 256, 2, 4, 4, __VA_ARGS__>(a, batch, st)
  const bool fast_ok = !loop8;
#endif
  const bool two_acts = d->mulz && d->act != ACT_NONE;               // not a fast combination
  if (d->c_split3) {
    // C is bf16 [M, 3N]: hi at column n, lo at N + n, hi again at 2N + n -- the three 16-bit outputs of the packed epilogue
    MART_CHECK(!d->c_f32 && !d->preact && !d->C2 && !d->res_f32 && !d->res_bf16 && !d->mulz && !d->in_f16 && batch == 1 && aligned && cfg == 256 && d->ldc >= 3 * d->N && d->N % 8 == 0,
               "gemm_nt: c_split3 needs a bf16 C of >= 3N columns, N a multiple of the tile, 16-byte aligned rows and none of c_f32 / preact / C2 / residual / mulz / fp16 operands");
    a.preact = (bf16*)d->C + d->N; a.C2 = (bf16*)d->C + 2 * d->N; a.ldc2 = d->ldc;
    const int m3 = F_PREACT | F_C2 | F_SPLIT3 | (d->act != ACT_NONE ? F_ACT : 0);
    MART_CHECK(fast_ok, "gemm_nt: c_split3 is not available with b_blocked / operands beyond 2^31 elements");
    if (m3 == (F_PREACT | F_C2 | F_SPLIT3 | F_ACT) && d->act == ACT_QGELU) return L256(F_PREACT | F_C2 | F_SPLIT3 | F_ACT, ACT_QGELU);
    if (m3 == (F_PREACT | F_C2 | F_SPLIT3 | F_ACT) && d->act == ACT_GELU) return L256(F_PREACT | F_C2 | F_SPLIT3 | F_ACT, ACT_GELU);
    return L256(F_PREACT | F_C2 | F_SPLIT3, ACT_NONE);
  }
  const int mask = (two_acts ? (1 << 20) : 0) | (d->res_f32 ? F_RES : 0) | (d->mulz ? F_MULZ : 0) | (d->preact ? F_PREACT : 0) | (d->preact_grad ? F_PGRAD : 0) | (d->act != ACT_NONE ? F_ACT : 0) |
                   (d->c_f32 ? F_CF32 : 0) | (d->C2 ? F_C2 : 0) | (d->row_stats ? F_STATS : 0) | (d->ln_mean ? F_LNFOLD : 0);
  MART_CHECK(!d->ln_mean || (d->ln_rstd && d->ln_colsum && d->bias && !d->bias2 && !d->bias_by_brow), "gemm_nt: the LayerNorm fold needs ln_mean, ln_rstd, ln_colsum and the folded bias");
  MART_CHECK(!d->row_stats || (d->c_f32 && d->res_f32 && d->C2 && d->N % 64 == 0 && batch == 1), "gemm_nt: row_stats is an option of the f32 + residual + bf16-copy epilogue (N a multiple of 64)");
  // Start stagger for the f32-residual epilogues (out-proj, fc2: 512 KB of HBM traffic per tile against a 12-48 K-tile loop):
  // every CU runs the same loop, so all 256 reach their epilogue together and the memory side sees a 134 MB burst per round
  // while the matrix cores idle.  Groups of workgroups that share an A panel start 3 us apart (8 phases): +15 % on the
  // out-proj shape, +2-3 % on fc2 (tools/nt_harness time 5 0 70300); only where there are rounds enough to pay for the delay.
  if (d->res_f32 && d->c_f32 && cfg == 256 && a.stagger_ticks == 0 && t256 >= 4 * 256 && batch == 1) a.stagger_ticks = 300;
  if (aligned && fast_ok) {
    // persistent loop: +6-7 % where the epilogue is light (bf16 out); with the fp32 residual or two bf16 outputs it is
    // neutral at best (re-measured after the epilogue work: fc1 0.571 vs 0.572 ms, step +0.3 %) -> only for the light masks
#ifdef MART_EXPERIMENTS
    const bool persist = d->tile_cfg != 2562 && (mask == 0 || mask == F_MULZ) && (a.dbg == 0 || stamp_persist);
#else
    const bool persist = d->tile_cfg != 2562 && (mask == 0 || mask == F_MULZ || mask == F_LNFOLD) && a.dbg == 0;
#endif
#ifdef MART_EXPERIMENTS
#define MART_FAST(M_, K_)                                                                   \
    if (dt == 0 && mask == (M_) && kind == (K_)) {                                                       \
      if (tile == 256 && !old_loop) return persist ? L256((M_), (K_), true) : L256((M_), (K_)); \
      if (persist) return tile == 256 ? launch<256, 256, 2, 4, 0, (M_), (K_), true>(a, batch, st) : launch<128, 128, 2, 2, 0, (M_), (K_), true>(a, batch, st); \
      return tile == 256 ? launch<256, 256, 2, 4, 0, (M_), (K_)>(a, batch, st) : launch<128, 128, 2, 2, 0, (M_), (K_)>(a, batch, st); }
#else
#define MART_FAST(M_, K_)                                                                   \
    if (dt == 0 && mask == (M_) && kind == (K_)) {                                                       \
      if (tile == 256) return persist ? L256((M_), (K_), true) : L256((M_), (K_)); \
      return persist ? launch<128, 128, 2, 2, 0, (M_), (K_), true>(a, batch, st) : launch<128, 128, 2, 2, 0, (M_), (K_)>(a, batch, st); }
#endif
    const int kind = d->mulz ? d->mul_act : d->act;
    MART_FAST(0, ACT_NONE)                        // bf16 out (+bias): QKV, data gradients
    MART_FAST(F_CF32, ACT_NONE)                   // f32 out: scores, head
    MART_FAST(F_CF32 | F_RES, ACT_NONE)           // residual stream: out-proj, fc2
    MART_FAST(F_CF32 | F_RES | F_C2, ACT_NONE)    // ... with a bf16 copy for the fusion layers
    MART_FAST(F_MULZ, ACT_STORED)                 // data gradient through the activation: * act'(z) saved by the forward pass
    MART_FAST(F_PREACT | F_ACT | F_PGRAD, ACT_QGELU)   // fc1 / intermediate: act(z) and act'(z)
    MART_FAST(F_PREACT | F_ACT | F_PGRAD, ACT_GELU)
    MART_FAST(F_ACT, ACT_QGELU)                   // ... activation only (forward under no_grad)
    MART_FAST(F_ACT, ACT_GELU)
    MART_FAST(F_PREACT | F_ACT, ACT_QGELU)        // ... z itself (callers that differentiate later with mul_act = the activation)
    MART_FAST(F_MULZ, ACT_QGELU)
    MART_FAST(F_CF32 | F_ACT, ACT_GELU)           // head transform / precise-path GELU (f32 out)
    MART_FAST(F_CF32 | F_ACT, ACT_QGELU)
    MART_FAST(F_CF32 | F_RES | F_C2 | F_STATS, ACT_NONE)   // LayerNorm fold, producer: residual stream + its bf16 copy + per-row partial sums
    MART_FAST(F_LNFOLD, ACT_NONE)                  // LayerNorm fold, consumer: QKV on the bf16 copy of x and the folded weight
    MART_FAST(F_ACT | F_LNFOLD, ACT_QGELU)         // ... fc1 (forward passes that keep nothing for a backward pass)
#undef MART_FAST
    // fp16 operands: the forward linear layers of the text stream (engine.text_f16)
#define MART_FAST_H(M_, K_, DT_, P_)                                                           \
    if (dt == (DT_) && mask == (M_) && kind == (K_)) {                                         \
      if (tile == 256) return L256((M_), (K_), (P_), (DT_)); \
      return launch<128, 128, 2, 2, 0, (M_), (K_), (P_), (DT_)>(a, batch, st); }
    MART_FAST_H(0, ACT_NONE, 1, true)                                 // Q/K/V (bf16 out, the attention kernels' operand type)
    MART_FAST_H(F_CF32, ACT_NONE, 1, false)                           // attention.output.dense / output.dense: f32 into the LayerNorm
    MART_FAST_H(F_CF32 | F_RES, ACT_NONE, 1, false)                   // ... of the pre-LN blocks (FLAVA): + the f32 residual stream
    MART_FAST_H(F_CF32 | F_ACT, ACT_GELU, 1, false)                   // head transform
    MART_FAST_H(F_PREACT | F_ACT | F_PGRAD | F_C2, ACT_GELU, 2, false)   // intermediate: fp16 GELU output + its bf16 copy + act'(z)
    MART_FAST_H(F_ACT, ACT_GELU, 2, false)                            // ... under no_grad
#undef MART_FAST_H
  }
  MART_CHECK((mask & (F_STATS | F_LNFOLD)) == 0, "gemm_nt: row_stats / the LayerNorm fold exist for full-width, 16-byte aligned bf16 products only (out-proj / fc2 -> QKV / fc1 of the vision stream)");
  // general epilogue: both loops exist in every build
#define G256(DT_) (loop8 ? launch<256, 256, 2, 4, 2, -1, 0, false, (DT_)>(a, batch, st) : launch<256, 256, 2, 4, 4, -1, 0, false, (DT_)>(a, batch, st))
  if (dt == 1) return (cfg == 256 || cfg == 2561) ? G256(1) : launch<128, 128, 2, 2, 0, -1, 0, false, 1>(a, batch, st);
  if (dt == 2) return (cfg == 256 || cfg == 2561) ? G256(2) : launch<128, 128, 2, 2, 0, -1, 0, false, 2>(a, batch, st);
  if ((cfg == 256 || cfg == 2561) && !old_loop) return G256(0);
#undef G256
#undef L256
#ifdef MART_EXPERIMENTS
  if (cfg == 256 || cfg == 2561) return launch<256, 256, 2, 4>(a, batch, st);
#endif
  return launch<128, 128, 2, 2>(a, batch, st);
}

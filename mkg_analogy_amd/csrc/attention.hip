// Flash-style multi-head attention for head_dim 64 on gfx950 (forward, dQ pass, dK/dV pass).
//
// One skeleton, three kernels.  A wave OWNS 32 rows (queries in fwd/dQ, keys in dK/dV) whose fragments stay in
// registers as MFMA B operands, so every MFMA result has "lane = owner row": online-softmax state, rescaling and
// the final stores are lane-local.  The other side is STREAMED in 64-row tiles through a double-buffered LDS
// ring filled by LDS-DMA (global_load_lds); first-GEMM fragments are plain ds_read_b128 on an XOR-swizzled
// image, second-GEMM fragments (contraction over the streamed rows) use the LDS transpose read.
//
//  vision (CLIPAttention, modeling_unimo.py:212-272): softmax(q k^T * dh^-0.5), no mask, keys = [text prefix | own]
//  text   (BertSelfAttention, :317-377): scores/8 -> adaptive analogy reweight (:342-349) -> + (1-mask)*-1e4
//          (:355,:55-56) -> softmax -> dropout(p) on the probabilities (:362)
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "mart_hip.h"

namespace {

#ifndef ATTN_PK
#define ATTN_PK 1                        // packed f32x2 softmax arithmetic (0: one scalar VALU instruction per score)
#endif
constexpr int NTH = 256;                 // 4 waves
constexpr float LOG2E = 1.4426950408889634f;
constexpr int TILE_BYTES = 64 * 128;     // 64 rows x 64 bf16
constexpr int STAGE_BYTES = 2 * TILE_BYTES + 512;   // two tiles + 2x64 floats (lse, delta)
constexpr int LDS_BYTES = 2 * STAGE_BYTES;

// Cycle stamps of ONE workgroup (variant builds only: tools/build_variant.sh attention.hip <out.so> -DATTN_STAMPS; tools/attn_stamps.py): [kernel 0 fwd / 1 fused bwd][wave][stamp]
#ifdef ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[2 * 8 * 32];
__device__ __forceinline__ unsigned long long attn_memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");   // the wait belongs INSIDE: the result lands asynchronously
  return t;
}
#define ATTN_STAMP(kern, on, k) do { if ((on) && lane == 0 && (k) < 32) g_attn_stamps[((kern) * 8 + wave) * 32 + (k)] = attn_memtime(); } while (0)
#else
#define ATTN_STAMP(kern, on, k) do { } while (0)
#endif

struct Side {                            // a [rows, nh*64] bf16 view with an optional prefix block in front
  const bf16* own; int ld_own; int n_own;          // rows of this batch element: own[(b*n_own + j)*ld_own + h*64 + d]
  const bf16* pre; int ld_pre; int n_pre;
};
__device__ __forceinline__ const bf16* side_row(const Side& s, int b, int h, int j) {
  // j in [0, n_pre + n_own); clamps to the last valid row
  j = min(j, s.n_pre + s.n_own - 1);
  if (j < s.n_pre) return s.pre + ((long long)b * s.n_pre + j) * s.ld_pre + h * 64;
  return s.own + ((long long)b * s.n_own + (j - s.n_pre)) * s.ld_own + h * 64;
}

// Workgroup -> (part, head, batch).  The gridDim.x workgroups of one (batch, head) pair stream the SAME K/V (or Q/dO) rows.
// Workgroups are dealt to the 8 XCDs round-robin by linear id, so with the natural order (x fastest) the parts of a pair
// land on different XCDs and each pulls the rows through its own L2.  Remapped in groups of 8 * gridDim.x ids: ids j,
// 8 + j, 16 + j, ... of a group (same XCD, dispatched together) are the parts of ONE pair -- the second reader hits L2.
struct WgId { int part, h, b; };
__device__ __forceinline__ WgId wg_id(int nh, int B) {
  // Division-free (round 5; the 64-bit quotient / remainder forms were ~320 scalar instructions at the head of every wave).  With
  // P = y + nh z the pair index in launch order, L = x + G P, and x < G:  L / (8 G) = P >> 3 and L mod (8 G) = G (P & 7) + x.
  const int G = gridDim.x;
  const int x = blockIdx.x, y = blockIdx.y, z = blockIdx.z;
  const unsigned pairs = (unsigned)nh * (unsigned)B;
  if (G == 1 || (pairs & 7u) != 0) return WgId{x, y, z};
  const unsigned P = (unsigned)y + (unsigned)nh * (unsigned)z;
  const int c = (int)(P & 7u), r = G * c + x;
  // pair = 8 (P >> 3) + (r & 7) = P + ((r & 7) - c): the head / batch indices move by less than 8 positions from (y, z)
  int hh = y + ((r & 7) - c), bb = z;
  while (hh < 0) { hh += nh; --bb; }
  while (hh >= nh) { hh -= nh; ++bb; }
  return WgId{r >> 3, hh, bb};
}

// Loop-top barrier of the double-buffered tile loops.  The explicit vmcnt(0) is REQUIRED: the LDS-DMA of the tile about
// to be read was issued by all four waves, and waves that skip the compute body (query / key rows past the end) would
// otherwise reach the barrier with their quarter of the tile still in flight -- the compiler only places its own wait
// in front of the LDS reads of the compute body, not in front of s_barrier.  (Seen as rare NaNs in dQ of the last,
// partial query tile at B=256.)
__device__ __forceinline__ void tile_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

// Swizzle key of a 128-byte tile row: the 16-byte chunk index is XORed with the BIT-REVERSED pair index (row >> 1) & 7.
// Plain fragment reads (ds_read_b128, lane = row) only need the key to be a bijection of the pair index; the transposed reads
// (ds_read_b64_tr_b16: 4 consecutive rows x two 16-column halves per 32-lane pass) need rows r and r + 2 in different 64-byte
// halves of the row -- with the plain key (row >> 1) & 7 they differed by ONE chunk, inside the same 32 bytes: a 2-way conflict
// on every transposed read (SQ_LDS_BANK_CONFLICT = 25 % of the LDS cycles of the dK/dV pass in round 1).
__device__ __forceinline__ int swz_key(int row) {
  const int k = (row >> 1) & 7;
  return ((k & 1) << 2) | (k & 2) | (k >> 2);
}
// stage one 64x64 tile (rows r0..r0+63 of `s`) into LDS with the chunk swizzle c' = c ^ swz_key(row)
// Each thread copies chunk (row = tid>>3 [+32], lc) of both 32-row halves.  When a half lies entirely inside the prefix
// or inside the own rows (prefix length a multiple of 32, half not cut by the end) its source is a wave-uniform row
// pointer (scalar unit) plus ONE loop-invariant lane offset; the general form (per-lane clamp, prefix/own select and a
// 64-bit multiply: ~28 VALU instructions per copy, four copies per tile iteration) is left to the cut halves.
// (Measured and dropped, twice each: the copy as opaque assembly -- no compiler-forced vmcnt(0) in front of the
// transposed reads -- is 2 % slower; a 4-stage ring with three tiles in flight and counted waits changes nothing:
// the loop does not wait for the DMA, see docs/LAB_r01-r05.md section 4.2.)
#ifndef STAGE_RAW
#define STAGE_RAW 0
#endif
__device__ __forceinline__ void stage16(const void* src, void* dst) {
  if (STAGE_RAW) glds16_raw(src, dst); else glds16(src, dst);
}
__device__ __forceinline__ void stage_tile(const Side& s, int b, int h, int r0, char* lds, int tid, int wave) {
  const int rowh = tid >> 3, pc = tid & 7, lc = pc ^ swz_key(rowh);               // the key is the same for row and row + 32
  const unsigned off_own = (unsigned)(rowh * s.ld_own + lc * 8), off_pre = (unsigned)(rowh * s.ld_pre + lc * 8);
  const int n_tot = s.n_pre + s.n_own;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int j0 = r0 + r * 32;                                                       // first row of this half (wave-uniform)
    char* dst = lds + (r * NTH + wave * 64) * 16;
    if ((s.n_pre & 31) == 0 && j0 + 32 <= n_tot) {
      if (j0 < s.n_pre) stage16(s.pre + ((long long)b * s.n_pre + j0) * s.ld_pre + h * 64 + off_pre, dst);
      else stage16(s.own + ((long long)b * s.n_own + (j0 - s.n_pre)) * s.ld_own + h * 64 + off_own, dst);
    } else {
      stage16(side_row(s, b, h, j0 + rowh) + lc * 8, dst);
    }
  }
}
// ---- strength-reduced staging for the vision kernels (round 3).  The general stage_tile above compiles to ~20 VALU + ~25 SALU + 6 branches
// per 1-KB copy (per-lane clamp, prefix / own select and a 64-bit multiply, if-converted into vector code): four copies per tile were a third
// of the forward kernel's VALU and most of its SALU instructions (16.5 VALU + 7.3 SALU per MFMA, profiles/r03_pmc_attn_fwd.txt).  Here the
// source of a 32-row half is a WAVE-UNIFORM pointer (scalar unit: prefix or own block, rows j0 ..) plus ONE loop-invariant 32-bit lane offset,
// issued in the saddr form; only a half that is cut by the last key (per-lane clamp) or that straddles the prefix / own boundary (prefix length not a multiple of 32: per-lane select) leaves it.
struct Stager {
  const bf16* own_b; const bf16* pre_b;   // first row of this (batch, head) in the own / prefix block
  unsigned voff_own, voff_pre;            // byte offset of this lane's chunk inside a 32-row half
  int ld_own, ld_pre, n_own, n_pre;
};
__device__ __forceinline__ Stager make_stager(const Side& s, int b, int h, int tid) {
  const int rowh = tid >> 3, pc = tid & 7, lc = pc ^ swz_key(rowh);
  Stager g;
  g.own_b = s.own + ((long long)b * s.n_own) * s.ld_own + h * 64;
  g.pre_b = s.n_pre ? s.pre + ((long long)b * s.n_pre) * s.ld_pre + h * 64 : s.own;
  g.voff_own = (unsigned)(rowh * s.ld_own + lc * 8) * 2u;
  g.voff_pre = (unsigned)(rowh * s.ld_pre + lc * 8) * 2u;
  g.ld_own = s.ld_own; g.ld_pre = s.ld_pre; g.n_own = s.n_own; g.n_pre = s.n_pre;
  return g;
}
__device__ __forceinline__ void dma_saddr(const bf16* sbase, unsigned voff, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)LDS_PTR(lds_wave_base));
  const unsigned long long a = (unsigned long long)(__UINTPTR_TYPE__)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sb) : "memory", "m0");
}
__device__ __forceinline__ void stage_tile_fast(const Stager& g, int r0, char* lds, int tid, int wave) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int j0 = r0 + r * 32;                                                       // first row of this half (wave-uniform)
    char* dst = lds + (r * NTH + wave * 64) * 16;
    if (j0 + 32 <= g.n_pre) {
      dma_saddr(g.pre_b + (long long)j0 * g.ld_pre, g.voff_pre, dst);
    } else if (j0 < g.n_pre) {
      // the ONE half that straddles the prefix / own boundary (prefix length not a multiple of 32: text rows as long as the batch's longest example,
      // data_module.py:113-119): per-lane source select, general 64-bit address form
      const int rowh = tid >> 3, pc = tid & 7, lc = pc ^ swz_key(rowh), j = j0 + rowh;
      const bf16* src = j < g.n_pre ? g.pre_b + (long long)j * g.ld_pre : g.own_b + (long long)min(j - g.n_pre, g.n_own - 1) * g.ld_own;
      glds16(src + lc * 8, dst);
    } else {
      const int jj = j0 - g.n_pre;
      if (jj + 32 <= g.n_own) {
        dma_saddr(g.own_b + (long long)jj * g.ld_own, g.voff_own, dst);
      } else {                                                                        // half cut by (or past) the last key: clamped rows
        const int rowh = tid >> 3, pc = tid & 7, lc = pc ^ swz_key(rowh);
        const int row = min(jj + rowh, g.n_own - 1);
        dma_saddr(g.own_b, (unsigned)(row * g.ld_own + lc * 8) * 2u, dst);
      }
    }
  }
}
// A tile whose 64 rows all exist (every tile but possibly the last): both halves are wave-uniform, the prefix / own choice is a scalar select --
// no branch per copy (the branchy general form above was ~100 scalar instructions and a dozen taken / not-taken branches per tile iteration).
__device__ __forceinline__ void stage_tile_full(const Stager& g, int r0, char* lds, int wave) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int j0 = r0 + r * 32;
    const bool pre = j0 < g.n_pre;                                                    // wave-uniform; n_pre is a multiple of 32
    const bf16* base = pre ? g.pre_b + (long long)j0 * g.ld_pre : g.own_b + (long long)(j0 - g.n_pre) * g.ld_own;
    dma_saddr(base, pre ? g.voff_pre : g.voff_own, lds + (r * NTH + wave * 64) * 16);
  }
}
// ---- round 6: the same copies with the row offset folded into the LANE offset.  The ISA of the forms above was ~120 VALU + ~330 SALU per key tile and
// wave (64-bit row multiplies, prefix / own selects, three v_readfirstlane per copy, branch ladders) -- more instructions than the tile's softmax.  Here a
// copy of a 32-row half that lies inside one block is: one scalar multiply (row * bytes per row), one v_add (lane offset + that), s_mov m0, the DMA --
// against FOUR loop-invariant scalar bases (K / V x own / prefix block of this (batch, head)).  Offsets are 32-bit byte offsets inside the block
// (host-checked: rows * ld * 2 < 2^32).  Halves cut by the last key or straddling the prefix / own boundary take the per-lane form.
struct Lean {
  const char* own_b; const char* pre_b;   // row 0 of this (batch, head) in the own / prefix block (wave-uniform)
  unsigned vo_own, vo_pre;                // byte offset of this lane's chunk inside a 32-row half
  unsigned rb_own, rb_pre;                // bytes per row
  int n_own, n_pre;
};
__device__ __forceinline__ Lean make_lean(const Side& s, int b, int h, int tid) {
  const int rowh = tid >> 3, pc = tid & 7, lc = pc ^ swz_key(rowh);
  Lean g;
  g.own_b = (const char*)(s.own + ((long long)b * s.n_own) * s.ld_own + h * 64);
  g.pre_b = s.n_pre ? (const char*)(s.pre + ((long long)b * s.n_pre) * s.ld_pre + h * 64) : g.own_b;
  g.rb_own = (unsigned)s.ld_own * 2u; g.rb_pre = (unsigned)s.ld_pre * 2u;
  g.vo_own = (unsigned)rowh * g.rb_own + (unsigned)lc * 16u;
  g.vo_pre = (unsigned)rowh * g.rb_pre + (unsigned)lc * 16u;
  g.n_own = s.n_own; g.n_pre = s.n_pre;
  return g;
}
__device__ __forceinline__ void dma_lean(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory", "m0");
}
// one 32-row half (rows j0 .. j0 + 31 of the key stream [prefix | own]) -> 4 KB at LDS byte address lds_dst (this wave's 1 KB slice: + wave * 1024 by the caller)
__device__ __forceinline__ void stage_half_lean(const Lean& g, int j0, unsigned lds_dst, int tid) {
  const int jj = j0 - g.n_pre;
  if (j0 + 32 <= g.n_pre) {
    dma_lean(g.pre_b, g.vo_pre + (unsigned)j0 * g.rb_pre, lds_dst);
  } else if (jj >= 0 && jj + 32 <= g.n_own) {
    dma_lean(g.own_b, g.vo_own + (unsigned)jj * g.rb_own, lds_dst);
  } else {                                                    // cut by the last key (clamped rows) or straddling the prefix / own boundary: per-lane row
    const int rowh = tid >> 3, pc = tid & 7, lc = pc ^ swz_key(rowh), j = j0 + rowh;
    if (j < g.n_pre) dma_lean(g.pre_b, (unsigned)j * g.rb_pre + (unsigned)lc * 16u, lds_dst);
    else dma_lean(g.own_b, (unsigned)min(j - g.n_pre, g.n_own - 1) * g.rb_own + (unsigned)lc * 16u, lds_dst);
  }
}
// Per-lane byte offsets of the fragment reads inside a 64x64 tile -- loop invariant, computed once per kernel so the
// tile loops carry no address arithmetic (the first version spent 36 VALU instructions per MFMA, mostly on this).
struct LaneOffs {
  int pf[4];        // plain fragment, k-step ks: row l31 (+32 t), chunk (2ks+hh) ^ key(row); key(32t + l31) == key(l31)
  int tr1[2], tr2[2];   // transposed fragment, dim tile dt: rows rr and rr+8 (+ base), rr = 4hh + (pp>>2)
};
__device__ __forceinline__ LaneOffs make_offs(int lane) {
  LaneOffs o;
  const int l31 = lane & 31, hh = lane >> 5, pp = lane & 15, g1 = (lane >> 4) & 1;
  const int key = swz_key(l31);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) o.pf[ks] = l31 * 128 + (((ks * 2 + hh) ^ key) << 4);
  const int rr = 4 * hh + (pp >> 2), k1 = swz_key(rr), k2 = swz_key(rr + 8);   // keys repeat every 16 rows
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int col = dt * 32 + g1 * 16 + (pp & 3) * 4, lc = col >> 3, bo = (col & 7) * 2;
    o.tr1[dt] = rr * 128 + ((lc ^ k1) << 4) + bo;
    o.tr2[dt] = (rr + 8) * 128 + ((lc ^ k2) << 4) + bo;
  }
  return o;
}
// plain fragment: rows t*32 + l31, k-step ks (16 of the 64 head dims)
__device__ __forceinline__ bf16x8 tile_frag(const char* lds, int t, int ks, const LaneOffs& o) {
  return *(const bf16x8*)(lds + t * 4096 + o.pf[ks]);
}
// transposed fragment for the second GEMM: A[i = dim dt*32 + l31][k = 8 streamed rows]; k order matches the
// register order of an MFMA result tile: e<4 -> row base+4hh+e, e>=4 -> row base+8+4hh+(e-4); base multiple of 16
__device__ __forceinline__ bf16x8 tile_frag_tr(const char* lds, int base, int dt, const LaneOffs& o) {
  s16x4 lo = lds_tr_read(lds + base * 128 + o.tr1[dt]);
  s16x4 hi = lds_tr_read(lds + base * 128 + o.tr2[dt]);
  return join_tr(lo, hi);
}
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
  return o;
}

struct TextCtl {                          // per-(b) text controls, all optional
  const int64_t* mask_row;                // attention_mask row of this batch element or NULL
  int sep;                                // reweight split, <0 = off
  int skip0;                              // FLAVA: query row 0 keeps factor 1
  float c0, c1;
  float p_drop, inv_keep; uint64_t seed;
  uint32_t s2, thr;                       // dropout: folded seed and 16-bit threshold (common.h: dropout_keep32)
};
__device__ __forceinline__ TextCtl make_ctl(const mart_attn_fwd_desc& p, int b, int Sk) {
  TextCtl c;
  c.mask_row = p.attn_mask ? p.attn_mask + (long long)b * Sk : nullptr;
  c.sep = p.sep ? (int)p.sep[(long long)b * p.sep_stride] : -1;
  c.skip0 = p.rw_skip_row0;
  c.c0 = p.w0 ? fminf(fmaxf(p.w0[0], 0.f), 0.5f) : 1.f;
  c.c1 = p.w1 ? fminf(fmaxf(p.w1[0], 0.5f), 1.f) : 1.f;
  c.p_drop = p.p_drop; c.inv_keep = 1.f / (1.f - p.p_drop); c.seed = p.seed;
  c.s2 = rng_seedmix(p.seed, 0); c.thr = dropout_thr16(p.p_drop);
  return c;
}
__device__ __forceinline__ float reweight(const TextCtl& c, int qi, int kj) {
  if (c.sep < 0 || kj < c.sep) return 1.f;
  if (qi >= c.sep) return c.c1;
  return (c.skip0 && qi == 0) ? 1.f : c.c0;
}

// Query-owner kernels (forward, dQ): everything about the adaptive reweight that depends on the QUERY is lane-constant -- the factor this
// query applies to keys >= sep and the two selectors of the d(w0) / d(w1) sums -- so a score costs one compare + select instead of the
// branches of reweight() (the per-score divergent `if`s compiled to ~80 s_and_saveexec in the text dQ kernel).
struct QueryRw { float fq, sel0, sel1; int sep; };
__device__ __forceinline__ QueryRw make_qrw(const TextCtl& c, int qi) {
  QueryRw q;
  q.sep = c.sep < 0 ? 0x7fffffff : c.sep;                    // no reweight: no key is >= sep
  const bool hi = qi >= q.sep, lo_ok = !hi && !(c.skip0 && qi == 0);
  q.fq = hi ? c.c1 : (lo_ok ? c.c0 : 1.f);
  q.sel1 = hi ? 1.f : 0.f; q.sel0 = lo_ok ? 1.f : 0.f;
  return q;
}
// Key-padding mask of one 64-key tile as a wave-uniform bit set (bit j = key kt*64 + j is attended; keys past the end: 1, they are handled by
// the caller).  Lane j loads mask[kt*64 + j] once (one coalesced 512-byte request); `mask_row[kj]` per score -- 32 loads per lane whose address
// depends on the lane only through one bit -- had been scalarised by the compiler into waterfall loops of scalar loads (86 v_readlane + 160
// scalar multiplies in the text forward kernel) and made the 64 x 64 text attention kernels latency-bound on them.  The mask holds 0 / 1
// (extended mask (1 - m) * -10000, modeling_unimo.py:55-56): any non-zero value counts as 1.
__device__ __forceinline__ uint64_t mask_bits(const TextCtl& c, int kt, int Stot, int lane) {
  if (!c.mask_row) return ~0ull;
  const int kl = kt * 64 + lane;
  const long long mv = kl < Stot ? c.mask_row[kl] : 1;
  return __ballot(mv != 0);
}
// additive mask term of key (t*32 + row-in-block) for this lane: hsel = the lane's half select (mfma_row adds 4 * hh)
__device__ __forceinline__ float mask_add(uint64_t mb_lane, int c) { return ((mb_lane >> c) & 1ull) ? 0.f : -10000.0f; }

// =========================================================================== forward
// TPW = query tiles (32 rows) per wave.  With TPW = 2 a wave carries two independent softmax / accumulator chains, so the
// exp-heavy VALU work of one tile overlaps the MFMAs of the other inside the wave, a workgroup covers 256 query rows, and
// the K/V stream of a head is read by half as many workgroups.
// Vision forward, round 3: ONE query tile per wave compiled for FOUR waves per SIMD (128 VGPRs, no spills) beats two tiles per wave at two
// waves per SIMD (256 VGPRs, 6 spills) by 6-7 % (0.263 vs 0.283 ms at 393 keys, tools/ab_attn2.sh): twice the K/V streams, but four
// independent dependence chains per SIMD instead of two hide more of the MFMA -> exp -> cvt -> MFMA latency.  (One tile per wave at 168 VGPRs
// = three waves per SIMD: 0.275-0.279.)  The text instantiation (mask, reweight, dropout state) would spill at 128 and stays at two.
#ifndef ATTN_FWD_MINW1
#define ATTN_FWD_MINW1 4
#endif
// RES (vision, round 4; an experiment, compiled only with -DMART_EXPERIMENTS -- measured 40 % SLOWER, docs/LAB_r01-r05.md section 4.2): the K / V of the whole head RESIDENT in LDS.  One 16-wave workgroup per (batch, head): every thread brings in one 16-byte
// chunk of every 64-key tile of K (waves 0-7) or V (waves 8-15) with one LDS-DMA instruction per tile -- the head is staged ONCE instead of once
// per 128-query part (four times at 393 queries, the fourth for 9 rows), by straight-line address code in a prologue instead of the branchy
// per-tile staging of the ring -- then ONE barrier, and every wave walks the image with its 32 queries at its own pace: no barrier, no DMA wait and
// no staging instruction in the tile loop.  Same 128 VGPRs / four waves per SIMD as the ring kernel.  LDS: 16 KB per 64 keys (112 KB at 393 keys,
// 128 KB at 457: one workgroup per CU).
constexpr int RES_NTH = 1024;
#ifndef FWD_NW
#define FWD_NW 4                         // waves per workgroup of the vision ring kernel (one query tile each); waves 0-3 stage the K / V tiles
#endif
constexpr int fwd_waves(bool text, int tpw, bool res) { return (!text && tpw == 1 && !res) ? FWD_NW : 4; }
template <bool TEXT, int TPW, bool RES = false>
__global__ __launch_bounds__(RES ? RES_NTH : 64 * fwd_waves(TEXT, TPW, RES), RES ? 1 : ((TPW == 1 && !TEXT) ? ATTN_FWD_MINW1 : 2)) void attn_fwd_k(mart_attn_fwd_desc p) {
  constexpr int NWV = fwd_waves(TEXT, TPW, RES);
  static_assert(!RES || (TPW == 1 && !TEXT), "the resident form is the vision kernel, one query tile per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WgId wg = RES ? WgId{0, (int)blockIdx.x, (int)blockIdx.y} : wg_id(p.nh, p.B);
  const int b = wg.b, h = wg.h;
  const int Stot = p.Lp + p.Sk;
  const Side K{(const bf16*)p.k, p.ldk, p.Sk, (const bf16*)p.pk, p.ldp, p.Lp};
  const Side V{(const bf16*)p.v, p.ldv, p.Sk, (const bf16*)p.pv, p.ldp, p.Lp};
  const TextCtl ctl = make_ctl(p, b, p.Sk);
  constexpr bool text = TEXT;          // vision instantiation: no mask / reweight / dropout code at all
  const bool stamp_on = !TEXT && !RES && b == p.B / 2 && h == 5 && wg.part == 1;
  ATTN_STAMP(0, stamp_on, 0);
  const LaneOffs lo = make_offs(lane);

  int q0[TPW], qi[TPW];
  bool active[TPW];
  bf16x8 qf[TPW][4];
  f32x16 ot[TPW][2];
  float m_run[TPW], l_run[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    q0[u] = RES ? wave * 32 : wg.part * (32 * NWV * TPW) + (wave * TPW + u) * 32;
    qi[u] = q0[u] + l31;
    active[u] = q0[u] < p.Sq;                          // wave-uniform
    const bf16* qp = (const bf16*)p.q + ((long long)b * p.Sq + min(qi[u], p.Sq - 1)) * p.ldq + h * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[u][ks] = *(const bf16x8*)(qp + ks * 16 + hh * 8);
#pragma unroll
    for (int r = 0; r < 16; ++r) { ot[u][0][r] = 0.f; ot[u][1][r] = 0.f; }
    m_run[u] = -1.0e30f; l_run[u] = 0.f;
  }
  const float c2 = p.scale * LOG2E;

  const int ntiles = (Stot + 63) / 64;
  Stager gK, gV;
  if constexpr (!TEXT) { gK = make_stager(K, b, h, tid); gV = make_stager(V, b, h, tid); }
  // vision: uniform-pointer staging; a prefix length that is not a multiple of 32 has ONE 32-row half that straddles the prefix / own boundary
  // (index strad): its tile takes the per-half form, the half itself a per-lane select (stage_tile_fast)
  const int strad = (p.Lp & 31) ? (p.Lp >> 5) : -1;
#ifndef FWD_STAGE_FULL
#define FWD_STAGE_FULL 1
#endif
#ifndef FWD_LEAN_STAGE
#define FWD_LEAN_STAGE 1
#endif
  Lean lK, lV;
  unsigned lds_w = 0;                                  // LDS byte address of this wave's 1 KB slice of K half 0 of ring stage 0
  if constexpr (!TEXT && !RES && FWD_LEAN_STAGE) {
    lK = make_lean(K, b, h, tid); lV = make_lean(V, b, h, tid);
    lds_w = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)LDS_PTR(smem)) + (unsigned)wave * 1024u;
  }
#define stage_kv(r0_, buf_)                                                                           \
  do {                                                                                                 \
    if constexpr (!TEXT && !RES && FWD_LEAN_STAGE) {                                                   \
      const unsigned d0_ = lds_w + (unsigned)((buf_) - smem);                                          \
      stage_half_lean(lK, (r0_), d0_, tid);                                                            \
      stage_half_lean(lK, (r0_) + 32, d0_ + 4096u, tid);                                               \
      stage_half_lean(lV, (r0_), d0_ + (unsigned)TILE_BYTES, tid);                                     \
      stage_half_lean(lV, (r0_) + 32, d0_ + (unsigned)TILE_BYTES + 4096u, tid);                        \
    } else if constexpr (!TEXT) {                                                                      \
      if (FWD_STAGE_FULL && (r0_) + 64 <= Stot && ((r0_) >> 5) != strad && ((r0_) >> 5) + 1 != strad) {  \
        stage_tile_full(gK, (r0_), (buf_), wave);                                                      \
        stage_tile_full(gV, (r0_), (buf_) + TILE_BYTES, wave);                                         \
      } else {                                                                                         \
        stage_tile_fast(gK, (r0_), (buf_), tid, wave);                                                 \
        stage_tile_fast(gV, (r0_), (buf_) + TILE_BYTES, tid, wave);                                    \
      }                                                                                                \
    } else {                                                                                           \
      stage_tile(K, b, h, (r0_), (buf_), tid, wave);                                                   \
      stage_tile(V, b, h, (r0_), (buf_) + TILE_BYTES, tid, wave);                                      \
    }                                                                                                  \
  } while (0)
  if constexpr (RES) {
    const bool isv = wave >= 8;                        // wave-uniform: waves 0-7 bring in K, waves 8-15 V
    const Side& S = isv ? V : K;
    const int c = tid & 511, row = c >> 3, lc = (c & 7) ^ swz_key(row);
    char* img = smem + (isv ? ntiles * TILE_BYTES : 0) + ((wave & 7) * 64) * 16;
    for (int kt = 0; kt < ntiles; ++kt) glds16(side_row(S, b, h, kt * 64 + row) + lc * 8, img + kt * TILE_BYTES);   // rows past the last key: clamped copies (finite)
    tile_barrier();
  } else {
    if (NWV == 4 || wave < 4) stage_kv(0, smem);
  }
  // The Q fragments (plain global loads, above) are first USED inside the loop, so the compiler put its own s_waitcnt vmcnt(3..0) for them in front
  // of the first four MFMAs of EVERY iteration -- and the hardware counter also counts the LDS-DMA of the next tile, issued (as opaque assembly) at the
  // top of the iteration: every S phase waited for the prefetch it had just started, i.e. the ring never ran ahead.  A compiler-visible wait here
  // retires the Q loads before the loop; the tile loop then carries no vmcnt wait but the one in tile_barrier().
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0), expcnt / lgkmcnt untouched
  ATTN_STAMP(0, stamp_on, 1);
#ifndef FWD_HALF_TAIL
#define FWD_HALF_TAIL 1
#endif
#ifndef FWD_NOMAX
#define FWD_NOMAX 1                                      // vision forward: probabilities against the running reference maximum, no per-tile maximum (round 6)
#endif
#ifndef FWD_PEEL
#define FWD_PEEL 1                                       // the last key tile peeled out of the loop (vision ring kernel)
#endif
  // One tile.  LAST (compile time, vision ring kernel with FWD_PEEL): the tile that may be cut by the last key -- the only one that carries the partial-tile
  // mask and the block guard NT (32-key blocks that hold keys: the last tile of the vision shapes, 393 = 6 x 64 + 9 and 457 = 7 x 64 + 9 keys, runs
  // as ONE block: half the MFMAs and exponentials of a tile that was 86 % padding).  The loop body proper is straight-line.
  // MODE (vision ring kernel with FWD_NOMAX): 0 = classic tile (tile maximum, new reference, rescale), 1 = fast tile (probabilities against the running
  // reference; returns false -- nothing changed -- when the wave vote finds a sum that says the reference is too low)
  auto tile = [&](const int kt, auto last_c, auto mode_c) -> bool {
    constexpr bool LAST = decltype(last_c)::value;
    constexpr int MODE = decltype(mode_c)::value;
    const int NT = (LAST && FWD_HALF_TAIL && (!TEXT && !RES) && kt * 64 + 32 >= Stot) ? 1 : 2;      // wave-uniform
    const char* sK = RES ? smem + kt * TILE_BYTES : smem + (kt & 1) * STAGE_BYTES;
    const char* sV = RES ? smem + (ntiles + kt) * TILE_BYTES : sK + TILE_BYTES;
    // S^T[key][q] = K q^T
    f32x16 st[TPW][2];
    float pv[TPW][2][16];
    float alpha[TPW];
    constexpr bool NOMAX = FWD_NOMAX && !TEXT && TPW == 1;
    if constexpr (NOMAX) {
      // Round 6 (vision ring kernel): NO per-tile maximum on the fast path.  The probabilities of a tile are formed against the running reference
      // maximum m_run at once (fma, exp2, add per score, in place of the scores) and the per-row partial sums -- needed anyway -- tell whether that
      // was legal: a lane's sum stays below 2^60 unless some score exceeds the reference by ~60 binary orders.  If one does (wave vote; never on
      // activations this network produces) the tile is REDONE in the classic form -- S recomputed from the K tile still in LDS, tile maximum, new
      // reference, O / l rescaled -- which is also how the first tile runs (m_run has no value yet).  The reference may lag the true row maximum by up
      // to 2^60: harmless -- P is a floating-point number (bf16 keeps its 8 bits at any magnitude), O and l are f32 sums bounded by 2^60 x keys x |V|, and
      // the saved statistic m + log2(l) does not depend on the reference.  Gone from every tile but the first: 22 v_max3 / v_max, a cross-lane
      // exchange, a wave vote IN FRONT of the exponentials (they used to wait for the whole maximum chain) and the threshold logic of the
      // deferred rescale (rounds 1-5 deferred only the rescale, threshold 2^8).
      auto scores = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) st[0][t][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) st[0][t] = mfma32(tile_frag(sK, t, ks, lo), qf[0][ks], st[0][t]);
        }
        if (LAST && kt * 64 + 64 > Stot) {                  // partial last key tile: mask in place; -1e30 * c2 - m underflows exp2 to exactly 0
          const int lim = Stot - kt * 64;
#pragma unroll
          for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (t * 32 + mfma_row(r, hh) >= lim) st[0][t][r] = -1.0e30f;
        }
      };
      auto probs = [&](const float m_ref) {                  // P = exp2(S c2 - m_ref) in the log2 domain, packed two scores per instruction; returns the lane's sum
        f32x2 rs2 = {0.f, 0.f};
        const f32x2 c22 = {c2, c2}, mn2 = {m_ref, m_ref};
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 x = f32x2{st[0][t][r], st[0][t][r + 1]} * c22 - mn2;
            const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            rs2 += e;
            st[0][t][r] = e[0]; st[0][t][r + 1] = e[1];          // in place: the scores are dead (the classic form recomputes them from the K tile)
          }
        return rs2[0] + rs2[1];
      };
      float rs = 0.f;
      alpha[0] = 1.f;
      scores();
      if constexpr (MODE == 1) {
        rs = probs(m_run[0]);
        if (!__all(rs < 1.152921504606846976e18f)) return false;    // 2^60; inf / NaN fail the compare as well.  Nothing of the wave's state was touched.
      } else {                                               // first tile, or the redo of a tile that failed the vote (and whatever follows it)
        float mx = st[0][0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[0][t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;
        const float m_new = fmaxf(m_run[0], mx);
        alpha[0] = __builtin_amdgcn_exp2f(m_run[0] - m_new);
        rs = probs(m_new);
        m_run[0] = m_new;
      }
      rs += __shfl_xor(rs, 32, 64);
      l_run[0] = l_run[0] * alpha[0] + rs;
      if (!__all(alpha[0] == 1.f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { ot[0][0][r] *= alpha[0]; ot[0][1][r] *= alpha[0]; }
      }
    } else {
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      if (u > 0 && !active[u]) continue;
#pragma unroll
      for (int t = 0; t < 2; ++t) if (t < NT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[u][t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st[u][t] = mfma32(tile_frag(sK, t, ks, lo), qf[u][ks], st[u][t]);
      }
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      if (u > 0 && !active[u]) continue;
      // online softmax in the log2 domain: m_run, the saved statistic and every exponent are base-2 (one v_exp_f32 each)
      float rs = 0.f;
      if constexpr (!TEXT) {                            // vision: fma, exp2, add per score (packed two at a time)
        if (LAST && kt * 64 + 64 > Stot) {
          // last (partial) key tile -- one in seven at 393 keys: mask in place (compare + select per score), then the
          // same fast path; -1e30 * c2 - m underflows exp2 to exactly 0
          const int lim = Stot - kt * 64;
#pragma unroll
          for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (t * 32 + mfma_row(r, hh) >= lim) st[u][t][r] = -1.0e30f;
        }
        float mx = st[u][0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[u][t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;
        // deferred rescale: while no row's maximum grew by more than 2^8 keep the old reference maximum (probabilities then
        // reach at most 256, harmless in bf16/f32) and skip the O-wide rescale; decided per wave, before any P is formed
        float m_new = m_run[u];
        alpha[u] = 1.f;
        if (!__all(mx - m_run[u] <= 8.0f)) {
          m_new = fmaxf(m_run[u], mx);
          alpha[u] = __builtin_amdgcn_exp2f(m_run[u] - m_new);
        }
        // two scores per VALU instruction where the ISA has a packed form (v_pk_fma_f32, v_pk_add_f32): a wave64 VALU
        // instruction occupies the SIMD for 4 cycles, and this loop -- not the MFMAs -- is what bounds the kernel
#if ATTN_PK
        f32x2 rs2 = {0.f, 0.f};
        const f32x2 c22 = {c2, c2}, mn2 = {m_new, m_new};
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 x = f32x2{st[u][t][r], st[u][t][r + 1]} * c22 - mn2;
            const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            rs2 += e;
            pv[u][t][r] = e[0]; pv[u][t][r + 1] = e[1];
          }
        rs = rs2[0] + rs2[1];
#else
        float rsa = 0.f, rsb = 0.f;
        const float nm = -m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][t][r], c2, nm));
            const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][t][r + 1], c2, nm));
            rsa += e0; rsb += e1;
            pv[u][t][r] = e0; pv[u][t][r + 1] = e1;
          }
        rs = rsa + rsb;
#endif
        m_run[u] = m_new;
      } else {
        float mx = -1.0e30f;
        const uint64_t mb = mask_bits(ctl, kt, Stot, lane) >> (4 * hh);   // this lane's keys: bit (t*32 + (r&3) + 8*(r>>2))
        const QueryRw qr = make_qrw(ctl, qi[u]);
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kj = kt * 64 + t * 32 + mfma_row(r, hh);
            float s = st[u][t][r] * p.scale;
            if (text) {
              s *= kj >= qr.sep ? qr.fq : 1.f;
              s += mask_add(mb, t * 32 + (r & 3) + 8 * (r >> 2));
            }
            s *= LOG2E;
            if (kj >= Stot) s = -1.0e30f;
            pv[u][t][r] = s;
            mx = fmaxf(mx, s);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[u], mx);
        alpha[u] = __builtin_amdgcn_exp2f(m_run[u] - m_new);
        // dropout on the probabilities: index = ((b nh + h) Sq + q) Stot + key (< 2^32, host-checked); registers r, r + 1 (r even) are
        // adjacent keys, so with an even Stot they share one hash (common.h)
        const uint32_t rowbase = (uint32_t)(((uint32_t)b * p.nh + h) * p.Sq + (uint32_t)qi[u]) * (uint32_t)Stot;
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float e0 = __builtin_amdgcn_exp2f(pv[u][t][r] - m_new), e1 = __builtin_amdgcn_exp2f(pv[u][t][r + 1] - m_new);
            rs += e0 + e1;
            float u0 = e0, u1 = e1;
            if (TEXT && ctl.p_drop > 0.f) {
              const uint32_t idx = rowbase + (uint32_t)(kt * 64 + t * 32 + mfma_row(r, hh));
              bool k0, k1;
              if ((Stot & 1) == 0) {                       // wave-uniform: idx is even, idx + 1 its pair partner
                const uint32_t hsh = rng_pair(ctl.s2, idx >> 1);
                k0 = (hsh & 0xffffU) >= ctl.thr; k1 = (hsh >> 16) >= ctl.thr;
              } else {
                k0 = dropout_keep32(ctl.s2, idx, ctl.thr); k1 = dropout_keep32(ctl.s2, idx + 1, ctl.thr);
              }
              u0 = k0 ? e0 * ctl.inv_keep : 0.f; u1 = k1 ? e1 * ctl.inv_keep : 0.f;
            }
            pv[u][t][r] = u0; pv[u][t][r + 1] = u1;
          }
        m_run[u] = m_new;
      }
      rs += __shfl_xor(rs, 32, 64);
      l_run[u] = l_run[u] * alpha[u] + rs;
      if (!__all(alpha[u] == 1.f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { ot[u][0][r] *= alpha[u]; ot[u][1][r] *= alpha[u]; }
      }
    }
    }
    // O^T[d][q] += V^T P^T   (the transposed V fragments are shared by the wave's tiles)
#pragma unroll
    for (int t = 0; t < 2; ++t) if (t < NT)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        bf16x8 vf[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) vf[dt] = tile_frag_tr(sV, t * 32 + 16 * a, dt, lo);
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
          if (u > 0 && !active[u]) continue;
          bf16x8 pf;
          if constexpr (NOMAX) {
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (bf16)st[u][t][8 * a + e];
          } else {
            pf = pack8(&pv[u][t][8 * a]);
          }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) ot[u][dt] = mfma32(vf[dt], pf, ot[u][dt]);
        }
      }
    return true;
  };
  constexpr bool PEEL = FWD_PEEL && !TEXT && !RES;
  using Classic = std::integral_constant<int, 0>;
  using Fast = std::integral_constant<int, 1>;
  if constexpr (PEEL && FWD_NOMAX && TPW == 1) {
    // tile 0 classic (it gives the reference maximum its first value), the others fast; a wave whose vote fails redoes THAT tile in the classic form and
    // finishes the key stream in it -- every wave still passes the same barriers and stages the same tiles, whichever form it runs
    int kt = 0;
    bool bad = false;
    if (ntiles > 1) {
      tile_barrier();
      ATTN_STAMP(0, stamp_on, 2);
      if (NWV == 4 || wave < 4) stage_kv(64, smem + STAGE_BYTES);
      if (active[0]) tile(0, std::false_type{}, Classic{});
      for (kt = 1; kt < ntiles - 1; ++kt) {
        tile_barrier();
        ATTN_STAMP(0, stamp_on, 2 + kt);
        if (NWV == 4 || wave < 4) stage_kv((kt + 1) * 64, smem + ((kt + 1) & 1) * STAGE_BYTES);
        if (!active[0]) continue;                     // wave past the last query row: only stages tiles and keeps the barriers
        if (!tile(kt, std::false_type{}, Fast{})) { bad = true; break; }
      }
      if (!bad) {
        tile_barrier();
        ATTN_STAMP(0, stamp_on, 1 + ntiles);
        if (active[0] && !tile(ntiles - 1, std::true_type{}, Fast{})) bad = true;     // (kt == ntiles - 1 here)
      }
    } else {
      tile_barrier();
      if (active[0]) tile(0, std::true_type{}, Classic{});
    }
    if (__builtin_expect(bad, 0)) {
      if (kt < ntiles - 1) {
        tile(kt, std::false_type{}, Classic{});       // its barrier and the staging of tile kt + 1 were done by the fast loop
        for (++kt; kt < ntiles - 1; ++kt) {
          tile_barrier();
          if (NWV == 4 || wave < 4) stage_kv((kt + 1) * 64, smem + ((kt + 1) & 1) * STAGE_BYTES);
          tile(kt, std::false_type{}, Classic{});
        }
        tile_barrier();
      }
      tile(ntiles - 1, std::true_type{}, Classic{});  // (failed in the last tile itself: its barrier was passed above)
    }
  } else {
    for (int kt = 0; kt < ntiles - (PEEL ? 1 : 0); ++kt) {
      if constexpr (!RES) {
        tile_barrier();
        ATTN_STAMP(0, stamp_on, 2 + kt);
        if (kt + 1 < ntiles && (NWV == 4 || wave < 4)) stage_kv((kt + 1) * 64, smem + ((kt + 1) & 1) * STAGE_BYTES);
      }
      if (!active[0]) continue;                       // wave past the last query row: only stages tiles and keeps the barriers
      if constexpr (PEEL) tile(kt, std::false_type{}, Classic{}); else tile(kt, std::true_type{}, Classic{});
    }
    if constexpr (PEEL) {
      tile_barrier();
      ATTN_STAMP(0, stamp_on, 1 + ntiles);
      if (active[0]) tile(ntiles - 1, std::true_type{}, Classic{});
    }
  }
#ifndef FWD_LDS_EPI
#define FWD_LDS_EPI 1
#endif
  if (FWD_LDS_EPI && TPW == 1 && !TEXT && (((uintptr_t)p.ctx & 15) == 0) && (((uintptr_t)p.ctx_f16 & 15) == 0) && p.ldctx % 8 == 0) {   // (text shape, two active waves: 1.6 us SLOWER)
    // The output tile has "lane = query, registers = 4 consecutive head dims": stored from registers, an instruction writes 8 bytes into each of 32
    // rows that lie 2 * ld bytes apart (32 partial cache lines).  Each wave stages its 32 rows in its own 4 KB of the (now dead) K / V ring and
    // stores them as 16-byte chunks, eight lanes per 128-byte row segment.
    ATTN_STAMP(0, stamp_on, 2 + ntiles);
    __syncthreads();                                       // every wave is through its last K / V tile
    ATTN_STAMP(0, stamp_on, 3 + ntiles);
    char* so = smem + wave * 4096;
    if (active[0]) {
      const float inv = 1.f / l_run[0];
      const int okey = swz_key(l31);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x4 v = {ot[0][dt][4 * qd] * inv, ot[0][dt][4 * qd + 1] * inv, ot[0][dt][4 * qd + 2] * inv, ot[0][dt][4 * qd + 3] * inv};
          *(bf16x4*)(so + l31 * 128 + (((4 * dt + qd) ^ okey) << 4) + 8 * hh) = f4_to_bf4(v);
        }
      if (hh == 0 && p.lse && qi[0] < p.Sq) p.lse[((long long)b * p.nh + h) * p.Sq + qi[0]] = m_run[0] + __builtin_amdgcn_logf(l_run[0]);   // log2-domain LSE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int pc = lane & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3), lc = pc ^ swz_key(row);
        if (q0[0] + row < p.Sq)
          *(bf16x8*)((bf16*)p.ctx + ((long long)b * p.Sq + q0[0] + row) * p.ldctx + h * 64 + lc * 8) = *(const bf16x8*)(so + row * 128 + pc * 16);
      }
      if (p.ctx_f16) {                                     // fp16 twin (rounded once from f32) through the same wave-private staging tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const f32x4 v = {ot[0][dt][4 * qd] * inv, ot[0][dt][4 * qd + 1] * inv, ot[0][dt][4 * qd + 2] * inv, ot[0][dt][4 * qd + 3] * inv};
            *(bf16x4*)(so + l31 * 128 + (((4 * dt + qd) ^ okey) << 4) + 8 * hh) = f4_to_h4raw(v);
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * i + (lane >> 3), lc = pc ^ swz_key(row);
          if (q0[0] + row < p.Sq)
            *(bf16x8*)((bf16*)p.ctx_f16 + ((long long)b * p.Sq + q0[0] + row) * p.ldctx + h * 64 + lc * 8) = *(const bf16x8*)(so + row * 128 + pc * 16);
        }
      }
    }
    ATTN_STAMP(0, stamp_on, 4 + ntiles);
    return;
  }
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    if (qi[u] < p.Sq) {
      const float inv = 1.f / l_run[u];
      bf16* op = (bf16*)p.ctx + ((long long)b * p.Sq + qi[u]) * p.ldctx + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          f32x4 v = {ot[u][dt][4 * qd] * inv, ot[u][dt][4 * qd + 1] * inv, ot[u][dt][4 * qd + 2] * inv, ot[u][dt][4 * qd + 3] * inv};
          *(bf16x4*)(op + dt * 32 + 8 * qd + 4 * hh) = f4_to_bf4(v);
          if (p.ctx_f16) *(bf16x4*)((bf16*)p.ctx_f16 + (op - (bf16*)p.ctx) + dt * 32 + 8 * qd + 4 * hh) = f4_to_h4raw(v);   // fp16 twin, rounded once from f32
        }
      if (hh == 0 && p.lse) p.lse[((long long)b * p.nh + h) * p.Sq + qi[u]] = m_run[u] + __builtin_amdgcn_logf(l_run[u]);   // log2-domain LSE
    }
  }
}

#undef stage_kv

// =========================================================================== backward, dQ pass (owner = queries)
// TPW query tiles per wave as in the forward kernel (two independent exp / dS chains per wave, half the K/V streams).
template <bool TEXT, int TPW>
__global__ __launch_bounds__(NTH) void attn_bwd_dq_k(mart_attn_bwd_desc pb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mart_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WgId wg = wg_id(p.nh, p.B);
  const int b = wg.b, h = wg.h;
  const int Stot = p.Lp + p.Sk;
  const Side K{(const bf16*)p.k, p.ldk, p.Sk, (const bf16*)p.pk, p.ldp, p.Lp};
  const Side V{(const bf16*)p.v, p.ldv, p.Sk, (const bf16*)p.pv, p.ldp, p.Lp};
  const TextCtl ctl = make_ctl(p, b, p.Sk);
  constexpr bool text = TEXT;          // vision instantiation: no mask / reweight / dropout code at all
  const LaneOffs lo = make_offs(lane);

  int q0[TPW], qi[TPW];
  bool qvalid[TPW];
  bf16x8 qf[TPW][4], gf[TPW][4];
  float lse[TPW], delta[TPW];
  f32x16 dq[TPW][2];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    q0[u] = wg.part * (128 * TPW) + (wave * TPW + u) * 32;
    qi[u] = q0[u] + l31;
    const int qc = min(qi[u], p.Sq - 1);
    qvalid[u] = qi[u] < p.Sq;
    const bf16* qp = (const bf16*)p.q + ((long long)b * p.Sq + qc) * p.ldq + h * 64;
    const bf16* gp = (const bf16*)pb.dctx + ((long long)b * p.Sq + qc) * pb.lddctx + h * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[u][ks] = *(const bf16x8*)(qp + ks * 16 + hh * 8);
      gf[u][ks] = *(const bf16x8*)(gp + ks * 16 + hh * 8);
    }
    const long long li = ((long long)b * p.nh + h) * p.Sq + qc;
    lse[u] = p.lse[li];                                 // log2-domain statistic saved by the forward pass
    // delta = rowsum(dO * O): this lane holds half of the row's dO already; the matching half of O is one more 64-byte
    // read.  Written out for the dK/dV pass (replaces a separate kernel over both tensors).
    const bf16* op = (const bf16*)p.ctx + ((long long)b * p.Sq + qc) * p.ldctx + h * 64;
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 of = *(const bf16x8*)(op + ks * 16 + hh * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum += (float)of[e] * (float)gf[u][ks][e];
    }
    delta[u] = dsum + __shfl_xor(dsum, 32, 64);
    if (qvalid[u] && hh == 0) pb.delta[li] = delta[u];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[u][0][r] = 0.f; dq[u][1][r] = 0.f; }
  }
  const float c2 = p.scale * LOG2E;
  float dc0 = 0.f, dc1 = 0.f;

  const int ntiles = (Stot + 63) / 64;
  stage_tile(K, b, h, 0, smem, tid, wave);
  stage_tile(V, b, h, 0, smem + TILE_BYTES, tid, wave);
  for (int kt = 0; kt < ntiles; ++kt) {
    tile_barrier();
    if (kt + 1 < ntiles) {
      char* nb = smem + ((kt + 1) & 1) * STAGE_BYTES;
      stage_tile(K, b, h, (kt + 1) * 64, nb, tid, wave);
      stage_tile(V, b, h, (kt + 1) * 64, nb + TILE_BYTES, tid, wave);
    }
    const char* sK = smem + (kt & 1) * STAGE_BYTES;
    const char* sV = sK + TILE_BYTES;
    if (q0[0] >= p.Sq) continue;                      // idle wave (rows past Sq): staging + barriers only
    const bool fastp = !text && kt * 64 + 64 <= Stot; // fast path: fma, exp2, sub, mul per score
#pragma unroll
    for (int t = 0; t < 2; ++t) {                      // one 32-key half at a time: one S / dP accumulator pair live per tile
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        if (u > 0 && q0[u] >= p.Sq) continue;
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st = mfma32(tile_frag(sK, t, ks, lo), qf[u][ks], st);
          dp = mfma32(tile_frag(sV, t, ks, lo), gf[u][ks], dp);
        }
        float dsv[16];                                     // d/d(raw q.k) / scale  (scale applied once to the accumulators)
        if constexpr (!TEXT) {
          if (!fastp) {                                    // partial key tile: masked scores -> p = exp2(-huge) = 0 -> dS = 0
            const int lim = Stot - kt * 64 - t * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (mfma_row(r, hh) >= lim) st[r] = -1.0e30f;
          }
          const f32x2 c22 = {c2, c2}, l2 = {lse[u], lse[u]}, d2 = {delta[u], delta[u]};
#pragma unroll
          for (int r = 0; r < 16; r += 2) {                // packed fma / sub / mul: two scores per VALU instruction
            const f32x2 x = f32x2{st[r], st[r + 1]} * c22 - l2;
            const f32x2 pr = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            const f32x2 ds = pr * (f32x2{dp[r], dp[r + 1]} - d2);
            dsv[r] = ds[0]; dsv[r + 1] = ds[1];
          }
        } else {
          const uint64_t mb = mask_bits(ctl, kt, Stot, lane) >> (4 * hh);
          const QueryRw qr = make_qrw(ctl, qi[u]);
          const uint32_t rowbase = (uint32_t)(((uint32_t)b * p.nh + h) * p.Sq + (uint32_t)qi[u]) * (uint32_t)Stot;
          const float lse_u = qvalid[u] ? lse[u] : 1.0e30f;    // rows past Sq: p = exp2(x - 1e30) = 0
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kj = kt * 64 + t * 32 + mfma_row(r, hh);
            const float spre = st[r] * p.scale;
            const bool rw = kj >= qr.sep;                      // key in the reweighted block
            const float f = rw ? qr.fq : 1.f;
            const float sc = spre * f + mask_add(mb, t * 32 + (r & 3) + 8 * (r >> 2));
            const float pr = kj < Stot ? __builtin_amdgcn_exp2f(sc * LOG2E - lse_u) : 0.f;
            float dpd = dp[r];
            if (TEXT && ctl.p_drop > 0.f) dpd = dropout_keep32(ctl.s2, rowbase + (uint32_t)kj, ctl.thr) ? dpd * ctl.inv_keep : 0.f;
            const float ds = pr * (dpd - delta[u]);        // d/d(post-reweight, pre-mask score)
            const float tw = rw ? ds * spre : 0.f;           // branch-free d(w0) / d(w1) sums: the selectors are lane constants
            dc1 = __builtin_fmaf(tw, qr.sel1, dc1); dc0 = __builtin_fmaf(tw, qr.sel0, dc0);
            dsv[r] = ds * f;
          }
        }
        // dQ^T[d][q] += K^T dS^T
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const bf16x8 df = pack8(&dsv[8 * a]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) dq[u][dt] = mfma32(tile_frag_tr(sK, t * 32 + 16 * a, dt, lo), df, dq[u][dt]);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    if (qvalid[u]) {
      bf16* op = (bf16*)pb.dq + ((long long)b * p.Sq + qi[u]) * pb.lddq + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          f32x4 v = {dq[u][dt][4 * qd] * p.scale, dq[u][dt][4 * qd + 1] * p.scale, dq[u][dt][4 * qd + 2] * p.scale, dq[u][dt][4 * qd + 3] * p.scale};
          *(bf16x4*)(op + dt * 32 + 8 * qd + 4 * hh) = f4_to_bf4(v);
        }
    }
  }
  if (TEXT && pb.dw && ctl.sep >= 0) {
    dc0 = wave_sum(dc0); dc1 = wave_sum(dc1);
    if (lane == 0) {
      if (pb.dw_ws) {
        // one private slot per wave; attn_dw_reduce_k sums them.  (Every wave of every workgroup adding to the same two
        // floats = 6144 contended device-scope atomics per launch: 0.15 ms, more than the rest of the kernel.)
        const long long slot = (((long long)b * p.nh + h) * gridDim.x + wg.part) * (NTH / 64) + wave;
        pb.dw_ws[2 * slot] = dc0; pb.dw_ws[2 * slot + 1] = dc1;
      } else {
        const float w0 = p.w0[0], w1 = p.w1[0];
        if (w0 >= 0.f && w0 <= 0.5f) atomicAdd(pb.dw + 0, dc0);     // clamp sub-gradient: passes inside and AT the bounds
        if (w1 >= 0.5f && w1 <= 1.f) atomicAdd(pb.dw + 1, dc1);
      }
    }
  } else if (TEXT && pb.dw && pb.dw_ws && lane == 0) {
    const long long slot = (((long long)b * p.nh + h) * gridDim.x + wg.part) * (NTH / 64) + wave;
    pb.dw_ws[2 * slot] = 0.f; pb.dw_ws[2 * slot + 1] = 0.f;
  }
}

// sum of the per-wave partials of d(w0), d(w1) -> gradient (clamp sub-gradient: passes inside and AT the bounds)
__global__ void attn_dw_reduce_k(const float* __restrict__ ws, long long n, const float* w0p, const float* w1p, float* dw) {
  float a = 0.f, c = 0.f;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) { a += ws[2 * i]; c += ws[2 * i + 1]; }
  a = wave_sum(a); c = wave_sum(c);
  __shared__ float red[2][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = a; red[1][w] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sa = 0.f, sc = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { sa += red[0][i]; sc += red[1][i]; }
    const float w0 = w0p[0], w1 = w1p[0];
    if (w0 >= 0.f && w0 <= 0.5f) dw[0] += sa;
    if (w1 >= 0.5f && w1 <= 1.f) dw[1] += sc;
  }
}

// =========================================================================== backward, dK/dV pass (owner = keys)
template <bool TEXT>
__global__ __launch_bounds__(NTH, 2) void attn_bwd_dkv_k(mart_attn_bwd_desc pb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mart_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WgId wg = wg_id(p.nh, p.B);
  const int b = wg.b, h = wg.h;
  const int k0 = wg.part * 128 + wave * 32;
  const int Stot = p.Lp + p.Sk;
  const Side K{(const bf16*)p.k, p.ldk, p.Sk, (const bf16*)p.pk, p.ldp, p.Lp};
  const Side V{(const bf16*)p.v, p.ldv, p.Sk, (const bf16*)p.pv, p.ldp, p.Lp};
  const Side Q{(const bf16*)p.q, p.ldq, p.Sq, nullptr, 0, 0};
  const Side G{(const bf16*)pb.dctx, pb.lddctx, p.Sq, nullptr, 0, 0};
  const TextCtl ctl = make_ctl(p, b, p.Sk);
  constexpr bool text = TEXT;          // vision instantiation: no mask / reweight / dropout code at all
  const LaneOffs lo = make_offs(lane);

  const int kj = k0 + l31;
  const bool kvalid = kj < Stot;
  const bf16* kp = side_row(K, b, h, kj);
  const bf16* vp = side_row(V, b, h, kj);
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = *(const bf16x8*)(kp + ks * 16 + hh * 8);
    vf[ks] = *(const bf16x8*)(vp + ks * 16 + hh * 8);
  }
  const float maskadd = (ctl.mask_row && kvalid) ? (ctl.mask_row[kj] != 0 ? 0.f : -10000.0f) : 0.f;   // any non-zero mask value = attended, as mask_bits() in the forward pass
  const float c2 = p.scale * LOG2E;

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }

  const float* lse_b = p.lse + ((long long)b * p.nh + h) * p.Sq;
  const float* del_b = pb.delta + ((long long)b * p.nh + h) * p.Sq;
  auto stage_stats = [&](int q0, char* base) {       // 64 lse + 64 delta floats via 4-byte LDS-DMA (waves 0,1)
    if (wave < 2) {
      const float* src = (wave == 0 ? lse_b : del_b) + min(q0 + lane, p.Sq - 1);
      if (STAGE_RAW) {
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)LDS_PTR(base + 2 * TILE_BYTES + wave * 256));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(src) : "memory", "m0");
      } else {
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base + 2 * TILE_BYTES + wave * 256), 4, 0, 0);
      }
    }
  };

  const int ntiles = (p.Sq + 63) / 64;
  stage_tile(Q, b, h, 0, smem, tid, wave);
  stage_tile(G, b, h, 0, smem + TILE_BYTES, tid, wave);
  stage_stats(0, smem);
  for (int qt = 0; qt < ntiles; ++qt) {
    tile_barrier();
    if (qt + 1 < ntiles) {
      char* nb = smem + ((qt + 1) & 1) * STAGE_BYTES;
      stage_tile(Q, b, h, (qt + 1) * 64, nb, tid, wave);
      stage_tile(G, b, h, (qt + 1) * 64, nb + TILE_BYTES, tid, wave);
      stage_stats((qt + 1) * 64, nb);
    }
    const char* sQ = smem + (qt & 1) * STAGE_BYTES;
    const char* sG = sQ + TILE_BYTES;
    const float* sLse = (const float*)(sQ + 2 * TILE_BYTES);
    const float* sDel = sLse + 64;
    if (k0 >= Stot) continue;                         // idle wave (keys past the end): staging + barriers only
    // S[q][key] = Q k^T ; dP[q][key] = dO v^T      (lane = key), one 32-row half of the query tile at a time so that only
    // one S / dP accumulator pair is live (register budget: a third wave per SIMD)
    const bool fast = !text && qt * 64 + 64 <= p.Sq;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        st = mfma32(tile_frag(sQ, t, ks, lo), kf[ks], st);
        dp = mfma32(tile_frag(sG, t, ks, lo), vf[ks], dp);
      }
      if (!TEXT && !fast) {                              // vision, partial query tile: rows past Sq masked -> p = 0, dS = 0 (dO, delta staged from clamped rows: finite)
        const int lim = p.Sq - qt * 64 - t * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (mfma_row(r, hh) >= lim) st[r] = -1.0e30f;
      }
      // dV^T[d][key] += dO^T Pd ; dK^T[d][key] += Q^T dS, eight q rows (one MFMA k-slice) at a time so the probabilities
      // live only as packed bf16 (dS carries no `scale`: it is applied once to dK at the end)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float pd8[8], ds8[8];
        if constexpr (!TEXT) {
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            const int qd = 2 * a + q2;
            const f32x4 l4 = *(const f32x4*)(sLse + t * 32 + 8 * qd + 4 * hh);   // rows 4qd..4qd+3 of this lane-half
            const f32x4 d4 = *(const f32x4*)(sDel + t * 32 + 8 * qd + 4 * hh);
            const f32x2 c22 = {c2, c2};
#pragma unroll
            for (int e = 0; e < 4; e += 2) {                 // packed fma / sub / mul: two scores per VALU instruction
              const int r = 4 * qd + e;
              const f32x2 x = f32x2{st[r], st[r + 1]} * c22 - f32x2{l4[e], l4[e + 1]};
              const f32x2 pr = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
              const f32x2 ds = pr * (f32x2{dp[r], dp[r + 1]} - f32x2{d4[e], d4[e + 1]});
              pd8[4 * q2 + e] = pr[0]; pd8[4 * q2 + e + 1] = pr[1];
              ds8[4 * q2 + e] = ds[0]; ds8[4 * q2 + e + 1] = ds[1];
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int r = 8 * a + i;
            const int ql = t * 32 + mfma_row(r, hh);
            const int qi = qt * 64 + ql;
            float f = 1.f, sc = st[r] * p.scale;
            if (text) { f = reweight(ctl, qi, kj); sc = sc * f + maskadd; }
            const float pr = (kvalid && qi < p.Sq) ? __builtin_amdgcn_exp2f(sc * LOG2E - sLse[ql]) : 0.f;
            float keep = 1.f;
            if (TEXT && ctl.p_drop > 0.f) {
              const uint32_t idx = (uint32_t)(((uint32_t)b * p.nh + h) * p.Sq + (uint32_t)qi) * (uint32_t)Stot + (uint32_t)kj;
              keep = dropout_keep32(ctl.s2, idx, ctl.thr) ? ctl.inv_keep : 0.f;
            }
            pd8[i] = pr * keep;
            ds8[i] = pr * (dp[r] * keep - sDel[ql]) * f;
          }
        }
        const bf16x8 pf = pack8(pd8);
        const bf16x8 df = pack8(ds8);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = mfma32(tile_frag_tr(sG, t * 32 + 16 * a, dt, lo), pf, dv[dt]);
          dk[dt] = mfma32(tile_frag_tr(sQ, t * 32 + 16 * a, dt, lo), df, dk[dt]);
        }
      }
    }
  }
  if (!kvalid) return;
  bf16* okp; bf16* ovp; bool acc = false;
  if (kj < p.Lp) {
    okp = (bf16*)pb.dpk + ((long long)b * p.Lp + kj) * pb.lddp + h * 64;
    ovp = (bf16*)pb.dpv + ((long long)b * p.Lp + kj) * pb.lddp + h * 64;
  } else {
    okp = (bf16*)pb.dk + ((long long)b * p.Sk + (kj - p.Lp)) * pb.lddk + h * 64;
    ovp = (bf16*)pb.dv + ((long long)b * p.Sk + (kj - p.Lp)) * pb.lddv + h * 64;
    acc = pb.accum_dkv != 0;
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int off = dt * 32 + 8 * qd + 4 * hh;
      f32x4 a = {dk[dt][4 * qd] * p.scale, dk[dt][4 * qd + 1] * p.scale, dk[dt][4 * qd + 2] * p.scale, dk[dt][4 * qd + 3] * p.scale};
      f32x4 c = {dv[dt][4 * qd], dv[dt][4 * qd + 1], dv[dt][4 * qd + 2], dv[dt][4 * qd + 3]};
      if (acc) { a += bf4_to_f4(*(const bf16x4*)(okp + off)); c += bf4_to_f4(*(const bf16x4*)(ovp + off)); }
      *(bf16x4*)(okp + off) = f4_to_bf4(a);
      *(bf16x4*)(ovp + off) = f4_to_bf4(c);
    }
}


// =========================================================================== backward, fused (vision: no mask / reweight / dropout)
// ONE 8-wave workgroup per (batch, head), all keys of the head at once (<= 512): S, P and dS are computed ONCE (the two-pass
// kernels above recompute S and the exponentials in the dK/dV pass: 7 GEMM passes and 2 exp per score for 5 and 1).
//   * wave w OWNS the keys [64w, 64w+64) (two 32-key blocks): its V fragments and its dK / dV accumulators (128 VGPRs) stay in
//     registers for the whole kernel; the head's K sits in LDS (64 KB image, swizzled) for the S fragments and for dQ
//   * queries stream through LDS in 32-row tiles (Q, dO: 4 KB each, double-buffered LDS-DMA); per tile and key block:
//     S = Q k^T, dP = dO v^T (lane = key), P = exp2(S c - lse), dS = P (dP - delta), dV += dO^T P, dK += Q^T dS
//   * dQ needs the contraction over KEYS, i.e. dS with the keys along the MFMA k dimension: every wave writes its dS block
//     transposed-ready into an LDS image [key][32 q] (bf16, 8-byte chunks XOR-swizzled by key), and after one barrier wave
//     (db, qb) computes the 16(d) x 16(q) sub-block of dQ^T = K^T dS^T over ALL keys with v_mfma_f32_16x16x32_bf16, both
//     operands by transposed LDS reads: no cross-wave reduction, no atomics, results independent of scheduling.
//     The dQ of tile t-1 is computed at the start of iteration t (dS image double-buffered): one barrier per tile.
#ifndef FUSED_STRAIGHT
#define FUSED_STRAIGHT 0                      // 1: no early-out for key blocks past the end (both blocks of a wave in one basic block)
#endif
constexpr int FQ = 32;                        // query rows per tile
constexpr int F_KIMG = 512 * 128;             // K image
constexpr int F_QT = FQ * 128;                // one 32-row tile
constexpr int F_DS = 512 * 64;                // dS image of one tile: [512 keys][32 q] bf16
constexpr int F_OFF_Q = F_KIMG;               // Q tiles [2], dO tiles [2]
constexpr int F_OFF_DS = F_OFF_Q + 4 * F_QT;
constexpr int F_OFF_STAT = F_OFF_DS + 2 * F_DS;   // lse[512], delta[512]
constexpr int F_LDS = F_OFF_STAT + 2 * 512 * 4;

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// dS image [key][32 q] (64-byte rows, 8-byte chunks).  Writers: lane = key row, one chunk per ds_write_b64 -- 16 consecutive rows
// per pass, 8 of them in each 64-byte half of the 128-byte bank window: the key must be a bijection of (row >> 1) & 7.  Readers:
// transposed reads of rows r..r+7 per 32-lane pass, 32 bytes each: rows r and r + 4 must use different 32-byte halves: bit 2 of
// the key = bit 2 of the row.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ int ds_swz(int row) { return (((row >> 2) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 1) & 1); }

__global__ __launch_bounds__(512) void attn_bwd_fused_k(mart_attn_bwd_desc pb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mart_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x, b = blockIdx.y;
  const int Stot = p.Lp + p.Sk;
  const Side K{(const bf16*)p.k, p.ldk, p.Sk, (const bf16*)p.pk, p.ldp, p.Lp};
  const Side V{(const bf16*)p.v, p.ldv, p.Sk, (const bf16*)p.pv, p.ldp, p.Lp};
  const bool stamp_on = b == p.B / 2 && h == 5;
  ATTN_STAMP(1, stamp_on, 0);
  const LaneOffs lo = make_offs(lane);
  char* sK = smem;
  float* sLse = (float*)(smem + F_OFF_STAT);
  float* sDel = sLse + 512;

  // ---- prologue: K image, lse / delta of every query row, first Q / dO tile
  const int nkrows = 512;                            // ALL 512 image rows (rows past the last key: clamped copies, finite): the dQ
                                                     // contraction multiplies them by the zero rows of the dS image, and 0 x garbage could be NaN
#ifndef FUSED_PRO_FAST
#define FUSED_PRO_FAST 1
#endif
  {
    // A wave-instruction copies the 8 rows j0 .. j0 + 7, j0 = r0 + 8 wave.  Where they all exist and lie on one side of the prefix boundary (prefix
    // length a multiple of 8) the source is a WAVE-UNIFORM pointer plus one loop-invariant lane offset (saddr form; swz_key(row) depends on the row
    // only through (row >> 1) & 7 = (4 wave + (lane >> 4)) & 7): no per-lane clamp / select / 64-bit multiply (8 copies x ~30 VALU in rounds 2-4).
    const int rl = lane >> 3, pc = tid & 7, lcw = pc ^ swz_key(wave * 8 + rl);
    const unsigned voff_own = (unsigned)(rl * p.ldk + lcw * 8) * 2u, voff_pre = (unsigned)(rl * p.ldp + lcw * 8) * 2u;
    const bf16* own_b = (const bf16*)p.k + ((long long)b * p.Sk) * p.ldk + h * 64;
    const bf16* pre_b = p.Lp ? (const bf16*)p.pk + ((long long)b * p.Lp) * p.ldp + h * 64 : own_b;
    const bool fast = FUSED_PRO_FAST && (p.Lp & 7) == 0;
    for (int r0 = 0; r0 < nkrows; r0 += 64) {
      const int j0 = r0 + wave * 8;
      char* dst = sK + (size_t)(r0 * 8 + wave * 64) * 16;
      if (fast && j0 + 8 <= Stot) {
        if (j0 < p.Lp) dma_saddr(pre_b + (long long)j0 * p.ldp, voff_pre, dst);
        else dma_saddr(own_b + (long long)(j0 - p.Lp) * p.ldk, voff_own, dst);
      } else {
        const int row = r0 + (tid >> 3), lc = pc ^ swz_key(row);
        glds16(side_row(K, b, h, row) + lc * 8, dst);
      }
    }
  }
  auto stage_q = [&](int t, int buf) {               // waves 0-3: Q tile, waves 4-7: dO tile (one 16-byte chunk per thread)
    const int c = tid & 255, row = c >> 3, pc = c & 7, lc = pc ^ swz_key(row);
    const int q = min(t * FQ + row, p.Sq - 1);
    const bf16* src = wave < 4 ? (const bf16*)p.q + ((long long)b * p.Sq + q) * p.ldq + h * 64 + lc * 8
                               : (const bf16*)pb.dctx + ((long long)b * p.Sq + q) * pb.lddctx + h * 64 + lc * 8;
#ifndef FUSED_RAW_Q
#define FUSED_RAW_Q 1
#endif
    char* dstq = smem + F_OFF_Q + ((wave < 4 ? 0 : 2) + buf) * F_QT + ((wave & 3) * 64) * 16;
    if (FUSED_RAW_Q) glds16_raw(src, dstq); else glds16(src, dstq);
  };
  stage_q(0, 0);
  {
    // dS images: the rows of key blocks past the last key are never written and must read as zeros in the dQ contraction (every row of a block that
    // holds a key is rewritten each tile, masked keys as zeros).  A wave-instruction covers 16 image rows: skipped when all of them get written.
    f32x4* z = (f32x4*)(smem + F_OFF_DS);
    const int first_dead = 32 * ((Stot + 31) / 32);
#pragma unroll
    for (int i = 0; i < (2 * F_DS / 16) / 512; ++i) {
      const int row_hi = (((i * 512 + wave * 64 + 63) * 16) & (F_DS - 1)) >> 6;      // last row this wave-instruction touches (wave-uniform)
      if (row_hi >= first_dead) z[i * 512 + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  {
    // row statistics.  delta = rowsum(dO * O): EIGHT lanes per query row, one 16-byte chunk of O and dO each, so that a load instruction reads whole
    // 128-byte row segments (one thread per row = 64 partial cache lines per load instruction, 16 instructions per wave, in a prologue nothing overlaps)
    const int pc = tid & 7;
    bf16x8 ov[8], gv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                        // all loads first (rows past Sq: clamped, discarded below)
      const long long r = (long long)b * p.Sq + min(64 * i + (tid >> 3), p.Sq - 1);
      ov[i] = *(const bf16x8*)((const bf16*)p.ctx + r * p.ldctx + h * 64 + pc * 8);
      gv[i] = *(const bf16x8*)((const bf16*)pb.dctx + r * pb.lddctx + h * 64 + pc * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = 64 * i + (tid >> 3);
      float dsum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e += 2)                     // v_dot2c_f32_bf16: two products per instruction, f32 accumulate (16 unpacks + 8 fma per chunk before)
        dsum = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{ov[i][e], ov[i][e + 1]}, bf16x2{gv[i][e], gv[i][e + 1]}, dsum, false);
      dsum += dpp_f32<0xB1>(dsum);                     // lanes ^1, ^2 (quad permutes), then the other quad of the 8-lane group (row_half_mirror)
      dsum += dpp_f32<0x4E>(dsum);
      dsum += dpp_f32<0x141>(dsum);
      if (pc == 0) {
#ifndef FUSED_CINIT
#define FUSED_CINIT 1
#endif
        // FUSED_CINIT: the row statistics enter the score / dP products as the INITIAL accumulator of the MFMA chains (S - lse / c2, dP - delta):
        // their LDS reads move in front of the MFMAs, out of the dependent chain MFMA -> read -> exp2, and one packed op per score pair goes away
        sDel[q] = q < p.Sq ? (FUSED_CINIT ? -dsum : dsum) : 0.f;
        if (pb.delta && q < p.Sq) pb.delta[((long long)b * p.nh + h) * p.Sq + q] = dsum;
      }
    }
    {
      const float l = tid < p.Sq ? p.lse[((long long)b * p.nh + h) * p.Sq + tid] : 1.0e30f;   // rows past Sq: p = exp2(x - 1e30) = 0, they contribute nothing
      sLse[tid] = FUSED_CINIT ? -l * (1.0f / (p.scale * LOG2E)) : l;
    }
  }
  // own keys: V fragments in registers, validity
  bf16x8 vf[2][4];
  int nval[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int kbase = wave * 64 + kb * 32;
    nval[kb] = min(max(Stot - kbase, 0), 32);
    const bf16* vp = side_row(V, b, h, kbase + l31);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vf[kb][ks] = *(const bf16x8*)(vp + ks * 16 + hh * 8);
  }
  f32x16 dk[2][2], dv[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[kb][dt][r] = 0.f; dv[kb][dt][r] = 0.f; }
  const float c2 = p.scale * LOG2E;
  const int ntiles = (p.Sq + FQ - 1) / FQ;
  const int nkc = (Stot + 31) / 32;                   // 32-key chunks of the dQ contraction
  // dQ sub-block of this wave: d rows db*16.., q columns qb*16..
  const int db = wave & 3, qb = wave >> 2;
  const int g16 = lane >> 4, p16 = lane & 15;
  ATTN_STAMP(1, stamp_on, 1);
  tile_barrier();
  ATTN_STAMP(1, stamp_on, 2);

  // dQ of one tile: contraction over 16 chunks of 32 keys, four chunks (16 transposed reads, then 4 MFMAs) at a time so that one
  // LDS latency is exposed per group instead of per chunk; chunks past the last key multiply zeros (dS image zero-filled once).
  // Key order inside a 32-key chunk (the contraction index is a dummy: A and B only have to agree): element e of lane group g16
  // is key 4 g16 + e for e < 4 and 16 + 4 g16 + (e - 4) for e >= 4, so that the two lane groups of a 32-lane pass read rows
  // r..r+7 (conflict-free with both swizzles).  Every swizzle term depends on the row only through row mod 32, so the four
  // per-lane byte offsets are loop invariant and a chunk is an immediate offset (no per-read address arithmetic).
  const int rb = 4 * g16 + (p16 >> 2);
  const int kchunk = db * 2 + ((p16 & 3) >> 1), kbo = (p16 & 1) * 8, dchunk = qb * 4 + (p16 & 3);
  const int ka1 = rb * 128 + ((kchunk ^ swz_key(rb)) << 4) + kbo, ka2 = (rb + 16) * 128 + ((kchunk ^ swz_key(rb + 16)) << 4) + kbo;
  const int da1 = rb * 64 + ((dchunk ^ ds_swz(rb)) << 3), da2 = (rb + 16) * 64 + ((dchunk ^ ds_swz(rb + 16)) << 3);
#ifndef FUSED_DQ_DEPTH
#define FUSED_DQ_DEPTH 3                                   // 32-key chunks whose reads are in flight ahead of the MFMA that consumes them
#endif
  // STRAIGHT-LINE software pipeline over NCH 32-key chunks (14 covers the 393 keys of a prefix-free layer, 16 everything up to 512; chunks past the
  // last key multiply the zero rows of the dS image): the four transposed reads of chunks c + 1 .. c + DQD are in flight under the MFMA of chunk c
  // (DQD + 1 register slots of 8 VGPRs).  Rounds 2-4 read in groups of two chunks, ONE group ahead, behind a runtime group count (a branch per
  // group): the phase was eight exposed LDS latencies long -- 1300-1400 cycles per tile and wave for 256 cycles of matrix work (tools/attn_stamps.py).
  auto dq_tile_n = [&](int tq, auto nch_c) {
    constexpr int NCH = decltype(nch_c)::value;
    constexpr int DQD = FUSED_DQ_DEPTH;
    const char* sD = smem + F_OFF_DS + (tq & 1) * F_DS;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;       // two accumulation chains (even / odd chunks)
    bf16x8 af[DQD + 1], bfv[DQD + 1];
    auto load = [&](int c, int slot) {
      af[slot] = join_tr(lds_tr_read(sK + c * 4096 + ka1), lds_tr_read(sK + c * 4096 + ka2));
      bfv[slot] = join_tr(lds_tr_read(sD + c * 2048 + da1), lds_tr_read(sD + c * 2048 + da2));
    };
#pragma unroll
    for (int c = 0; c < DQD; ++c) load(c, c % (DQD + 1));
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // the scheduling barriers pin the order (left alone, the compiler re-sorts the reads to save registers and waits with ~1 chunk of slack)
      __builtin_amdgcn_sched_barrier(0);
      if (c + DQD < NCH) load(c + DQD, (c + DQD) % (DQD + 1));
      __builtin_amdgcn_sched_barrier(0);
      if (c & 1) acc1 = mfma16(af[c % (DQD + 1)], bfv[c % (DQD + 1)], acc1);
      else acc0 = mfma16(af[c % (DQD + 1)], bfv[c % (DQD + 1)], acc0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 acc = acc0 + acc1;
    const int q = tq * FQ + qb * 16 + p16;
    if (q < p.Sq) {
      bf16* op = (bf16*)pb.dq + ((long long)b * p.Sq + q) * pb.lddq + h * 64 + db * 16 + 4 * g16;
      *(bf16x4*)op = f4_to_bf4(f32x4{acc[0] * p.scale, acc[1] * p.scale, acc[2] * p.scale, acc[3] * p.scale});
    }
  };
  auto dq_tile = [&](int tq) {
    if (nkc <= 14) dq_tile_n(tq, std::integral_constant<int, 14>{}); else dq_tile_n(tq, std::integral_constant<int, 16>{});
  };
  // S, P, dS of tile t for the wave's key blocks; dV, dK accumulate; dS goes to the image of tile t
  auto s_tile = [&](int t) {
    const char* sQ = smem + F_OFF_Q + (t & 1) * F_QT;
    const char* sG = smem + F_OFF_Q + (2 + (t & 1)) * F_QT;
    char* sD = smem + F_OFF_DS + (t & 1) * F_DS;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#if !FUSED_STRAIGHT
      if (nval[kb] == 0) continue;                   // wave-uniform: key block past the end
#endif
      const int key = wave * 64 + kb * 32 + l31;
      const char* krow = sK + key * 128;
      const int kkey = swz_key(key);
      f32x16 st, dp;
      if (FUSED_CINIT) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {             // accumulator rows 4qd .. 4qd+3 of this lane-half are query rows 8qd + 4hh + 0..3 of the tile
          const f32x4 l4 = *(const f32x4*)(sLse + t * FQ + 8 * qd + 4 * hh);
          const f32x4 d4 = *(const f32x4*)(sDel + t * FQ + 8 * qd + 4 * hh);
#pragma unroll
          for (int e = 0; e < 4; ++e) { st[4 * qd + e] = l4[e]; dp[4 * qd + e] = d4[e]; }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(krow + (((ks * 2 + hh) ^ kkey) << 4));
        st = mfma32(tile_frag(sQ, 0, ks, lo), kf, st);
        dp = mfma32(tile_frag(sG, 0, ks, lo), vf[kb][ks], dp);
      }
      ATTN_STAMP(1, stamp_on && t == 5, 23 + 3 * kb);
      if (nval[kb] < 32) {                           // partially valid block: masked keys get p = 0, dS = 0
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (l31 >= nval[kb]) st[r] = -1.0e30f;
      }
      const int swz = ds_swz(key);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float pd8[8], ds8[8];
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int qd = 2 * a + q2;
          f32x4 l4 = {0.f, 0.f, 0.f, 0.f}, d4 = l4;
          if (!FUSED_CINIT) {
            l4 = *(const f32x4*)(sLse + t * FQ + 8 * qd + 4 * hh);
            d4 = *(const f32x4*)(sDel + t * FQ + 8 * qd + 4 * hh);
          }
#if ATTN_PK
          const f32x2 c22 = {c2, c2};
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const int r = 4 * qd + e;
            const f32x2 x = FUSED_CINIT ? f32x2{st[r], st[r + 1]} * c22 : f32x2{st[r], st[r + 1]} * c22 - f32x2{l4[e], l4[e + 1]};
            const f32x2 pr = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            const f32x2 ds = FUSED_CINIT ? pr * f32x2{dp[r], dp[r + 1]} : pr * (f32x2{dp[r], dp[r + 1]} - f32x2{d4[e], d4[e + 1]});
            pd8[4 * q2 + e] = pr[0]; pd8[4 * q2 + e + 1] = pr[1];
            ds8[4 * q2 + e] = ds[0]; ds8[4 * q2 + e + 1] = ds[1];
          }
#else
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * qd + e;
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -l4[e]));
            pd8[4 * q2 + e] = pr;
            ds8[4 * q2 + e] = pr * (dp[r] - d4[e]);
          }
#endif
        }
        const bf16x8 pf = pack8(pd8);
        const bf16x8 df = pack8(ds8);
        // dS, transposed-ready: row = key, 8-byte chunk (4a [+2] + hh) holds the four q rows 16a [+8] + 4hh + 0..3
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        const u32x4_t dw = __builtin_bit_cast(u32x4_t, df);
        *(u32x2_t*)(sD + key * 64 + (((4 * a + hh) ^ swz) << 3)) = u32x2_t{dw[0], dw[1]};
        *(u32x2_t*)(sD + key * 64 + (((4 * a + 2 + hh) ^ swz) << 3)) = u32x2_t{dw[2], dw[3]};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[kb][dt] = mfma32(tile_frag_tr(sG, 16 * a, dt, lo), pf, dv[kb][dt]);
          dk[kb][dt] = mfma32(tile_frag_tr(sQ, 16 * a, dt, lo), df, dk[kb][dt]);
        }
        ATTN_STAMP(1, stamp_on && t == 5, 24 + 3 * kb + a);
      }
    }
  };
  // Iteration t: the dQ of tile t-1 (LDS-read heavy) and the S / dV / dK work of tile t (MFMA + exp heavy) are independent, so
  // the two waves of a SIMD (w and w + 4) run them in OPPOSITE order: one reads while the other computes.
  for (int t = 0; t <= ntiles; ++t) {
    ATTN_STAMP(1, stamp_on && t == 5, 20);
    if (t + 1 < ntiles) stage_q(t + 1, (t + 1) & 1);
    ATTN_STAMP(1, stamp_on && t == 5, 21);
#ifndef FUSED_KO
#define FUSED_KO 0
#endif
    if (wave < 4 && t > 0 && !(FUSED_KO & 1)) dq_tile(t - 1);
    ATTN_STAMP(1, stamp_on && t == 5, 22);
    if (t < ntiles && !(FUSED_KO & 2)) s_tile(t);
    ATTN_STAMP(1, stamp_on && t == 5, 29);
    if (wave >= 4 && t > 0 && !(FUSED_KO & 1)) dq_tile(t - 1);
    ATTN_STAMP(1, stamp_on && t == 5, 30);
    tile_barrier();                                    // dS image of tile t complete, Q / dO tile t+1 landed, tile t's buffers free
    ATTN_STAMP(1, stamp_on, 3 + t);
  }

  // ---- dK, dV of the wave's keys.  From registers ("lane = key, registers = 4 consecutive head dims") a store instruction writes 8 bytes into each
  // of 32 rows that lie 2 * ld bytes apart: 32 partial cache lines per instruction, 32 such instructions per wave, at the end of a workgroup that
  // has the CU to itself.  Staged as bf16 rows in the K image (dK) and the dS images (dV) -- both dead by now -- eight lanes store one 128-byte row
  // segment (8 lines per instruction).  The accumulating form (rounds once from f32) keeps the register path.
#ifndef FUSED_LDS_EPI
#define FUSED_LDS_EPI 1
#endif
  const bool al16 = ((((uintptr_t)pb.dk | (uintptr_t)pb.dv | (uintptr_t)pb.dpk | (uintptr_t)pb.dpv) & 15) == 0) && pb.lddk % 8 == 0 && pb.lddv % 8 == 0 &&
                    (p.Lp == 0 || pb.lddp % 8 == 0);
  if (FUSED_LDS_EPI && !pb.accum_dkv && al16) {
    char* sV2 = smem + F_OFF_DS;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int kj = wave * 64 + kb * 32 + l31, kkey = swz_key(kj);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int off = kj * 128 + (((4 * dt + qd) ^ kkey) << 4) + 8 * hh;
          *(bf16x4*)(sK + off) = f4_to_bf4(f32x4{dk[kb][dt][4 * qd] * p.scale, dk[kb][dt][4 * qd + 1] * p.scale, dk[kb][dt][4 * qd + 2] * p.scale, dk[kb][dt][4 * qd + 3] * p.scale});
          *(bf16x4*)(sV2 + off) = f4_to_bf4(f32x4{dv[kb][dt][4 * qd], dv[kb][dt][4 * qd + 1], dv[kb][dt][4 * qd + 2], dv[kb][dt][4 * qd + 3]});
        }
    }
#ifndef FUSED_EPI_WAVE
#define FUSED_EPI_WAVE 1
#endif
    // FUSED_EPI_WAVE: every wave stores the 64 rows it staged itself -- the images were dead for ALL waves at the loop's last barrier, so the round trip
    // is wave-private and needs no workgroup barrier (rounds 3-4: rows dealt across the workgroup behind a __syncthreads)
    if (FUSED_EPI_WAVE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
    ATTN_STAMP(1, stamp_on, 4 + ntiles);
    const int pc = tid & 7;
    // destinations = (batch, head) base pointers (64-bit, once) + 32-bit element offsets per row (the row-by-row 64-bit products were ~130 VALU per wave here)
    bf16* const dk_b = (bf16*)pb.dk + ((long long)b * p.Sk) * pb.lddk + h * 64;
    bf16* const dv_b = (bf16*)pb.dv + ((long long)b * p.Sk) * pb.lddv + h * 64;
    bf16* const dpk_b = p.Lp ? (bf16*)pb.dpk + ((long long)b * p.Lp) * pb.lddp + h * 64 : dk_b;
    bf16* const dpv_b = p.Lp ? (bf16*)pb.dpv + ((long long)b * p.Lp) * pb.lddp + h * 64 : dv_b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = FUSED_EPI_WAVE ? wave * 64 + 8 * i + (lane >> 3) : 64 * i + (tid >> 3);
      if (row >= Stot) continue;
      const int lc = pc ^ swz_key(row);
      const bool pre = row < p.Lp;
      const unsigned ok_off = pre ? (unsigned)(row * pb.lddp + lc * 8) : (unsigned)((row - p.Lp) * pb.lddk + lc * 8);
      const unsigned ov_off = pre ? ok_off : (unsigned)((row - p.Lp) * pb.lddv + lc * 8);
      *(bf16x8*)((pre ? dpk_b : dk_b) + ok_off) = *(const bf16x8*)(sK + row * 128 + pc * 16);
      *(bf16x8*)((pre ? dpv_b : dv_b) + ov_off) = *(const bf16x8*)(sV2 + row * 128 + pc * 16);
    }
    ATTN_STAMP(1, stamp_on, 5 + ntiles);
    return;
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int kj = wave * 64 + kb * 32 + l31;
    if (kj >= Stot) continue;
    bf16* okp; bf16* ovp; bool acc = false;
    if (kj < p.Lp) {
      okp = (bf16*)pb.dpk + ((long long)b * p.Lp + kj) * pb.lddp + h * 64;
      ovp = (bf16*)pb.dpv + ((long long)b * p.Lp + kj) * pb.lddp + h * 64;
    } else {
      okp = (bf16*)pb.dk + ((long long)b * p.Sk + (kj - p.Lp)) * pb.lddk + h * 64;
      ovp = (bf16*)pb.dv + ((long long)b * p.Sk + (kj - p.Lp)) * pb.lddv + h * 64;
      acc = pb.accum_dkv != 0;
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int off = dt * 32 + 8 * qd + 4 * hh;
        f32x4 a = {dk[kb][dt][4 * qd] * p.scale, dk[kb][dt][4 * qd + 1] * p.scale, dk[kb][dt][4 * qd + 2] * p.scale, dk[kb][dt][4 * qd + 3] * p.scale};
        f32x4 c = {dv[kb][dt][4 * qd], dv[kb][dt][4 * qd + 1], dv[kb][dt][4 * qd + 2], dv[kb][dt][4 * qd + 3]};
        if (acc) { a += bf4_to_f4(*(const bf16x4*)(okp + off)); c += bf4_to_f4(*(const bf16x4*)(ovp + off)); }
        *(bf16x4*)(okp + off) = f4_to_bf4(a);
        *(bf16x4*)(ovp + off) = f4_to_bf4(c);
      }
  }
}

// =========================================================================== fp32-accurate forward on the bf16 matrix pipe (evaluation path)
// The multi-head attentions of the fp32-accurate EVALUATION pass (engine_precise, no_grad) -- the vision tower incl. its text K / V prefix (24 % of
// that pass on the f32 matrix pipe) and, with the text options of attn_f32_mfma_k, the text / multimodal stacks (attn_f32_mfma_k: 2.0 ms per layer at the bench shape, the f32 MFMA runs at 1/16 of the
// bf16 rate).  Here both contractions run on TWO-TERM bf16 splits of their f32 operands, three products each (hi*hi + hi*lo + lo*hi, f32 accumulate:
// 2^-16 relative, the same arithmetic as the path's GEMMs, mart_split_bf16x3): S^T = K q^T with K and Q split, O^T = V^T P^T with V and the
// probabilities split; softmax statistics in f32.  Skeleton of attn_fwd_k<vision>: a wave owns 32 queries (lane = query), K / V stream in 64-key
// tiles -- f32 rows are fetched into registers one tile ahead, split, and written as four swizzled bf16 images (K hi / lo, V hi / lo: 32 KB per
// workgroup) that the same fragment readers as the bf16 kernel consume.  48 MFMAs per tile and wave against 16 in the bf16 kernel.
constexpr int SP_IMG = 64 * 128;                  // one 64 x 64 bf16 image
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (bf16)a[e]; lo[e] = (bf16)(a[e] - (float)hi[e]);
    hi[4 + e] = (bf16)b[e]; lo[4 + e] = (bf16)(b[e] - (float)hi[4 + e]);
  }
}
template <int NW>      // waves per workgroup: 4 (128 queries) or 8 (256 queries: the K / V of a head are fetched and split half as often)
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_split_fwd_k(mart_attn_f32_desc p) {
  __shared__ __attribute__((aligned(16))) char sm[4 * SP_IMG];
  char* sKh = sm; char* sKl = sm + SP_IMG; char* sVh = sm + 2 * SP_IMG; char* sVl = sm + 3 * SP_IMG;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y;
  const long long b = blockIdx.z;
  const int Stot = p.Lp + p.Sk;
  const LaneOffs lo = make_offs(lane);
  const int q0 = blockIdx.x * (32 * NW) + wave * 32, qi = q0 + l31;
  const bool active = q0 < p.Sq;                       // wave-uniform
  bf16x8 qh[4], ql[4];
  {
    const float* qp = p.q + (b * p.Sq + min(qi, p.Sq - 1)) * p.ldq + h * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) split8(*(const f32x4*)(qp + ks * 16), *(const f32x4*)(qp + ks * 16 + 4), qh[ks], ql[ks]);
  }
  f32x16 ot[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { ot[0][r] = 0.f; ot[1][r] = 0.f; }
  float m_run = -1.0e30f, l_run = 0.f;
  const float c2 = p.scale * LOG2E;
  // text options (BertSelfAttention / the FLAVA variant), exactly as attn_f32_mfma_k: scale, adaptive reweight of the keys >= sep, additive key mask
  const bool textopt = p.sep != nullptr || p.attn_mask != nullptr;
  int sp = 0x7fffffff; float rw = 1.f;
  if (p.sep) {
    sp = (int)p.sep[b * p.sep_stride];
    const float w0 = fminf(fmaxf(*p.w0, 0.f), 0.5f), w1 = fminf(fmaxf(*p.w1, 0.5f), 1.f);
    rw = (p.rw_skip_row0 && qi == 0) ? 1.f : (qi < sp ? w0 : w1);
  }
  // staging: thread -> key row, NC 16-byte bf16 chunks (8 dims each) of K and of V: f32 rows kept in registers one tile ahead
  constexpr int NC = 8 / NW;                                           // chunks per thread: 64 rows x 8 chunks over 64 * NW threads
  const int srow = tid / (8 / NC), sc8 = (tid % (8 / NC)) * NC;        // first chunk
  f32x4 kp[2 * NC], vp[2 * NC];
  auto fetch = [&](int t) {
    const int j = min(t * 64 + srow, Stot - 1);                       // rows past the last key: clamped copies (masked below)
    const float* kr = j < p.Lp ? p.pk + (b * p.Lp + j) * p.ldp + h * 64 : p.k + (b * p.Sk + (j - p.Lp)) * p.ldk + h * 64;
    const float* vr = j < p.Lp ? p.pv + (b * p.Lp + j) * p.ldp + h * 64 : p.v + (b * p.Sk + (j - p.Lp)) * p.ldv + h * 64;
#pragma unroll
    for (int u = 0; u < 2 * NC; ++u) { kp[u] = *(const f32x4*)(kr + sc8 * 8 + 4 * u); vp[u] = *(const f32x4*)(vr + sc8 * 8 + 4 * u); }
  };
  auto stash = [&]() {
    const int key = swz_key(srow);
#pragma unroll
    for (int u = 0; u < NC; ++u) {
      const int off = srow * 128 + (((sc8 + u) ^ key) << 4);
      bf16x8 a, c;
      split8(kp[2 * u], kp[2 * u + 1], a, c);
      *(bf16x8*)(sKh + off) = a; *(bf16x8*)(sKl + off) = c;
      split8(vp[2 * u], vp[2 * u + 1], a, c);
      *(bf16x8*)(sVh + off) = a; *(bf16x8*)(sVl + off) = c;
    }
  };
  const int ntiles = (Stot + 63) / 64;
  fetch(0);
  stash();
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    if (kt + 1 < ntiles) fetch(kt + 1);
    if (active) {
      f32x16 st[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 kh = tile_frag(sKh, t, ks, lo), kl = tile_frag(sKl, t, ks, lo);
          st[t] = mfma32(kl, qh[ks], st[t]);                           // small terms first
          st[t] = mfma32(kh, ql[ks], st[t]);
          st[t] = mfma32(kh, qh[ks], st[t]);
        }
      }
      // scores in the log2 domain: st <- log2(e) * (scale * s [* reweight] [+ mask]); keys past the end: -1e30
      if (!textopt && kt * 64 + 64 <= Stot) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[t][r] *= c2;
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = kt * 64 + t * 32 + mfma_row(r, hh);
            float v = st[t][r] * p.scale;
            if (j >= sp) v *= rw;
            if (p.attn_mask) v += (p.attn_mask[b * p.Sk + min(j, Stot - 1)] == 0) ? -10000.0f : 0.f;
            st[t][r] = j >= Stot ? -1.0e30f : v * LOG2E;
          }
      }
      float mx = st[0][0];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float rs = 0.f;
      float pv[2][16];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(st[t][r] - m_new);
          rs += e;
          pv[t][r] = e;
        }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run * alpha + rs;
      m_run = m_new;
      if (!__all(alpha == 1.f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { ot[0][r] *= alpha; ot[1][r] *= alpha; }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          bf16x8 ph, pl;
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float x = pv[t][8 * a + e]; ph[e] = (bf16)x; pl[e] = (bf16)(x - (float)ph[e]); }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const bf16x8 vh = tile_frag_tr(sVh, t * 32 + 16 * a, dt, lo), vl = tile_frag_tr(sVl, t * 32 + 16 * a, dt, lo);
            ot[dt] = mfma32(vl, ph, ot[dt]);
            ot[dt] = mfma32(vh, pl, ot[dt]);
            ot[dt] = mfma32(vh, ph, ot[dt]);
          }
        }
    }
    __syncthreads();                                                   // every wave is through tile kt
    if (kt + 1 < ntiles) { stash(); __syncthreads(); }
  }
  if (qi < p.Sq) {
    const float inv = 1.f / l_run;
    float* op = p.ctx + (b * p.Sq + qi) * p.ldctx + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
      {
        const f32x4 y = {ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv, ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv};
        const int col = h * 64 + dt * 32 + 8 * g + 4 * hh;
        if (p.ctx) *(f32x4*)(op + dt * 32 + 8 * g + 4 * hh) = y;
        if (p.ctx_split3) {                                            // [hi | lo | hi] operand rows for the output projection
          const int HD = p.nh * 64;
          const bf16x4 hi = f4_to_bf4(y), lw = f4_to_bf4(y - bf4_to_f4(hi));
          bf16* d3 = (bf16*)p.ctx_split3 + (b * p.Sq + qi) * p.ldctx3 + col;
          *(bf16x4*)d3 = hi; *(bf16x4*)(d3 + HD) = lw; *(bf16x4*)(d3 + 2 * HD) = hi;
        }
      }
  }
}

int check_fwd(const mart_attn_fwd_desc* d) {
  MART_CHECK(d && d->q && d->k && d->v && d->ctx, "attn: null pointer");
  MART_CHECK(d->B > 0 && d->nh > 0 && d->Sq > 0 && d->Sk > 0 && d->Lp >= 0, "attn: bad shape");
  MART_CHECK(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldctx % 4 == 0, "attn: leading dims must keep 16-byte rows");
  MART_CHECK(d->Lp == 0 || (d->pk && d->pv && d->ldp % 8 == 0), "attn: prefix needs pk/pv/ldp");
  MART_CHECK((d->w0 == nullptr) == (d->w1 == nullptr), "attn: w0/w1 must come together");
  MART_CHECK(!d->sep || (d->w0 && d->Lp == 0), "attn: reweight needs w0/w1 and no prefix");
  MART_CHECK(!d->attn_mask || d->Lp == 0, "attn: mask with prefix unsupported");
  MART_CHECK(d->p_drop >= 0.f && d->p_drop < 1.f, "attn: bad dropout p");
  MART_CHECK(d->p_drop == 0.f || (long long)d->B * d->nh * d->Sq * (d->Lp + d->Sk) < (1LL << 32), "attn: dropout indices are 32-bit (B nh Sq Sk < 2^32)");
  return 0;
}

// =========================================================================== backward, fused, text shape (<= 64 queries x <= 64 keys per head)
// The two-pass kernels above are built for long key / query streams: at the text shape (BertSelfAttention at L = 64: ONE 64 x 64 score tile per
// head) half of their waves own no rows, both passes stage tiles through the double-buffered ring behind two barriers, S and the exponentials
// are computed twice and `delta` makes a round trip through HBM.  Here ONE 2-wave workgroup does a head in one pass:
//   * Q, dO and K sit in LDS as swizzled 64 x 64 tiles (LDS-DMA, one barrier); lse and delta = rowsum(dO * O) of the 64 query rows beside them
//   * wave w OWNS keys [32w, 32w + 32): S = Q k^T, dP = dO v^T (lane = key), the text options (reweight, key-padding mask, dropout,
//     d(w0) / d(w1)) exactly as in the two-pass kernels, dV += dO^T Pd, dK += Q^T dS; dS goes transposed-ready into a fourth tile [key][q]
//   * after the second barrier wave w OWNS queries [32w, 32w + 32): dQ^T = K^T dS^T over all 64 keys, both operands by transposed LDS reads
// No atomics, no cross-wave reduction: results are independent of scheduling.
// NB = number of 32-row blocks (= waves): 2 covers the fine-tune shape (L = 64), 4 the pre-train shape (L = 96) up to 128 x 128.
//   R = 32 NB rows per tile; dS is stored as R / 64 sub-tiles [R keys][64 queries] so that every tile keeps the 128-byte row of the fragment
//   readers; the K tile lands over rows [0, R/2) of the Q and the dO tile once the first NB / 2 query blocks are through.
template <int NB> struct T64 {
  static constexpr int R = 32 * NB, NT = 64 * NB;
  static constexpr int TILE = R * 128;                    // one [R][64] bf16 tile
  static constexpr int DS = ((R + 63) / 64) * TILE;       // the dS sub-tiles
  static constexpr int LDS = 2 * TILE + DS + 2 * R * 4;   // Q, dO, dS, lse / delta
  static constexpr int TK = (R / 2 + 31) / 32 - 1;        // query block after which rows [0, R/2) of the Q / dO tiles are dead
};
// swap a value between lanes 2m and 2m + 1 (DPP quad_perm [1,0,3,2]: a full-rate VALU move, no LDS)
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
template <bool DROP, int NB>
__global__ __launch_bounds__(64 * NB, NB == 2 ? 3 : 2) void attn_bwd_text64_k(mart_attn_bwd_desc pb) {
  using G = T64<NB>;
  constexpr int R = G::R;
  __shared__ __attribute__((aligned(16))) char smem[G::LDS];
  const mart_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x, b = blockIdx.y;
  const int Stot = p.Sk;                                  // host: no prefix, Sq <= R, Sk <= R
  const TextCtl ctl = make_ctl(p, b, p.Sk);
  const LaneOffs lo = make_offs(lane);
  char* sQ = smem;
  char* sG = smem + G::TILE;
  char* sD = smem + 2 * G::TILE;
  float* sLse = (float*)(smem + 2 * G::TILE + G::DS);
  float* sDel = sLse + R;

  // staging: one DMA instruction of the workgroup covers 8 NB rows (eight lanes per 128-byte row); the swizzle key repeats every 16 rows
  constexpr int RPI = 8 * NB;
  const int rowh = tid >> 3;
  auto lc_of = [&](int i) { return (tid & 7) ^ swz_key(RPI * i + rowh); };   // logical chunk held by this lane's physical chunk (NB = 3: 24 rows per instruction)
#pragma unroll
  for (int i = 0; i < 4; ++i) {                           // Q, dO tiles (rows past the end: clamped copies)
    const long long rq = (long long)b * p.Sq + min(RPI * i + rowh, p.Sq - 1);
    const int doff = (i * G::NT + wave * 64) * 16, lc = lc_of(i);
    glds16_raw((const bf16*)p.q + rq * p.ldq + h * 64 + lc * 8, sQ + doff);
    glds16_raw((const bf16*)pb.dctx + rq * pb.lddctx + h * 64 + lc * 8, sG + doff);
  }
  {                                                       // row statistics: eight lanes per query row (whole 128-byte row segments per load instruction)
    const int pc = tid & 7;
    bf16x8 ov[4], gv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                         // rows past Sq: clamped, discarded below
      const long long r = (long long)b * p.Sq + min(RPI * i + rowh, p.Sq - 1);
      ov[i] = *(const bf16x8*)((const bf16*)p.ctx + r * p.ldctx + h * 64 + pc * 8);
      gv[i] = *(const bf16x8*)((const bf16*)pb.dctx + r * pb.lddctx + h * 64 + pc * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = RPI * i + rowh;
      float dsum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum += (float)ov[i][e] * (float)gv[i][e];
      dsum += dpp_f32<0xB1>(dsum);
      dsum += dpp_f32<0x4E>(dsum);
      dsum += dpp_f32<0x141>(dsum);
      if (pc == 0) {
        sDel[q] = q < p.Sq ? dsum : 0.f;
        if (q < p.Sq) pb.delta[((long long)b * p.nh + h) * p.Sq + q] = dsum;
      }
    }
    if (tid < R) sLse[tid] = tid < p.Sq ? p.lse[((long long)b * p.nh + h) * p.Sq + tid] : 1.0e30f;   // rows past Sq: p = exp2(x - 1e30) = 0
  }
  // own keys
  const int kj = wave * 32 + l31;
  const bool kvalid = kj < Stot;
  // K / V fragments of the own keys: loaded for each query block (after the first time from L1 / L2) instead of held across the softmax arithmetic --
  // 32 VGPRs that decide between two and three waves per SIMD
  bf16x8 kf[4], vf[4];
  const bf16* kp = (const bf16*)p.k + ((long long)b * p.Sk + min(kj, p.Sk - 1)) * p.ldk + h * 64 + hh * 8;
  const bf16* vp = (const bf16*)p.v + ((long long)b * p.Sk + min(kj, p.Sk - 1)) * p.ldv + h * 64 + hh * 8;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const bf16x8*)(kp + ks * 16); vf[ks] = *(const bf16x8*)(vp + ks * 16); }
  // lane constants of the key: additive mask in the log2 domain (keys past the end: p = 0), membership in the reweighted block
  const float madd2 = !kvalid ? -1.0e30f : (ctl.mask_row ? (ctl.mask_row[kj] != 0 ? 0.f : -10000.0f * LOG2E) : 0.f);   // same convention as mask_bits(): non-zero = attended
  const int sep = ctl.sep < 0 ? 0x7fffffff : ctl.sep;
  const bool rw = kj >= sep;                               // this lane's key is in the reweighted block
  const float rwf = rw ? 1.f : 0.f;
  // dropout (common.h dropout_keep32): element idx = (head row) * Stot + key takes the low / high 16 bits of ONE hash per index pair.  With an
  // even key count the two keys of a pair sit in lanes 2m, 2m + 1 of the same query row: the even lane hashes rows 0-3 of each 8-row group, the
  // odd lane rows 4-7, and a DPP swap hands each the other half -- four hashes per eight decisions.
  const uint32_t headbase = ((uint32_t)b * p.nh + h) * (uint32_t)p.Sq;
  const bool shared = (Stot & 1) == 0;                     // wave-uniform
  const int odd = lane & 1;
  const uint32_t ib_own = headbase * (uint32_t)Stot + (uint32_t)kj + (uint32_t)(4 * hh) * (uint32_t)Stot;                    // + row * Stot
  const uint32_t ib_pair = headbase * (uint32_t)Stot + (uint32_t)(kj & ~1) + (uint32_t)(4 * hh + 8 * odd) * (uint32_t)Stot;  // this lane's four rows of a group
  const uint32_t hsh = (uint32_t)(kj & 1) * 16u;
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
  float dc0 = 0.f, dc1 = 0.f;
  tile_barrier();

  const int dkey = swz_key(kj);
  char* drow = sD + kj * 128 + 8 * hh;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    if (t * 32 < p.Sq) {                                   // wave-uniform: a block without queries is skipped (its dS columns are never read)
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        st = mfma32(tile_frag(sQ, t, ks, lo), kf[ks], st);
        dp = mfma32(tile_frag(sG, t, ks, lo), vf[ks], dp);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float pd8[8], ds8[8], l8[8], d8[8];
        {
          const f32x4 la = *(const f32x4*)(sLse + t * 32 + 16 * a + 4 * hh), lb = *(const f32x4*)(sLse + t * 32 + 16 * a + 8 + 4 * hh);
          const f32x4 da = *(const f32x4*)(sDel + t * 32 + 16 * a + 4 * hh), db = *(const f32x4*)(sDel + t * 32 + 16 * a + 8 + 4 * hh);
#pragma unroll
          for (int e = 0; e < 4; ++e) { l8[e] = la[e]; l8[4 + e] = lb[e]; d8[e] = da[e]; d8[4 + e] = db[e]; }
        }
        float keep8[8];
        if (DROP) {
          const int g0 = t * 32 + 16 * a;                  // first query row of the group (rows g0 + 4hh + {0..3, 8..11})
          uint32_t hv[8];
          if (shared) {
            uint32_t mine[4], other[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) mine[j] = rng_pair(ctl.s2, (ib_pair + (uint32_t)(g0 + j) * (uint32_t)Stot) >> 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) other[j] = pair_swap(mine[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { hv[j] = odd ? other[j] : mine[j]; hv[4 + j] = odd ? mine[j] : other[j]; }
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) hv[i] = rng_pair(ctl.s2, (ib_own + (uint32_t)(g0 + (i & 3) + 8 * (i >> 2)) * (uint32_t)Stot) >> 1);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const uint32_t idx_par = shared ? 0u : ((ib_own + (uint32_t)(g0 + (i & 3) + 8 * (i >> 2)) * (uint32_t)Stot) & 1u) * 16u;
            const uint32_t u = (hv[i] >> (shared ? hsh : idx_par)) & 0xffffu;
            keep8[i] = u >= ctl.thr ? ctl.inv_keep : 0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = 8 * a + i;
          const int qi = t * 32 + mfma_row(r, hh);
          const float spre = st[r] * p.scale;
          const bool qhi = qi >= sep;
          const bool qlo = !qhi && !(ctl.skip0 && qi == 0);
          const float fq = qhi ? ctl.c1 : (qlo ? ctl.c0 : 1.f);
          const float f = rw ? fq : 1.f;
          const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(spre * f, LOG2E, madd2 - l8[i]));
          const float dpd = DROP ? dp[r] * keep8[i] : dp[r];
          const float ds = pr * (dpd - d8[i]);             // d/d(post-reweight, pre-mask score)
          const float tw = ds * (spre * rwf);
          dc1 = __builtin_fmaf(tw, qhi ? 1.f : 0.f, dc1);
          dc0 = __builtin_fmaf(tw, qlo ? 1.f : 0.f, dc0);
          pd8[i] = DROP ? pr * keep8[i] : pr;
          ds8[i] = ds * f;
        }
        const bf16x8 pf = pack8(pd8);
        const bf16x8 df = pack8(ds8);
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        const u32x4_t dw4 = __builtin_bit_cast(u32x4_t, df);
        // dS sub-tile t/2, [key][64 q]: elements 0-3 are queries 16a + 4hh + 0..3 of the block (16-byte chunk 4(t&1) + 2a), elements 4-7 the same
        // rows of the next chunk
        char* dsub = drow + (t >> 1) * G::TILE;
        *(u32x2_t*)(dsub + (((4 * (t & 1) + 2 * a) ^ dkey) << 4)) = u32x2_t{dw4[0], dw4[1]};
        *(u32x2_t*)(dsub + (((4 * (t & 1) + 2 * a + 1) ^ dkey) << 4)) = u32x2_t{dw4[2], dw4[3]};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = mfma32(tile_frag_tr(sG, t * 32 + 16 * a, dt, lo), pf, dv[dt]);
          dk[dt] = mfma32(tile_frag_tr(sQ, t * 32 + 16 * a, dt, lo), df, dk[dt]);
        }
      }
    }
    if (t + 1 < NB) {                                      // K / V fragments for the next query block
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { kf[ks] = *(const volatile bf16x8*)(kp + ks * 16); vf[ks] = *(const volatile bf16x8*)(vp + ks * 16); }
    }
    if (t == G::TK) {
      // rows [0, R/2) of the Q and dO tiles are dead once EVERY wave is through the first NB / 2 query blocks: the K tile of the dQ contraction lands
      // there (keys [0, R/2) over Q, [R/2, R) over dO) while the other blocks compute -- no fourth tile in LDS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long rk = (long long)b * p.Sk + min(RPI * i + rowh, p.Sk - 1);        // K row R/2 + r sits in LDS row r of the dO tile: R/2 is a multiple of 16, same key
        glds16_raw((const bf16*)p.k + rk * p.ldk + h * 64 + lc_of(i) * 8, (i < 2 ? sQ : sG) + ((i & 1) * G::NT + wave * 64) * 16);
      }
    }
  }
  // ---- d(w0), d(w1): one private slot per wave (four slots per head as in the two-pass layout; with two waves slots 2, 3 are zero)
  if (pb.dw) {
    const bool on = ctl.sep >= 0;
    dc0 = wave_sum(dc0); dc1 = wave_sum(dc1);
    if (lane == 0) {
      if (pb.dw_ws) {
        const long long slot = ((long long)b * p.nh + h) * 4 + wave;
        pb.dw_ws[2 * slot] = on ? dc0 : 0.f; pb.dw_ws[2 * slot + 1] = on ? dc1 : 0.f;
        if (NB == 2) { pb.dw_ws[2 * (slot + 2)] = 0.f; pb.dw_ws[2 * (slot + 2) + 1] = 0.f; }
        if (NB == 3 && wave == 0) { pb.dw_ws[2 * (slot + 3)] = 0.f; pb.dw_ws[2 * (slot + 3) + 1] = 0.f; }
      } else if (on) {
        const float w0 = p.w0[0], w1 = p.w1[0];
        if (w0 >= 0.f && w0 <= 0.5f) atomicAdd(pb.dw + 0, dc0);     // clamp sub-gradient: passes inside and AT the bounds
        if (w1 >= 0.5f && w1 <= 1.f) atomicAdd(pb.dw + 1, dc1);
      }
    }
  }
  tile_barrier();                                          // dS tiles complete (LDS writes of every wave), K tile landed

  // ---- dQ^T[d][q] = sum over keys K^T[d][key] dS^T[key][q] for the wave's 32 queries
  const bool qwave = wave * 32 < p.Sq;                     // wave-uniform
  f32x16 dq[2];
  if (qwave) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    const int tq1 = (wave & 1) ? lo.tr1[1] : lo.tr1[0], tq2 = (wave & 1) ? lo.tr2[1] : lo.tr2[0];   // (a run-time index into the offset arrays would put them in scratch)
    const char* dsub = sD + (wave >> 1) * G::TILE;
#pragma unroll
    for (int kk = 0; kk < 2 * NB; ++kk) {
      const bf16x8 df = join_tr(lds_tr_read(dsub + 16 * kk * 128 + tq1), lds_tr_read(dsub + 16 * kk * 128 + tq2));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(tile_frag_tr(kk < NB ? sQ : sG, 16 * (kk % NB), dt, lo), df, dq[dt]);
    }
  }
  // ---- results leave through LDS.  A result tile has "lane = row, registers = 4 consecutive head dims": stored from registers a wave instruction
  // writes 8 bytes into each of 32 rows that lie 2 * ld bytes apart -- 32 partial cache lines per instruction, 24 such instructions per wave (the
  // stores were 21 of the kernel's 57 us).  Staged [row][64 dims] in the tiles that are dead by now, eight lanes write one 128-byte row segment.
  //   dQ: bf16 tile over the dS tiles;  dK, then dV: f32 tiles over the Q | dO tiles (f32 so that the optional accumulation into the destination
  //   rounds once, exactly as the two-pass kernels do)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                         // every LDS read of the dQ contraction is done
  char* sF = smem;                                         // R rows x 256 bytes
  const int fkey = (kj & 15);
  auto stage_f32 = [&](const f32x16* acc2, float mul) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c16 = 8 * dt + 2 * qd + hh;              // 16-byte chunk of dims dt*32 + 8qd + 4hh .. +3
        *(f32x4*)(sF + kj * 256 + ((c16 ^ fkey) << 4)) =
            f32x4{acc2[dt][4 * qd] * mul, acc2[dt][4 * qd + 1] * mul, acc2[dt][4 * qd + 2] * mul, acc2[dt][4 * qd + 3] * mul};
      }
  };
  auto copy_f32 = [&](void* dst, int ld) {                 // thread: rows RPI i + rowh, dims 8g .. 8g+7 (g = tid & 7)
    const int g = tid & 7;
    const bool acc = pb.accum_dkv != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = RPI * i + rowh;
      if (row >= p.Sk) continue;
      f32x4 a = *(const f32x4*)(sF + row * 256 + (((2 * g) ^ (row & 15)) << 4));
      f32x4 c = *(const f32x4*)(sF + row * 256 + (((2 * g + 1) ^ (row & 15)) << 4));
      bf16* op = (bf16*)dst + ((long long)b * p.Sk + row) * ld + h * 64 + 8 * g;
      if (acc) {
        const bf16x8 old = *(const bf16x8*)op;
        a += f32x4{(float)old[0], (float)old[1], (float)old[2], (float)old[3]};
        c += f32x4{(float)old[4], (float)old[5], (float)old[6], (float)old[7]};
      }
      const bf16x4 ra = f4_to_bf4(a), rc = f4_to_bf4(c);
      *(bf16x8*)op = bf16x8{ra[0], ra[1], ra[2], ra[3], rc[0], rc[1], rc[2], rc[3]};
    }
  };
  if (qwave) {
    const int qi = wave * 32 + l31, qkey = swz_key(qi);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = {dq[dt][4 * qd] * p.scale, dq[dt][4 * qd + 1] * p.scale, dq[dt][4 * qd + 2] * p.scale, dq[dt][4 * qd + 3] * p.scale};
        *(bf16x4*)(sD + qi * 128 + (((4 * dt + qd) ^ qkey) << 4) + 8 * hh) = f4_to_bf4(v);
      }
  }
  stage_f32(dk, p.scale);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {                            // dQ: physical chunk pc of row RPI i + rowh holds the logical chunk lc (same map as the staging DMA)
    const int row = RPI * i + rowh;
    if (row < p.Sq) *(bf16x8*)((bf16*)pb.dq + ((long long)b * p.Sq + row) * pb.lddq + h * 64 + lc_of(i) * 8) = *(const bf16x8*)(sD + row * 128 + (tid & 7) * 16);
  }
  copy_f32(pb.dk, pb.lddk);
  __syncthreads();
  stage_f32(dv, 1.f);
  __syncthreads();
  copy_f32(pb.dv, pb.lddv);
}

MartAttrOnce g_attr_once;
int set_attrs() {
  bool* g_attr_set = g_attr_once.slot();
  if (*g_attr_set) return 0;
  const void* ks[8] = {(const void*)attn_fwd_k<false, 1>, (const void*)attn_fwd_k<false, 2>, (const void*)attn_fwd_k<true, 1>,
                       (const void*)attn_bwd_dq_k<false, 1>, (const void*)attn_bwd_dq_k<false, 2>, (const void*)attn_bwd_dq_k<true, 1>, (const void*)attn_bwd_dkv_k<false>,
                       (const void*)attn_bwd_dkv_k<true>};
  bool ok = true;
  for (int i = 0; i < 8; ++i) ok = ok && hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
  ok = ok && hipFuncSetAttribute((const void*)attn_bwd_fused_k, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS) == hipSuccess;
#ifdef MART_EXPERIMENTS
  ok = ok && hipFuncSetAttribute((const void*)attn_fwd_k<false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * TILE_BYTES) == hipSuccess;
#endif
  if (!ok) {
    mart_set_error("attn: hipFuncSetAttribute failed");
    return -2;
  }
  *g_attr_set = true;
  return 0;
}
}  // namespace

// internal launcher for csrc/precise.hip (mart_attn_fwd_f32 routes its unmasked head-dim-64 calls of evaluation passes here)
int mart_attn_split_launch(const mart_attn_f32_desc* d, void* stream) {
  static const int nw = getenv("MART_ATTN_SPLIT_WAVES") ? atoi(getenv("MART_ATTN_SPLIT_WAVES")) : 8;   // 0.858 vs 0.910 ms at the bench shape
  if (nw == 8 && d->Sq > 128) hipLaunchKernelGGL(attn_split_fwd_k<8>, dim3((d->Sq + 255) / 256, d->nh, d->B), dim3(512), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL(attn_split_fwd_k<4>, dim3((d->Sq + 127) / 128, d->nh, d->B), dim3(256), 0, (hipStream_t)stream, *d);
  MART_LAUNCH_CHECK();
  return 0;
}

#ifdef ATTN_STAMPS
extern "C" int mart_debug_attn_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * 2 * 8 * 32) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int mart_attn_fwd(const mart_attn_fwd_desc* d, void* stream) {
  if (int rc = check_fwd(d)) return rc;
  if (int rc = set_attrs()) return rc;
  const bool text = d->attn_mask || d->sep || d->p_drop > 0.f;   // general instantiation: every text option; the vision kernel takes any prefix length (round 6)
  static const int tpw = getenv("MART_ATTN_TPW") ? atoi(getenv("MART_ATTN_TPW")) : 1;
#ifdef MART_EXPERIMENTS   // (tools/build_variant.sh attention.hip <out.so> -DMART_EXPERIMENTS; MART_ATTN_RES=1)
  static const int res = getenv("MART_ATTN_RES") ? atoi(getenv("MART_ATTN_RES")) : 0;
  const int Stot = d->Lp + d->Sk;
  if (!text && res && d->Sq > 128 && d->Sq <= 512 && Stot <= 512) {          // K / V of the head resident in LDS, one 16-wave workgroup per head
    const int nt = (Stot + 63) / 64, lds = nt * 2 * TILE_BYTES < 16 * 4096 ? 16 * 4096 : nt * 2 * TILE_BYTES;   // (>= the 16 output staging tiles)
    hipLaunchKernelGGL((attn_fwd_k<false, 1, true>), dim3(d->nh, d->B), dim3(RES_NTH), lds, (hipStream_t)stream, *d);
    MART_LAUNCH_CHECK();
    return 0;
  }
#endif
  if (text) hipLaunchKernelGGL((attn_fwd_k<true, 1>), dim3((d->Sq + 127) / 128, d->nh, d->B), dim3(NTH), LDS_BYTES, (hipStream_t)stream, *d);
  else if (tpw == 2 && d->Sq > 128) hipLaunchKernelGGL((attn_fwd_k<false, 2>), dim3((d->Sq + 255) / 256, d->nh, d->B), dim3(NTH), LDS_BYTES, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL((attn_fwd_k<false, 1>), dim3((d->Sq + 32 * FWD_NW - 1) / (32 * FWD_NW), d->nh, d->B), dim3(64 * FWD_NW), LDS_BYTES, (hipStream_t)stream, *d);
  MART_LAUNCH_CHECK();
  return 0;
}

extern "C" int mart_attn_bwd(const mart_attn_bwd_desc* d, void* stream) {
  MART_CHECK(d != nullptr, "attn_bwd: null descriptor");
  if (int rc = check_fwd(&d->f)) return rc;
  MART_CHECK(d->f.lse && d->dctx && d->delta && d->dq && d->dk && d->dv, "attn_bwd: null pointer");
  MART_CHECK(d->lddctx % 8 == 0 && d->lddq % 4 == 0 && d->lddk % 4 == 0 && d->lddv % 4 == 0, "attn_bwd: bad leading dims");
  MART_CHECK(d->f.Lp == 0 || (d->dpk && d->dpv && d->lddp % 4 == 0), "attn_bwd: prefix grads need dpk/dpv");
  if (int rc = set_attrs()) return rc;
  hipStream_t st = (hipStream_t)stream;
  const mart_attn_fwd_desc& f = d->f;
  const bool text = f.attn_mask || f.sep || f.p_drop > 0.f;
  static const int tpw = getenv("MART_ATTN_TPW_DQ") ? atoi(getenv("MART_ATTN_TPW_DQ")) : 2;
  static const int fused = getenv("MART_ATTN_FUSED") ? atoi(getenv("MART_ATTN_FUSED")) : 1;
  if (!text && fused && f.Lp + f.Sk <= 512 && f.Sq <= 512 && f.Sq > 128) {      // one workgroup per head: every key and query row fits (at 100 queries: 0.183 vs 0.117 ms one-pass small kernel, 0.196 vs 0.207 two-pass with a 64-key prefix)
    hipLaunchKernelGGL(attn_bwd_fused_k, dim3(f.nh, f.B), dim3(512), F_LDS, st, *d);
    MART_LAUNCH_CHECK();
    return 0;
  }
  static const int text_fused = getenv("MART_ATTN_TEXT_FUSED") ? atoi(getenv("MART_ATTN_TEXT_FUSED")) : 1;
  const bool al16 = d->lddq % 8 == 0 && d->lddk % 8 == 0 && d->lddv % 8 == 0 && (((uintptr_t)d->dq | (uintptr_t)d->dk | (uintptr_t)d->dv) & 15) == 0;
  // the whole score matrix of a head in one workgroup: one pass.  Text shapes (L = 64 / 96) and the short prefix-free vision shapes (CLIP-B/32
  // geometry: 100 tokens) alike -- every text option of the kernel is off when its operand is null
  if (text_fused && al16 && f.Lp == 0 && f.Sq <= 128 && f.Sk <= 128) {
    const int mx = f.Sq > f.Sk ? f.Sq : f.Sk;                                             // 2 waves (64 x 64), 3 (96 x 96) or 4 (128 x 128)
    const bool drop = f.p_drop > 0.f;
    if (mx > 96) { if (drop) hipLaunchKernelGGL((attn_bwd_text64_k<true, 4>), dim3(f.nh, f.B), dim3(256), 0, st, *d); else hipLaunchKernelGGL((attn_bwd_text64_k<false, 4>), dim3(f.nh, f.B), dim3(256), 0, st, *d); }
    else if (mx > 64) { if (drop) hipLaunchKernelGGL((attn_bwd_text64_k<true, 3>), dim3(f.nh, f.B), dim3(192), 0, st, *d); else hipLaunchKernelGGL((attn_bwd_text64_k<false, 3>), dim3(f.nh, f.B), dim3(192), 0, st, *d); }
    else { if (drop) hipLaunchKernelGGL((attn_bwd_text64_k<true, 2>), dim3(f.nh, f.B), dim3(128), 0, st, *d); else hipLaunchKernelGGL((attn_bwd_text64_k<false, 2>), dim3(f.nh, f.B), dim3(128), 0, st, *d); }
    MART_LAUNCH_CHECK();
    if (d->dw && d->dw_ws && f.sep) {
      hipLaunchKernelGGL(attn_dw_reduce_k, dim3(1), dim3(1024), 0, st, d->dw_ws, (long long)f.B * f.nh * 4, f.w0, f.w1, d->dw);
      MART_LAUNCH_CHECK();
    }
    return 0;
  }
  if (text) hipLaunchKernelGGL((attn_bwd_dq_k<true, 1>), dim3((f.Sq + 127) / 128, f.nh, f.B), dim3(NTH), LDS_BYTES, st, *d);
  else if (tpw == 2 && f.Sq > 128) hipLaunchKernelGGL((attn_bwd_dq_k<false, 2>), dim3((f.Sq + 255) / 256, f.nh, f.B), dim3(NTH), LDS_BYTES, st, *d);
  else hipLaunchKernelGGL((attn_bwd_dq_k<false, 1>), dim3((f.Sq + 127) / 128, f.nh, f.B), dim3(NTH), LDS_BYTES, st, *d);
  MART_LAUNCH_CHECK();
  if (text && d->dw && d->dw_ws && f.sep) {
    const long long n = (long long)f.B * f.nh * ((f.Sq + 127) / 128) * (NTH / 64);
    hipLaunchKernelGGL(attn_dw_reduce_k, dim3(1), dim3(1024), 0, st, d->dw_ws, n, f.w0, f.w1, d->dw);
    MART_LAUNCH_CHECK();
  }
  if (text) hipLaunchKernelGGL(attn_bwd_dkv_k<true>, dim3((f.Lp + f.Sk + 127) / 128, f.nh, f.B), dim3(NTH), LDS_BYTES, st, *d);
  else hipLaunchKernelGGL(attn_bwd_dkv_k<false>, dim3((f.Lp + f.Sk + 127) / 128, f.nh, f.B), dim3(NTH), LDS_BYTES, st, *d);
  MART_LAUNCH_CHECK();
  return 0;
}

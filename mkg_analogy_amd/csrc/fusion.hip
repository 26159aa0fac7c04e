// BertFusion (modeling_unimo.py:400-414) as ONE kernel per direction on gfx950:
//     scores = hidden @ visual^T   (no scale)      probs = softmax(scores)      fusion_output = probs @ visual
// One 8-wave workgroup per (batch element, block of 64 text positions).  The [Lq, Nv] scores never leave the chip; the
// probabilities are written once (bf16) because the backward pass reads them.
//
//   forward   phase 1  S^T[key][q] = sum_d V[key][d] Q[q][d]       d streamed in 64-wide chunks (LDS-DMA, double buffered);
//                                                                   a wave owns key tiles {w, w + 8}: lane = q, registers = keys
//             phase 2  softmax over the keys of a lane's query      lane-local + one LDS exchange between the 8 waves
//             phase 3  O^T[d][q] = sum_key V[key][d] P[q][key]      keys streamed in 32-row chunks, a wave owns 96 of the 768 d
//   backward  B1 dP^T = V dO^T (as phase 1)   B2 dS = P o (dP - rowsum(P o dP))   B3 dQ = dS V (as phase 3)
//             B4 dV[key][d] = sum_q dS[q][key] Q[q][d] + P[q][key] dO[q][d], added to the fp32 vision-stream gradient in place
//
// All LDS operand images are 128-byte rows of 64 bf16 with the 16-byte chunk index XORed by a bit-reversed key of the row pair
// (same image as the attention kernels): plain fragment reads (ds_read_b128, lane = row) and transposed fragment reads
// (ds_read_b64_tr_b16, lane = column) are both conflict-free on it.
#include <cstdlib>
#include "common.h"
#include "mart_hip.h"

namespace {

constexpr int FT = 512;                    // 8 waves
constexpr int LDS_MAX = 160 * 1024;
constexpr int RED_BYTES = 4096;            // [2][8 waves][64 queries] floats
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int swz_key(int row) {
  const int k = (row >> 1) & 7;
  return ((k & 1) << 2) | (k & 2) | (k >> 2);
}
__device__ __forceinline__ void dma_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
}
// The LDS-DMA goes out as opaque assembly: through the builtin the compiler, which cannot tell which LDS bytes a DMA writes, puts s_waitcnt vmcnt(0)
// in front of the next LDS read -- i.e. right behind the prefetch of the NEXT chunk, which it thereby serialises with the compute of this one
// (attention.hip, docs/LAB_r01-r05.md section 4.2).  Every consumer sits behind dma_wait_barrier(), which carries the wait explicitly.
#ifndef FUSION_RAW
#define FUSION_RAW 0   // measured neutral on both kernels (65.6 vs 66.6 us forward, 288 vs 287 us backward): the builtin stays
#endif
__device__ __forceinline__ void fdma16(const void* src, void* dst) {
  if (FUSION_RAW) glds16_raw(src, dst); else glds16(src, dst);
}
// 64 x 64 image of rows r0 .. r0+63 (clamped to n-1), columns c0 .. c0+63 of a row-major bf16 matrix: one 16-byte chunk per thread
__device__ __forceinline__ void stage64(const bf16* base, int ld, int r0, int n, int c0, char* img, int tid, int wave) {
  const int row = tid >> 3, pc = tid & 7, lc = pc ^ swz_key(row);
  fdma16(base + (long long)min(r0 + row, n - 1) * ld + c0 + lc * 8, img + wave * 1024);
}
// per-lane byte offsets of the fragment reads (loop invariant)
struct Offs {
  int pf[4];            // plain fragment, k-step ks (16 columns): row l31, chunk (2 ks + hh) ^ key
  int trl[2], trh[2];   // transposed fragment, 32-column half dt: rows 8 hh + (p >> 2) and + 4
};
__device__ __forceinline__ Offs make_offs(int lane) {
  Offs o;
  const int l31 = lane & 31, hh = lane >> 5, pp = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) o.pf[ks] = l31 * 128 + (((ks * 2 + hh) ^ swz_key(l31)) << 4);
  const int rr = 8 * hh + (pp >> 2);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int col = dt * 32 + g1 * 16 + (pp & 3) * 4, lc = col >> 3, bo = (col & 7) * 2;
    o.trl[dt] = rr * 128 + ((lc ^ swz_key(rr)) << 4) + bo;
    o.trh[dt] = (rr + 4) * 128 + ((lc ^ swz_key(rr + 4)) << 4) + bo;
  }
  return o;
}
// X[i = row half*32 + l31][k = 16 ks + 8 hh + 0..7] of an image (rows on lanes)
__device__ __forceinline__ bf16x8 frag(const char* img, int half, int ks, const Offs& o) {
  return *(const bf16x8*)(img + half * 4096 + o.pf[ks]);
}
// X^T[i = column dt*32 + l31][k = rows base + 8 hh + 0..7] (columns on lanes); base a multiple of 16
__device__ __forceinline__ bf16x8 frag_tr(const char* img, int base, int dt, const Offs& o) {
  // dt is a run-time value in the d-tile loops (d tile = wave * DT + i): indexing o.trl[dt] put the four offsets into scratch memory and a
  // scratch_load in front of every transposed read; a select keeps them in registers
  const int lo_ = dt ? o.trl[1] : o.trl[0], hi_ = dt ? o.trh[1] : o.trh[0];
  return join_tr(lds_tr_read(img + base * 128 + lo_), lds_tr_read(img + base * 128 + hi_));
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

struct Geo {                               // shared index arithmetic of the two kernels
  int b, q0, nq, QT, KT32, KT64, NDC, SB, PB, VC;
};
__device__ __forceinline__ Geo make_geo(int Lq, int Nv, int H, int qblock) {
  Geo g;
  g.b = blockIdx.x; g.q0 = qblock * 64; g.nq = min(64, Lq - g.q0); g.QT = (g.nq + 31) / 32;
  g.KT32 = (Nv + 31) / 32; g.KT64 = (Nv + 63) / 64; g.NDC = H / 64;
  g.SB = (g.KT64 + 1) * 8192;              // one d-chunk: V images + one Q image
  g.PB = 2 * g.KT64 * 4096;                // [2 query tiles][KT64] images of 32 x 64
  g.VC = g.NDC * 4096;                     // one 32-key chunk of V: NDC images of 32 x 64
  return g;
}

// ---- phase 1 / B1: acc[t][u] = S^T tile (keys kt[t]*32.., queries u*32..) = sum_d V Q^T
__device__ __forceinline__ void scores_T(const bf16* Vb, int ldv, int Nv, const bf16* Qb, int ldq, const Geo& g, char* smem, int tid, int wave,
                                         const Offs& o, const int (&kt)[2], const bool (&val)[2], f32x16 (&acc)[2][2]) {
  auto stage = [&](int dc, char* buf) {
    for (int j = 0; j < g.KT64; ++j) stage64(Vb, ldv, j * 64, Nv, dc * 64, buf + j * 8192, tid, wave);
    stage64(Qb, ldq, 0, g.nq, dc * 64, buf + g.KT64 * 8192, tid, wave);
  };
#pragma unroll
  for (int t = 0; t < 2; ++t) { acc[t][0] = zero16(); acc[t][1] = zero16(); }
  stage(0, smem);
  for (int dc = 0; dc < g.NDC; ++dc) {
    dma_wait_barrier();
    if (dc + 1 < g.NDC) stage(dc + 1, smem + ((dc + 1) & 1) * g.SB);
    const char* buf = smem + (dc & 1) * g.SB;
    const char* qi = buf + g.KT64 * 8192;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 q0f = frag(qi, 0, ks, o);
      bf16x8 q1f = q0f;
      if (g.QT > 1) q1f = frag(qi, 1, ks, o);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (!val[t]) continue;
        const bf16x8 kf = frag(buf + (kt[t] >> 1) * 8192, kt[t] & 1, ks, o);
        acc[t][0] = mfma32(kf, q0f, acc[t][0]);
        if (g.QT > 1) acc[t][1] = mfma32(kf, q1f, acc[t][1]);
      }
    }
  }
}
// sum / max over the 8 waves of a per-query value held by lanes hh == 0
template <bool MAX>
__device__ __forceinline__ void cross_wave(float (&v)[2], float* red, int wave, int l31, int hh) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float w = __shfl_xor(v[u], 32, 64);
    v[u] = MAX ? fmaxf(v[u], w) : v[u] + w;
    if (hh == 0) red[wave * 64 + u * 32 + l31] = v[u];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float a = red[u * 32 + l31];
#pragma unroll
    for (int w = 1; w < 8; ++w) { const float x = red[w * 64 + u * 32 + l31]; a = MAX ? fmaxf(a, x) : a + x; }   // wave order: fixed
    v[u] = a;
  }
}
// registers (lane = q, regs = keys) -> image [q tile u][key tile64] rows of 64 keys
__device__ __forceinline__ void put_image(char* pimg, const Geo& g, const int (&kt)[2], const bool (&val)[2], const f32x16 (&x)[2][2], int l31, int hh) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (kt[t] >= 2 * g.KT64) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u >= g.QT) continue;
      char* img = pimg + (u * g.KT64 + (kt[t] >> 1)) * 4096 + l31 * 128 + 8 * hh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (val[t]) v = f32x4{x[t][u][4 * q], x[t][u][4 * q + 1], x[t][u][4 * q + 2], x[t][u][4 * q + 3]};
        *(bf16x4*)(img + ((((kt[t] & 1) * 4 + q) ^ swz_key(l31)) << 4)) = f4_to_bf4(v);
      }
    }
  }
}
// ---- phase 3 / B3: o[i][u] = O^T tile (d = (wave*DT + i)*32.., queries u*32..) = sum_key V^T P^T, P from the image at pimg
template <int DT>
__device__ __forceinline__ void apply_V(const bf16* Vb, int ldv, int Nv, const Geo& g, const char* pimg, char* vb0, int nbuf, int tid, int wave,
                                        const Offs& o, int l31, int hh, f32x16 (&acc)[DT][2]) {
  auto stage = [&](int kc, char* buf) {
    const int row = (tid >> 3) & 31, pc = tid & 7, lc = pc ^ swz_key(row);
    const bf16* src = Vb + (long long)min(kc * 32 + row, Nv - 1) * ldv + lc * 8;
    for (int s = 0; s < g.NDC / 2; ++s) {
      const int img = s * 2 + (tid >> 8);
      fdma16(src + img * 64, buf + img * 4096 + (wave & 3) * 1024);
    }
  };
#pragma unroll
  for (int i = 0; i < DT; ++i) { acc[i][0] = zero16(); acc[i][1] = zero16(); }
  stage(0, vb0);
  for (int kc = 0; kc < g.KT32; ++kc) {
    dma_wait_barrier();
    if (nbuf == 2 && kc + 1 < g.KT32) stage(kc + 1, vb0 + ((kc + 1) & 1) * g.VC);
    const char* vb = vb0 + (nbuf == 2 ? (kc & 1) * g.VC : 0);
    const char* pi = pimg + (kc >> 1) * 4096;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int po = l31 * 128 + ((((kc & 1) * 4 + ks * 2 + hh) ^ swz_key(l31)) << 4);
      const bf16x8 p0 = *(const bf16x8*)(pi + po);
      bf16x8 p1 = p0;
      if (g.QT > 1) p1 = *(const bf16x8*)(pi + g.KT64 * 4096 + po);
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const int dtile = wave * DT + i;
        const bf16x8 vf = frag_tr(vb + (dtile >> 1) * 4096, ks * 16, dtile & 1, o);
        acc[i][0] = mfma32(vf, p0, acc[i][0]);
        if (g.QT > 1) acc[i][1] = mfma32(vf, p1, acc[i][1]);
      }
    }
    if (nbuf == 1) { __syncthreads(); if (kc + 1 < g.KT32) stage(kc + 1, vb0); }
  }
}
template <int DT, bool F16 = false>
__device__ __forceinline__ void store_rows(bf16* out, int ldo, const Geo& g, int Lq, int wave, int l31, int hh, const f32x16 (&acc)[DT][2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = u * 32 + l31;
    if (u >= g.QT || q >= g.nq) continue;
    bf16* row = out + ((long long)g.b * Lq + g.q0 + q) * ldo;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c)
      {
        const f32x4 v = {acc[i][u][4 * c], acc[i][u][4 * c + 1], acc[i][u][4 * c + 2], acc[i][u][4 * c + 3]};
        *(bf16x4*)(row + (wave * DT + i) * 32 + 8 * c + 4 * hh) = F16 ? f4_to_h4raw(v) : f4_to_bf4(v);
      }
  }
}

// =========================================================================== forward
template <int DT>
__global__ __launch_bounds__(FT, 1) void fusion_fwd_k(mart_fusion_fwd_desc p, int nbuf, int red_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = make_geo(p.Lq, p.Nv, p.H, blockIdx.y);
  const Offs o = make_offs(lane);
  const bf16* Qb = (const bf16*)p.q + ((long long)g.b * p.Lq + g.q0) * p.ldq;
  const bf16* Vb = (const bf16*)p.v + (long long)g.b * p.Nv * p.ldv;
  float* red = (float*)(smem + red_off);
  const int kt[2] = {wave, wave + 8};
  const bool val[2] = {kt[0] < g.KT32, kt[1] < g.KT32};

  f32x16 acc[2][2];
  scores_T(Vb, p.ldv, p.Nv, Qb, p.ldq, g, smem, tid, wave, o, kt, val, acc);

  // ---- softmax over the keys of each query (columns of S^T): lane-local, then across the two lane halves and the 8 waves
  float mx[2] = {-3.0e38f, -3.0e38f};
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (val[t] && kt[t] * 32 + mfma_row(r, hh) < p.Nv) mx[u] = fmaxf(mx[u], acc[t][u][r]);
  cross_wave<true>(mx, red, wave, l31, hh);              // its barrier also retires the phase-1 buffers
  float sum[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = val[t] && kt[t] * 32 + mfma_row(r, hh) < p.Nv;
        const float e = ok ? __builtin_amdgcn_exp2f((acc[t][u][r] - mx[u]) * LOG2E) : 0.f;
        acc[t][u][r] = e; sum[u] += e;
      }
  cross_wave<false>(sum, red + 512, wave, l31, hh);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float inv = 1.f / sum[u];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] *= inv;
  }
  char* pimg = smem;
  put_image(pimg, g, kt, val, acc, l31, hh);
  __syncthreads();
  // probabilities -> HBM, row-major [Lq][ldp] (columns Nv .. ldp are zero): 16 bytes per thread, 8 threads per 128-byte row segment
  for (int i = tid; i < g.QT * g.KT64 * 256; i += FT) {
    const int u = i / (g.KT64 * 256), rem = i - u * (g.KT64 * 256), kj = rem >> 8, row = (rem >> 3) & 31, pc = rem & 7;
    const int col = kj * 64 + ((pc ^ swz_key(row)) << 3);
    if (col < p.ldp && u * 32 + row < g.nq)
      *(bf16x8*)((bf16*)p.probs + ((long long)g.b * p.Lq + g.q0 + u * 32 + row) * p.ldp + col) = *(const bf16x8*)(pimg + (u * g.KT64 + kj) * 4096 + row * 128 + pc * 16);
  }
  f32x16 ov[DT][2];
  apply_V<DT>(Vb, p.ldv, p.Nv, g, pimg, smem + g.PB, nbuf, tid, wave, o, l31, hh, ov);
  store_rows<DT>((bf16*)p.out, p.ldo, g, p.Lq, wave, l31, hh, ov);
  if (p.out_f16) store_rows<DT, true>((bf16*)p.out_f16, p.ldo, g, p.Lq, wave, l31, hh, ov);
}


// =========================================================================== backward
// One launch per 64-query block (the host loops: the blocks of a batch element add into the same rows of the vision gradient, and
// sequential launches keep that sum in a fixed order without atomics).
template <int DT>
__global__ __launch_bounds__(FT, 1) void fusion_bwd_k(mart_fusion_bwd_desc p, int qblock, int nbuf, int red_off, int dbg_) {
#ifdef MART_EXPERIMENTS
  const int dbg = dbg_;                                  // timing knock-outs (harness builds only): 1 no dV loads, 2 no dV stores, 4 no B4, 8 no B3
#else
  constexpr int dbg = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = make_geo(p.Lq, p.Nv, p.H, qblock);
  const Offs o = make_offs(lane);
  const bf16* Qb = (const bf16*)p.q + ((long long)g.b * p.Lq + g.q0) * p.ldq;
  const bf16* Ob = (const bf16*)p.dout + ((long long)g.b * p.Lq + g.q0) * p.lddo;
  const bf16* Vb = (const bf16*)p.v + (long long)g.b * p.Nv * p.ldv;
  float* red = (float*)(smem + red_off);
  const int kt[2] = {wave, wave + 8};
  const bool val[2] = {kt[0] < g.KT32, kt[1] < g.KT32};

  // ---- B1: dP^T = V dO^T
  f32x16 dp[2][2];
  scores_T(Vb, p.ldv, p.Nv, Ob, p.lddo, g, smem, tid, wave, o, kt, val, dp);
  // ---- B2: dS = P o (dP - rowsum(P o dP)); P comes back from HBM in the register layout of dP (8 bytes per lane and group)
  f32x16 pr[2][2];
  float rs[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bf16* prow = (const bf16*)p.probs + ((long long)g.b * p.Lq + g.q0 + u * 32 + l31) * p.ldp;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = kt[t] * 32 + 8 * c + 4 * hh;
        bf16x4 v = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        // query rows past the end of a partial block (Lq not a multiple of 32: the reference pads a batch to its longest example, data_module.py:113-119)
        // keep P = 0, hence dS = 0: their staged Q / dO rows are clamped copies of the last row and must not reach dV
        if (val[t] && u < g.QT && u * 32 + l31 < g.nq && col < p.ldp) v = *(const bf16x4*)(prow + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) { pr[t][u][4 * c + e] = (float)v[e]; rs[u] += (float)v[e] * dp[t][u][4 * c + e]; }
      }
    }
  cross_wave<false>(rs, red, wave, l31, hh);             // its barrier also retires the B1 buffers
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[t][u][r] = pr[t][u][r] * (dp[t][u][r] - rs[u]);
  char* simg = smem;                                     // dS image
  char* pimg = smem + g.PB;                              // P image
  put_image(simg, g, kt, val, dp, l31, hh);
  put_image(pimg, g, kt, val, pr, l31, hh);
  __syncthreads();

  // ---- B4: dV[key][d] += sum_q dS[q][key] Q[q][d] + P[q][key] dO[q][d].  The wave keeps its key tiles' dS^T / P^T fragments
  // (contraction over the 32 / 64 queries) in registers and walks the 24 d tiles; Q and dO come through LDS, 64 columns at a time.
  bf16x8 aS[2][4], aP[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      aS[t][ks] = bf16x8{}; aP[t][ks] = bf16x8{};
      if (val[t] && ks < 2 * g.QT) {
        const int off = ((ks >> 1) * g.KT64 + (kt[t] >> 1)) * 4096;
        aS[t][ks] = frag_tr(simg + off, (ks & 1) * 16, kt[t] & 1, o);
        aP[t][ks] = frag_tr(pimg + off, (ks & 1) * 16, kt[t] & 1, o);
      }
    }
  // The P image is dead once every wave holds its fragments: its place becomes eight wave-private [32][36] fp32 scratch tiles through
  // which a result tile goes from the MFMA layout (lane = key, 4 d per register group) to rows (8 lanes x 16 bytes per 128-byte row
  // segment), so that the read-modify-write of the fp32 gradient is fully coalesced: 8 line requests per instruction instead of 32
  // (lane = d with 4-byte accesses was bound by the store issue rate, lane = key with 16-byte accesses by the request rate).
  __syncthreads();
  float* scr = (float*)(smem + g.PB) + wave * (32 * 36);
  char* st0 = smem + g.PB + max(g.PB, 8 * 32 * 36 * 4);
  auto stage4 = [&](int dc, char* buf) {
    stage64(Qb, p.ldq, 0, g.nq, dc * 64, buf, tid, wave);
    stage64(Ob, p.lddo, 0, g.nq, dc * 64, buf + 8192, tid, wave);
  };
  const int rrow = lane >> 3, rcol = (lane & 7) * 4;            // row layout: 8 rows per instruction
  // dv_f32 == NULL (round 6: the vision-stream gradient carried in bf16): the read-modify-write goes to the bf16 tensor alone (2 + 2 bytes per element
  // instead of 4 + 4 + 2)
  const bool gb16 = p.dv_f32 == nullptr;
  float* dvf = gb16 ? nullptr : p.dv_f32 + (long long)g.b * p.Nv * p.lddv + rcol;
  bf16* dvb = p.dv_bf16 ? (bf16*)p.dv_bf16 + (long long)g.b * p.Nv * p.lddvb + rcol : nullptr;
  stage4(0, st0);
  for (int dc = 0; dc < ((dbg & 4) ? 0 : g.NDC); ++dc) {
    // The gradient rows of this chunk are requested BEFORE the wait for the chunk's Q / dO images: vmcnt is one counter for the DMA,
    // these loads and the stores of the previous chunk, so the wait below drains all three together -- one memory round trip per chunk.
    f32x4 old[2][2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int key = kt[t] * 32 + it * 8 + rrow;
          old[dt][t][it] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (val[t] && key < p.Nv && !(dbg & 1))
            old[dt][t][it] = gb16 ? bf4_to_f4(*(const bf16x4*)(dvb + (long long)key * p.lddvb + dc * 64 + dt * 32)) : *(const f32x4*)(dvf + (long long)key * p.lddv + dc * 64 + dt * 32);
        }
    dma_wait_barrier();
    if (dc + 1 < g.NDC) stage4(dc + 1, st0 + ((dc + 1) & 1) * 16384);
    const char* buf = st0 + (dc & 1) * 16384;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int dcol = dc * 64 + dt * 32;
      f32x16 a[2] = {zero16(), zero16()};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks >= 2 * g.QT) continue;
        const bf16x8 bq = frag_tr(buf, ks * 16, dt, o), bo = frag_tr(buf + 8192, ks * 16, dt, o);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (!val[t]) continue;
          a[t] = mfma32(bq, aS[t][ks], a[t]);              // D[i = d][j = key]
          a[t] = mfma32(bo, aP[t][ks], a[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (!val[t]) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) *(f32x4*)(scr + l31 * 36 + 8 * c + 4 * hh) = f32x4{a[t][4 * c], a[t][4 * c + 1], a[t][4 * c + 2], a[t][4 * c + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int key = kt[t] * 32 + it * 8 + rrow;
          const f32x4 x = old[dt][t][it] + *(const f32x4*)(scr + (it * 8 + rrow) * 36 + rcol);
          if (key < p.Nv && !(dbg & 2)) {
            if (!gb16) *(f32x4*)(dvf + (long long)key * p.lddv + dcol) = x;
            if (dvb) *(bf16x4*)(dvb + (long long)key * p.lddvb + dcol) = f4_to_bf4(x);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  // ---- B3: dQ = dS V (the V chunks land where the P image was: every wave is done with it)
  __syncthreads();
  if (dbg & 8) return;
  f32x16 dq[DT][2];
  apply_V<DT>(Vb, p.ldv, p.Nv, g, simg, smem + g.PB, nbuf, tid, wave, o, l31, hh, dq);
  store_rows<DT>((bf16*)p.dq, p.lddq, g, p.Lq, wave, l31, hh, dq);
}

struct Plan { int lds, nbuf, red_off; bool ok; };
Plan make_plan(int Lq, int Nv, int H, bool bwd = false) {
  Plan pl{0, 0, 0, false};
  if (H != 768 || Lq <= 0 || Nv <= 0 || Nv > 512) return pl;      // any Lq: the last 32-row query block may be partial (rows clamped on load, masked in B2, skipped on store)
  const int KT64 = (Nv + 63) / 64, SB = (KT64 + 1) * 8192, PB = 2 * KT64 * 4096, VC = (H / 64) * 4096;
  const int p1 = 2 * SB;
  pl.nbuf = PB + 2 * VC + RED_BYTES <= LDS_MAX ? 2 : 1;
  const int p3 = PB + pl.nbuf * VC;
  pl.red_off = p1 > p3 ? p1 : p3;
  const int b4 = PB + (PB > 36864 ? PB : 36864) + 32768;   // B4: dS image, P image / transposition scratch, two Q / dO chunk buffers
  if (bwd && b4 > pl.red_off) pl.red_off = b4;
  pl.lds = pl.red_off + RED_BYTES;
  pl.ok = pl.lds <= LDS_MAX;
  return pl;
}
MartAttrOnce g_attr_fwd, g_attr_bwd;
}  // namespace

extern "C" int mart_fusion_supported(int Lq, int Nv, int H) { return make_plan(Lq, Nv, H).ok && make_plan(Lq, Nv, H, true).ok ? 1 : 0; }

extern "C" int mart_fusion_fwd(const mart_fusion_fwd_desc* d, void* stream) {
  MART_CHECK(d && d->q && d->v && d->out && d->probs && d->B > 0, "fusion_fwd: bad args");
  const Plan pl = make_plan(d->Lq, d->Nv, d->H);
  MART_CHECK(pl.ok, "fusion_fwd: unsupported shape (H = 768, Nv <= 512: mart_fusion_supported)");
  const int KT64 = (d->Nv + 63) / 64;
  MART_CHECK(d->ldq >= d->H && d->ldv >= d->H && d->ldo >= d->H && d->ldq % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 4 == 0, "fusion_fwd: row strides must cover H (ldq, ldv multiples of 8)");
  MART_CHECK(d->ldp >= d->Nv && d->ldp % 8 == 0 && d->ldp <= KT64 * 64, "fusion_fwd: ldp must be a multiple of 8 in [Nv, 64 * ceil(Nv / 64)]");
  MART_CHECK(((uintptr_t)d->q | (uintptr_t)d->v | (uintptr_t)d->probs) % 16 == 0 && (uintptr_t)d->out % 8 == 0, "fusion_fwd: operands must be 16-byte aligned");
  bool* once = g_attr_fwd.slot();
  if (!*once) {
    if (hipFuncSetAttribute((const void*)fusion_fwd_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess) { mart_set_error("fusion_fwd: cannot raise the dynamic LDS limit"); return -2; }
    *once = true;
  }
  hipLaunchKernelGGL((fusion_fwd_k<3>), dim3(d->B, (d->Lq + 63) / 64), dim3(FT), pl.lds, (hipStream_t)stream, *d, pl.nbuf, pl.red_off);
  MART_LAUNCH_CHECK();
  return 0;
}

extern "C" int mart_fusion_bwd(const mart_fusion_bwd_desc* d, void* stream) {
  MART_CHECK(d && d->q && d->v && d->dout && d->probs && d->dq && (d->dv_f32 || d->dv_bf16) && d->B > 0, "fusion_bwd: bad args");
  const Plan pl = make_plan(d->Lq, d->Nv, d->H, true);
  MART_CHECK(pl.ok, "fusion_bwd: unsupported shape (mart_fusion_supported)");
  MART_CHECK(d->ldq >= d->H && d->ldv >= d->H && d->lddo >= d->H && d->lddq >= d->H && (!d->dv_f32 || d->lddv >= d->H) && d->ldq % 8 == 0 && d->ldv % 8 == 0 && d->lddo % 8 == 0 &&
             d->lddq % 4 == 0 && (!d->dv_bf16 || d->lddvb >= d->H), "fusion_bwd: row strides must cover H (ldq, ldv, lddo multiples of 8)");
  MART_CHECK(d->ldp >= d->Nv && d->ldp % 8 == 0 && d->ldp <= ((d->Nv + 63) / 64) * 64, "fusion_bwd: ldp must be a multiple of 8 in [Nv, 64 * ceil(Nv / 64)]");
  MART_CHECK(((uintptr_t)d->q | (uintptr_t)d->v | (uintptr_t)d->dout) % 16 == 0 && ((uintptr_t)d->dq | (uintptr_t)d->probs) % 8 == 0, "fusion_bwd: operands must be 16-byte aligned");
  // the vision-stream gradient is read and written as f32x4 (and its bf16 copy as bf16x4) at columns that are multiples of 4
  MART_CHECK(!d->dv_f32 || (d->lddv % 4 == 0 && (uintptr_t)d->dv_f32 % 16 == 0), "fusion_bwd: dv_f32 must be 16-byte aligned with lddv a multiple of 4");
  MART_CHECK(!d->dv_bf16 || (d->lddvb % 4 == 0 && (uintptr_t)d->dv_bf16 % 8 == 0), "fusion_bwd: dv_bf16 must be 8-byte aligned with lddvb a multiple of 4");
  bool* once = g_attr_bwd.slot();
  if (!*once) {
    if (hipFuncSetAttribute((const void*)fusion_bwd_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess) { mart_set_error("fusion_bwd: cannot raise the dynamic LDS limit"); return -2; }
    *once = true;
  }
  for (int qb = 0; qb < (d->Lq + 63) / 64; ++qb) {
#ifdef MART_EXPERIMENTS
    static const int dbg = getenv("MART_FUSION_DBG") ? atoi(getenv("MART_FUSION_DBG")) : 0;
#else
    const int dbg = 0;
#endif
    hipLaunchKernelGGL((fusion_bwd_k<3>), dim3(d->B), dim3(FT), pl.lds, (hipStream_t)stream, *d, qb, pl.nbuf, pl.red_off, dbg);
    MART_LAUNCH_CHECK();
  }
  return 0;
}

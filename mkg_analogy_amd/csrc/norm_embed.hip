// HBM-bound row kernels: LayerNorm family, embeddings, row softmax, transposes.
// One wave64 per row, 16-byte vector accesses, fp32 statistics (two-pass in registers).
#include <cstdlib>
#include "common.h"
#include "mart_hip.h"

namespace {
// Streaming hints of the LayerNorm kernels.  bit 0: nontemporal loads in ln_fwd, bit 1: nontemporal f32 stores, bit 2: nontemporal bf16
// stores.  Measured in the step (same box, pipelined ln_bwd): 0 -> 93.0 ms, 1 -> 92.6, 2 -> 92.55, 3 -> 92.3; the bf16 outputs are the next
// GEMM's A operand and stay cacheable.
#ifndef LN_NT
#define LN_NT 3
#endif
template <typename T> __device__ __forceinline__ T ldx(const T* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <typename T> __device__ __forceinline__ void stx(T* p, T v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
constexpr int WPB = 4;            // waves (rows in flight) per workgroup
constexpr int TPB = 64 * WPB;
constexpr int VMAX_ALL = 4;       // float4 per lane cached in registers -> H <= 1024 (kernels are instantiated for 3 = H 768, and 4)

inline int row_grid(int M) {
  int g = (M + WPB - 1) / WPB;
  return g > 4096 ? 4096 : (g < 1 ? 1 : g);
}

// Dropout decisions of the four consecutive elements o .. o+3 (o % 4 == 0: H is a multiple of 256) as keep factors: TWO pair hashes and no
// seed fold per element (common.h dropout_keep costs two mix32 per element; the text-stream LayerNorm kernels spent more VALU time on it than on
// the normalisation).  Identical decisions: elements 2j, 2j+1 take the low / high 16 bits of rng_pair(s2, pair j); s2 folds the upper pair bits,
// which are 0 for every tensor below 2^33 elements (s2_lo = rng_seedmix(seed, 0), computed once per kernel).
__device__ __forceinline__ f32x4 dropout_scale4(uint64_t seed, uint32_t s2_lo, long long o, uint32_t thr, float inv_keep) {
  const uint64_t pair = (uint64_t)o >> 1;
  const uint32_t hi = (uint32_t)(pair >> 32);
  const uint32_t s2 = hi == 0 ? s2_lo : rng_seedmix(seed, hi);
  const uint32_t h0 = rng_pair(s2, (uint32_t)pair), h1 = rng_pair(s2, (uint32_t)pair + 1u);
  return f32x4{(h0 & 0xffffu) >= thr ? inv_keep : 0.f, (h0 >> 16) >= thr ? inv_keep : 0.f, (h1 & 0xffffu) >= thr ? inv_keep : 0.f, (h1 >> 16) >= thr ? inv_keep : 0.f};
}

// ------------------------------------------------------------------ LayerNorm forward
template <int VMAX>
__global__ __launch_bounds__(TPB) void ln_fwd_k(mart_ln_fwd_desc p) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * WPB;
  const int nv = p.H / 256;                       // float4 per lane (H multiple of 256), nv <= VMAX
  const float inv_keep = 1.f / (1.f - p.p_drop);
  const uint32_t s2_lo = rng_seedmix(p.seed, 0), thr16 = dropout_thr16(p.p_drop);
  const bf16* yb = (const bf16*)p.y_bf16;
  for (int m = wave_g; m < p.M; m += nwaves) {
    f32x4 x[VMAX];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      if (v < nv) {
        const int c = (v * 64 + lane) * 4;
        const long long o = (long long)m * p.H + c;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (p.x_f32) t = ldx((const f32x4*)(p.x_f32 + (p.x_rows ? (long long)p.x_rows[m] * p.H + c : o)), (LN_NT & 1) != 0);
        if (yb) {
          f32x4 y = bf4_to_f4(ldx((const bf16x4*)(yb + o), (LN_NT & 1) != 0));
          if (p.p_drop > 0.f) {
            y = y * dropout_scale4(p.seed, s2_lo, o, thr16, inv_keep);
          }
          t += y;
        }
        if (p.y_f32) {
          f32x4 y = ldx((const f32x4*)(p.y_f32 + o), (LN_NT & 1) != 0);
          if (p.p_drop > 0.f) {
            y = y * dropout_scale4(p.seed, s2_lo, o, thr16, inv_keep);
          }
          t += y;
        }
        x[v] = t;
        sum += t[0] + t[1] + t[2] + t[3];
        if (p.s_out) stx((f32x4*)(p.s_out + o), t, (LN_NT & 2) != 0);
      }
    }
    const float mean = wave_sum(sum) / (float)p.H;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      if (v < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { float dlt = x[v][e] - mean; sq += dlt * dlt; }
      }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.H + p.eps);
    if (lane == 0) { p.mean[m] = mean; p.rstd[m] = rstd; }
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      if (v < nv) {
        const int c = (v * 64 + lane) * 4;
        const long long o = (long long)m * p.H + c;
        f32x4 g = *(const f32x4*)(p.gamma + c), b = *(const f32x4*)(p.beta + c), y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (x[v][e] - mean) * rstd * g[e] + b[e];
        if (p.out_f32) stx((f32x4*)(p.out_f32 + o), y, (LN_NT & 2) != 0);
        if (p.out_bf16) stx((bf16x4*)((bf16*)p.out_bf16 + o), f4_to_bf4(y), (LN_NT & 4) != 0);
        if (p.out_f16) stx((bf16x4*)((bf16*)p.out_f16 + o), f4_to_h4raw(y), (LN_NT & 4) != 0);
        if (p.out_split3) {                                   // [hi | lo | hi]: the A operand of the fp32-accurate path's split GEMMs (mart_split_bf16x3, role 0)
          const bf16x4 hi = f4_to_bf4(y);
          const bf16x4 lo = f4_to_bf4(y - bf4_to_f4(hi));
          bf16* d3 = (bf16*)p.out_split3 + (long long)m * 3 * p.H + c;
          *(bf16x4*)d3 = hi; *(bf16x4*)(d3 + p.H) = lo; *(bf16x4*)(d3 + 2 * p.H) = hi;
        }
      }
  }
}

// ---- the two LayerNorm shapes of the vision stream (24 launches each per step, 100 608 rows), straight-line and software-pipelined.
// The general kernels take every operand as an option: each `if (p.x)` around a load or a store is a join at which the compiler can no
// longer count the outstanding memory operations and waits vmcnt(0) -- which also drains the row that was prefetched.  Here the operand set
// is fixed, rows past the end are clamped duplicates of the last row (same values to the same addresses; their dgamma / dbeta share is
// multiplied by 0), every wave runs the same trip count, and the waits are counted: two rows are really in flight per wave.
template <int VMAX, int OUT = 0>      // OUT: 0 = bf16 output, 1 = fp16 output (out_h), 2 = both (the fp16 forward operand and the bf16 one the backward pass reads)
__global__ __launch_bounds__(TPB) void ln_fwd_fast_k(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                     int M, bf16* __restrict__ out, float* __restrict__ mean_o, float* __restrict__ rstd_o, bf16* __restrict__ out_h = nullptr, int rev = 0) {
  constexpr int H = VMAX * 256;
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * WPB;
  f32x4 gam[VMAX], bet[VMAX];
#pragma unroll
  for (int v = 0; v < VMAX; ++v) { gam[v] = *(const f32x4*)(gamma + (v * 64 + lane) * 4); bet[v] = *(const f32x4*)(beta + (v * 64 + lane) * 4); }
  const int iters = (M + nwaves - 1) / nwaves;
  float keep_mean = 0.f, keep_rstd = 0.f;                 // lane k keeps the statistics of the wave's k-th row (written after the loop: no store under a lane predicate inside it)
  // rev: the sweep runs from the LAST row to the first -- the rows the producing kernel wrote last are the ones the memory-side cache still holds
  auto row_of = [&](int k) { const int r = min(wave_g + k * nwaves, M - 1); return rev ? M - 1 - r : r; };
  auto fetch = [&](int k, f32x4 (&r)[VMAX]) {
    const float* src = x + (long long)row_of(k) * H + lane * 4;
#pragma unroll
    for (int v = 0; v < VMAX; ++v) r[v] = __builtin_nontemporal_load((const f32x4*)(src + v * 256));
  };
  auto process = [&](int k, const f32x4 (&r)[VMAX]) {
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v) sum += r[v][0] + r[v][1] + r[v][2] + r[v][3];
    const float mean = wave_sum(sum) * (1.f / H);
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = r[v][e] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) * (1.f / H) + eps);
    keep_mean = lane == (k & 63) ? mean : keep_mean;
    keep_rstd = lane == (k & 63) ? rstd : keep_rstd;
    const long long ro = (long long)row_of(k) * H + lane * 4;
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (r[v][e] - mean) * rstd * gam[v][e] + bet[v][e];
      if constexpr (OUT != 1) *(bf16x4*)(out + ro + v * 256) = f4_to_bf4(y);
      if constexpr (OUT != 0) *(bf16x4*)(out_h + ro + v * 256) = f4_to_h4raw(y);
    }
  };
  f32x4 ra[VMAX], rb[VMAX];
  fetch(0, ra);
  for (int k = 0; k < iters; k += 2) {
    fetch(k + 1, rb);
    process(k, ra);
    fetch(k + 2, ra);
    process(k + 1, rb);
  }
  if (lane < iters && wave_g + lane * nwaves < M) {                                      // iters <= 64 (launcher)
    const int r = wave_g + lane * nwaves, m = rev ? M - 1 - r : r;
    mean_o[m] = keep_mean; rstd_o[m] = keep_rstd;
  }
}

// ------------------------------------------------------------------ LayerNorm backward
template <int VMAX>
__global__ __launch_bounds__(TPB) void ln_bwd_k(mart_ln_bwd_desc p) {
  __shared__ float red[WPB][2][VMAX * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave_g = blockIdx.x * WPB + w;
  const int nwaves = gridDim.x * WPB;
  const int nv = p.H / 256;
  const float inv_keep = 1.f / (1.f - p.p_drop);
  const uint32_t s2_lo = rng_seedmix(p.seed, 0), thr16 = dropout_thr16(p.p_drop);
  const bf16* dyb = (const bf16*)p.dy_bf16;
  f32x4 dg[VMAX], db[VMAX], gam[VMAX];
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    dg[v] = f32x4{0.f, 0.f, 0.f, 0.f}; db[v] = dg[v]; gam[v] = dg[v];
    if (v < nv) gam[v] = *(const f32x4*)(p.gamma + (v * 64 + lane) * 4);
  }
  // Two rows in flight per wave: the operands of row m + nwaves are requested before row m is reduced and stored, so the wave always
  // has a row's worth of loads outstanding (with one row at a time the loads of the next row were only issued after the stores of this
  // one: 4.9 TB/s; a plain streaming kernel with the same read / write mix reaches 6.1-6.3 TB/s on this part, tools/copy_bw.hip).
  struct Row { f32x4 add[VMAX], dyf[VMAX], s[VMAX]; bf16x4 dyh[VMAX]; float mean, rstd; };
  auto fetch = [&](int m, Row& r) {
    r.mean = p.mean[m]; r.rstd = p.rstd[m];
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      if (v < nv) {
        const long long o = (long long)m * p.H + (v * 64 + lane) * 4;
        r.add[v] = f32x4{0.f, 0.f, 0.f, 0.f}; r.dyf[v] = r.add[v]; r.dyh[v] = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        if (p.add_f32) r.add[v] = __builtin_nontemporal_load((const f32x4*)(p.add_f32 + o));
        if (p.add_bf16) r.add[v] = bf4_to_f4(__builtin_nontemporal_load((const bf16x4*)((const bf16*)p.add_bf16 + o)));
        if (p.add2_f32) r.add[v] += __builtin_nontemporal_load((const f32x4*)(p.add2_f32 + o));
        if (p.dy_f32) r.dyf[v] = __builtin_nontemporal_load((const f32x4*)(p.dy_f32 + o));
        if (dyb) r.dyh[v] = __builtin_nontemporal_load((const bf16x4*)(dyb + o));
        r.s[v] = __builtin_nontemporal_load((const f32x4*)(p.s + o));
      }
  };
  auto process = [&](int m, const Row& r) {
    const float mean = r.mean, rstd = r.rstd;
    f32x4 dy[VMAX], xh[VMAX];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      if (v < nv) {
        f32x4 d = r.dyf[v];
        if (dyb) d += bf4_to_f4(r.dyh[v]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xhat = (r.s[v][e] - mean) * rstd;
          xh[v][e] = xhat;
          float g = d[e] * gam[v][e];
          s1 += g; s2 += g * xhat;
          dg[v][e] += d[e] * xhat; db[v][e] += d[e];
        }
        dy[v] = d;
      }
    const float c1 = wave_sum(s1) / (float)p.H, c2 = wave_sum(s2) / (float)p.H;
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      if (v < nv) {
        const long long o = (long long)m * p.H + (v * 64 + lane) * 4;
        f32x4 ds;
#pragma unroll
        for (int e = 0; e < 4; ++e) ds[e] = rstd * (dy[v][e] * gam[v][e] - c1 - xh[v][e] * c2);
        if (p.ds_bf16 && !p.bf16_total) {
          f32x4 dd = ds;
          if (p.p_drop > 0.f) dd = dd * dropout_scale4(p.seed, s2_lo, o, thr16, inv_keep);
          stx((bf16x4*)((bf16*)p.ds_bf16 + o), f4_to_bf4(dd), (LN_NT & 4) != 0);
        }
        if (p.add_f32 || p.add_bf16) ds += r.add[v];
        if (p.ds_f32) stx((f32x4*)(p.ds_f32 + o), ds, (LN_NT & 2) != 0);
        if (p.ds_bf16 && p.bf16_total) stx((bf16x4*)((bf16*)p.ds_bf16 + o), f4_to_bf4(ds), (LN_NT & 4) != 0);
      }
  };
  {
    Row ra, rb;
    int m = wave_g;
    if (m < p.M) fetch(m, ra);
    for (; m < p.M; m += 2 * nwaves) {
      const int m2 = m + nwaves, m3 = m2 + nwaves;
      if (m2 < p.M) fetch(m2, rb);
      process(m, ra);
      if (m3 < p.M) fetch(m3, ra);
      if (m2 < p.M) process(m2, rb);
    }
  }
  if (!p.dgamma && !p.dbeta) return;
#pragma unroll
  for (int v = 0; v < VMAX; ++v)
    if (v < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[w][0][(v * 64 + lane) * 4 + e] = dg[v][e];
        red[w][1][(v * 64 + lane) * 4 + e] = db[v][e];
      }
    }
  __syncthreads();
  for (int c = threadIdx.x; c < p.H; c += TPB) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int ww = 0; ww < WPB; ++ww) { a += red[ww][0][c]; b += red[ww][1][c]; }
    if (p.ws) {                                     // deterministic: this workgroup's partial row, summed in workgroup order by ln_dgb_reduce_k
      p.ws[((long long)blockIdx.x * 2 + 0) * p.H + c] = a;
      p.ws[((long long)blockIdx.x * 2 + 1) * p.H + c] = b;
    } else {
      if (p.dgamma) atomicAdd(p.dgamma + c, a);
      if (p.dbeta) atomicAdd(p.dbeta + c, b);
    }
  }
}

// vision-stream shape of the backward pass: dy bf16, x f32, residual gradient f32 in; f32 total + its bf16 copy out; dgamma / dbeta partials to the
// workspace.  Straight-line and pipelined like ln_fwd_fast_k (rows past the end: clamped duplicates, weight 0 in dgamma / dbeta).
// GB16 (round 6): the residual GRADIENT stream lives in bf16 -- `add` is a bf16 tensor and only the bf16 total is written (no f32 copy): 10 bytes per element
// instead of 16 (dy 2 + x 4 + add 2 in, 2 out against dy 2 + x 4 + add 4 in, 4 + 2 out).
template <int VMAX, bool ADD2 = false, bool GB16 = false>     // ADD2: a second f32 residual operand (the fusion op's d(visual) side buffer, engine.backward)
__global__ __launch_bounds__(TPB) void ln_bwd_fast_k(const bf16* __restrict__ dy, const float* __restrict__ x, const void* __restrict__ add_, const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, const float* __restrict__ gamma, int M, float* __restrict__ ds_f32, bf16* __restrict__ ds_bf16,
                                                     float* __restrict__ ws, const float* __restrict__ add2 = nullptr, int rev = 0) {
  constexpr int H = VMAX * 256;
  static_assert(!(ADD2 && GB16), "the side-buffer operand belongs to the f32 gradient stream");
  __shared__ float red[WPB][2][VMAX * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave_g = blockIdx.x * WPB + w;
  const int nwaves = gridDim.x * WPB;
  const float* add = (const float*)add_;
  const bf16* addb = (const bf16*)add_;
  f32x4 dg[VMAX], db[VMAX], gam[VMAX];
#pragma unroll
  for (int v = 0; v < VMAX; ++v) { dg[v] = f32x4{0.f, 0.f, 0.f, 0.f}; db[v] = dg[v]; gam[v] = *(const f32x4*)(gamma + (v * 64 + lane) * 4); }
  const int iters = (M + nwaves - 1) / nwaves;
  struct Row { f32x4 a[GB16 ? 1 : VMAX], a2[ADD2 ? VMAX : 1], s[VMAX]; bf16x4 d[VMAX], ab[GB16 ? VMAX : 1]; float mean, rstd; };
  auto row_of = [&](int k) { const int r = min(wave_g + k * nwaves, M - 1); return rev ? M - 1 - r : r; };     // rev: last row first (see ln_fwd_fast_k)
  auto fetch = [&](int k, Row& r) {
    const int m = row_of(k);
    const long long o = (long long)m * H + lane * 4;
    r.mean = mean_i[m]; r.rstd = rstd_i[m];
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      if constexpr (GB16) r.ab[v] = __builtin_nontemporal_load((const bf16x4*)(addb + o + v * 256));
      else r.a[v] = __builtin_nontemporal_load((const f32x4*)(add + o + v * 256));
      if constexpr (ADD2) r.a2[v] = __builtin_nontemporal_load((const f32x4*)(add2 + o + v * 256));
      r.d[v] = __builtin_nontemporal_load((const bf16x4*)(dy + o + v * 256));
      r.s[v] = __builtin_nontemporal_load((const f32x4*)(x + o + v * 256));
    }
  };
  auto process = [&](int k, const Row& r) {
    const float valid = wave_g + k * nwaves < M ? 1.f : 0.f;
    f32x4 d[VMAX], xh[VMAX];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      d[v] = bf4_to_f4(r.d[v]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xhat = (r.s[v][e] - r.mean) * r.rstd;
        xh[v][e] = xhat;
        const float g = d[v][e] * gam[v][e];
        s1 += g; s2 += g * xhat;
        dg[v][e] += valid * (d[v][e] * xhat); db[v][e] += valid * d[v][e];
      }
    }
    const float c1 = wave_sum(s1) * (1.f / H), c2 = wave_sum(s2) * (1.f / H);
    const long long o = (long long)row_of(k) * H + lane * 4;
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      f32x4 t, av;
      if constexpr (GB16) av = bf4_to_f4(r.ab[v]); else av = r.a[v];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = r.rstd * (d[v][e] * gam[v][e] - c1 - xh[v][e] * c2) + av[e];
      if constexpr (ADD2) t += r.a2[v];
      if constexpr (!GB16) __builtin_nontemporal_store(t, (f32x4*)(ds_f32 + o + v * 256));
      *(bf16x4*)(ds_bf16 + o + v * 256) = f4_to_bf4(t);
    }
  };
  {
    Row ra, rb;
    fetch(0, ra);
    for (int k = 0; k < iters; k += 2) {
      fetch(k + 1, rb);
      process(k, ra);
      fetch(k + 2, ra);
      process(k + 1, rb);
    }
  }
#pragma unroll
  for (int v = 0; v < VMAX; ++v)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[w][0][(v * 64 + lane) * 4 + e] = dg[v][e];
      red[w][1][(v * 64 + lane) * 4 + e] = db[v][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += TPB) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int ww = 0; ww < WPB; ++ww) { a += red[ww][0][c]; b += red[ww][1][c]; }
    ws[((long long)blockIdx.x * 2 + 0) * H + c] = a;
    ws[((long long)blockIdx.x * 2 + 1) * H + c] = b;
  }
}

// dgamma[c] += sum_g ws[g][0][c], dbeta[c] += sum_g ws[g][1][c], in a FIXED order (run-to-run identical): a 1024-thread workgroup
// owns 64 columns; wave w sums the partial rows g = w, w + 16, ... (64 consecutive floats per row: one 256-byte request per wave),
// the 16 wave sums are then combined in wave order.  (A first version with one thread per column walking all 768 rows serially
// took 0.21 ms per call -- 10 ms per step.)
__global__ __launch_bounds__(1024) void ln_dgb_reduce_k(const float* __restrict__ ws, int nwg, int H, float* dgamma, float* dbeta) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;                  // column of the [2H] row: [0,H) dgamma, [H,2H) dbeta
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int g = w;
  for (; g + 48 < nwg; g += 64) {                        // four independent loads in flight per lane
    a0 += ws[(long long)g * 2 * H + i];
    a1 += ws[(long long)(g + 16) * 2 * H + i];
    a2 += ws[(long long)(g + 32) * 2 * H + i];
    a3 += ws[(long long)(g + 48) * 2 * H + i];
  }
  for (; g < nwg; g += 16) a0 += ws[(long long)g * 2 * H + i];
  red[w][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (w == 0) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[k][lane];
    float* dst = i < H ? dgamma : dbeta;
    if (dst) dst[i < H ? i : i - H] += tot;
  }
}

// ------------------------------------------------------------------ LayerNorm folded into the consuming product (gemm_nt F_STATS / F_LNFOLD)
// Operands of the fold, from the f32 master weights: Wf = bf16(gamma o W) [N, K], s[n] = sum_k Wf[n][k] (of the ROUNDED values: the mean term of
// x Wf^T then cancels against exactly what the matrix pipe summed), bf[n] = b[n] + sum_k beta[k] W[n][k].  One wave per output row.
__global__ __launch_bounds__(TPB) void ln_fold_prep_k(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16* __restrict__ Wf, float* __restrict__ s, float* __restrict__ bf, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (n >= N) return;
  float ssum = 0.f, bsum = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const f32x4 w = *(const f32x4*)(W + (long long)n * K + k), g = *(const f32x4*)(gamma + k), b = *(const f32x4*)(beta + k);
    const bf16x4 wf = f4_to_bf4(w * g);
    *(bf16x4*)(Wf + (long long)n * K + k) = wf;
    const f32x4 r = bf4_to_f4(wf);
    ssum += (r[0] + r[1]) + (r[2] + r[3]);
    bsum += (w[0] * b[0] + w[1] * b[1]) + (w[2] * b[2] + w[3] * b[3]);
  }
  ssum = wave_sum(ssum); bsum = wave_sum(bsum);
  if (lane == 0) { s[n] = ssum; bf[n] = (bias ? bias[n] : 0.f) + bsum; }
}
// mean / rstd of every row from the S = H / 64 partial (sum, sum of squares) pairs the producing epilogue wrote.  One thread per row.
__global__ __launch_bounds__(256) void ln_stats_finalize_k(const float* __restrict__ part, int M, int S, float inv_h, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float* p = part + (long long)m * S * 2;
  float s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < S; i += 2) {                                 // S is even (H % 128 == 0, launcher): 16-byte loads
    const f32x4 v = *(const f32x4*)(p + 2 * i);
    s1 += v[0] + v[2]; s2 += v[1] + v[3];
  }
  const float mu = s1 * inv_h;
  const float var = fmaxf(s2 * inv_h - mu * mu, 0.f);
  mean[m] = mu; rstd[m] = rsqrtf(var + eps);
}

// ------------------------------------------------------------------ text embeddings (gather + LN + dropout)
__global__ __launch_bounds__(TPB) void text_embed_k(mart_text_embed_desc p) {
  constexpr int VMAX = VMAX_ALL;
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * WPB + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * WPB;
  const int nv = p.H / 256, M = p.B * p.L;
  const float inv_keep = 1.f / (1.f - p.p_drop);
  for (int m = wave_g; m < M; m += nwaves) {
    const long long id = p.ids[m], tt = p.tt[m];
    const int pos = m % p.L;
    f32x4 x[VMAX];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      if (v < nv) {
        const int c = (v * 64 + lane) * 4;
        f32x4 t = *(const f32x4*)(p.word + id * p.H + c);
        t += *(const f32x4*)(p.type + tt * p.H + c);
        t += *(const f32x4*)(p.pos + (long long)pos * p.H + c);
        x[v] = t;
        sum += t[0] + t[1] + t[2] + t[3];
        if (p.s_out) *(f32x4*)(p.s_out + (long long)m * p.H + c) = t;
      }
    const float mean = wave_sum(sum) / (float)p.H;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      if (v < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { float dlt = x[v][e] - mean; sq += dlt * dlt; }
      }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.H + p.eps);
    if (lane == 0) { p.mean[m] = mean; p.rstd[m] = rstd; }
#pragma unroll
    for (int v = 0; v < VMAX; ++v)
      if (v < nv) {
        const int c = (v * 64 + lane) * 4;
        const long long o = (long long)m * p.H + c;
        f32x4 g = *(const f32x4*)(p.gamma + c), b = *(const f32x4*)(p.beta + c), y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (x[v][e] - mean) * rstd * g[e] + b[e];
        if (p.p_drop > 0.f) y = y * dropout_scale4(p.seed, rng_seedmix(p.seed, 0), o, dropout_thr16(p.p_drop), inv_keep);
        if (p.out_f32) stx((f32x4*)(p.out_f32 + o), y, (LN_NT & 2) != 0);
        if (p.out_bf16) stx((bf16x4*)((bf16*)p.out_bf16 + o), f4_to_bf4(y), (LN_NT & 4) != 0);
        if (p.out_f16) stx((bf16x4*)((bf16*)p.out_f16 + o), f4_to_h4raw(y), (LN_NT & 4) != 0);
      }
  }
}

// grid (L positions, batch slices): the word rows are scattered with atomics (ids are arbitrary); the position and
// token-type rows are summed over the slice in registers first -- one atomic per (slice, position[, type], column)
// instead of one per token (the two token-type rows used to take 8192 contended adds per column).
__global__ __launch_bounds__(256) void text_embed_scatter_k(const float* __restrict__ ds, const int64_t* __restrict__ ids, const int64_t* __restrict__ tt,
                                                           float* dword, float* dpos, float* dtype, int B, int L, int H) {
  const int pos = blockIdx.x;
  const int nsl = gridDim.y, per = (B + nsl - 1) / nsl;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  for (int c = threadIdx.x; c < H; c += blockDim.x) {                // lane = column: every atomic instruction covers 64 consecutive floats
    float ap = 0.f, a0 = 0.f, a1 = 0.f;
    for (int b = b0; b < b1; ++b) {
      const long long m = (long long)b * L + pos;
      const float g = ds[m * H + c];
      atomicAdd(dword + ids[m] * H + c, g);
      ap += g;
      const long long ty = tt[m];
      if (ty == 0) a0 += g;
      else if (ty == 1) a1 += g;
      else atomicAdd(dtype + ty * H + c, g);
    }
    atomicAdd(dpos + (long long)pos * H + c, ap);
    if (a0 != 0.f) atomicAdd(dtype + c, a0);
    if (a1 != 0.f) atomicAdd(dtype + H + c, a1);
  }
}


// ------------------------------------------------------------------ deterministic embedding gradients (no float atomics)
// Word rows: the caller passes `order`, the token positions sorted by id (stable sort of ids, index preparation on the host
// side).  Pass 1 -- one workgroup per 64 consecutive SORTED positions walks them in order and sums runs of equal ids: a run that
// lies inside the workgroup's window is added to its word row directly (it is the only writer of that row); the run that
// touches the window's first / last position may continue in the neighbouring windows and goes to a partial row (head / tail
// slot of the window).  Pass 2 -- the workgroup whose window holds the FIRST piece of such a run walks the following windows and
// adds the pieces in window order.  Every row is summed in sorted-position order: run-to-run identical.
constexpr int ES = 64;                               // sorted positions per window
__global__ __launch_bounds__(256) void embed_word_pass1_k(const float* __restrict__ ds, const int64_t* __restrict__ ids, const int64_t* __restrict__ order,
                                                         float* dword, float* ws, int* meta, int n, int H) {
  const int w0 = blockIdx.x * ES, w1 = min(n, w0 + ES);
  const long long id_first = ids[order[w0]], id_last = ids[order[w1 - 1]];
  const bool head_open = w0 > 0 && ids[order[w0 - 1]] == id_first;          // first run continues a run of the previous window
  const bool tail_open = w1 < n && ids[order[w1]] == id_last;               // last run continues into the next window
  for (int c = threadIdx.x; c < H; c += blockDim.x) {
    float acc = 0.f;
    long long cur = id_first;
    bool first_run = true;
    for (int j = w0; j < w1; ++j) {
      const long long tok = order[j], id = ids[tok];
      if (id != cur) {                                                     // flush the finished run
        if (first_run && head_open) ws[((long long)blockIdx.x * 2 + 0) * H + c] = acc;
        else dword[cur * H + c] += acc;
        first_run = false; cur = id; acc = 0.f;
      }
      acc += ds[tok * H + c];
    }
    // last run of the window (it may also be the first one: a window inside a long run)
    if (first_run && head_open) ws[((long long)blockIdx.x * 2 + 0) * H + c] = acc;      // continuation piece (whole window or its head)
    else if (tail_open) ws[((long long)blockIdx.x * 2 + 1) * H + c] = acc;               // first piece of a run that continues
    else dword[cur * H + c] += acc;
  }
  if (threadIdx.x == 0) {
    // kind of the window's pieces: bit 0 = head slot holds a continuation piece, bit 1 = tail slot holds the first piece of a run,
    // bit 2 = the head piece is the whole window and the run goes on (same id at both ends, both open)
    int m = 0;
    if (head_open) m |= 1;
    if (tail_open && !(head_open && id_first == id_last)) m |= 2;
    if (head_open && tail_open && id_first == id_last) m |= 4;
    meta[blockIdx.x] = m;
  }
}
__global__ __launch_bounds__(256) void embed_word_pass2_k(const int64_t* __restrict__ ids, const int64_t* __restrict__ order, float* dword, const float* __restrict__ ws,
                                                         const int* __restrict__ meta, int n, int nwin, int H) {
  const int k = blockIdx.x;
  if (!(meta[k] & 2)) return;                                              // this window does not start a multi-window run
  const long long id = ids[order[min(n, (k + 1) * ES) - 1]];
  for (int c = threadIdx.x; c < H; c += blockDim.x) {
    float acc = ws[((long long)k * 2 + 1) * H + c];
    for (int q = k + 1; q < nwin; ++q) {
      const int m = meta[q];
      if (!(m & 1)) break;
      acc += ws[((long long)q * 2 + 0) * H + c];
      if (!(m & 4)) break;                                                 // the run ended inside window q
    }
    dword[id * H + c] += acc;
  }
}
// position / token-type rows: per (position, batch slice) partial sums, then an ordered reduction
__global__ __launch_bounds__(256) void embed_postype_partial_k(const float* __restrict__ ds, const int64_t* __restrict__ tt, float* ws, int B, int L, int H) {
  const int pos = blockIdx.x, nsl = gridDim.y, per = (B + nsl - 1) / nsl;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  for (int c = threadIdx.x; c < H; c += blockDim.x) {
    float ap = 0.f, a1 = 0.f;
    for (int b = b0; b < b1; ++b) {
      const long long m = (long long)b * L + pos;
      const float g = ds[m * H + c];
      ap += g;
      if (tt[m] == 1) a1 += g;
    }
    float* o = ws + (((long long)pos * nsl + blockIdx.y) * 2) * H;
    o[c] = ap; o[H + c] = a1;                                              // all tokens, type-1 tokens (embed_type_reduce_k forms the type-0 row as all - type 1)
  }
}
__global__ __launch_bounds__(256) void embed_postype_reduce_k(const float* __restrict__ ws, float* dpos, int L, int nsl, int H) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= H) return;
  const int row = blockIdx.y;                                              // position row
  float tot = 0.f;
  for (int s = 0; s < nsl; ++s) tot += ws[(((long long)row * nsl + s) * 2) * H + c];
  dpos[(long long)row * H + c] += tot;
}
// token-type rows: all L * nsl partial pairs of a column, 16 groups of 64 columns per workgroup striding over the pairs with
// independent loads, combined through LDS in group order (fixed order -> run-to-run identical; a single thread walking the
// 2 * L * nsl loads serially took 0.27 ms)
__global__ __launch_bounds__(1024) void embed_type_reduce_k(const float* __restrict__ ws, float* dtype, int n_pairs, int H) {
  __shared__ float sa[16][64], so[16][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + tx;
  float all = 0.f, one = 0.f;
  if (c < H)
    for (int i = g; i < n_pairs; i += 16) { all += ws[((long long)i * 2) * H + c]; one += ws[((long long)i * 2 + 1) * H + c]; }
  sa[g][tx] = all; so[g][tx] = one;
  __syncthreads();
  if (g == 0 && c < H) {
    all = 0.f; one = 0.f;
    for (int k = 0; k < 16; ++k) { all += sa[k][tx]; one += so[k][tx]; }
    dtype[c] += all - one; dtype[H + c] += one;
  }
}
// vision: slice partials of vision_assemble_bwd_k (ws given) -> dcls / dpos in slice order
__global__ __launch_bounds__(256) void vision_assemble_reduce_k(const float* __restrict__ ws, float* dcls, float* dpos, int P, int nsl, int H, int tail_shift) {
  const int c = blockIdx.x * 256 + threadIdx.x, u = blockIdx.y;            // output position row u = 0..P
  if (c >= H) return;
  float tot = 0.f;
  for (int s = 0; s < nsl; ++s) tot += ws[(((long long)u * nsl + s) * 2) * H + c];                 // first-image row u (u = 0: class token)
  if (u == 0) dcls[c] += tot;
  const int t2 = u + tail_shift;                                           // second-image rows whose position row is u
  if (t2 >= 1 && t2 <= P)
    for (int s = 0; s < nsl; ++s) tot += ws[(((long long)t2 * nsl + s) * 2 + 1) * H + c];
  dpos[(long long)u * H + c] += tot;
}

// ------------------------------------------------------------------ vision embeddings
// pixels [B,2,3,S,S] f32 -> patch matrix [(b,img,py,px), (c,ky,kx)] bf16
__global__ void patchify_k(const float* __restrict__ pix, bf16* __restrict__ out, int B, int S, int p) {
  const int g = S / p, P = g * g, K = 3 * p * p;
  const long long row = blockIdx.x;                  // (b*2+img)*P + py*g + px
  const int patch = (int)(row % P);
  const long long bi = row / P;
  const int py = patch / g, px = patch % g;
  const float* src = pix + bi * 3LL * S * S;
  for (int k4 = threadIdx.x * 4; k4 < K; k4 += blockDim.x * 4) {
    const int c = k4 / (p * p), rem = k4 % (p * p), ky = rem / p, kx = rem % p;   // p multiple of 4 -> 4 kx contiguous
    f32x4 v = *(const f32x4*)(src + ((long long)c * S + py * p + ky) * S + px * p + kx);
    *(bf16x4*)(out + row * K + k4) = f4_to_bf4(v);
  }
}

__global__ void patchify_gather_k(const float* __restrict__ table, const int32_t* __restrict__ index, bf16* __restrict__ out, int B, int S, int p) {
  const int g = S / p, P = g * g, K = 3 * p * p;
  const long long row = blockIdx.x;                  // (b*2+slot)*P + py*g + px
  const int patch = (int)(row % P);
  const long long bi = row / P;
  const int py = patch / g, px = patch % g;
  const int src_row = index[bi];
  const float* src = table + (long long)src_row * 3LL * S * S;
  for (int k4 = threadIdx.x * 4; k4 < K; k4 += blockDim.x * 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src_row >= 0) {
      const int c = k4 / (p * p), rem = k4 % (p * p), ky = rem / p, kx = rem % p;
      v = *(const f32x4*)(src + ((long long)c * S + py * p + ky) * S + px * p + kx);
    }
    *(bf16x4*)(out + row * K + k4) = f4_to_bf4(v);
  }
}
__global__ void gather_images_k(const float* __restrict__ table, const int32_t* __restrict__ index, float* __restrict__ out, long long per) {
  const long long slot = blockIdx.y;
  const int src_row = index[slot];
  const float* src = table + (long long)src_row * per;
  float* dst = out + slot * per;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < per; i += (long long)gridDim.x * blockDim.x * 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src_row >= 0) v = *(const f32x4*)(src + i);
    *(f32x4*)(dst + i) = v;
  }
}

__global__ void vision_assemble_k(const bf16* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                  float* __restrict__ s, int B, int P, int H, int tail_shift) {
  const int Nv = 1 + 2 * P;
  const long long row = blockIdx.x;                  // b*Nv + t
  const int t = (int)(row % Nv);
  const long long b = row / Nv;
  const int pidx = t == 0 ? 0 : (t <= P ? t : t - P - tail_shift);
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
    f32x4 v;
    if (t == 0) v = *(const f32x4*)(cls + c);
    else v = bf4_to_f4(*(const bf16x4*)(patch + ((b * 2 * P) + (t - 1)) * H + c));     // (b,0,t-1) or (b,1,t-1-P): contiguous
    v += *(const f32x4*)(pos + (long long)pidx * H + c);
    *(f32x4*)(s + row * H + c) = v;
  }
}

// dpatch = bf16(ds rows 1..2P); dcls += sum_b ds[b,0]; dpos[t] += sum_b (ds[b,t] + ds[b,t+P])
// grid (P+1 rows, batch slices), 16-byte accesses; the slice sums reach dcls / dpos with one atomic per (slice, row, column)
__global__ __launch_bounds__(192) void vision_assemble_bwd_k(const float* __restrict__ ds, bf16* __restrict__ dpatch, float* dcls, float* dpos, int B, int P, int H, int tail_shift, float* ws) {
  const int Nv = 1 + 2 * P;
  const int t = blockIdx.x;                          // 0..P : row of the first image (and of the class token)
  const int nsl = gridDim.y, per = (B + nsl - 1) / nsl;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {          // H % 4 == 0 (checked by the host)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_tail = acc;
    for (int b = b0; b < b1; ++b) {
      const float* base = ds + (long long)b * Nv * H;
      if (t == 0) acc += *(const f32x4*)(base + c);
      else {
        const f32x4 g0 = *(const f32x4*)(base + (long long)t * H + c), g1 = *(const f32x4*)(base + (long long)(t + P) * H + c);
        acc += g0; acc_tail += g1;
        *(bf16x4*)(dpatch + ((long long)b * 2 * P + (t - 1)) * H + c) = f4_to_bf4(g0);
        *(bf16x4*)(dpatch + ((long long)b * 2 * P + (t - 1 + P)) * H + c) = f4_to_bf4(g1);
      }
    }
    // acc: the first image's row t (position row t); acc_tail: the second image's row t (position row t - tail_shift)
    if (ws) {                                        // deterministic: slice partials, reduced in slice order by vision_assemble_reduce_k
      float* o = ws + (((long long)t * nsl + blockIdx.y) * 2) * H;
      *(f32x4*)(o + c) = acc; *(f32x4*)(o + H + c) = acc_tail;
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(dpos + (long long)t * H + c + e, acc[e]);
      if (t > 0) atomicAdd(dpos + (long long)(t - tail_shift) * H + c + e, acc_tail[e]);
      if (t == 0) atomicAdd(dcls + c + e, acc[e]);
    }
  }
}

// ------------------------------------------------------------------ row softmax (one wave per row)
__global__ __launch_bounds__(TPB) void softmax_fwd_k(const float* __restrict__ sc, int ld, bf16* __restrict__ pr, int ldp, int R, int C) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* s = sc + (long long)r * ld;
  float mx = -3.0e38f;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, s[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += __expf(s[c] - mx);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  bf16* o = pr + (long long)r * ldp;
  for (int c = lane; c < ldp; c += 64) o[c] = c < C ? f2bf(__expf(s[c] - mx) * inv) : (bf16)0.f;
}
__global__ __launch_bounds__(TPB) void softmax_bwd_k(const bf16* __restrict__ pr, int ldp, const float* __restrict__ dp, int ldd,
                                                     bf16* __restrict__ dsout, int ldo, int R, int C) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (r >= R) return;
  const bf16* p = pr + (long long)r * ldp;
  const float* d = dp + (long long)r * ldd;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += bf2f(p[c]) * d[c];
  dot = wave_sum(dot);
  bf16* o = dsout + (long long)r * ldo;
  for (int c = lane; c < ldo; c += 64) o[c] = c < C ? f2bf(bf2f(p[c]) * (d[c] - dot)) : (bf16)0.f;
}

// ------------------------------------------------------------------ tiled 2-D transposes (64x64 through LDS)
// 16-byte global accesses on both sides (8 lanes cover one 128-byte row segment of the source / of the destination);
// the transposition itself is 2-byte LDS gathers from a tile padded to 66 columns (33-dword rows: odd stride).
// out[c][r] = in[r][c] for r < R, c < C; destination rows are Rp long, columns R..Rp-1 are written as zeros.
// (The 32x32 version with 2-byte global accesses ran at 0.8 TB/s.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TT = 64, TT_LD = TT + 2;
__device__ __forceinline__ void transpose_tile(const bf16* __restrict__ src, int ldi, bf16* __restrict__ dst, int Rp, int R, int C,
                                               int r0, int c0, bf16 (*tile)[TT_LD], bool vec_in, bool vec_out) {
  const int t = threadIdx.x, sub = t & 7, line = t >> 3;         // 256 threads: 32 lines x 8 chunks per pass
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rl = line + 32 * pass, r = r0 + rl, c = c0 + sub * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    if (r < R) {
      if (vec_in && c + 8 <= C) v = *(const bf16x8*)(src + (long long)r * ldi + c);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (c + e < C) v[e] = src[(long long)r * ldi + c + e];
      }
    }
    unsigned* d = (unsigned*)&tile[rl][sub * 8];                 // 4-byte aligned (row stride 132 B)
    const u32x4 w = __builtin_bit_cast(u32x4, v);
    d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; d[3] = w[3];
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = line + 32 * pass, c = c0 + cl, r = r0 + sub * 8;  // destination row c, columns r..r+7
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[sub * 8 + e][cl];
    if (c < C) {
      if (vec_out && r + 8 <= Rp) *(bf16x8*)(dst + (long long)c * Rp + r) = v;
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (r + e < Rp) dst[(long long)c * Rp + r + e] = v[e];
      }
    }
  }
}
__global__ __launch_bounds__(256) void transpose_k(const bf16* __restrict__ in, int ldi, long long si, bf16* __restrict__ out, int Rp, long long so, int R, int C) {
  __shared__ bf16 tile[TT][TT_LD];
  const bf16* src = in + (long long)blockIdx.z * si;
  bf16* dst = out + (long long)blockIdx.z * so;
  const bool vec_in = (ldi % 8 == 0) && (si % 8 == 0) && ((uintptr_t)in % 16 == 0);
  const bool vec_out = (Rp % 8 == 0) && (so % 8 == 0) && ((uintptr_t)out % 16 == 0);
  transpose_tile(src, ldi, dst, Rp, R, C, blockIdx.y * TT, blockIdx.x * TT, tile, vec_in, vec_out);
}
__global__ __launch_bounds__(256) void transpose_table_k(const bf16* __restrict__ src, bf16* __restrict__ dst, const int64_t* __restrict__ table, int n) {
  __shared__ bf16 tile[TT][TT_LD];
  // blockIdx.y = matrix; blockIdx.x = linear tile id (grid-stride over that matrix's tiles)
  const int64_t* e = table + 4LL * blockIdx.y;
  const bf16* s = src + e[0];
  bf16* d = dst + e[1];
  const int R = (int)e[2], C = (int)e[3];
  const int tr = (R + TT - 1) / TT, tc = (C + TT - 1) / TT;
  const bool vec_in = (C % 8 == 0) && (e[0] % 8 == 0) && ((uintptr_t)src % 16 == 0);
  const bool vec_out = (R % 8 == 0) && (e[1] % 8 == 0) && ((uintptr_t)dst % 16 == 0);
  for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    transpose_tile(s, C, d, R, R, C, (t / tc) * TT, (t % tc) * TT, tile, vec_in, vec_out);
    __syncthreads();
  }
}
__global__ void block_table_k(const bf16* __restrict__ src, bf16* __restrict__ dst, const int64_t* __restrict__ table, int n) {
  const int64_t* e = table + 4LL * blockIdx.y;
  const bf16* s = src + e[0];
  bf16* d = dst + e[1];
  const int R = (int)e[2], C = (int)e[3];
  const long long chunks = (long long)R * C / 8;                 // 16-byte chunks, destination order
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
    const long long blk = i >> 11;                               // 2048 chunks per 256x64 block
    const int within = (int)(i & 2047), row = within >> 3, ch = within & 7;
    const int kb = (int)(blk % (C >> 6)), nb = (int)(blk / (C >> 6));
    *(bf16x8*)(d + i * 8) = *(const bf16x8*)(s + ((long long)nb * 256 + row) * C + kb * 64 + ch * 8);
  }
}
}  // namespace

extern "C" int mart_ln_fwd(const mart_ln_fwd_desc* d, void* stream) {
  MART_CHECK(d && (d->x_f32 || d->y_bf16 || d->y_f32), "ln_fwd: need x_f32, y_bf16 or y_f32");
  MART_CHECK(!(d->y_bf16 && d->y_f32), "ln_fwd: y_bf16 and y_f32 are alternatives");
  MART_CHECK(d->M > 0 && d->H % 256 == 0 && d->H <= 256 * VMAX_ALL, "ln_fwd: H must be a multiple of 256 and <= 1024");
  MART_CHECK(d->gamma && d->beta && d->mean && d->rstd && (d->out_f32 || d->out_bf16 || d->out_f16 || d->out_split3), "ln_fwd: null pointer");
  MART_CHECK(!d->x_rows || d->x_f32, "ln_fwd: x_rows gathers x_f32");
  MART_CHECK(d->p_drop >= 0.f && d->p_drop < 1.f, "ln_fwd: bad dropout p");
  static const int fast = getenv("MART_LN_FAST") ? atoi(getenv("MART_LN_FAST")) : 1;
  static const int fcap = getenv("MART_LN_FWD_GRID") ? atoi(getenv("MART_LN_FWD_GRID")) : 512;   // 6.15 TB/s (768 / 1024 / 2048: 5.8-5.95)
  if (fast && d->x_f32 && !d->x_rows && !d->y_bf16 && !d->y_f32 && !d->s_out && !d->out_f32 && !d->out_split3 && (d->out_bf16 || d->out_f16) && d->p_drop == 0.f && (d->H == 768 || d->H == 1024) && d->M >= 4096) {
    int g = (d->M + WPB - 1) / WPB;
    if (g > fcap) g = fcap;
    const int gmin = (d->M + WPB * 64 - 1) / (WPB * 64);          // at most 64 rows per wave (their statistics live in one register across the lanes)
    if (g < gmin) g = gmin;
    const int ov = d->out_f16 ? (d->out_bf16 ? 2 : 1) : 0;
    static const int rev = getenv("MART_LN_REV") ? atoi(getenv("MART_LN_REV")) & 1 : 0;       // sweep from the last row (tools/mall_probe.py: neutral, off)
#define LN_FAST(V_, O_) hipLaunchKernelGGL((ln_fwd_fast_k<V_, O_>), dim3(g), dim3(TPB), 0, (hipStream_t)stream, d->x_f32, d->gamma, d->beta, d->eps, d->M, (bf16*)d->out_bf16, d->mean, d->rstd, (bf16*)d->out_f16, rev)
    if (d->H == 768) { if (ov == 0) LN_FAST(3, 0); else if (ov == 1) LN_FAST(3, 1); else LN_FAST(3, 2); }
    else { if (ov == 0) LN_FAST(4, 0); else if (ov == 1) LN_FAST(4, 1); else LN_FAST(4, 2); }
#undef LN_FAST
    MART_LAUNCH_CHECK();
    return 0;
  }
  if (d->H <= 768) hipLaunchKernelGGL(ln_fwd_k<3>, dim3(row_grid(d->M)), dim3(TPB), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL(ln_fwd_k<4>, dim3(row_grid(d->M)), dim3(TPB), 0, (hipStream_t)stream, *d);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_ln_bwd(const mart_ln_bwd_desc* d, void* stream) {
  MART_CHECK(d && (d->dy_f32 || d->dy_bf16), "ln_bwd: need dy_f32 or dy_bf16");
  MART_CHECK(d->M > 0 && d->H % 256 == 0 && d->H <= 256 * VMAX_ALL, "ln_bwd: H must be a multiple of 256 and <= 1024");
  MART_CHECK(d->s && d->mean && d->rstd && d->gamma && (d->ds_f32 || d->ds_bf16), "ln_bwd: null pointer");
  MART_CHECK(!d->add2_f32 || d->add_f32, "ln_bwd: add2_f32 is a second residual operand (needs add_f32)");
  MART_CHECK(!d->add_bf16 || (!d->add_f32 && !d->add2_f32), "ln_bwd: add_bf16 replaces add_f32 (the residual gradient stream in bf16)");
  MART_CHECK(!d->defer_reduce || d->ws, "ln_bwd: defer_reduce needs the workspace");
  int g = row_grid(d->M);
  // Two rows in flight per wave cost 206 VGPRs: two waves per SIMD, i.e. two 4-wave workgroups per CU -> grid = 512 is exactly ONE round
  // (768, one and a half rounds: 4.65 TB/s; 512: 5.66 TB/s at M = 100608; the one-row kernel ran 4.97 TB/s at its best grid of 768)
  static const int cap_env = getenv("MART_LN_BWD_GRID") ? atoi(getenv("MART_LN_BWD_GRID")) : 0;
  static const int fastb = getenv("MART_LN_FAST") ? atoi(getenv("MART_LN_FAST")) : 1;
  const bool fast_shape = fastb && d->dy_bf16 && !d->dy_f32 && d->add_f32 && d->ds_f32 && d->ds_bf16 && d->bf16_total && d->p_drop == 0.f && d->ws && d->dgamma && d->dbeta &&
                          (d->H == 768 || d->H == 1024) && d->M >= 4096 &&
                          d->add_f32 != d->ds_f32 && d->s != d->ds_f32 && d->dy_bf16 != d->ds_bf16 && d->add2_f32 != d->ds_f32;   // clamped duplicate rows re-read their inputs: no in-place operands
  const int cap = cap_env ? cap_env : 512;      // two workgroups per CU in one round: 6.1 TB/s for the straight-line kernel (768: 5.9, 384: 5.6, 256: 5.0), 5.66 for the general one
  if (g > cap) g = cap;
  static const int revb = getenv("MART_LN_REV") ? (atoi(getenv("MART_LN_REV")) >> 1) & 1 : 0;
  MART_CHECK(!d->ws || d->ws_bytes >= (long long)g * 2 * d->H * (long long)sizeof(float), "ln_bwd: workspace too small (768 * 2 * H floats always suffice)");
  // the same shape with the residual gradient stream in bf16: bf16 residual operand in, bf16 total out, nothing else
  const bool fast_gb16 = fastb && d->dy_bf16 && !d->dy_f32 && d->add_bf16 && !d->ds_f32 && d->ds_bf16 && d->bf16_total && d->p_drop == 0.f && d->ws && d->dgamma && d->dbeta &&
                         (d->H == 768 || d->H == 1024) && d->M >= 4096 && d->add_bf16 != d->ds_bf16 && d->dy_bf16 != d->ds_bf16;
#define LNB_FAST(V_, A_) hipLaunchKernelGGL((ln_bwd_fast_k<V_, A_>), dim3(g), dim3(TPB), 0, (hipStream_t)stream, (const bf16*)d->dy_bf16, d->s, (const void*)d->add_f32, d->mean, d->rstd, d->gamma, d->M, d->ds_f32, (bf16*)d->ds_bf16, d->ws, d->add2_f32, revb)
#define LNB_FAST16(V_) hipLaunchKernelGGL((ln_bwd_fast_k<V_, false, true>), dim3(g), dim3(TPB), 0, (hipStream_t)stream, (const bf16*)d->dy_bf16, d->s, d->add_bf16, d->mean, d->rstd, d->gamma, d->M, (float*)nullptr, (bf16*)d->ds_bf16, d->ws, (const float*)nullptr, revb)
  if (fast_gb16 && d->H == 768) LNB_FAST16(3);
  else if (fast_gb16) LNB_FAST16(4);
  else if (fast_shape && d->H == 768) { if (d->add2_f32) LNB_FAST(3, true); else LNB_FAST(3, false); }
  else if (fast_shape) { if (d->add2_f32) LNB_FAST(4, true); else LNB_FAST(4, false); }
#undef LNB_FAST
#undef LNB_FAST16
  else if (d->H <= 768) hipLaunchKernelGGL(ln_bwd_k<3>, dim3(g), dim3(TPB), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL(ln_bwd_k<4>, dim3(g), dim3(TPB), 0, (hipStream_t)stream, *d);
  MART_LAUNCH_CHECK();
  if (d->ws && (d->dgamma || d->dbeta) && !d->defer_reduce) {
    hipLaunchKernelGGL(ln_dgb_reduce_k, dim3(2 * d->H / 64), dim3(1024), 0, (hipStream_t)stream, d->ws, g, d->H, d->dgamma, d->dbeta);   // H % 256 == 0
    MART_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int mart_ln_bwd_partials(int M) {           // the grid rule of mart_ln_bwd
  static const int cap_env = getenv("MART_LN_BWD_GRID") ? atoi(getenv("MART_LN_BWD_GRID")) : 0;
  const int cap = cap_env ? cap_env : 512;
  const int g = row_grid(M);
  return g > cap ? cap : g;
}
extern "C" int mart_ln_dgb_reduce(const float* ws, int partials, int H, float* dgamma, float* dbeta, void* stream) {
  MART_CHECK(ws && partials > 0 && H > 0 && H % 256 == 0 && (dgamma || dbeta), "ln_dgb_reduce: bad args");
  hipLaunchKernelGGL(ln_dgb_reduce_k, dim3(2 * H / 64), dim3(1024), 0, (hipStream_t)stream, ws, partials, H, dgamma, dbeta);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_ln_fold_prep(const float* W, const float* bias, const float* gamma, const float* beta, void* Wf, float* s, float* bf, int N, int K, void* stream) {
  MART_CHECK(W && gamma && beta && Wf && s && bf && N > 0 && K > 0 && K % 4 == 0, "ln_fold_prep: bad args (K must be a multiple of 4)");
  hipLaunchKernelGGL(ln_fold_prep_k, dim3((N + WPB - 1) / WPB), dim3(TPB), 0, (hipStream_t)stream, W, bias, gamma, beta, (bf16*)Wf, s, bf, N, K);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_ln_stats_finalize(const float* partials, int M, int H, float eps, float* mean, float* rstd, void* stream) {
  MART_CHECK(partials && mean && rstd && M > 0 && H > 0 && H % 128 == 0, "ln_stats_finalize: bad args (H must be a multiple of 128)");
  hipLaunchKernelGGL(ln_stats_finalize_k, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, partials, M, H / 64, 1.f / (float)H, eps, mean, rstd);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_text_embed_fwd(const mart_text_embed_desc* d, void* stream) {
  MART_CHECK(d && d->ids && d->tt && d->word && d->pos && d->type && d->gamma && d->beta, "text_embed: null pointer");
  MART_CHECK(d->B > 0 && d->L > 0 && d->H % 256 == 0 && d->H <= 256 * VMAX_ALL, "text_embed: bad shape");
  MART_CHECK(d->mean && d->rstd && (d->out_f32 || d->out_bf16 || d->out_f16), "text_embed: null output");
  hipLaunchKernelGGL(text_embed_k, dim3(row_grid(d->B * d->L)), dim3(TPB), 0, (hipStream_t)stream, *d);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_text_embed_scatter(const float* ds, const int64_t* ids, const int64_t* tt, float* dword, float* dpos, float* dtype,
                                       int B, int L, int H, void* stream) {
  MART_CHECK(ds && ids && tt && dword && dpos && dtype && B > 0 && L > 0 && H > 0, "text_embed_scatter: bad args");
  hipLaunchKernelGGL(text_embed_scatter_k, dim3(L, B >= 64 ? 16 : 1), dim3(256), 0, (hipStream_t)stream, ds, ids, tt, dword, dpos, dtype, B, L, H);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_patchify(const float* pixels, void* out_bf16, int B, int S, int p, void* stream) {
  MART_CHECK(pixels && out_bf16 && B > 0 && S > 0 && p > 0 && S % p == 0 && p % 4 == 0, "patchify: bad args");
  const int g = S / p;
  hipLaunchKernelGGL(patchify_k, dim3(B * 2 * g * g), dim3(192), 0, (hipStream_t)stream, pixels, (bf16*)out_bf16, B, S, p);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_patchify_gather(const float* table, const int32_t* index, void* out_bf16, int B, int S, int p, void* stream) {
  MART_CHECK(table && index && out_bf16 && B > 0 && S > 0 && p > 0 && S % p == 0 && p % 4 == 0, "patchify_gather: bad args");
  const int g = S / p;
  hipLaunchKernelGGL(patchify_gather_k, dim3(B * 2 * g * g), dim3(192), 0, (hipStream_t)stream, table, index, (bf16*)out_bf16, B, S, p);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_gather_images(const float* table, const int32_t* index, float* out, int B, int S, void* stream) {
  MART_CHECK(table && index && out && B > 0 && S > 0 && (S * S * 3) % 4 == 0, "gather_images: bad args");
  hipLaunchKernelGGL(gather_images_k, dim3(64, B * 2), dim3(256), 0, (hipStream_t)stream, table, index, out, 3LL * S * S);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_vision_assemble(const void* patch_bf16, const float* cls, const float* pos, float* s, int B, int P, int H, int tail_shift, void* stream) {
  MART_CHECK(patch_bf16 && cls && pos && s && B > 0 && P > 0 && H % 4 == 0 && (tail_shift == 0 || tail_shift == 1), "vision_assemble: bad args");
  hipLaunchKernelGGL(vision_assemble_k, dim3(B * (1 + 2 * P)), dim3(192), 0, (hipStream_t)stream, (const bf16*)patch_bf16, cls, pos, s, B, P, H, tail_shift);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_vision_assemble_bwd(const float* ds, void* dpatch_bf16, float* dcls, float* dpos, int B, int P, int H, int tail_shift, void* stream) {
  MART_CHECK(ds && dpatch_bf16 && dcls && dpos && B > 0 && P > 0 && H > 0 && H % 4 == 0 && (tail_shift == 0 || tail_shift == 1), "vision_assemble_bwd: bad args (H must be a multiple of 4)");
  hipLaunchKernelGGL(vision_assemble_bwd_k, dim3(P + 1, B >= 64 ? 16 : 1), dim3(192), 0, (hipStream_t)stream, ds, (bf16*)dpatch_bf16, dcls, dpos, B, P, H, tail_shift, (float*)nullptr);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_vision_assemble_bwd_det(const float* ds, void* dpatch_bf16, float* dcls, float* dpos, int B, int P, int H, int tail_shift,
                                            float* ws, long long ws_bytes, void* stream) {
  MART_CHECK(ds && dpatch_bf16 && dcls && dpos && ws && B > 0 && P > 0 && H > 0 && H % 4 == 0 && (tail_shift == 0 || tail_shift == 1), "vision_assemble_bwd_det: bad args");
  const int nsl = B >= 64 ? 16 : 1;
  MART_CHECK(ws_bytes >= (long long)(P + 1) * nsl * 2 * H * (long long)sizeof(float), "vision_assemble_bwd_det: workspace too small ((P+1) * 16 * 2 * H floats suffice)");
  hipLaunchKernelGGL(vision_assemble_bwd_k, dim3(P + 1, nsl), dim3(192), 0, (hipStream_t)stream, ds, (bf16*)dpatch_bf16, dcls, dpos, B, P, H, tail_shift, ws);
  MART_LAUNCH_CHECK();
  hipLaunchKernelGGL(vision_assemble_reduce_k, dim3((H + 255) / 256, P + 1), dim3(256), 0, (hipStream_t)stream, ws, dcls, dpos, P, nsl, H, tail_shift);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_text_embed_scatter_det(const float* ds, const int64_t* ids, const int64_t* tt, const int64_t* order, float* dword, float* dpos, float* dtype,
                                           int B, int L, int H, float* ws, long long ws_bytes, int32_t* meta, void* stream) {
  MART_CHECK(ds && ids && tt && order && dword && dpos && dtype && ws && meta && B > 0 && L > 0 && H > 0, "text_embed_scatter_det: bad args");
  const int n = B * L, nwin = (n + ES - 1) / ES, nsl = B >= 64 ? 16 : 1;
  const long long need = ((long long)nwin * 2 + (long long)L * nsl * 2) * H * (long long)sizeof(float);
  MART_CHECK(ws_bytes >= need, "text_embed_scatter_det: workspace too small ((ceil(B*L/64) * 2 + L * 16 * 2) * H floats suffice)");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_word_pass1_k, dim3(nwin), dim3(256), 0, st, ds, ids, order, dword, ws, meta, n, H);
  MART_LAUNCH_CHECK();
  hipLaunchKernelGGL(embed_word_pass2_k, dim3(nwin), dim3(256), 0, st, ids, order, dword, ws, meta, n, nwin, H);
  MART_LAUNCH_CHECK();
  float* ws2 = ws + (long long)nwin * 2 * H;
  hipLaunchKernelGGL(embed_postype_partial_k, dim3(L, nsl), dim3(256), 0, st, ds, tt, ws2, B, L, H);
  MART_LAUNCH_CHECK();
  hipLaunchKernelGGL(embed_postype_reduce_k, dim3((H + 255) / 256, L), dim3(256), 0, st, ws2, dpos, L, nsl, H);
  MART_LAUNCH_CHECK();
  hipLaunchKernelGGL(embed_type_reduce_k, dim3((H + 63) / 64), dim3(1024), 0, st, ws2, dtype, L * nsl, H);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_softmax_fwd(const float* scores, int lds_, void* probs_bf16, int ldp, int R, int C, void* stream) {
  MART_CHECK(scores && probs_bf16 && R > 0 && C > 0 && ldp >= C && lds_ >= C, "softmax_fwd: bad args");
  hipLaunchKernelGGL(softmax_fwd_k, dim3((R + WPB - 1) / WPB), dim3(TPB), 0, (hipStream_t)stream, scores, lds_, (bf16*)probs_bf16, ldp, R, C);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_softmax_bwd(const void* probs_bf16, int ldp, const float* dprobs, int ldd, void* dscores_bf16, int ldo, int R, int C, void* stream) {
  MART_CHECK(probs_bf16 && dprobs && dscores_bf16 && R > 0 && C > 0 && ldp >= C && ldd >= C && ldo >= C, "softmax_bwd: bad args");
  hipLaunchKernelGGL(softmax_bwd_k, dim3((R + WPB - 1) / WPB), dim3(TPB), 0, (hipStream_t)stream, (const bf16*)probs_bf16, ldp, dprobs, ldd,
                     (bf16*)dscores_bf16, ldo, R, C);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_transpose_bf16(const void* in, int ldi, long long stride_i, void* out, int Rp, long long stride_o, int R, int C, int batch, void* stream) {
  MART_CHECK(in && out && R > 0 && C > 0 && Rp >= R && ldi >= C && batch > 0, "transpose_bf16: bad args");
  hipLaunchKernelGGL(transpose_k, dim3((C + TT - 1) / TT, (Rp + TT - 1) / TT, batch), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, ldi, stride_i,
                     (bf16*)out, Rp, stride_o, R, C);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_transpose_table(const void* src_bf16, void* dst_bf16, const int64_t* table, int n, void* stream) {
  MART_CHECK(src_bf16 && dst_bf16 && table && n > 0, "transpose_table: bad args");
  hipLaunchKernelGGL(transpose_table_k, dim3(128, n), dim3(256), 0, (hipStream_t)stream, (const bf16*)src_bf16, (bf16*)dst_bf16, table, n);
  MART_LAUNCH_CHECK();
  return 0;
}

extern "C" int mart_block_table(const void* src_bf16, void* dst_bf16, const int64_t* table, int n, void* stream) {
  MART_CHECK(src_bf16 && dst_bf16 && table && n > 0, "block_table: bad args");
  hipLaunchKernelGGL(block_table_k, dim3(64, n), dim3(256), 0, (hipStream_t)stream, (const bf16*)src_bf16, (bf16*)dst_bf16, table, n);
  MART_LAUNCH_CHECK();
  return 0;
}

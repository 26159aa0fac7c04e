// fp32-accurate evaluation path (north_star: logits within 1e-3 of the fp32 reference, ranked indices exact).
//
// The reference computes in fp32 end to end.  gfx950's fast matrix path is bf16, so the precise mode keeps
// every activation in fp32 and feeds the SAME bf16 MFMA GEMM kernel with operands split into two bf16 terms
//     x = hi + lo,  hi = bf16(x),  lo = bf16(x - hi)            (16 mantissa bits)
// laid out K-concatenated so that one NT GEMM over K' = 3K accumulates  hi*hi + lo*hi + hi*lo  in fp32:
//     A' = [ hi | lo | hi ]   (activations, role 0)        B' = [ hi | hi | lo ]   (weights, role 1)
// (the dropped lo*lo term is 2^-16 relative).  Attention and the image-text fusion softmax(ctx vis^T) vis run in plain
// fp32 FMA arithmetic in attn_f32_k (no MFMA: 7 % of the FLOPs, and the softmax path is the precision-critical part).
// Evaluation runs the forward half; the backward half (end of this file) exists for verification: gradients against the fp32 reference.
#include <cstdlib>
#include "common.h"
#include "mart_hip.h"

int mart_attn_split_launch(const mart_attn_f32_desc* d, void* stream);   // csrc/attention.hip

namespace {

// terms = 2: the layout above (3 blocks).  terms = 3 (verification mode: gradients on chaotic weights need products exact to ~2^-24):
// x = h + m + l (24 mantissa bits), six products  hh + hm + mh + mm + hl + lh  =>  role 0 [h|h|m|m|h|l], role 1 [h|m|h|m|l|h]  (K' = 6K).
__device__ __forceinline__ void split_terms(const f32x4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
  h = f4_to_bf4(x);
  const f32x4 hf = bf4_to_f4(h);
  const f32x4 r1 = {x[0] - hf[0], x[1] - hf[1], x[2] - hf[2], x[3] - hf[3]};
  m = f4_to_bf4(r1);
  const f32x4 mf = bf4_to_f4(m);
  l = f4_to_bf4(f32x4{r1[0] - mf[0], r1[1] - mf[1], r1[2] - mf[2], r1[3] - mf[3]});
}
__global__ void split3_k(const float* __restrict__ src, long long ld, bf16* __restrict__ dst, int rows, int K, int role,
                         const int32_t* __restrict__ gather, int terms) {
  const long long total = (long long)rows * (K / 4);
  const int W = terms == 3 ? 6 : 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / (K / 4);
    const int c = (int)(i % (K / 4)) * 4;
    const long long sr = gather ? (long long)gather[r] : r;
    const f32x4 x = *(const f32x4*)(src + sr * ld + c);
    bf16x4 h, m, l;
    split_terms(x, h, m, l);
    bf16* o = dst + r * (long long)W * K + c;
    if (terms == 3) {
      const bf16x4 a[6] = {h, h, m, m, h, l}, b[6] = {h, m, h, m, l, h};
#pragma unroll
      for (int t = 0; t < 6; ++t) *(bf16x4*)(o + (long long)t * K) = role == 0 ? a[t] : b[t];
    } else {
      *(bf16x4*)(o) = h;
      *(bf16x4*)(o + K) = role == 0 ? m : h;
      *(bf16x4*)(o + 2 * K) = role == 0 ? h : m;
    }
  }
}

// The two-term split of a dense [rows, K] matrix (K % 8 == 0): the shape behind every GEMM of the fp32-accurate evaluation pass (140 launches, 13 ms of
// its 108 at one element quad per thread and iteration: a dependent load -> three 8-byte stores chain at 1.3 TB/s).  Here a thread has four quads in
// flight, 256 quads apart, so that every wave-instruction still covers one contiguous kilobyte (sixteen CONSECUTIVE elements per thread put the lanes of a
// load 64 bytes apart: twice as slow as the plain kernel).
__global__ __launch_bounds__(256) void split3_fast_k(const float* __restrict__ src, long long ld, bf16* __restrict__ dst, int rows, int K, int role) {
  const int qpr = K / 4;                                   // element quads per row
  const long long total = (long long)rows * qpr;
  constexpr int U = 4;                                     // quads per thread and iteration: four independent 16-byte loads in flight, each wave-instruction contiguous
  for (long long base = (long long)blockIdx.x * (256 * U); base < total; base += (long long)gridDim.x * (256 * U)) {
    f32x4 x[U];
    long long r[U]; int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = min(base + u * 256 + threadIdx.x, total - 1);
      r[u] = i / qpr; c[u] = (int)(i % qpr) * 4;
      x[u] = __builtin_nontemporal_load((const f32x4*)(src + r[u] * ld + c[u]));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + u * 256 + threadIdx.x >= total) continue;
      const bf16x4 h = f4_to_bf4(x[u]);
      const f32x4 hf = bf4_to_f4(h);
      const bf16x4 m = f4_to_bf4(f32x4{x[u][0] - hf[0], x[u][1] - hf[1], x[u][2] - hf[2], x[u][3] - hf[3]});
      bf16* o = dst + r[u] * 3LL * K + c[u];
      *(bf16x4*)(o) = h;
      *(bf16x4*)(o + K) = role == 0 ? m : h;
      *(bf16x4*)(o + 2 * K) = role == 0 ? h : m;
    }
  }
}

// pixels [B,2,3,S,S] f32 (or table rows through index) -> f32 patch matrix [(b,img,py,px), (c,ky,kx)]
__global__ void patchify_f32_k(const float* __restrict__ pix, const int32_t* __restrict__ index, float* __restrict__ out, int S, int p) {
  const int g = S / p, P = g * g, K = 3 * p * p;
  const long long row = blockIdx.x;
  const int patch = (int)(row % P);
  const long long bi = row / P;
  const int py = patch / g, px = patch % g;
  const long long src_row = index ? index[bi] : bi;
  const float* src = pix + src_row * 3LL * S * S;
  for (int k4 = threadIdx.x * 4; k4 < K; k4 += blockDim.x * 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src_row >= 0) {
      const int c = k4 / (p * p), rem = k4 % (p * p), ky = rem / p, kx = rem % p;
      v = *(const f32x4*)(src + ((long long)c * S + py * p + ky) * S + px * p + kx);
    }
    *(f32x4*)(out + row * K + k4) = v;
  }
}

__global__ void vision_assemble_f32_k(const float* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                      float* __restrict__ s, int P, int H, int tail_shift) {
  const int Nv = 1 + 2 * P;
  const long long row = blockIdx.x;
  const int t = (int)(row % Nv);
  const long long b = row / Nv;
  const int pidx = t == 0 ? 0 : (t <= P ? t : t - P - tail_shift);
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
    f32x4 v = t == 0 ? *(const f32x4*)(cls + c) : *(const f32x4*)(patch + ((b * 2 * P) + (t - 1)) * H + c);
    v += *(const f32x4*)(pos + (long long)pidx * H + c);
    *(f32x4*)(s + row * H + c) = v;
  }
}

// Generic fp32 attention: one workgroup = QT query rows of one (batch, head).
//   phase 1  scores[r][j] = q_r . key_j   (thread j streams its key row once, the QT query rows sit in LDS)
//   phase 2  scale, adaptive reweight (modeling_unimo.py:342-349), additive mask (-10000), row softmax (one wave per row)
//   phase 3  ctx[r][d] = sum_j p[r][j] v[j][d]   (thread = column d, coalesced V reads)
// Keys = optional prefix [Lp] (text K/V feeding a vision layer, :227-229) followed by the Sk own keys.
constexpr int QT = 16;
template <int D>
__global__ __launch_bounds__(256) void attn_f32_k(mart_attn_f32_desc p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Stot = p.Lp + p.Sk;
  float* qs = sm;                    // [QT][D]
  float* sc = sm + QT * D;           // [QT][Stot]
  const int q0 = blockIdx.x * QT, h = blockIdx.y;
  const long long b = blockIdx.z;
  const int tid = threadIdx.x;
  const int nq = min(QT, p.Sq - q0);
  for (int i = tid; i < QT * D; i += 256) {
    const int r = i / D, d = i % D;
    qs[i] = r < nq ? p.q[(b * p.Sq + q0 + r) * p.ldq + h * D + d] : 0.f;
  }
  __syncthreads();
  for (int j = tid; j < Stot; j += 256) {
    const float* kr = j < p.Lp ? p.pk + (b * p.Lp + j) * p.ldp + h * D : p.k + (b * p.Sk + (j - p.Lp)) * p.ldk + h * D;
    float acc[QT];
#pragma unroll
    for (int r = 0; r < QT; ++r) acc[r] = 0.f;
    for (int d = 0; d < D; d += 4) {
      const f32x4 kv = *(const f32x4*)(kr + d);
#pragma unroll
      for (int r = 0; r < QT; ++r) {
        const f32x4 qv = *(const f32x4*)(qs + r * D + d);
        acc[r] = fmaf(qv[0], kv[0], acc[r]); acc[r] = fmaf(qv[1], kv[1], acc[r]);
        acc[r] = fmaf(qv[2], kv[2], acc[r]); acc[r] = fmaf(qv[3], kv[3], acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < QT; ++r) sc[r * Stot + j] = acc[r];
  }
  __syncthreads();
  {
    const int wave = tid >> 6, lane = tid & 63;
    float w0 = 1.f, w1 = 1.f; int s = 0x7fffffff;
    if (p.sep) {
      s = (int)p.sep[b * p.sep_stride];
      w0 = fminf(fmaxf(*p.w0, 0.f), 0.5f);
      w1 = fminf(fmaxf(*p.w1, 0.5f), 1.f);
    }
    for (int r = wave; r < nq; r += 4) {
      const int qi = q0 + r;
      float mx = -3.0e38f;
      for (int j = lane; j < Stot; j += 64) {
        float v = sc[r * Stot + j] * p.scale;
        if (p.sep && j >= s && !(p.rw_skip_row0 && qi == 0)) v *= (qi < s ? w0 : w1);
        if (p.attn_mask && p.attn_mask[b * p.Sk + j] == 0) v += -10000.0f;
        sc[r * Stot + j] = v;
        mx = fmaxf(mx, v);
      }
      mx = wave_max(mx);
      float sum = 0.f;
      for (int j = lane; j < Stot; j += 64) {
        const float e = expf(sc[r * Stot + j] - mx);
        sc[r * Stot + j] = e;
        sum += e;
      }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      for (int j = lane; j < Stot; j += 64) sc[r * Stot + j] *= inv;
    }
  }
  __syncthreads();
  constexpr int RG = D >= 256 ? 1 : 256 / D;       // row groups working side by side on narrow heads
  constexpr int RPG = QT / RG;                      // rows per group
  for (int d0 = 0; d0 < D; d0 += 256 / RG) {
    const int d = d0 + tid % (256 / RG), rg = tid / (256 / RG);
    if (d >= D) continue;
    float acc[RPG];
#pragma unroll
    for (int r = 0; r < RPG; ++r) acc[r] = 0.f;
    for (int j = 0; j < Stot; ++j) {
      const float vv = j < p.Lp ? p.pv[(b * p.Lp + j) * p.ldp + h * D + d] : p.v[(b * p.Sk + (j - p.Lp)) * p.ldv + h * D + d];
#pragma unroll
      for (int r = 0; r < RPG; ++r) acc[r] = fmaf(sc[(rg * RPG + r) * Stot + j], vv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
      const int rr = rg * RPG + r;
      if (rr < nq) p.ctx[(b * p.Sq + q0 + rr) * p.ldctx + h * D + d] = acc[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------- fp32 attention on the matrix pipe
// v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulate, bitwise an fmaf chain, at the f32 vector rate (157 TF/s per chip, 1/16 of bf16) -- 10-20x what the
// FMA loops of attn_f32_k reach, which is what made the fp32-accurate evaluation pass 6.4x the bf16 one.  Head dim 64 (every multi-head attention of the
// path; the D = 768 fusion keeps attn_f32_k).
//   one wave = 32 queries, "lane = owner query": S^T[key][q] = sum_d K[key][d] Q[q][d] has the wave's queries as COLUMNS, so the online softmax state of a
//   query is lane-local up to one exchange between the two lane halves (which hold complementary key rows of a tile);
//   Q lives in registers (32 per lane: d = half + 2s), K / V stream through LDS in 32-key tiles shared by the workgroup's 4 waves (row stride 65 floats:
//   the per-lane scalar reads K[key = lane][d] and V[key][d = lane] are both bank-conflict free), the next tile's global loads are in flight during the
//   64 MFMAs (4096 cycles) of the current one;
//   O^T[d][q] += sum_key V[key][d] P[q][key] contracts the keys in the order the S^T registers hold them (register s of half h is key (s&3) + 8(s>>2) + 4h):
//   the probabilities never leave their registers.
// Text options exactly as attn_f32_k (scale, adaptive reweight incl. the FLAVA row-0 variant, additive key mask), vision options: the [prefix | own] key set.
constexpr int AM_WAVES = 4, AM_KT = 32, AM_LD = 65;
__global__ __launch_bounds__(64 * AM_WAVES) void attn_f32_mfma_k(mart_attn_f32_desc p) {
  __shared__ float Ks[AM_KT * AM_LD], Vs[AM_KT * AM_LD];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y;
  const long long b = blockIdx.z;
  const int Stot = p.Lp + p.Sk;
  const int qi = blockIdx.x * (32 * AM_WAVES) + wave * 32 + l31;
  float qreg[32];
  {
    const float* qp = p.q + (b * p.Sq + min(qi, p.Sq - 1)) * p.ldq + h * 64 + hf;
#pragma unroll
    for (int s = 0; s < 32; ++s) qreg[s] = qp[2 * s];
  }
  float w0 = 1.f, w1 = 1.f; int sp = 0x7fffffff;
  if (p.sep) {
    sp = (int)p.sep[b * p.sep_stride];
    w0 = fminf(fmaxf(*p.w0, 0.f), 0.5f);
    w1 = fminf(fmaxf(*p.w1, 0.5f), 1.f);
  }
  const float rw = (p.sep && !(p.rw_skip_row0 && qi == 0)) ? (qi < sp ? w0 : w1) : 1.f;     // factor of this query's scores at keys >= sp
  f32x16 ot[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { ot[0][r] = 0.f; ot[1][r] = 0.f; }
  float m_run = -3.0e38f, l_run = 0.f;
  // staging: thread -> key row tid >> 3, 8 consecutive dims (tid & 7) * 8 of K and of V
  const int srow = tid >> 3, scol = (tid & 7) * 8;
  f32x4 kp[2], vp[2];
  auto fetch = [&](int t) {
    const int j = min(t * AM_KT + srow, Stot - 1);                      // rows past the last key: clamped copies (masked below)
    const float* kr = j < p.Lp ? p.pk + (b * p.Lp + j) * p.ldp + h * 64 : p.k + (b * p.Sk + (j - p.Lp)) * p.ldk + h * 64;
    const float* vr = j < p.Lp ? p.pv + (b * p.Lp + j) * p.ldp + h * 64 : p.v + (b * p.Sk + (j - p.Lp)) * p.ldv + h * 64;
    kp[0] = *(const f32x4*)(kr + scol); kp[1] = *(const f32x4*)(kr + scol + 4);
    vp[0] = *(const f32x4*)(vr + scol); vp[1] = *(const f32x4*)(vr + scol + 4);
  };
  auto stash = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) { Ks[srow * AM_LD + scol + 4 * u + e] = kp[u][e]; Vs[srow * AM_LD + scol + 4 * u + e] = vp[u][e]; }
  };
  const int ntiles = (Stot + AM_KT - 1) / AM_KT;
  fetch(0);
  stash();
  __syncthreads();
  const bool has_mask = p.attn_mask != nullptr;
  const bool wave_on = blockIdx.x * (32 * AM_WAVES) + wave * 32 < p.Sq;      // a wave past the last query only helps staging (393 queries = 12.3 waves of 16)
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) fetch(t + 1);
    if (wave_on) {
    // ---- S^T tile
    // operands of a whole MFMA chain are fetched from LDS BEFORE the chain (the compiler otherwise places each ds_read right in front of the two MFMAs
    // that use it and waits for it there: one exposed LDS round trip per 128 cycles of matrix work); the V operands of the second chain are requested
    // here as well, so that their latency hides behind the softmax arithmetic
    float ka[32], va[2][16];
#pragma unroll
    for (int s = 0; s < 32; ++s) ka[s] = Ks[l31 * AM_LD + hf + 2 * s];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int key = (s & 3) + 8 * (s >> 2) + 4 * hf;
      va[0][s] = Vs[key * AM_LD + l31];
      va[1][s] = Vs[key * AM_LD + 32 + l31];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s], qreg[s], st, 0, 0, 0);
    // f32 MFMA shares the SIMD's f32 ALUs with VALU work (PMC: 10 VALU per MFMA held the matrix pipe at 45 %), so the softmax of a tile is kept lean:
    // the text options and the key-range test run only where they can matter (wave-uniform branches), the exponential is exp2 of a pre-scaled
    // argument (v_exp_f32, 1 ulp; the scaling adds |x| * 6e-8 relative: 1e-6 at the score ranges of this model).
    const bool plain_tile = !p.sep && !has_mask && (t + 1) * AM_KT <= Stot;
    float mx = -3.0e38f;
    if (plain_tile) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] *= p.scale; mx = fmaxf(mx, st[r]); }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = t * AM_KT + (r & 3) + 8 * (r >> 2) + 4 * hf;        // key of this register
        float v = st[r] * p.scale;
        if (j >= sp) v *= rw;
        if (has_mask) v += (p.attn_mask[b * p.Sk + min(j, Stot - 1)] == 0) ? -10000.0f : 0.f;
        if (j >= Stot) v = -3.0e38f;
        st[r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    constexpr float L2E = 1.44269504088896340736f;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f((st[r] - m_new) * L2E);     // masked / out-of-range keys: exp2(-huge) = 0
      st[r] = e;
      sum += e;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * L2E);
    l_run = l_run * alpha + sum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ot[0][r] *= alpha; ot[1][r] *= alpha; }
    // ---- O^T += V^T P^T, keys in register order
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      ot[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[0][s], st[s], ot[0], 0, 0, 0);
      ot[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[1][s], st[s], ot[1], 0, 0, 0);
    }
    }
    __syncthreads();                                                   // every wave is through tile t
    if (t + 1 < ntiles) { stash(); __syncthreads(); }
  }
  if (qi < p.Sq) {
    const float inv = 1.f / l_run;
    float* op = p.ctx + (b * p.Sq + qi) * p.ldctx + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(f32x4*)(op + dt * 32 + 8 * g + 4 * hf) = f32x4{ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv, ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv};
  }
}

// ---- the image-text fusion op (BertFusion, modeling_unimo.py:400-414: one head of 768, unscaled, unmasked) on the f32 matrix pipe.
// One 8-wave workgroup per (example, 32 queries).  Phase 1: S^T[key][q] = sum_d vis[key][d] ctx[q][d], d streamed through LDS in 32-wide chunks, wave w
// owning the 32-key tiles {w, w + 8} (lane = query, registers = keys).  Phase 2: softmax over all keys of a query = lane-local + one exchange between the
// lane halves + one through LDS between the waves; the probabilities go to an LDS image P[key][q].  Phase 3: O^T[d][q] = sum_key vis[key][d] P[key][q],
// keys streamed in 16-row chunks, wave w owning the 32-wide output column tiles {3w, 3w + 1, 3w + 2}.  ~90 us per layer at B = 256 against 1.4 ms for
// the FMA loops of attn_f32_k<768>.
constexpr int FM_DC = 32, FM_KC = 16, FM_LD = 33, FM_VLD = 769;
__global__ __launch_bounds__(512) void fusion_f32_mfma_k(mart_attn_f32_desc p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  const int NKP = ntiles * 32;
  const int areg = max(NKP * FM_LD, FM_KC * FM_VLD);
  float* Va = fsm;                         // phase 1: vis chunk [NKP][33]; phase 3: vis chunk [16][769]
  float* Ps = fsm + areg;                  // [NKP][33] probabilities
  float* Qs = Ps + NKP * FM_LD;            // [32][33] query chunk
  float* red = Qs + 32 * FM_LD;            // [2][8][32] cross-wave max / sum
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long b = blockIdx.z;
  const int q0 = blockIdx.x * 32;
  const int Sk = p.Sk;
  const bool own1 = wave + 8 < ntiles;     // (wave < ntiles always holds for tile 0 when ntiles >= 8; guarded below for small key counts)
  const bool own0 = wave < ntiles;
  f32x16 st[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { st[0][r] = 0.f; st[1][r] = 0.f; }
  // ---------------- phase 1
  for (int dc = 0; dc < 768 / FM_DC; ++dc) {
    __syncthreads();
    {
      const int q = tid >> 4, dd = (tid & 15) * 2;
      const float* qp = p.q + (b * p.Sq + min(q0 + q, p.Sq - 1)) * p.ldq + dc * FM_DC + dd;
      Qs[q * FM_LD + dd] = qp[0]; Qs[q * FM_LD + dd + 1] = qp[1];
    }
    for (int i = tid; i < NKP * 8; i += 512) {
      const int key = i >> 3, c4 = (i & 7) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (key < Sk) v = *(const f32x4*)(p.k + (b * Sk + key) * p.ldk + dc * FM_DC + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) Va[key * FM_LD + c4 + e] = v[e];
    }
    __syncthreads();
    float qa[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) qa[s] = Qs[l31 * FM_LD + hf + 2 * s];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 0 ? !own0 : !own1) continue;
      const int t = wave + 8 * u;
      float ka[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) ka[s] = Va[(t * 32 + l31) * FM_LD + hf + 2 * s];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 16; ++s) st[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s], qa[s], st[u], 0, 0, 0);
    }
  }
  // ---------------- phase 2: softmax over keys (rows of S^T), per query column
  float mx = -3.0e38f;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (wave + 8 * u) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
      const bool ok = (u == 0 ? own0 : own1) && key < Sk;
      st[u][r] = ok ? st[u][r] * p.scale : -3.0e38f;
      mx = fmaxf(mx, st[u][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (hf == 0) red[wave * 32 + l31] = mx;
  __syncthreads();
  float gm = -3.0e38f;
#pragma unroll
  for (int w = 0; w < 8; ++w) gm = fmaxf(gm, red[w * 32 + l31]);
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = st[u][r] > -1.0e38f ? expf(st[u][r] - gm) : 0.f;
      st[u][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  if (hf == 0) red[256 + wave * 32 + l31] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) tot += red[256 + w * 32 + l31];
  const float inv = 1.f / tot;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (u == 0 ? !own0 : !own1) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) Ps[((wave + 8 * u) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf) * FM_LD + l31] = st[u][r] * inv;
  }
  // ---------------- phase 3
  f32x16 ot[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[j][r] = 0.f;
  const int nkc = (Sk + FM_KC - 1) / FM_KC;
  for (int kc = 0; kc < nkc; ++kc) {
    __syncthreads();                                       // previous chunk consumed (first pass: the P image is complete, the phase-1 chunk is dead)
#pragma unroll
    for (int u = 0; u < (FM_KC * 192) / 512; ++u) {
      const int i = u * 512 + tid, key = i / 192, c4 = (i % 192) * 4;
      const int kk = kc * FM_KC + key;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kk < Sk) v = *(const f32x4*)(p.v + (b * Sk + kk) * p.ldv + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) Va[key * FM_VLD + c4 + e] = v[e];
    }
    __syncthreads();
    float pb[8], va[3][8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k2 = 2 * s + hf;
      pb[s] = (kc * FM_KC + k2 < NKP) ? Ps[(kc * FM_KC + k2) * FM_LD + l31] : 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) va[j][s] = Va[k2 * FM_VLD + (3 * wave + j) * 32 + l31];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int j = 0; j < 3; ++j) ot[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[j][s], pb[s], ot[j], 0, 0, 0);
  }
  const int qi = q0 + l31;
  if (qi < p.Sq) {
    float* op = p.ctx + (b * p.Sq + qi) * p.ldctx;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(f32x4*)(op + (3 * wave + j) * 32 + 8 * g + 4 * hf) = f32x4{ot[j][4 * g], ot[j][4 * g + 1], ot[j][4 * g + 2], ot[j][4 * g + 3]};
  }
}

// ---------------------------------------------------------------------------------------------------- fp32-accurate BACKWARD (verification mode)
// Gradients of the precise forward, so that the training step can be held to the fp32 reference at ~1e-3 where the bf16 path can only be
// compared with a control (plain N(0,0.02) weights: tests/test_parity_full_gpu.py).  Dense contractions reuse the split-operand GEMMs
// (dgrad: K-concatenated splits through mart_gemm_nt; wgrad: ROW-concatenated splits through mart_gemm_tn, contraction over 3M rows); the
// kernels below are the pieces without an MFMA form.  Speed is not a goal here (atomics, plain FMA loops).

// dst[3M, K] bf16: role 0 (X of X^T Y) = [hi ; lo ; hi], role 1 (Y) = [hi ; hi ; lo]  =>  X'^T Y' = hi^T hi + lo^T hi + hi^T lo
__global__ void split3_rows_k(const float* __restrict__ src, long long ld, bf16* __restrict__ dst, int M, int K, int role, int terms) {
  const long long total = (long long)M * (K / 4);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / (K / 4);
    const int c = (int)(i % (K / 4)) * 4;
    const f32x4 x = *(const f32x4*)(src + r * ld + c);
    bf16x4 h, m, l;
    split_terms(x, h, m, l);
    if (terms == 3) {
      const bf16x4 a[6] = {h, h, m, m, h, l}, b[6] = {h, m, h, m, l, h};
#pragma unroll
      for (int t = 0; t < 6; ++t) *(bf16x4*)(dst + ((long long)t * M + r) * K + c) = role == 0 ? a[t] : b[t];
    } else {
      *(bf16x4*)(dst + r * K + c) = h;
      *(bf16x4*)(dst + ((long long)M + r) * K + c) = role == 0 ? m : h;
      *(bf16x4*)(dst + (2LL * M + r) * K + c) = role == 0 ? h : m;
    }
  }
}
__global__ void act_f32_k(const float* __restrict__ z, float* __restrict__ a, int act, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) a[i] = act_fwd(z[i], act);
}
__global__ void act_bwd_f32_k(const float* __restrict__ dy, const float* __restrict__ z, int act, float* __restrict__ dz, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dz[i] = dy[i] * act_grad(z[i], act);
}
// out[c] += sum_r src[r, c]   (bias gradients; 64 columns x 4 row lanes per workgroup, one atomic per column and workgroup)
__global__ __launch_bounds__(256) void colsum_f32_k(const float* __restrict__ src, long long ld, float* __restrict__ out, int R, int C) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int per = (R + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(R, r0 + per);
  float acc = 0.f;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) acc += src[(long long)r * ld + c];
  part[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < C) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// Backward of attn_f32_k: one workgroup = QTB query rows of one (batch, head); probabilities are recomputed exactly as the forward does.
//   P = softmax(v), v = s * scale * f (+ mask);  dP = dctx V^T;  dv_ = P o (dP - rowsum(P o dP));  d(q.k) = dv_ * f * scale
//   dq (written), dk / dv / prefix gradients (atomically accumulated: callers zero them), d(adaptive weights) (atomically accumulated,
//   clamp sub-gradient = 1 inside AND at the bounds, as torch.clamp's backward, modeling_unimo.py:342-349).
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_f32_k(mart_attn_bwd_f32_desc pb) {
  constexpr int QTB = D >= 256 ? 8 : 16;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const mart_attn_f32_desc& p = pb.f;
  const int Stot = p.Lp + p.Sk;
  float* qs = sm;                        // [QTB][D] queries
  float* gs = qs + QTB * D;              // [QTB][D] d(ctx)
  float* pr = gs + QTB * D;              // [QTB][Stot] probabilities
  float* ds = pr + QTB * Stot;           // [QTB][Stot] dP, then d(q.k)
  float* sp = ds + QTB * Stot;           // [QTB][Stot] scaled raw scores (for the adaptive-weight gradients)
  const int q0 = blockIdx.x * QTB, h = blockIdx.y;
  const long long b = blockIdx.z;
  const int tid = threadIdx.x;
  const int nq = min(QTB, p.Sq - q0);
  for (int i = tid; i < QTB * D; i += 256) {
    const int r = i / D, d = i % D;
    qs[i] = r < nq ? p.q[(b * p.Sq + q0 + r) * p.ldq + h * D + d] : 0.f;
    gs[i] = r < nq ? pb.dctx[(b * p.Sq + q0 + r) * pb.lddctx + h * D + d] : 0.f;
  }
  __syncthreads();
  for (int j = tid; j < Stot; j += 256) {
    const float* kr = j < p.Lp ? p.pk + (b * p.Lp + j) * p.ldp + h * D : p.k + (b * p.Sk + (j - p.Lp)) * p.ldk + h * D;
    const float* vr = j < p.Lp ? p.pv + (b * p.Lp + j) * p.ldp + h * D : p.v + (b * p.Sk + (j - p.Lp)) * p.ldv + h * D;
    float acc[QTB], dpa[QTB];
#pragma unroll
    for (int r = 0; r < QTB; ++r) { acc[r] = 0.f; dpa[r] = 0.f; }
    for (int d = 0; d < D; d += 4) {
      const f32x4 kv = *(const f32x4*)(kr + d), vv = *(const f32x4*)(vr + d);
#pragma unroll
      for (int r = 0; r < QTB; ++r) {
        const f32x4 qv = *(const f32x4*)(qs + r * D + d), gv = *(const f32x4*)(gs + r * D + d);
        acc[r] = fmaf(qv[0], kv[0], acc[r]); acc[r] = fmaf(qv[1], kv[1], acc[r]);
        acc[r] = fmaf(qv[2], kv[2], acc[r]); acc[r] = fmaf(qv[3], kv[3], acc[r]);
        dpa[r] = fmaf(gv[0], vv[0], dpa[r]); dpa[r] = fmaf(gv[1], vv[1], dpa[r]);
        dpa[r] = fmaf(gv[2], vv[2], dpa[r]); dpa[r] = fmaf(gv[3], vv[3], dpa[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < QTB; ++r) { pr[r * Stot + j] = acc[r]; ds[r * Stot + j] = dpa[r]; }
  }
  __syncthreads();
  {
    const int wave = tid >> 6, lane = tid & 63;
    float w0 = 1.f, w1 = 1.f; int s = 0x7fffffff;
    if (p.sep) {
      s = (int)p.sep[b * p.sep_stride];
      w0 = fminf(fmaxf(*p.w0, 0.f), 0.5f);
      w1 = fminf(fmaxf(*p.w1, 0.5f), 1.f);
    }
    float dc0 = 0.f, dc1 = 0.f;
    for (int r = wave; r < nq; r += 4) {
      const int qi = q0 + r;
      float mx = -3.0e38f;
      for (int j = lane; j < Stot; j += 64) {
        const float spre = pr[r * Stot + j] * p.scale;
        float v = spre;
        if (p.sep && j >= s && !(p.rw_skip_row0 && qi == 0)) v *= (qi < s ? w0 : w1);
        if (p.attn_mask && p.attn_mask[b * p.Sk + j] == 0) v += -10000.0f;
        sp[r * Stot + j] = spre;
        pr[r * Stot + j] = v;
        mx = fmaxf(mx, v);
      }
      mx = wave_max(mx);
      float sum = 0.f;
      for (int j = lane; j < Stot; j += 64) {
        const float e = expf(pr[r * Stot + j] - mx);
        pr[r * Stot + j] = e;
        sum += e;
      }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      float dl = 0.f;
      for (int j = lane; j < Stot; j += 64) {
        const float pj = pr[r * Stot + j] * inv;
        pr[r * Stot + j] = pj;
        dl += pj * ds[r * Stot + j];
      }
      dl = wave_sum(dl);
      for (int j = lane; j < Stot; j += 64) {
        const float dv_ = pr[r * Stot + j] * (ds[r * Stot + j] - dl);        // gradient w.r.t. the post-reweight score
        float f = 1.f;
        if (p.sep && j >= s && !(p.rw_skip_row0 && qi == 0)) {
          f = qi < s ? w0 : w1;
          if (qi < s) dc0 += dv_ * sp[r * Stot + j]; else dc1 += dv_ * sp[r * Stot + j];
        }
        ds[r * Stot + j] = dv_ * f * p.scale;                               // gradient w.r.t. q . k
      }
    }
    if (p.sep && pb.dw) {
      dc0 = wave_sum(dc0); dc1 = wave_sum(dc1);
      if (lane == 0) {
        const float r0 = *p.w0, r1 = *p.w1;
        if (r0 >= 0.f && r0 <= 0.5f && dc0 != 0.f) atomicAdd(pb.dw, dc0);
        if (r1 >= 0.5f && r1 <= 1.f && dc1 != 0.f) atomicAdd(pb.dw + 1, dc1);
      }
    }
  }
  __syncthreads();
  // dq[r][d] = sum_j ds[r][j] key_j[d]   (thread = column d)
  for (int d = tid; d < D; d += 256) {
    float acc[QTB];
#pragma unroll
    for (int r = 0; r < QTB; ++r) acc[r] = 0.f;
    for (int j = 0; j < Stot; ++j) {
      const float kk = j < p.Lp ? p.pk[(b * p.Lp + j) * p.ldp + h * D + d] : p.k[(b * p.Sk + (j - p.Lp)) * p.ldk + h * D + d];
#pragma unroll
      for (int r = 0; r < QTB; ++r) acc[r] = fmaf(ds[r * Stot + j], kk, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < QTB; ++r)
      if (r < nq) pb.dq[(b * p.Sq + q0 + r) * pb.lddq + h * D + d] = acc[r];
  }
  // dk_j[d] += sum_r ds[r][j] q_r[d];  dv_j[d] += sum_r P[r][j] dctx_r[d]
  for (long long i = tid; i < (long long)Stot * D; i += 256) {
    const int j = (int)(i / D), d = (int)(i % D);
    float ak = 0.f, av = 0.f;
#pragma unroll
    for (int r = 0; r < QTB; ++r) {
      ak = fmaf(ds[r * Stot + j], qs[r * D + d], ak);
      av = fmaf(pr[r * Stot + j], gs[r * D + d], av);
    }
    if (j < p.Lp) {
      atomicAdd(pb.dpk + (b * p.Lp + j) * pb.lddp + h * D + d, ak);
      atomicAdd(pb.dpv + (b * p.Lp + j) * pb.lddp + h * D + d, av);
    } else {
      atomicAdd(pb.dk + (b * p.Sk + (j - p.Lp)) * pb.lddk + h * D + d, ak);
      atomicAdd(pb.dv + (b * p.Sk + (j - p.Lp)) * pb.lddv + h * D + d, av);
    }
  }
}

template <int D>
int launch_attn_bwd(const mart_attn_bwd_f32_desc* d, hipStream_t st) {
  constexpr int QTB = D >= 256 ? 8 : 16;
  const int Stot = d->f.Lp + d->f.Sk;
  const size_t lds = (size_t)(2 * QTB * D + 3 * QTB * Stot) * sizeof(float);
  MART_CHECK(lds <= 160 * 1024, "attn_bwd_f32: keys do not fit the LDS score tiles");
  static MartAttrOnce once;
  bool* attr_set = once.slot();
  auto kern = attn_bwd_f32_k<D>;
  if (!*attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      mart_set_error("attn_bwd_f32: hipFuncSetAttribute failed");
      return -2;
    }
    *attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((d->f.Sq + QTB - 1) / QTB, d->f.nh, d->f.B), dim3(256), lds, st, *d);
  MART_LAUNCH_CHECK();
  return 0;
}

template <int D>
int launch_attn(const mart_attn_f32_desc* d, hipStream_t st) {
  const int Stot = d->Lp + d->Sk;
  const size_t lds = (size_t)(QT * D + QT * Stot) * sizeof(float);
  MART_CHECK(lds <= 160 * 1024, "attn_f32: keys do not fit the LDS score tile");
  static MartAttrOnce once;
  bool* attr_set = once.slot();
  auto kern = attn_f32_k<D>;
  if (!*attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      mart_set_error("attn_f32: hipFuncSetAttribute failed");
      return -2;
    }
    *attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((d->Sq + QT - 1) / QT, d->nh, d->B), dim3(256), lds, st, *d);
  MART_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int mart_split_bf16x3_rows(const float* src, long long ld, const int32_t* gather, void* dst_bf16, int rows, int K, int role, int terms, void* stream) {
  MART_CHECK((terms == 2 || terms == 3) && src && gather && dst_bf16 && rows > 0 && K > 0 && K % 4 == 0 && ld >= K && ld % 4 == 0 && (role == 0 || role == 1), "split_bf16x3_rows: bad args");
  const long long total = (long long)rows * (K / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(split3_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld, (bf16*)dst_bf16, rows, K, role, gather, terms);
  MART_LAUNCH_CHECK();
  return 0;
}

extern "C" int mart_split_bf16x3(const float* src, long long ld, void* dst_bf16, int rows, int K, int role, int terms, void* stream) {
  MART_CHECK((terms == 2 || terms == 3) && src && dst_bf16 && rows > 0 && K > 0 && K % 4 == 0 && ld >= K && ld % 4 == 0 && (role == 0 || role == 1), "split_bf16x3: bad args");
  if (terms == 2 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst_bf16 & 7) == 0) {
    const long long groups = ((long long)rows * (K / 4) + 1023) / 1024;
    const int blk = (int)(groups < 8192 ? groups : 8192);
    hipLaunchKernelGGL(split3_fast_k, dim3(blk), dim3(256), 0, (hipStream_t)stream, src, ld, (bf16*)dst_bf16, rows, K, role);
    MART_LAUNCH_CHECK();
    return 0;
  }
  const long long total = (long long)rows * (K / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(split3_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld, (bf16*)dst_bf16, rows, K, role, (const int32_t*)nullptr, terms);
  MART_LAUNCH_CHECK();
  return 0;
}

extern "C" int mart_patchify_f32(const float* pixels, const int32_t* index, float* out, int B, int S, int p, void* stream) {
  MART_CHECK(pixels && out && B > 0 && S > 0 && p > 0 && S % p == 0 && p % 4 == 0, "patchify_f32: bad args");
  const int g = S / p;
  hipLaunchKernelGGL(patchify_f32_k, dim3(B * 2 * g * g), dim3(192), 0, (hipStream_t)stream, pixels, index, out, S, p);
  MART_LAUNCH_CHECK();
  return 0;
}

extern "C" int mart_vision_assemble_f32(const float* patch, const float* cls, const float* pos, float* s, int B, int P, int H, int tail_shift,
                                        void* stream) {
  MART_CHECK(patch && cls && pos && s && B > 0 && P > 0 && H % 4 == 0 && (tail_shift == 0 || tail_shift == 1), "vision_assemble_f32: bad args");
  hipLaunchKernelGGL(vision_assemble_f32_k, dim3(B * (1 + 2 * P)), dim3(192), 0, (hipStream_t)stream, patch, cls, pos, s, P, H, tail_shift);
  MART_LAUNCH_CHECK();
  return 0;
}

extern "C" int mart_attn_fwd_f32(const mart_attn_f32_desc* d, void* stream) {
  MART_CHECK(d && d->q && d->k && d->v && (d->ctx || d->ctx_split3), "attn_f32: null pointer");
  MART_CHECK(d->B > 0 && d->nh > 0 && d->Sq > 0 && d->Sk > 0 && d->Lp >= 0, "attn_f32: bad shape");
  MART_CHECK(d->ldq % 4 == 0 && d->ldk % 4 == 0 && d->ldv % 4 == 0 && (d->Lp == 0 || (d->pk && d->pv && d->ldp % 4 == 0)), "attn_f32: bad strides / prefix");
  MART_CHECK((d->w0 == nullptr) == (d->w1 == nullptr) && (!d->sep || (d->w0 && d->Lp == 0)), "attn_f32: reweight needs w0/w1 and no prefix");
  MART_CHECK(!d->attn_mask || d->Lp == 0, "attn_f32: mask with prefix unsupported");
  static const int scalar = getenv("MART_ATTN_F32_SCALAR") ? atoi(getenv("MART_ATTN_F32_SCALAR")) : 0;   // 1: the FMA-loop kernel (A/B, tests)
  static const int split_ok = getenv("MART_ATTN_F32_SPLIT") ? atoi(getenv("MART_ATTN_F32_SPLIT")) : 1;   // 0: exact f32 for evaluation passes too
  if (d->fast && split_ok && !scalar && d->D == 64 && d->ldctx % 4 == 0 && d->ldctx3 % 4 == 0 &&
      (((uintptr_t)d->ctx | (uintptr_t)d->ctx_split3 | (uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v | (uintptr_t)d->pk | (uintptr_t)d->pv) & 15) == 0)
    return mart_attn_split_launch(d, stream);          // csrc/attention.hip: attn_split_fwd_k
  MART_CHECK(!d->ctx_split3 || d->ctx, "attn_f32: ctx_split3 without ctx needs the fast path (fast = 1, D = 64, 16-byte aligned operands)");
  MART_CHECK(!d->ctx_split3, "attn_f32: ctx_split3 is an output of the fast path only");
  if (d->D == 64 && !scalar && d->ldctx % 4 == 0 && ((uintptr_t)d->ctx & 15) == 0 && ((uintptr_t)d->k & 15) == 0 && ((uintptr_t)d->v & 15) == 0 &&
      (!d->pk || (((uintptr_t)d->pk & 15) == 0 && ((uintptr_t)d->pv & 15) == 0))) {
    hipLaunchKernelGGL(attn_f32_mfma_k, dim3((d->Sq + 32 * AM_WAVES - 1) / (32 * AM_WAVES), d->nh, d->B), dim3(64 * AM_WAVES), 0, (hipStream_t)stream, *d);
    MART_LAUNCH_CHECK();
    return 0;
  }
  if (d->D == 768 && !scalar && d->nh == 1 && d->Lp == 0 && !d->attn_mask && !d->sep && d->Sk <= 512 && d->ldq % 4 == 0 && d->ldctx % 4 == 0 &&
      ((uintptr_t)d->ctx & 15) == 0 && ((uintptr_t)d->k & 15) == 0 && ((uintptr_t)d->v & 15) == 0) {
    const int ntiles = (d->Sk + 31) / 32, NKP = ntiles * 32;
    const int areg = NKP * FM_LD > FM_KC * FM_VLD ? NKP * FM_LD : FM_KC * FM_VLD;
    const size_t lds = (size_t)(areg + NKP * FM_LD + 32 * FM_LD + 512) * sizeof(float);
    static MartAttrOnce once;
    bool* attr_set = once.slot();
    if (!*attr_set) {
      if (hipFuncSetAttribute((const void*)fusion_f32_mfma_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        mart_set_error("attn_f32: hipFuncSetAttribute failed");
        return -2;
      }
      *attr_set = true;
    }
    hipLaunchKernelGGL(fusion_f32_mfma_k, dim3((d->Sq + 31) / 32, 1, d->B), dim3(512), lds, (hipStream_t)stream, *d, ntiles);
    MART_LAUNCH_CHECK();
    return 0;
  }
  if (d->D == 64) return launch_attn<64>(d, (hipStream_t)stream);
  if (d->D == 768) return launch_attn<768>(d, (hipStream_t)stream);
  mart_set_error("attn_f32: head dim must be 64 (multi-head attention) or 768 (fusion)");
  return -1;
}

extern "C" int mart_split_bf16x3_stack(const float* src, long long ld, void* dst_bf16, int M, int K, int role, int terms, void* stream) {
  MART_CHECK((terms == 2 || terms == 3) && src && dst_bf16 && M > 0 && K > 0 && K % 4 == 0 && ld >= K && ld % 4 == 0 && (role == 0 || role == 1), "split_bf16x3_stack: bad args");
  const long long total = (long long)M * (K / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(split3_rows_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld, (bf16*)dst_bf16, M, K, role, terms);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_act_f32(const float* z, float* a, int act, long long n, void* stream) {
  MART_CHECK(z && a && n > 0 && act >= ACT_NONE && act <= ACT_QGELU, "act_f32: bad args");
  hipLaunchKernelGGL(act_f32_k, dim3((int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream, z, a, act, n);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_act_bwd_f32(const float* dy, const float* z, int act, float* dz, long long n, void* stream) {
  MART_CHECK(dy && z && dz && n > 0 && act >= ACT_NONE && act <= ACT_QGELU, "act_bwd_f32: bad args");
  hipLaunchKernelGGL(act_bwd_f32_k, dim3((int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream, dy, z, act, dz, n);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_colsum_f32(const float* src, long long ld, float* out, int R, int C, void* stream) {
  MART_CHECK(src && out && R > 0 && C > 0 && ld >= C, "colsum_f32: bad args");
  const int slices = R >= 4096 ? 64 : (R >= 256 ? 8 : 1);
  hipLaunchKernelGGL(colsum_f32_k, dim3((C + 63) / 64, slices), dim3(256), 0, (hipStream_t)stream, src, ld, out, R, C);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_attn_bwd_f32(const mart_attn_bwd_f32_desc* d, void* stream) {
  MART_CHECK(d && d->f.q && d->f.k && d->f.v && d->dctx && d->dq && d->dk && d->dv, "attn_bwd_f32: null pointer");
  MART_CHECK(d->f.B > 0 && d->f.nh > 0 && d->f.Sq > 0 && d->f.Sk > 0 && d->f.Lp >= 0, "attn_bwd_f32: bad shape");
  MART_CHECK(d->f.ldq % 4 == 0 && d->f.ldk % 4 == 0 && d->f.ldv % 4 == 0 && (d->f.Lp == 0 || (d->f.pk && d->f.pv && d->dpk && d->dpv && d->f.ldp % 4 == 0)), "attn_bwd_f32: bad strides / prefix");
  MART_CHECK((d->f.w0 == nullptr) == (d->f.w1 == nullptr) && (!d->f.sep || (d->f.w0 && d->f.Lp == 0)), "attn_bwd_f32: reweight needs w0/w1 and no prefix");
  if (d->f.D == 64) return launch_attn_bwd<64>(d, (hipStream_t)stream);
  if (d->f.D == 768) return launch_attn_bwd<768>(d, (hipStream_t)stream);
  mart_set_error("attn_bwd_f32: head dim must be 64 or 768");
  return -1;
}

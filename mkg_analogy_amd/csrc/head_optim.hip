// Scoring-head losses, ranking and the fused multi-tensor AdamW.
#include "common.h"
#include "mart_hip.h"

namespace {
constexpr int TPB = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float t = -3.0e38f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, sh[i]);
  return t;
}

// LabelSmoothSoftmaxCEV1 (lit_models/utils.py:42-66): target = eps/C everywhere, 1-eps at the label.
// loss_row = -(sum_c t_c * logp_c) = lse*T - (eps/C)*sum_c x_c - (1-eps-eps/C)*x_label ,  T = 1-eps + eps*(C-1)/C
// Rows whose label == ignore (utils.py:49-52, 58): loss 0, not counted in n_valid, zero gradient.  A label outside [0, C) that is not
// ``ignore`` (the reference's scatter_ raises there): the row's loss is NaN, nothing is read out of bounds, status[0] is set.
__device__ __forceinline__ int label_class(long long lab, long long ignore, int C) {      // 0: scored, 1: ignored, 2: out of range
  return lab == ignore ? 1 : ((lab < 0 || lab >= (long long)C) ? 2 : 0);
}
__global__ void lsce_fwd_k(const float* __restrict__ lg, int ld, const int64_t* __restrict__ label, long long ignore, float eps, float* loss_rows,
                           float* lse_out, int* status, int R, int C) {
  __shared__ float sh[8];
  const int r = blockIdx.x;
  const float* x = lg + (long long)r * ld;
  float mx = -3.0e38f;
  for (int c = threadIdx.x; c < C; c += TPB) mx = fmaxf(mx, x[c]);
  mx = block_max(mx, sh);
  float se = 0.f, sx = 0.f;
  for (int c = threadIdx.x; c < C; c += TPB) { se += __expf(x[c] - mx); sx += x[c]; }
  se = block_sum(se, sh);
  sx = block_sum(sx, sh);
  if (threadIdx.x == 0) {
    const float lse = mx + __logf(se);
    const float neg = eps / (float)C, pos = 1.f - eps;
    const float T = pos + neg * (float)(C - 1);
    const long long lab = label[r];
    const int cls = label_class(lab, ignore, C);
    float loss = 0.f;
    if (cls == 0) {
      const float xl = x[lab];
      loss = lse * T - neg * (sx - xl) - pos * xl;
    } else if (cls == 2) {
      loss = __builtin_nanf("");
      if (status) atomicOr(status, 1);
    }
    loss_rows[r] = loss;
    lse_out[r] = lse;
  }
}
// reduction over the rows (deterministic: one workgroup, fixed order): out[0] = sum(rows) / n_valid ('mean', utils.py:59-60; 0/0 = NaN as the
// reference when every row is ignored) or sum(rows) ('sum', :61-62); out[1] = n_valid (read by the backward pass)
__global__ void lsce_reduce_k(const float* __restrict__ loss_rows, const int64_t* __restrict__ label, long long ignore, int mean, float* out, int R) {
  __shared__ float sh[8];
  float s = 0.f, n = 0.f;
  for (int r = threadIdx.x; r < R; r += TPB) { s += loss_rows[r]; n += (label[r] != ignore) ? 1.f : 0.f; }
  s = block_sum(s, sh);
  n = block_sum(n, sh);
  if (threadIdx.x == 0) { out[0] = mean ? s / n : s; out[1] = n; }
}
__global__ void lsce_bwd_k(const float* __restrict__ lg, int ld, const int64_t* __restrict__ label, long long ignore, const float* __restrict__ lse, float eps,
                           const float* __restrict__ gscale, int g_per_row, float rowscale, const float* __restrict__ nvalid, bf16* ob, int ldo, float* of,
                           int R, int C) {
  const int r = blockIdx.x;
  const float* x = lg + (long long)r * ld;
  float g = (gscale ? gscale[g_per_row ? r : 0] : 1.f) * rowscale;
  if (nvalid) g /= nvalid[0];
  const float neg = eps / (float)C, pos = 1.f - eps;
  const float T = pos + neg * (float)(C - 1);
  const float l = lse[r];
  const long long lab = label[r];
  const int cls = label_class(lab, ignore, C);
  for (int c = threadIdx.x; c < ldo; c += TPB) {
    float v = 0.f;
    if (c < C && cls == 0) v = g * (__expf(x[c] - l) * T - ((long long)c == lab ? pos : neg));
    if (c < C && cls == 2) v = __builtin_nanf("");
    if (ob) ob[(long long)r * ldo + c] = f2bf(v);
    if (of && c < C) of[(long long)r * C + c] = v;
  }
}
// rank = 1 + #(logit > logit[label]); a label outside [0, C): rank 0 (never a valid rank), nothing read out of bounds
__global__ void rank_k(const float* __restrict__ lg, int ld, const int64_t* __restrict__ label, int64_t* rank, int R, int C) {
  __shared__ float sh[8];
  const int r = blockIdx.x;
  const float* x = lg + (long long)r * ld;
  const long long lab = label[r];
  if (lab < 0 || lab >= (long long)C) { if (threadIdx.x == 0) rank[r] = 0; return; }
  const float xl = x[lab];
  float cnt = 0.f;
  for (int c = threadIdx.x; c < C; c += TPB) cnt += (x[c] > xl) ? 1.f : 0.f;
  cnt = block_sum(cnt, sh);
  if (threadIdx.x == 0) rank[r] = (int64_t)(cnt + 0.5f) + 1;
}

// cosine similarity pieces (F.cosine_similarity eps = 1e-8 clamps each norm)
struct Cos { float dot, na, nb; };
__device__ __forceinline__ Cos cos_parts(const float* a, const float* b, int H, float* sh) {
  float d = 0.f, x = 0.f, y = 0.f;
  for (int c = threadIdx.x; c < H; c += TPB) { d += a[c] * b[c]; x += a[c] * a[c]; y += b[c] * b[c]; }
  Cos o;
  o.dot = block_sum(d, sh); o.na = sqrtf(block_sum(x, sh)); o.nb = sqrtf(block_sum(y, sh));
  return o;
}
// Row lookup of the relaxation loss.  Dense trans [B, L, H]: row b * L + pos.  Row-subset pass (rows != NULL: trans is the COMPACT [B * nr, H] tensor,
// rows[b * nr + j] the flat id b * L + position of slot j): the first slot of example b that names the position; a position the pass was not
// promised gives -1 (the loss row is NaN, nothing is read).
__device__ __forceinline__ long long sim_row(const int32_t* rows, int nr, int b, int L, long long pos) {
  if (pos < 0) pos += L;                                       // negative positions wrap like the reference's fancy indexing
  if (!rows) return (long long)b * L + pos;
  const int f = b * L + (int)pos;
  for (int j = 0; j < nr; ++j) if (rows[b * nr + j] == f) return (long long)b * nr + j;
  return -1;
}
__global__ void simloss_fwd_k(const float* __restrict__ tr, const int64_t* rel, const int64_t* qh, const int64_t* ah, const int32_t* rows, int nr,
                              float* loss_rows, int B, int L, int H) {
  __shared__ float sh[8];
  const int b = blockIdx.x;
  const long long rq = sim_row(rows, nr, b, L, qh[b]), ra = sim_row(rows, nr, b, L, ah[b]);
  const long long r0 = sim_row(rows, nr, b, L, rel[2 * b]), r1 = sim_row(rows, nr, b, L, rel[2 * b + 1]);
  if ((rq | ra | r0 | r1) < 0) { if (threadIdx.x == 0) loss_rows[b] = __builtin_nanf(""); return; }
  Cos e = cos_parts(tr + rq * H, tr + ra * H, H, sh);
  Cos r = cos_parts(tr + r0 * H, tr + r1 * H, H, sh);
  if (threadIdx.x == 0) {
    const float ce = e.dot / (fmaxf(e.na, 1e-8f) * fmaxf(e.nb, 1e-8f));
    const float cr = r.dot / (fmaxf(r.na, 1e-8f) * fmaxf(r.nb, 1e-8f));
    loss_rows[b] = fmaxf(ce, 0.f) + 1.f - cr;
  }
}
// d cos(a,b)/da = b/(|a||b|) - cos * a/|a|^2
__global__ void simloss_bwd_k(const float* __restrict__ tr, const int64_t* rel, const int64_t* qh, const int64_t* ah, const int32_t* rows, int nr,
                              const float* __restrict__ gscale, float rowscale, float* dtr, int B, int L, int H) {
  __shared__ float sh[8];
  const int b = blockIdx.x;
  const float g = (gscale ? gscale[0] : 1.f) * rowscale;
  for (int pair = 0; pair < 2; ++pair) {
    const long long ia = sim_row(rows, nr, b, L, pair == 0 ? qh[b] : rel[2 * b]), ib = sim_row(rows, nr, b, L, pair == 0 ? ah[b] : rel[2 * b + 1]);
    if ((ia | ib) < 0) continue;                                  // (the forward pass made this example's loss NaN)
    const float* a = tr + ia * H;
    const float* bb = tr + ib * H;
    Cos c = cos_parts(a, bb, H, sh);
    const float na = fmaxf(c.na, 1e-8f), nb = fmaxf(c.nb, 1e-8f);
    const float cs = c.dot / (na * nb);
    float coef = pair == 0 ? (cs > 0.f ? g : 0.f) : -g;          // relu(cos(q,a)) and -(cos(r0,r1))
    if (coef == 0.f) continue;
    for (int k = threadIdx.x; k < H; k += TPB) {
      const float da = coef * (bb[k] / (na * nb) - cs * a[k] / (na * na));
      const float db = coef * (a[k] / (na * nb) - cs * bb[k] / (nb * nb));
      atomicAdd(dtr + ia * H + k, da);
      atomicAdd(dtr + ib * H + k, db);
    }
  }
}

// ------------------------------------------------------------------ fused AdamW (torch.optim.AdamW semantics)
// One workgroup per 8192-element part of a chunk (a chunk is at most 65536 elements of one tensor): the streamed launches of a
// step cover ~230 chunks each, and one workgroup per chunk left most CUs with a single 4-wave workgroup (2.6 TB/s); eight
// parts per chunk put 7 workgroups on every CU.  Element-wise, so the split changes no result.
constexpr int ADAMW_PART = 8192;
__global__ __launch_bounds__(TPB) void adamw_k(mart_adamw_desc p) {
  const int parts = 65536 / ADAMW_PART;
  for (int wi = blockIdx.x; wi < p.n_chunks * parts; wi += gridDim.x) {
    const int ci = wi / parts, part = wi % parts;
    const int start = p.chunks[3 * ci], len = p.chunks[3 * ci + 1];
    const int flags = p.chunks[3 * ci + 2];
    const float wd = (flags & 1) ? p.weight_decay : 0.f;
    const float decay = 1.f - p.lr * wd;
    const float step_size = p.lr / p.bc1;
    const float inv_sqrt_bc2 = rsqrtf(p.bc2);
    bf16* sh = (bf16*)p.shadow_bf16;
    h16* sh16 = (flags & 2) ? (h16*)p.shadow_f16 : nullptr;       // fp16 forward shadow of the text-stream weights
    for (int base = part * ADAMW_PART; base < len; base += parts * ADAMW_PART)        // longer chunks (the ABI sets no limit): round-robin
    for (int i = base + threadIdx.x * 4, pend = min(len, base + ADAMW_PART); i < pend; i += TPB * 4) {
      const long long o = (long long)start + i;
      if (i + 3 < len) {
        f32x4 w = *(f32x4*)(p.master + o), g = *(const f32x4*)(p.grad + o), m = *(f32x4*)(p.m + o), v = *(f32x4*)(p.v + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ge = g[e] * p.grad_scale;
          w[e] *= decay;
          m[e] = p.beta1 * m[e] + (1.f - p.beta1) * ge;
          v[e] = p.beta2 * v[e] + (1.f - p.beta2) * ge * ge;
          w[e] -= step_size * m[e] / (sqrtf(v[e]) * inv_sqrt_bc2 + p.eps);
        }
        *(f32x4*)(p.master + o) = w; *(f32x4*)(p.m + o) = m; *(f32x4*)(p.v + o) = v;
        if (p.zero_grad) *(f32x4*)(p.grad + o) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sh) *(bf16x4*)(sh + o) = f4_to_bf4(w);
        if (sh16) *(bf16x4*)(sh16 + o) = f4_to_h4raw(w);
      } else {
        for (int e = 0; e < len - i; ++e) {
          const float ge = p.grad[o + e] * p.grad_scale;
          float w = p.master[o + e] * decay;
          const float m = p.beta1 * p.m[o + e] + (1.f - p.beta1) * ge;
          const float v = p.beta2 * p.v[o + e] + (1.f - p.beta2) * ge * ge;
          w -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + p.eps);
          p.master[o + e] = w; p.m[o + e] = m; p.v[o + e] = v;
          if (p.zero_grad) p.grad[o + e] = 0.f;
          if (sh) sh[o + e] = f2bf(w);
          if (sh16) sh16[o + e] = (h16)w;
        }
      }
    }
  }
}
}  // namespace

extern "C" int mart_lsce_fwd(const float* logits, int ld, const int64_t* label, long long ignore_index, float eps, float* loss_rows, float* lse,
                             float* loss_out, int reduction, int* status, int R, int C, void* stream) {
  MART_CHECK(logits && label && loss_rows && lse && R > 0 && C > 0 && ld >= C, "lsce_fwd: bad args");
  MART_CHECK(reduction == MART_REDUCE_NONE || ((reduction == MART_REDUCE_MEAN || reduction == MART_REDUCE_SUM) && loss_out), "lsce_fwd: reduction / loss_out");
  hipLaunchKernelGGL(lsce_fwd_k, dim3(R), dim3(TPB), 0, (hipStream_t)stream, logits, ld, label, ignore_index, eps, loss_rows, lse, status, R, C);
  MART_LAUNCH_CHECK();
  if (reduction != MART_REDUCE_NONE) {
    hipLaunchKernelGGL(lsce_reduce_k, dim3(1), dim3(TPB), 0, (hipStream_t)stream, loss_rows, label, ignore_index, reduction == MART_REDUCE_MEAN ? 1 : 0, loss_out, R);
    MART_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int mart_lsce_bwd(const float* logits, int ld, const int64_t* label, long long ignore_index, const float* lse, float eps, const float* gscale,
                             int gscale_per_row, float rowscale, const float* n_valid, void* dlogits_bf16, int ldo, float* dlogits_f32, int R, int C,
                             void* stream) {
  MART_CHECK(logits && label && lse && (dlogits_bf16 || dlogits_f32) && R > 0 && C > 0 && ld >= C && ldo >= C, "lsce_bwd: bad args");
  MART_CHECK(!gscale_per_row || gscale, "lsce_bwd: gscale_per_row needs gscale");
  hipLaunchKernelGGL(lsce_bwd_k, dim3(R), dim3(TPB), 0, (hipStream_t)stream, logits, ld, label, ignore_index, lse, eps, gscale, gscale_per_row, rowscale,
                     n_valid, (bf16*)dlogits_bf16, ldo, dlogits_f32, R, C);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_rank(const float* logits, int ld, const int64_t* label, int64_t* rank, int R, int C, void* stream) {
  MART_CHECK(logits && label && rank && R > 0 && C > 0 && ld >= C, "rank: bad args");
  hipLaunchKernelGGL(rank_k, dim3(R), dim3(TPB), 0, (hipStream_t)stream, logits, ld, label, rank, R, C);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_simloss_fwd(const float* trans, const int64_t* rel_idx, const int64_t* q_idx, const int64_t* a_idx, const int32_t* rows, int nr,
                                float* loss_rows, int B, int L, int H, void* stream) {
  MART_CHECK(trans && rel_idx && q_idx && a_idx && loss_rows && B > 0 && L > 0 && H > 0 && (!rows || (nr > 0 && nr <= 64)), "simloss_fwd: bad args");
  hipLaunchKernelGGL(simloss_fwd_k, dim3(B), dim3(TPB), 0, (hipStream_t)stream, trans, rel_idx, q_idx, a_idx, rows, nr, loss_rows, B, L, H);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_simloss_bwd(const float* trans, const int64_t* rel_idx, const int64_t* q_idx, const int64_t* a_idx, const int32_t* rows, int nr,
                                const float* gscale, float rowscale, float* dtrans, int B, int L, int H, void* stream) {
  MART_CHECK(trans && rel_idx && q_idx && a_idx && dtrans && B > 0 && L > 0 && H > 0 && (!rows || (nr > 0 && nr <= 64)), "simloss_bwd: bad args");
  hipLaunchKernelGGL(simloss_bwd_k, dim3(B), dim3(TPB), 0, (hipStream_t)stream, trans, rel_idx, q_idx, a_idx, rows, nr, gscale, rowscale, dtrans, B, L, H);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_adamw(const mart_adamw_desc* d, void* stream) {
  MART_CHECK(d && d->master && d->grad && d->m && d->v && d->chunks && d->n_chunks > 0, "adamw: bad args");
  MART_CHECK(d->bc1 > 0.f && d->bc2 > 0.f, "adamw: bias corrections must be positive");
  const long long wg = (long long)d->n_chunks * (65536 / ADAMW_PART);
  int g = wg < 8192 ? (int)wg : 8192;
  hipLaunchKernelGGL(adamw_k, dim3(g), dim3(TPB), 0, (hipStream_t)stream, *d);
  MART_LAUNCH_CHECK();
  return 0;
}

// Error plumbing, device check and small memory-bound utility kernels.
#include "common.h"
#include "mart_hip.h"
#include <string.h>

static thread_local char g_err[512] = "";
extern "C" void mart_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* mart_last_error(void) { return g_err; }
extern "C" int mart_abi_version(void) { return 10; }   // round 6: mart_lsce_* ignore_index / reduction / status; 9 = round 5: LayerNorm fold (gemm_nt row_stats / ln_*, mart_ln_fold_prep, mart_ln_stats_finalize); 8 = round 4: fp16 forward operands (gemm_nt in_f16 / c_f16, *_f16 outputs, adamw shadow_f16), row-subset helpers
extern "C" int mart_check_device(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { mart_set_error("no HIP device"); return -2; }
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, dev) != hipSuccess) { mart_set_error("hipGetDeviceProperties failed"); return -2; }
  if (strncmp(pr.gcnArchName, "gfx950", 6) != 0) {
    mart_set_error("libmart_hip is built for gfx950 (MI355X) only");
    return -1;
  }
  return 0;
}

namespace {
constexpr int TPB = 256;
inline int grid_for(long long n, int per_thread = 1) {
  long long b = (n + (long long)TPB * per_thread - 1) / ((long long)TPB * per_thread);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ void cast_f32_bf16_k(const float* __restrict__ s, bf16* __restrict__ d, long long n) {
  long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 4, stride = (long long)gridDim.x * TPB * 4;
  for (; i + 3 < n; i += stride) *(bf16x4*)(d + i) = f4_to_bf4(*(const f32x4*)(s + i));
  if (i < n && i + 3 >= n) for (long long j = i; j < n; ++j) d[j] = f2bf(s[j]);
}
__global__ void cast_bf16_f32_k(const bf16* __restrict__ s, float* __restrict__ d, long long n) {
  long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 4, stride = (long long)gridDim.x * TPB * 4;
  for (; i + 3 < n; i += stride) *(f32x4*)(d + i) = bf4_to_f4(*(const bf16x4*)(s + i));
  if (i < n && i + 3 >= n) for (long long j = i; j < n; ++j) d[j] = bf2f(s[j]);
}
__global__ void cast_f32_f16_k(const float* __restrict__ s, h16* __restrict__ d, long long n) {
  long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 4, stride = (long long)gridDim.x * TPB * 4;
  for (; i + 3 < n; i += stride) *(bf16x4*)(d + i) = f4_to_h4raw(*(const f32x4*)(s + i));
  if (i < n && i + 3 >= n) for (long long j = i; j < n; ++j) d[j] = (h16)s[j];
}
__global__ void cast_bf16_f16_k(const bf16* __restrict__ s, h16* __restrict__ d, long long n) {
  long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 4, stride = (long long)gridDim.x * TPB * 4;
  for (; i + 3 < n; i += stride) *(bf16x4*)(d + i) = f4_to_h4raw(bf4_to_f4(*(const bf16x4*)(s + i)));
  if (i < n && i + 3 >= n) for (long long j = i; j < n; ++j) d[j] = (h16)bf2f(s[j]);
}
// dst[r] = src[rows[r]] unless an earlier slot of the same group names the same row (then 0)
__global__ void gather_rows_first_k(const float* __restrict__ src, int ld, const int32_t* __restrict__ rows, int group, float* __restrict__ dst, int R, int H) {
  const int r = blockIdx.x, g0 = (r / group) * group, me = rows[r];
  bool dup = false;
  for (int i = g0; i < r; ++i) dup |= rows[i] == me;
  const float* s = src + (long long)me * ld;
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4)
    *(f32x4*)(dst + (long long)r * H + c) = dup ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(s + c);
}
// one workgroup per group, its slots applied in order.  ACC: dst[rows[r]] += src[r] (gradient rows; a row named twice receives both
// contributions).  !ACC: dst[rows[r]] = src[r], the FIRST slot that names a row wins (forward values: the later slots of a repeated row
// were computed under other dropout masks, and their gradient is dropped by gather_rows_first_k as well).
template <bool DBF, bool ACC>
__global__ void scatter_rows_k(const float* __restrict__ src, const int32_t* __restrict__ rows, int group, void* __restrict__ dst_, int ld, int R, int H) {
  const int g0 = blockIdx.x * group;
  for (int j = 0; j < group && g0 + j < R; ++j) {
    const long long drow = (long long)rows[g0 + j] * ld;
    bool seen = false;
    if (!ACC) for (int i = 0; i < j; ++i) seen |= rows[g0 + i] == rows[g0 + j];
    if (ACC || !seen)
      for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
        f32x4 v = *(const f32x4*)(src + (long long)(g0 + j) * H + c);
        if (DBF) {
          bf16* d = (bf16*)dst_ + drow + c;
          if (ACC) v += bf4_to_f4(*(const bf16x4*)d);
          *(bf16x4*)d = f4_to_bf4(v);
        } else {
          float* d = (float*)dst_ + drow + c;
          if (ACC) v += *(const f32x4*)d;
          *(f32x4*)d = v;
        }
      }
    __syncthreads();                                    // the next slot may name the same row: this slot's stores first
  }
}
__global__ void cast_pad_k(const float* __restrict__ s, int lds_, bf16* __restrict__ d, int ldd, int R, int C) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < ldd; c += blockDim.x) d[(long long)r * ldd + c] = c < C ? f2bf(s[(long long)r * lds_ + c]) : (bf16)0.f;
}
// out[i] = sum over the S split-K partial results parts[s * n + i], in split order (deterministic); n % 4 == 0
__global__ void sum_splits_k(const float* __restrict__ parts, int S, long long n, float* __restrict__ out) {
  long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 a = *(const f32x4*)(parts + i);
  for (int s = 1; s < S; ++s) a += *(const f32x4*)(parts + (long long)s * n + i);
  *(f32x4*)(out + i) = a;
}
__global__ void add_f32_bf16_k(const float* __restrict__ a, const bf16* __restrict__ b, float* __restrict__ of,
                               bf16* __restrict__ ob, long long n) {
  long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 4, stride = (long long)gridDim.x * TPB * 4;
  for (; i + 3 < n; i += stride) {
    f32x4 v = *(const f32x4*)(a + i);
    if (b) { f32x4 w = bf4_to_f4(*(const bf16x4*)(b + i)); v += w; }
    if (of) *(f32x4*)(of + i) = v;
    if (ob) *(bf16x4*)(ob + i) = f4_to_bf4(v);
  }
  if (i < n && i + 3 >= n) for (long long j = i; j < n; ++j) {
    float v = a[j] + (b ? bf2f(b[j]) : 0.f);
    if (of) of[j] = v;
    if (ob) ob[j] = f2bf(v);
  }
}
__global__ void dropout_mask_k(uint8_t* o, long long n, float p, uint64_t seed) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x, stride = (long long)gridDim.x * TPB;
  for (; i < n; i += stride) o[i] = dropout_keep(seed, (uint64_t)i, p) ? 1 : 0;
}
__global__ void dropout_bwd_k(const float* __restrict__ a, const bf16* __restrict__ b, float* __restrict__ o, long long n, float p, uint64_t seed) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x, stride = (long long)gridDim.x * TPB;
  const float sc = 1.f / (1.f - p);
  for (; i < n; i += stride) {
    float v = (a ? a[i] : 0.f) + (b ? bf2f(b[i]) : 0.f);
    o[i] = (p > 0.f) ? (dropout_keep(seed, (uint64_t)i, p) ? v * sc : 0.f) : v;
  }
}
__global__ void find_token_k(const int64_t* ids, int B, int L, int64_t tok, int32_t* pos, int32_t* row, int32_t* status) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int found = -1;
  for (int j = 0; j < L; ++j) if (ids[(long long)b * L + j] == tok) { found = j; break; }
  pos[b] = found;
  if (found < 0 && status) atomicOr(status, 2);       // an example without the token (the reference raises there): pos -1, row b*L+0, status bit 1
  if (row) row[b] = b * L + (found < 0 ? 0 : found);
}
// ---- row-subset passes (forward(needed_rows=...)): rows[b * nr + j] = flat id b * L + position of the j-th row example b was promised
// The rows of the step, built on the device in one launch: column 0 = first position of `tok` ([MASK]; absent: position 0 + status bit 1), then
// rel_idx[:, 0], rel_idx[:, 1], q_idx, a_idx when given (lit_models/transformer.py:94-95,103-107).  Negative positions wrap (+ L), then clamp to [0, L).
__global__ void needed_rows_k(const int64_t* __restrict__ ids, int B, int L, int64_t tok, const int64_t* __restrict__ rel, const int64_t* __restrict__ qi,
                              const int64_t* __restrict__ ai, int nr, int32_t* __restrict__ rows, int32_t* __restrict__ mask_row, int32_t* status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int found = -1;
  for (int j = 0; j < L; ++j) if (ids[(long long)b * L + j] == tok) { found = j; break; }
  if (found < 0 && status) atomicOr(status, 2);
  const int m = found < 0 ? 0 : found;
  rows[b * nr] = b * L + m;
  if (mask_row) mask_row[b] = b * L + m;
  if (nr >= 5) {
    const long long p[4] = {rel[2 * b], rel[2 * b + 1], qi[b], ai[b]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long v = p[j] < 0 ? p[j] + L : p[j];
      v = v < 0 ? 0 : (v >= L ? L - 1 : v);
      rows[b * nr + 1 + j] = b * L + (int)v;
    }
  }
}
// compact index of the flat row id f (example b = f / L): the FIRST slot of the example that names it (as scatter_rows_k / gather_rows_first_k);
// a row the pass was not promised: slot 0 of the example + status bit 2 (value 4) -- never an out-of-range index
__device__ __forceinline__ int rows_find(const int32_t* __restrict__ rows, int nr, int L, int f, int32_t* status) {
  const int b = f / L;
  for (int j = 0; j < nr; ++j) if (rows[b * nr + j] == f) return b * nr + j;
  if (status) atomicOr(status, 4);
  return b * nr;
}
__global__ void rows_lookup_k(const int32_t* __restrict__ flat, int n, const int32_t* __restrict__ rows, int nr, int L, int32_t* __restrict__ out, int32_t* status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rows_find(rows, nr, L, flat[i], status);
}
// the dense [B * L, H] f32 tensor of a row-subset pass: the promised rows from their compact copies (first slot wins), `fill` (NaN) elsewhere -- one pass
__global__ void rows_dense_k(const float* __restrict__ src, const int32_t* __restrict__ rows, int nr, int L, float* __restrict__ dst, float fill, int Mt, int H) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= Mt) return;
  const int b = r / L;
  int hit = -1;
  for (int j = nr - 1; j >= 0; --j) if (rows[b * nr + j] == r) hit = b * nr + j;
  const f32x4 fv = {fill, fill, fill, fill};
  for (int c = lane * 4; c < H; c += 256)
    *(f32x4*)(dst + (long long)r * H + c) = hit >= 0 ? *(const f32x4*)(src + (long long)hit * H + c) : fv;
}
__global__ void gather_rows_k(const float* __restrict__ src, int ld, const int32_t* __restrict__ rows, float* __restrict__ dst, int R, int H) {
  int r = blockIdx.x;
  const float* s = src + (long long)rows[r] * ld;
  for (int c = threadIdx.x; c < H; c += blockDim.x) dst[(long long)r * H + c] = s[c];
}
__global__ void gather_rows_bf16_k(const bf16* __restrict__ src, int ld, const int32_t* __restrict__ rows, bf16* __restrict__ dst, int R, int H) {
  int r = blockIdx.x;
  const bf16* s = src + (long long)rows[r] * ld;
  for (int c = threadIdx.x * 8; c < H; c += blockDim.x * 8) *(bf16x8*)(dst + (long long)r * H + c) = *(const bf16x8*)(s + c);
}
__global__ void act_bwd_k(const bf16* __restrict__ dy, const bf16* __restrict__ z, int act, bf16* __restrict__ o, long long n) {
  long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 4, stride = (long long)gridDim.x * TPB * 4;
  for (; i + 3 < n; i += stride) {
    f32x4 g = bf4_to_f4(*(const bf16x4*)(dy + i)), zz = bf4_to_f4(*(const bf16x4*)(z + i));
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] *= act_grad(zz[e], act);
    *(bf16x4*)(o + i) = f4_to_bf4(g);
  }
  if (i < n && i + 3 >= n) for (long long j = i; j < n; ++j) o[j] = f2bf(bf2f(dy[j]) * act_grad(bf2f(z[j]), act));
}
__global__ void scatter_add_rows_k(const float* __restrict__ src, const int32_t* __restrict__ rows, float* __restrict__ dst, int ld, int R, int H) {
  int r = blockIdx.x;
  float* d = dst + (long long)rows[r] * ld;
  for (int c = threadIdx.x; c < H; c += blockDim.x) atomicAdd(d + c, src[(long long)r * H + c]);
}
}  // namespace

extern "C" int mart_cast_f32_bf16(const float* src, void* dst, long long n, void* stream) {
  MART_CHECK(src && dst && n >= 0, "cast_f32_bf16: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cast_f32_bf16_k, dim3(grid_for(n, 4)), dim3(TPB), 0, (hipStream_t)stream, src, (bf16*)dst, n);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_cast_bf16_f32(const void* src, float* dst, long long n, void* stream) {
  MART_CHECK(src && dst && n >= 0, "cast_bf16_f32: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cast_bf16_f32_k, dim3(grid_for(n, 4)), dim3(TPB), 0, (hipStream_t)stream, (const bf16*)src, dst, n);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_cast_f32_f16(const float* src, void* dst, long long n, void* stream) {
  MART_CHECK(src && dst && n >= 0, "cast_f32_f16: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cast_f32_f16_k, dim3(grid_for(n, 4)), dim3(TPB), 0, (hipStream_t)stream, src, (h16*)dst, n);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_cast_bf16_f16(const void* src, void* dst, long long n, void* stream) {
  MART_CHECK(src && dst && n >= 0, "cast_bf16_f16: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cast_bf16_f16_k, dim3(grid_for(n, 4)), dim3(TPB), 0, (hipStream_t)stream, (const bf16*)src, (h16*)dst, n);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_gather_rows_first_f32(const float* src, int ld, const int32_t* rows, int group, float* dst, int R, int H, void* stream) {
  MART_CHECK(src && rows && dst && R > 0 && H > 0 && H % 4 == 0 && ld % 4 == 0 && group > 0 && group <= 64, "gather_rows_first_f32: bad args");
  hipLaunchKernelGGL(gather_rows_first_k, dim3(R), dim3(192), 0, (hipStream_t)stream, src, ld, rows, group, dst, R, H);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_scatter_rows(const float* src, const int32_t* rows, int group, void* dst, int ld, int dst_bf16, int accumulate, int R, int H, void* stream) {
  MART_CHECK(src && rows && dst && R > 0 && H > 0 && H % 4 == 0 && ld % 4 == 0 && group > 0 && group <= 64, "scatter_rows: bad args");
  const dim3 g((R + group - 1) / group), b(192);
  hipStream_t st = (hipStream_t)stream;
  if (dst_bf16) {
    if (accumulate) hipLaunchKernelGGL((scatter_rows_k<true, true>), g, b, 0, st, src, rows, group, dst, ld, R, H);
    else hipLaunchKernelGGL((scatter_rows_k<true, false>), g, b, 0, st, src, rows, group, dst, ld, R, H);
  } else {
    if (accumulate) hipLaunchKernelGGL((scatter_rows_k<false, true>), g, b, 0, st, src, rows, group, dst, ld, R, H);
    else hipLaunchKernelGGL((scatter_rows_k<false, false>), g, b, 0, st, src, rows, group, dst, ld, R, H);
  }
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_add_f32_bf16(const float* a, const void* b, float* of, void* ob, long long n, void* stream) {
  MART_CHECK(a && (of || ob) && n >= 0, "add_f32_bf16: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_f32_bf16_k, dim3(grid_for(n, 4)), dim3(TPB), 0, (hipStream_t)stream, a, (const bf16*)b, of, (bf16*)ob, n);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_dropout_mask(uint8_t* out, long long n, float p, uint64_t seed, void* stream) {
  MART_CHECK(out && n >= 0, "dropout_mask: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_mask_k, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, out, n, p, seed);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_dropout_bwd_f32(const float* dy_f32, const void* dy_bf16, float* out, long long n, float p, uint64_t seed, void* stream) {
  MART_CHECK((dy_f32 || dy_bf16) && out && n >= 0 && p >= 0.f && p < 1.f, "dropout_bwd_f32: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_bwd_k, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, dy_f32, (const bf16*)dy_bf16, out, n, p, seed);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_find_token(const int64_t* ids, int B, int L, int64_t token, int32_t* pos_out, int32_t* row_out, int32_t* status, void* stream) {
  MART_CHECK(ids && pos_out && B > 0 && L > 0, "find_token: bad args");
  hipLaunchKernelGGL(find_token_k, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, ids, B, L, token, pos_out, row_out, status);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_needed_rows(const int64_t* ids, int B, int L, int64_t token, const int64_t* rel_idx, const int64_t* q_idx, const int64_t* a_idx,
                                int32_t* rows_out, int32_t* mask_row_out, int32_t* status, void* stream) {
  MART_CHECK(ids && rows_out && B > 0 && L > 0, "needed_rows: bad args");
  MART_CHECK((rel_idx && q_idx && a_idx) || (!rel_idx && !q_idx && !a_idx), "needed_rows: rel_idx / q_idx / a_idx come together");
  MART_CHECK((long long)B * L < (1ll << 31), "needed_rows: B * L must fit 31 bits");
  hipLaunchKernelGGL(needed_rows_k, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, ids, B, L, token, rel_idx, q_idx, a_idx, rel_idx ? 5 : 1, rows_out,
                     mask_row_out, status);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_rows_lookup(const int32_t* flat, int n, const int32_t* rows, int nr, int L, int32_t* out, int32_t* status, void* stream) {
  MART_CHECK(flat && rows && out && n > 0 && nr > 0 && nr <= 64 && L > 0, "rows_lookup: bad args");
  hipLaunchKernelGGL(rows_lookup_k, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, flat, n, rows, nr, L, out, status);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_rows_dense(const float* src, const int32_t* rows, int nr, int B, int L, int H, float* dst, float fill, void* stream) {
  MART_CHECK(src && rows && dst && nr > 0 && nr <= 64 && B > 0 && L > 0 && H > 0 && H % 4 == 0, "rows_dense: bad args");
  const int Mt = B * L;
  hipLaunchKernelGGL(rows_dense_k, dim3((Mt + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, rows, nr, L, dst, fill, Mt, H);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_gather_rows_f32(const float* src, int ld, const int32_t* rows, float* dst, int R, int H, void* stream) {
  MART_CHECK(src && rows && dst && R > 0 && H > 0, "gather_rows_f32: bad args");
  hipLaunchKernelGGL(gather_rows_k, dim3(R), dim3(256), 0, (hipStream_t)stream, src, ld, rows, dst, R, H);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_scatter_add_rows_f32(const float* src, const int32_t* rows, float* dst, int ld, int R, int H, void* stream) {
  MART_CHECK(src && rows && dst && R > 0 && H > 0, "scatter_add_rows_f32: bad args");
  hipLaunchKernelGGL(scatter_add_rows_k, dim3(R), dim3(256), 0, (hipStream_t)stream, src, rows, dst, ld, R, H);
  MART_LAUNCH_CHECK();
  return 0;
}

extern "C" int mart_gather_rows_bf16(const void* src, int ld, const int32_t* rows, void* dst, int R, int H, void* stream) {
  MART_CHECK(src && rows && dst && R > 0 && H > 0 && H % 8 == 0 && ld % 8 == 0, "gather_rows_bf16: bad args");
  hipLaunchKernelGGL(gather_rows_bf16_k, dim3(R), dim3(128), 0, (hipStream_t)stream, (const bf16*)src, ld, rows, (bf16*)dst, R, H);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_act_bwd(const void* dy_bf16, const void* z_bf16, int act, void* out_bf16, long long n, void* stream) {
  MART_CHECK(dy_bf16 && z_bf16 && out_bf16 && n >= 0, "act_bwd: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_bwd_k, dim3(grid_for(n, 4)), dim3(TPB), 0, (hipStream_t)stream, (const bf16*)dy_bf16, (const bf16*)z_bf16, act, (bf16*)out_bf16, n);
  MART_LAUNCH_CHECK();
  return 0;
}

extern "C" int mart_sum_splits_f32(const float* parts, int S, long long n, float* out, void* stream) {
  MART_CHECK(parts && out && S > 0 && n > 0 && n % 4 == 0, "sum_splits_f32: bad args");
  hipLaunchKernelGGL(sum_splits_k, dim3((unsigned)((n / 4 + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, parts, S, n, out);
  MART_LAUNCH_CHECK();
  return 0;
}
extern "C" int mart_cast_pad_f32_bf16(const float* src, int lds_, void* dst, int ldd, int R, int C, void* stream) {
  MART_CHECK(src && dst && R > 0 && C > 0 && ldd >= C && lds_ >= C, "cast_pad_f32_bf16: bad args");
  hipLaunchKernelGGL(cast_pad_k, dim3(R), dim3(256), 0, (hipStream_t)stream, src, lds_, (bf16*)dst, ldd, R, C);
  MART_LAUNCH_CHECK();
  return 0;
}

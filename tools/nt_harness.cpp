// Standalone driver for mart_gemm_nt (no torch): correctness of the 8-phase K loop against the round-1 loop (bitwise) and a
// naive reference, a run-to-run race screen, and interleaved A/B timing on the step's shapes.
//   tools/build_variant.sh gemm_nt.hip tools/variants/libmart_hip.so -DMART_EXPERIMENTS      (the experiment tile_cfg codes live only in this build)
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/nt_harness.cpp -o tools/nt_harness -Ltools/variants -lmart_hip -Wl,-rpath,'$ORIGIN/variants'
//   tools/nt_harness [check|time|all] [rounds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <math.h>
#include <dlfcn.h>
#include "mart_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    float f = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;      // uniform [-scale, scale)
    uint32_t u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); p[i] = (uint16_t)(u >> 16);
  }
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    p[i] = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
  }
}
// naive reference: C[m,n] = sum_k A[m,k] B[n,k] (+ second pair), f32
__global__ void ref_nt(const uint16_t* A, const uint16_t* B, const uint16_t* A2, const uint16_t* B2, float* C, int M, int N, int K, int K2, int lda, int ldb) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= M) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += __uint_as_float((uint32_t)A[(size_t)m * lda + k] << 16) * __uint_as_float((uint32_t)B[(size_t)n * ldb + k] << 16);
  for (int k = 0; k < K2; ++k) acc += __uint_as_float((uint32_t)A2[(size_t)m * lda + k] << 16) * __uint_as_float((uint32_t)B2[(size_t)n * ldb + k] << 16);
  C[(size_t)m * N + n] = acc;
}

struct Case { const char* name; int M, N, K, K2; int epi; };   // epi: 0 plain bf16+bias, 1 f32 + res, 2 fc1 (preact grad + act qgelu), 3 mulz stored, 4 f32 plain, 5 f32+res+C2, 6 fc1 gelu dualK
static void* dalloc(size_t b) { void* p; CK(hipMalloc(&p, b)); return p; }

struct Bufs {
  uint16_t *A[12], *B, *A2, *B2, *mulz, *preact, *C2; float *bias, *bias2, *res; void* C[2]; size_t cbytes;
};

static void make_desc(mart_gemm_nt_desc& d, const Case& c, const Bufs& b, int rot, int which, int cfg) {
  memset(&d, 0, sizeof d);
  d.A = b.A[rot % 12]; d.B = b.B; d.lda = c.K > c.K2 ? c.K : c.K2; d.ldb = d.lda; d.M = c.M; d.N = c.N; d.K = c.K; d.K2 = c.K2;
  if (c.K2) { d.A2 = b.A2; d.B2 = b.B2; }
  d.batch = 1; d.alpha = 1.0f; d.C = b.C[which]; d.ldc = c.N; d.tile_cfg = cfg; d.bias = b.bias;
  static const int hot = getenv("NT_HOT") ? atoi(getenv("NT_HOT")) : 0;    // timing only (wrong numbers): 1: every A row aliases row 0, 2: B too, so the
  if (hot >= 1) d.lda = 0;                                                   // LDS-DMA stream hits L1 / L2 -- the K loop without memory-side latency
  if (hot >= 2) d.ldb = 0;
  static const int nobias = getenv("NT_NOBIAS") ? atoi(getenv("NT_NOBIAS")) : 0;     // the data-gradient products of the step carry no bias
  if (nobias) d.bias = nullptr;
  switch (c.epi) {
    case 0: break;
    case 1: d.c_f32 = 1; d.res_f32 = b.res; break;
    case 2: d.act = MART_ACT_QGELU; d.preact = b.preact; d.preact_grad = 1; break;
    case 3: d.mulz = b.mulz; d.mul_act = MART_ACT_STORED; d.bias = nullptr; break;
    case 4: d.c_f32 = 1; break;
    case 5: d.c_f32 = 1; d.res_f32 = b.res; d.C2 = b.C2; break;
    case 6: d.act = MART_ACT_GELU; d.preact = b.preact; d.preact_grad = 1; d.bias2 = b.bias2; break;
  }
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  int rounds = argc > 2 ? atoi(argv[2]) : 5;
  if (mart_check_device() != 0) { printf("not gfx950: %s\n", mart_last_error()); return 1; }
  const int MV = 256 * 393, MT = 256 * 64;
  std::vector<Case> timing = {
    {"v.qkv   fwd  [M,2304,768] bf16+bias", MV, 2304, 768, 0, 0},
    {"v.oproj fwd  [M,768,768] f32+res", MV, 768, 768, 0, 1},
    {"v.fc1   fwd  [M,3072,768] act+act'", MV, 3072, 768, 0, 2},
    {"v.fc2   fwd  [M,768,3072] f32+res", MV, 768, 3072, 0, 1},
    {"v.fc2   dgrad[M,3072,768] *act'", MV, 3072, 768, 0, 3},
    {"v.fc1   dgrad[M,768,3072] bf16", MV, 768, 3072, 0, 0},
    {"v.oproj dgrad[M,768,768] bf16", MV, 768, 768, 0, 0},
    {"v.qkv   dgrad[M,768,2304] bf16", MV, 768, 2304, 0, 0},
    {"t.fc1+fus fwd[Mt,3072,768+768] gelu", MT, 3072, 768, 768, 6},
    {"t.qkv   fwd  [Mt,2304,768]", MT, 2304, 768, 0, 0},
  };
  std::vector<Case> checks = {
    {"tiny K=64", 300, 256, 64, 0, 0}, {"K=128 f32", 777, 512, 128, 0, 4}, {"K=192 res", 1000, 768, 192, 0, 1},
    {"ragged M", 12576 + 77, 768, 768, 0, 1}, {"fc1", 12576, 3072, 768, 0, 2}, {"mulz", 5000, 3072, 768, 0, 3},
    {"dualK gelu", 4096, 3072, 768, 768, 6}, {"res+C2", 3000, 768, 3072, 0, 5}, {"plain K=2304", 9000, 768, 2304, 0, 0},
    {"ragged N (general)", 1000, 2063, 768, 0, 4},
  };
  size_t maxA = (size_t)MV * 3072, maxC = (size_t)MV * 3072;
  Bufs b;
  for (int i = 0; i < 12; ++i) { b.A[i] = (uint16_t*)dalloc(maxA * 2); fill_bf16<<<1024, 256>>>(b.A[i], maxA, 17 + i, 1.0f); }
  b.B = (uint16_t*)dalloc((size_t)3072 * 3072 * 2); fill_bf16<<<1024, 256>>>(b.B, (size_t)3072 * 3072, 101, 0.05f);
  b.A2 = (uint16_t*)dalloc((size_t)MT * 768 * 2); fill_bf16<<<1024, 256>>>(b.A2, (size_t)MT * 768, 102, 1.0f);
  b.B2 = (uint16_t*)dalloc((size_t)3072 * 768 * 2); fill_bf16<<<1024, 256>>>(b.B2, (size_t)3072 * 768, 103, 0.05f);
  b.mulz = (uint16_t*)dalloc(maxC * 2); fill_bf16<<<1024, 256>>>(b.mulz, maxC, 104, 1.0f);
  b.preact = (uint16_t*)dalloc(maxC * 2); b.C2 = (uint16_t*)dalloc(maxC * 2);
  b.bias = (float*)dalloc(4096 * 4); fill_f32<<<16, 256>>>(b.bias, 4096, 105, 0.5f);
  b.bias2 = (float*)dalloc(4096 * 4); fill_f32<<<16, 256>>>(b.bias2, 4096, 106, 0.5f);
  b.res = (float*)dalloc(maxC * 4); fill_f32<<<1024, 256>>>(b.res, maxC, 107, 1.0f);
  b.cbytes = maxC * 4; b.C[0] = dalloc(b.cbytes); b.C[1] = dalloc(b.cbytes);
  float* refC = (float*)dalloc((size_t)1024 * 4096 * 4);
  CK(hipDeviceSynchronize());
  hipStream_t st; CK(hipStreamCreate(&st));
  int bad = 0;

  if (!strcmp(mode, "check") || !strcmp(mode, "all")) {
    for (const Case& c : checks) {
      mart_gemm_nt_desc d;
      size_t obytes = (size_t)c.M * c.N * ((c.epi == 1 || c.epi == 4 || c.epi == 5) ? 4 : 2);
      std::vector<char> h0(obytes), h1(obytes), h2(obytes);
      // old loop
      CK(hipMemsetAsync(b.C[0], 0xff, obytes, st)); CK(hipMemsetAsync(b.C[1], 0xff, obytes, st));
      const int ccfg[2] = {argc > 3 ? atoi(argv[3]) : 2563, argc > 4 ? atoi(argv[4]) : 256};       // check <rounds> [cfgA cfgB]: bitwise A == B
      make_desc(d, c, b, 0, 0, ccfg[0]); if (mart_gemm_nt(&d, st)) { printf("old launch failed: %s\n", mart_last_error()); return 1; }
      make_desc(d, c, b, 0, 1, 0); d.tile_cfg = ccfg[1]; if (mart_gemm_nt(&d, st)) { printf("new launch failed: %s\n", mart_last_error()); return 1; }
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h0.data(), b.C[0], obytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), b.C[1], obytes, hipMemcpyDeviceToHost));
      bool same = !memcmp(h0.data(), h1.data(), obytes);
      // race screen: 20 more runs of the new loop, all identical
      int races = 0;
      for (int r = 0; r < 20; ++r) {
        CK(hipMemsetAsync(b.C[1], 0xff, obytes, st));
        if (mart_gemm_nt(&d, st)) return 1;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h2.data(), b.C[1], obytes, hipMemcpyDeviceToHost));
        if (memcmp(h1.data(), h2.data(), obytes)) ++races;
      }
      // naive reference on the first <=1024 rows for the plain f32 case
      double maxerr = -1;
      if (c.epi == 4) {
        int Mr = std::min(c.M, 1024);
        ref_nt<<<dim3((c.N + 255) / 256, Mr), 256, 0, st>>>(b.A[0], b.B, b.A2, b.B2, refC, Mr, c.N, c.K, c.K2, c.K, c.K);
        CK(hipStreamSynchronize(st));
        std::vector<float> hr((size_t)Mr * c.N), hb(4096);
        CK(hipMemcpy(hr.data(), refC, hr.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b.bias, 4096 * 4, hipMemcpyDeviceToHost));
        maxerr = 0;
        const float* o = (const float*)h1.data();
        for (int m = 0; m < Mr; ++m) for (int n = 0; n < c.N; ++n) maxerr = std::max(maxerr, (double)fabsf(o[(size_t)m * c.N + n] - hr[(size_t)m * c.N + n] - hb[n]));
      }
      printf("check %-22s M=%6d N=%5d K=%4d+%4d: new==old %s, races %d/20%s", c.name, c.M, c.N, c.K, c.K2, same ? "yes" : "NO", races, "");
      if (maxerr >= 0) printf(", vs naive max|err| %.3e", maxerr);
      printf("\n");
      if (!same || races || maxerr > 1e-2) ++bad;
    }
  }

  if (!strcmp(mode, "time") || !strcmp(mode, "all")) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int cfgs[2] = {argc > 3 ? atoi(argv[3]) : 2563, argc > 4 ? atoi(argv[4]) : 0};
    printf("timing: tile_cfg %d (\"old\") vs %d (\"new\")\n", cfgs[0], cfgs[1]);
    const int NIT = 12;
    for (const Case& c : timing) {
      std::vector<float> ms[2];
      mart_gemm_nt_desc d;
      for (int w = 0; w < 2; ++w) for (int k = 0; k < 2; ++k) { make_desc(d, c, b, w, 0, cfgs[k]); mart_gemm_nt(&d, st); }
      for (int r = 0; r < rounds; ++r)
        for (int k = 0; k < 2; ++k) {
          CK(hipEventRecord(e0, st));
          for (int it = 0; it < NIT; ++it) { make_desc(d, c, b, it, 0, cfgs[k]); if (mart_gemm_nt(&d, st)) { printf("launch failed: %s\n", mart_last_error()); return 1; } }
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[k].push_back(t / NIT);
        }
      double fl = 2.0 * c.M * c.N * (double)(c.K + c.K2);
      for (int k = 0; k < 2; ++k) std::sort(ms[k].begin(), ms[k].end());
      double m0 = ms[0][ms[0].size() / 2], m1 = ms[1][ms[1].size() / 2];
      printf("time %-40s old %.4f ms (%6.1f TF/s)   new %.4f ms (%6.1f TF/s)   x%.3f\n", c.name, m0, fl / m0 * 1e-9, m1, fl / m1 * 1e-9, m0 / m1);
    }
  }
  if (!strcmp(mode, "stamp")) {
    // cycle stamps of workgroup 200 (tile_cfg 2567: one tile per workgroup; 2568: the persistent loop where the product uses it; -DMART_EXPERIMENTS library)
    typedef int (*st_t)(unsigned long long*);
    st_t getst = (st_t)dlsym(RTLD_DEFAULT, "mart_debug_nt_stamps");
    if (!getst) { printf("library has no mart_debug_nt_stamps (build with -DMART_EXPERIMENTS)\n"); return 1; }
    const int scfg = argc > 2 ? atoi(argv[2]) : 2567;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Case& c : timing) {
      mart_gemm_nt_desc d;
      for (int w = 0; w < 3; ++w) { make_desc(d, c, b, w, 0, scfg); if (mart_gemm_nt(&d, st)) { printf("launch failed: %s\n", mart_last_error()); return 1; } }
      CK(hipEventRecord(e0, st));
      make_desc(d, c, b, 3, 0, scfg); if (mart_gemm_nt(&d, st)) return 1;
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long hs[3 * 8 * 9]; if (getst(hs)) { printf("stamp copy failed\n"); return 1; }
      const int tiles = ((c.M + 255) / 256) * ((c.N + 255) / 256);
      printf("stamp %-40s cfg %d: launch %.4f ms, %d tiles = %.2f rounds -> %.0f ns per tile-round (cycles below; workgroup 200)\n", c.name, scfg, ms, tiles, tiles / 256.0, ms * 1e6 / (tiles / 256.0));
      for (int it = 0; it < (scfg == 2568 ? 3 : 1); ++it)
        for (int w = 0; w < 8; w += 4) {
          const unsigned long long* t = hs + (it * 8 + w) * 9;
          if (!t[1]) continue;
          printf("   tile %d wave %d: entry -> loop start %6lld | K loop %6lld (%5.0f / K-tile) | -> epilogue start %5lld | blocks %5lld %5lld %5lld %5lld | tail %4lld | epilogue %6lld | tile total %6lld",
                 it, w, (long long)(t[0] - t[8]), (long long)(t[1] - t[0]), (double)(t[1] - t[0]) / ((c.K + c.K2) / 64), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]),
                 (long long)(t[4] - t[3]), (long long)(t[5] - t[4]), (long long)(t[6] - t[5]), (long long)(t[7] - t[6]), (long long)(t[7] - t[2]), (long long)(t[7] - t[8]));
          if (it > 0) { const unsigned long long* q = hs + ((it - 1) * 8 + w) * 9; printf(" | since previous tile's end %5lld", (long long)(t[8] - q[7])); }
          printf("\n");
        }
    }
  }
  if (!strcmp(mode, "ab")) {
    // interleaved A/B of two BUILDS of the library in one process:  nt_harness ab <rounds> <libA.so> <libB.so> [cfgA cfgB]
    typedef int (*fn_t)(const mart_gemm_nt_desc*, void*);
    void* ha = dlopen(argv[3], RTLD_NOW | RTLD_LOCAL); void* hb = dlopen(argv[4], RTLD_NOW | RTLD_LOCAL);
    if (!ha || !hb) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    fn_t fns[2] = {(fn_t)dlsym(ha, "mart_gemm_nt"), (fn_t)dlsym(hb, "mart_gemm_nt")};
    const int cfgs[2] = {argc > 5 ? atoi(argv[5]) : 0, argc > 6 ? atoi(argv[6]) : 0};
    printf("A = %s (cfg %d)   B = %s (cfg %d)\n", argv[3], cfgs[0], argv[4], cfgs[1]);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NIT = 12;
    double tot[2] = {0, 0};
    for (const Case& c : timing) {
      std::vector<float> ms[2];
      mart_gemm_nt_desc d;
      for (int w = 0; w < 2; ++w) for (int k = 0; k < 2; ++k) { make_desc(d, c, b, w, 0, cfgs[k]); fns[k](&d, st); }
      for (int r = 0; r < rounds; ++r)
        for (int k = 0; k < 2; ++k) {
          CK(hipEventRecord(e0, st));
          for (int it = 0; it < NIT; ++it) { make_desc(d, c, b, it, 0, cfgs[k]); if (fns[k](&d, st)) { printf("launch failed\n"); return 1; } }
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[k].push_back(t / NIT);
        }
      double fl = 2.0 * c.M * c.N * (double)(c.K + c.K2);
      for (int k = 0; k < 2; ++k) std::sort(ms[k].begin(), ms[k].end());
      double m0 = ms[0][ms[0].size() / 2], m1 = ms[1][ms[1].size() / 2];
      tot[0] += m0; tot[1] += m1;
      printf("ab   %-40s A %.4f ms (%6.1f TF/s)   B %.4f ms (%6.1f TF/s)   A/B x%.3f\n", c.name, m0, fl / m0 * 1e-9, m1, fl / m1 * 1e-9, m0 / m1);
    }
    printf("ab   sum over the shapes: A %.4f ms, B %.4f ms, A/B x%.3f\n", tot[0], tot[1], tot[0] / tot[1]);
  }
  printf(bad ? "HARNESS: %d FAILED\n" : "HARNESS: all checks passed\n", bad);
  return bad ? 1 : 0;
}

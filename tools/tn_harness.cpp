// Standalone driver for mart_gemm_tn (no torch): the deterministic path (8-phase loop + ordered split reduction through a
// workspace) against the atomic path and a naive reference, run-to-run bit equality, and interleaved A/B timing on the step's shapes.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/tn_harness.cpp -o tools/tn_harness -Lmkg_analogy_amd/lib -lmart_hip -Wl,-rpath,'$ORIGIN/../mkg_analogy_amd/lib'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "mart_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    float f = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
    uint32_t u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); p[i] = (uint16_t)(u >> 16);
  }
}
// naive reference: out[nx][ny] = sum_m X[m][nx] Y[m][ny], cs[nx] = sum_m X[m][nx]   (f64 accumulate)
__global__ void ref_tn(const uint16_t* X, const uint16_t* Y, double* out, double* cs, int M, int NX, int NY, int ldx, int ldy, int nx_lim) {
  int ny = blockIdx.x * blockDim.x + threadIdx.x, nx = blockIdx.y;
  if (ny >= NY || nx >= nx_lim) return;
  double acc = 0, c = 0;
  for (int m = 0; m < M; ++m) {
    double x = __uint_as_float((uint32_t)X[(size_t)m * ldx + nx] << 16);
    acc += x * __uint_as_float((uint32_t)Y[(size_t)m * ldy + ny] << 16);
    c += x;
  }
  out[(size_t)nx * NY + ny] = acc;
  if (ny == 0) cs[nx] = c;
}
static void* dalloc(size_t b) { void* p; CK(hipMalloc(&p, b)); return p; }
struct Case { const char* name; int M, NX, NY; };

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  int rounds = argc > 2 ? atoi(argv[2]) : 5;
  if (mart_check_device() != 0) { printf("not gfx950\n"); return 1; }
  const int MV = 256 * 393, MT = 256 * 64;
  std::vector<Case> cases = {
    {"v.fc2  dW [768,3072]  M=100608", MV, 768, 3072}, {"v.fc1  dW [3072,768]  M=100608", MV, 3072, 768},
    {"v.qkv  dW [2304,768]  M=100608", MV, 2304, 768}, {"v.out  dW [768,768]   M=100608", MV, 768, 768},
    {"t.fc2  dW [768,3072]  M=16384", MT, 768, 3072},  {"t.qkv  dW [2304,768]  M=16384", MT, 2304, 768},
    {"t.out  dW [768,768]   M=16384", MT, 768, 768},   {"ragged    [2063,768]  M=256", 256, 2063, 768},
    {"ragged    [700,1000]  M=1216", 1216, 700, 1000}, {"ragged M  [768,768]   M=1584", 1584, 768, 768}, {"tiny M    [768,2304]  M=40", 40, 768, 2304},
  };
  const size_t maxX = (size_t)MV * 3072;
  uint16_t* X[6]; uint16_t* Y[6];
  for (int i = 0; i < 6; ++i) {
    X[i] = (uint16_t*)dalloc(maxX * 2); fill_bf16<<<1024, 256>>>(X[i], maxX, 11 + i, 1.0f);
    Y[i] = (uint16_t*)dalloc(maxX * 2); fill_bf16<<<1024, 256>>>(Y[i], maxX, 31 + i, 1.0f);
  }
  const size_t maxO = (size_t)3072 * 3072;
  float* out[2] = {(float*)dalloc(maxO * 4), (float*)dalloc(maxO * 4)};
  float* cs[2] = {(float*)dalloc(4096 * 4), (float*)dalloc(4096 * 4)};
  double* rout = (double*)dalloc((size_t)64 * 3072 * 8); double* rcs = (double*)dalloc(4096 * 8);
  const size_t wsb = 256ull << 20;
  void* ws = dalloc(wsb);
  hipStream_t st; CK(hipStreamCreate(&st));
  CK(hipDeviceSynchronize());
  int bad = 0;
  auto desc = [&](const Case& c, int rot, int which, bool det) {
    mart_gemm_tn_desc d; memset(&d, 0, sizeof d);
    const int ldx = ((c.NX + 7) / 8) * 8, ldy = ((c.NY + 7) / 8) * 8;
    d.X = X[rot % 6]; d.Y = Y[rot % 6]; d.ldx = ldx; d.ldy = ldy; d.M = c.M; d.NX = c.NX; d.NY = c.NY;
    d.out = out[which]; d.ldo = c.NY; d.colsum = cs[which]; d.batch = 1; d.alpha = 1.0f;
    if (det) { d.workspace = ws; d.workspace_bytes = (long long)wsb; }
    return d;
  };
  if (!strcmp(mode, "check") || !strcmp(mode, "all")) {
    for (const Case& c : cases) {
      const size_t on = (size_t)c.NX * c.NY;
      std::vector<float> h0(on), h1(on), h2(on), c0(c.NX), c1(c.NX);
      if ((size_t)mart_gemm_tn_workspace_bytes(c.M, c.NX, c.NY, 0) > wsb) { printf("workspace too small for %s\n", c.name); return 1; }
      for (int w = 0; w < 2; ++w) { CK(hipMemsetAsync(out[w], 0, on * 4, st)); CK(hipMemsetAsync(cs[w], 0, c.NX * 4, st)); }
      mart_gemm_tn_desc d0 = desc(c, 0, 0, false), d1 = desc(c, 0, 1, true);
      if (mart_gemm_tn(&d0, st) || mart_gemm_tn(&d1, st)) { printf("launch failed: %s\n", mart_last_error()); return 1; }
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h0.data(), out[0], on * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), out[1], on * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(c0.data(), cs[0], c.NX * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), cs[1], c.NX * 4, hipMemcpyDeviceToHost));
      double dmax = 0, nrm = 0, cmax = 0;
      for (size_t i = 0; i < on; ++i) { dmax = std::max(dmax, (double)fabsf(h0[i] - h1[i])); nrm = std::max(nrm, (double)fabsf(h0[i])); }
      for (int i = 0; i < c.NX; ++i) cmax = std::max(cmax, (double)fabsf(c0[i] - c1[i]));
      // run-to-run bit equality of the deterministic path (accumulating into zeroed outputs each time)
      int diff = 0;
      for (int r = 0; r < 5; ++r) {
        CK(hipMemsetAsync(out[1], 0, on * 4, st)); CK(hipMemsetAsync(cs[1], 0, c.NX * 4, st));
        if (mart_gemm_tn(&d1, st)) return 1;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h2.data(), out[1], on * 4, hipMemcpyDeviceToHost));
        if (memcmp(h1.data(), h2.data(), on * 4)) ++diff;
      }
      // naive f64 reference on the first 64 output rows
      const int lim = std::min(c.NX, 64);
      const int ldx = ((c.NX + 7) / 8) * 8, ldy = ((c.NY + 7) / 8) * 8;
      ref_tn<<<dim3((c.NY + 255) / 256, lim), 256, 0, st>>>(X[0], Y[0], rout, rcs, c.M, c.NX, c.NY, ldx, ldy, lim);
      CK(hipStreamSynchronize(st));
      std::vector<double> hr((size_t)lim * c.NY), hc(lim);
      CK(hipMemcpy(hr.data(), rout, hr.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), rcs, lim * 8, hipMemcpyDeviceToHost));
      double rmax = 0, rc = 0;
      for (size_t i = 0; i < hr.size(); ++i) rmax = std::max(rmax, fabs(h1[i] - hr[i]));
      for (int i = 0; i < lim; ++i) rc = std::max(rc, fabs(c1[i] - hc[i]));
      const double tol = 2e-5 * nrm + 1e-3 * sqrt((double)c.M) * 1e-2;
      printf("check %-32s det-vs-atomic max|d| %.3e (|out| max %.1f)  colsum max|d| %.3e  vs f64 ref %.3e / colsum %.3e  run-to-run diffs %d/5\n",
             c.name, dmax, nrm, cmax, rmax, rc, diff);
      if (diff || dmax > 1e-4 * nrm + 1e-3 || rmax > 1e-4 * nrm + 1e-3 || rc > 1e-2) ++bad;
      (void)tol;
    }
  }
  if (!strcmp(mode, "time") || !strcmp(mode, "all")) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NIT = 8;
    for (const Case& c : cases) {
      std::vector<float> ms[2];
      for (int k = 0; k < 2; ++k) { mart_gemm_tn_desc d = desc(c, 0, 0, k == 1); mart_gemm_tn(&d, st); }
      for (int r = 0; r < rounds; ++r)
        for (int k = 0; k < 2; ++k) {
          CK(hipEventRecord(e0, st));
          for (int it = 0; it < NIT; ++it) { mart_gemm_tn_desc d = desc(c, it, 0, k == 1); if (mart_gemm_tn(&d, st)) { printf("launch failed: %s\n", mart_last_error()); return 1; } }
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[k].push_back(t / NIT);
        }
      double fl = 2.0 * c.M * (double)c.NX * c.NY;
      for (int k = 0; k < 2; ++k) std::sort(ms[k].begin(), ms[k].end());
      double m0 = ms[0][ms[0].size() / 2], m1 = ms[1][ms[1].size() / 2];
      printf("time %-32s atomic %.4f ms (%6.1f TF/s)   deterministic (+reduce kernel) %.4f ms (%6.1f TF/s)   x%.3f\n", c.name, m0, fl / m0 * 1e-9, m1, fl / m1 * 1e-9, m0 / m1);
    }
  }
  printf(bad ? "HARNESS: %d FAILED\n" : "HARNESS: all checks passed\n", bad);
  return bad ? 1 : 0;
}
